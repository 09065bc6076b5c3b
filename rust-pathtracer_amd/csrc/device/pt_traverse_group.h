// pt_traverse_group.h — closest-hit traversal with G lanes per ray (G = 4 or 8): the experiment SURVEY §7 step 5 and
// VERDICT r2 item 3 ask for ("8-lanes-per-ray node test vs 1-lane-per-ray"; the reference's own node test is 8 wide,
// accel/bvh.rs:171-223).  Same functions replaced as pt_traverse.h (Tlas::hit, Bvh8::traverse_closest, Instance::hit,
// Blas::hit, TriangleMesh::hit, intersect_triangle), same results bit for bit; measured against the per-ray kernels
// in DESIGN.md §4 "G lanes per ray" — and not used by any product path unless PTX_GROUP_LANES asks for it.
//
// A wave holds 64 / G rays.  The G lanes of a group keep identical copies of their ray's state; what they do in
// parallel:
//   NODE  lane k tests child k of the record (a record of the G-wide collapsed tree is read as ONE contiguous
//         32 * nchild-byte access by the group); the hit children are pushed in reverse list order — a child's
//         position is the number of hit children behind it, from one ballot — so they pop in list order: the
//         reference's order (pt_types.h says why the collapsed tree cannot change a hit);
//   LEAF  lane k tests triangle k of the leaf (intersect_triangle once per lane); the winner is the smallest t below
//         closest_t, the lowest index among equal t — what the reference's in-order loop with its strict `<` keeps;
//   ENTER / EXIT (Instance::hit head and tail) are per-ray arithmetic: every lane of the group computes the same values.
// One stack per group in LDS (all lanes read the same address: a broadcast), no HBM spill: a ray that needs more than
// GROUP_DEPTH entries raises the overflow flag, as the reference panics beyond 64 per level.
#pragma once
#include "pt_traverse.h"

namespace ptd {

template <int G> struct GroupCfg;
template <> struct GroupCfg<8> { static constexpr int DEPTH = 96; };
template <> struct GroupCfg<4> { static constexpr int DEPTH = 64; };

template <int G>
struct GroupStack {
    static constexpr int NG = 64 / G;
    static constexpr int DEPTH = GroupCfg<G>::DEPTH;
};
// 8 groups x 96 entries and 16 x 64: sized for the larger (8 KiB per wave)
static __shared__ uint2 s_gstack[16 * 64];

template <int G>
PT_DEV uint32_t group_bits(unsigned long long ballot, uint32_t g) {
    return (uint32_t)(ballot >> (g * (uint32_t)G)) & ((1u << G) - 1u);
}

// `closest`: Scene::hit for queue entries [0, n).  Src / Sink as in trace_queue.
template <int G, class Src, class Sink>
PT_DEV void trace_groups(const SceneDev& S, uint32_t* work, const uint32_t n, const Src& src, const Sink& sink, uint32_t* err) {
    constexpr int NG = 64 / G;
    constexpr int DEPTH = GroupCfg<G>::DEPTH;
    constexpr uint32_t REFILL_GROUPS = NG / 4 > 0 ? NG / 4 : 1;
    if ((unsigned long long)blockIdx.x * (unsigned long long)(NG / 4 + 1) >= (unsigned long long)n + 128ull) return;
    const uint32_t lane = threadIdx.x & 63u, g = lane / (uint32_t)G, k = lane % (uint32_t)G;
    uint2* const stk = &s_gstack[g * (uint32_t)DEPTH];
    const float tmin = RAY_EPSILON;

    // group-uniform state (every lane of a group holds the same values)
    V3 o = v3(0.0f, 0.0f, 0.0f), d = o, inv = o;
    float ct = 0.0f;
    int sp = 0;
    uint32_t act = ACT_IDLE, pw = 0, tri_cnt = 0, tri_base = 0, m_tri = PTX_MISS, my = 0;
    float m_u = 0.0f, m_v = 0.0f;
    bool in_mesh = false, found = false, exact = false;
    bool exhausted = false;

    for (;;) {
        // ---- refill: idle groups claim queue entries with one atomic per wave ---------------------------------------
        const unsigned long long idle_l = __ballot(act == ACT_IDLE && k == 0u);      // one bit per idle group (its leader lane)
        const uint32_t n_idle = (uint32_t)__popcll(idle_l);
        if (!exhausted && n_idle >= REFILL_GROUPS) {
            const int leader = __ffsll((long long)idle_l) - 1;
            uint32_t basei = 0;
            if ((int)lane == leader) basei = atomicAdd(work, n_idle);
            basei = (uint32_t)__builtin_amdgcn_readlane((int)basei, leader);
            if (act == ACT_IDLE) {
                // rank of this group among the idle ones: idle leaders below this group's leader lane
                const unsigned long long below = idle_l & ((1ull << (g * (uint32_t)G)) - 1ull);
                const uint32_t i = basei + (uint32_t)__popcll(below);
                if (i < n) {
                    float tmax;
                    my = src.key(i);
                    src.fetch(my, &o, &d, &tmax);
                    inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);          // Ray::new (ray.rs:13-21)
                    exact = ray_needs_exact(o, inv);
                    ct = tmax;
                    sp = 0; tri_base = 0; m_tri = PTX_MISS; m_u = 0.0f; m_v = 0.0f;
                    in_mesh = false; found = false; tri_cnt = 0;
                    pw = exact ? S.tlas_root_word : S.tlas_root_word_ray;
                    act = (S.n_tlas_nodes == 0 || 0.0f >= ct) ? ACT_DONE : ACT_NODE;     // accel/bvh.rs:322-329
                }
            }
            if (basei + n_idle >= n) exhausted = true;
        }

        // ---- pop (accel/bvh.rs:322-329): every lane of the group reads the same entry -------------------------------
        if (act == ACT_POP) {
            for (;;) {
                if (sp == 0) { act = ACT_DONE; break; }
                --sp;
                const uint2 e = stk[sp];
                if (__uint_as_float(e.y) >= ct) continue;
                pw = e.x;
                if (in_mesh) {
                    if (e.y == NEG_INF_BITS) act = ACT_EXIT;
                    else if (e.x & LEAF_FLAG) {
                        act = ACT_TRI;
                        pw = tri_base + (e.x & PRIM_START_MASK);
                        tri_cnt = ((e.x >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1u;
                    } else act = ACT_NODE;
                } else act = (e.x & LEAF_FLAG) ? ACT_ENTER : ACT_NODE;
                break;
            }
        }
        if (act == ACT_DONE) {
            if (k == 0u) sink.done(my, found);
            act = ACT_IDLE;
        }
        const unsigned long long any_work = __ballot(act != ACT_IDLE);
        if (any_work == 0ull) {
            if (exhausted) break;
            continue;
        }

        // ---- EXIT: tail of Instance::hit (accel/mod.rs:128-137) ------------------------------------------------------
        if (__ballot(act == ACT_EXIT) != 0ull) {
            if (act == ACT_EXIT) {
                const uint32_t cur_inst = pw & MARK_INST_MASK;
                const uint2 e1 = stk[sp - 1], e0 = stk[sp - 2];
                sp -= 2;
                V3 wo, wd;
                float unused_tmax;
                src.fetch(my, &wo, &wd, &unused_tmax);
                float ct_world = __uint_as_float(e1.y);
                if (m_tri != PTX_MISS) {
                    const float4* p = reinterpret_cast<const float4*>(&S.inst[cur_inst]);
                    const float4 ma = p[5], mb = p[6], mc = p[7];
                    const V3 local_hit_point = o + ct * d;
                    const V3 world_hit_point = xf_point(ma, mb, mc, local_hit_point);
                    const float t_world = dot(world_hit_point - wo, wd) / dot(wd, wd);
                    if (t_world < ct_world) {                               // traverse_closest accept (accel/bvh.rs:353-357)
                        ct_world = t_world;
                        if (k == 0u) sink.hit(my, cur_inst, m_tri, t_world, m_u, m_v);
                        found = true;
                    }
                }
                m_tri = PTX_MISS;
                in_mesh = false;
                o = wo; d = wd;
                inv = v3(__uint_as_float(e0.x), __uint_as_float(e0.y), __uint_as_float(e1.x));
                exact = (pw & MARK_EXACT) != 0u;
                ct = ct_world;
                act = ACT_POP;
            }
        }

        // ---- ENTER: Instance::hit head + the Blas-level box (accel/mod.rs:123-127, 59-65) ----------------------------
        if (__ballot(act == ACT_ENTER) != 0ull) {
            if (act == ACT_ENTER) {
                const uint32_t start = pw & PRIM_START_MASK;
                const uint32_t count = ((pw >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1u;
                const float4* p = reinterpret_cast<const float4*>(&S.inst_tlas[start]);
                const float4 ia = p[0], ib = p[1], ic = p[2], b0 = p[3], b1 = p[4];
                const uint32_t inst_idx = S.tlas_prims[start];
                if (sp + 4 > DEPTH) { if (k == 0u) atomicOr(err, 1u); found = false; act = ACT_DONE; }
                else {
                    if (count > 1u) { stk[sp] = make_uint2(LEAF_FLAG | ((count - 2u) << PRIM_COUNT_SHIFT) | (start + 1u), NEG_INF_BITS); ++sp; }
                    const V3 lo = xf_point(ia, ib, ic, o);
                    const V3 ld = xf_vector(ia, ib, ic, d);
                    const V3 linv = v3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
                    const bool lexact = ray_needs_exact(lo, linv);
                    float tb_enter;
                    const bool bhit = box_test<true>(b0, make_float4(b1.x, b1.y, 0.0f, 0.0f), lo, linv, tmin, PT_INF, &tb_enter);
                    const bool uncert = (__float_as_uint(b1.z) & INST_UNCERT) != 0u;
                    const float bound = (exact || lexact || uncert) ? PT_INF : forwarded_bound(S, o, d, ct);
                    if (!bhit || tb_enter >= bound) act = ACT_POP;
                    else {
                        // (every lane of the group writes the same words to the same addresses)
                        stk[sp] = make_uint2(__float_as_uint(inv.x), __float_as_uint(inv.y));
                        stk[sp + 1] = make_uint2(__float_as_uint(inv.z), __float_as_uint(ct));
                        stk[sp + 2] = make_uint2(inst_idx | (exact ? MARK_EXACT : 0u), NEG_INF_BITS);
                        sp += 3;
                        in_mesh = true;
                        ct = bound;
                        o = lo; d = ld; inv = linv;
                        exact = lexact;
                        tri_base = __float_as_uint(b1.w);
                        m_tri = PTX_MISS;
                        pw = exact ? S.inst_root_faithful[start] : (__float_as_uint(b1.z) & ~INST_UNCERT);
                        act = ACT_NODE;
                    }
                }
            }
        }

        // ---- LEAF: lane k tests triangle k (shape/triangle_mesh.rs:63-100 under accel/bvh.rs:348-357) ----------------
        if (__ballot(act == ACT_TRI) != 0ull) {
            if (act == ACT_TRI) {
                float best_t = PT_INF, best_u = 0.0f, best_v = 0.0f;
                uint32_t best_tri = PTX_MISS, best_k = 0xFFFFu;
                if (k < tri_cnt) {
                    const float4* p = reinterpret_cast<const float4*>(&S.tris[pw + k]);
                    const float4 ta = p[0], tb = p[1], tc = p[2];
                    const V3 v0 = v3(ta.x, ta.y, ta.z), edge1 = v3(ta.w, tb.x, tb.y), edge2 = v3(tb.z, tb.w, tc.x);
                    const V3 h = cross(d, edge2);
                    const float a = dot(edge1, h);
                    const float f = 1.0f / a;
                    const V3 s = o - v0;
                    const float u = f * dot(s, h);
                    const V3 q = cross(s, edge1);
                    const float v = f * dot(d, q);
                    const float t = f * dot(edge2, q);
                    const bool hit = !(fabsf(a) < 1e-8f) && (u >= 0.0f && u <= 1.0f) && !(v < 0.0f || u + v > 1.0f) && (t >= tmin && t <= PT_INF);
                    if (hit && t < ct) { best_t = t; best_u = u; best_v = v; best_tri = __float_as_uint(tc.y); best_k = k; }
                }
                // the group's winner: smallest t, lowest index among equals (a butterfly over the G lanes; lanes of other
                // groups and of other actions take part in the shuffles with their own harmless values)
#pragma unroll
                for (int off = 1; off < G; off <<= 1) {
                    const float ot = __shfl_xor(best_t, off);
                    const uint32_t ok = (uint32_t)__shfl_xor((int)best_k, off);
                    const float ou = __shfl_xor(best_u, off), ov = __shfl_xor(best_v, off);
                    const uint32_t otri = (uint32_t)__shfl_xor((int)best_tri, off);
                    const bool take = ot < best_t || (ot == best_t && ok < best_k);
                    if (take) { best_t = ot; best_k = ok; best_u = ou; best_v = ov; best_tri = otri; }
                }
                if (best_tri != PTX_MISS) { ct = best_t; m_tri = best_tri; m_u = best_u; m_v = best_v; }
                const uint32_t done = tri_cnt < (uint32_t)G ? tri_cnt : (uint32_t)G;
                pw += done; tri_cnt -= done;
                if (tri_cnt == 0u) act = ACT_POP;
            }
        }

        // ---- NODE: lane k tests child k; hit children pushed in reverse list order -----------------------------------
        if (__ballot(act == ACT_NODE) != 0ull) {
            const bool is_node = act == ACT_NODE;
            const uint32_t nchild = ((pw >> NODE_NCHILD_SHIFT) & NODE_NCHILD_MASK) + 1u;
            const float4* p = reinterpret_cast<const float4*>(&S.nodes[pw & NODE_INDEX_MASK]);
            if (is_node && sp + (int)nchild > DEPTH) { if (k == 0u) atomicOr(err, 1u); found = false; act = ACT_DONE; }
            const bool go = is_node && act == ACT_NODE;
            const bool wave_exact = __ballot(go && exact) != 0ull;
            // records wider than G (host trees walked by exact rays; G = 4 on an 8-wide host tree): last chunk first
            const uint32_t chunks = go ? (nchild + (uint32_t)G - 1u) / (uint32_t)G : 0u;
            uint32_t max_chunks = chunks;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const uint32_t oc = (uint32_t)__shfl_xor((int)max_chunks, off); max_chunks = oc > max_chunks ? oc : max_chunks; }
            for (uint32_t c = max_chunks; c-- > 0u;) {
                const uint32_t child = c * (uint32_t)G + k;
                const bool mine = go && c < chunks && child < nchild;
                float te = 0.0f;
                bool h = false;
                uint32_t w = 0;
                if (mine) {
                    const float4 lo4 = p[2u * child], hi4 = p[2u * child + 1u];
                    h = wave_exact ? box_test<true>(lo4, hi4, o, inv, tmin, ct, &te) : box_test<false>(lo4, hi4, o, inv, tmin, ct, &te);
                    w = __float_as_uint(hi4.z);
                }
                const uint32_t hits = group_bits<G>(__ballot(h), g);
                if (h) stk[sp + (int)__popc(hits >> (k + 1u))] = make_uint2(w, __float_as_uint(te));
                if (go && c < chunks) sp += (int)__popc(hits);
            }
            if (go) act = ACT_POP;
        }
    }
}

}  // namespace ptd
