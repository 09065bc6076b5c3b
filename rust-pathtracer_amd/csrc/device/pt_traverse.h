// pt_traverse.h — closest-hit and any-hit traversal of TLAS -> Blas -> mesh BVH.
//
// Replaces, on the device, `Scene::hit` / `Scene::hit_any` (scene.rs:44-50):
//   Tlas::hit/hit_any            accel/mod.rs:194-210
//   Bvh8::traverse_closest/_any  accel/bvh.rs:308-415   (slot order, LIFO, strict <)
//   Bvh8Node::intersect          accel/bvh.rs:171-223   (x86 min_ps/max_ps semantics)
//   Instance::hit/hit_any        accel/mod.rs:122-146
//   Blas::hit/hit_any            accel/mod.rs:59-72     (one item => one box test)
//   TriangleMesh::hit/hit_any    shape/triangle_mesh.rs:192-207
//   intersect_triangle           shape/triangle_mesh.rs:63-100
//
// One lane = one ray.  The reference nests three traversals (closures inside
// closures); on a 64-wide wave that nesting would idle every lane that is not
// inside an instance, so the three levels are flattened into ONE state machine
// whose step is "visit one node record", with one unified stack: TLAS entries
// stay at the bottom while a lane is inside an instance, mesh entries sit above
// `base`.  When the mesh level drains, the lane finishes `Instance::hit` and
// resumes the TLAS record it came from at the slot/primitive after the instance
// (the box tests of the remaining slots are redone with the current closest_t;
// a slot that passed with the older, larger closest_t but fails now would have
// been skipped by the reference's `dist >= closest_t` check anyway, so the
// visit order and the accepted hits are unchanged).
//
// A wave is persistent: lanes whose ray has finished pull the next ray from the
// queue (one wave-aggregated atomic) while their neighbours keep traversing, so
// long rays do not hold 63 idle lanes hostage.
//
// Node test: `a < b ? a : b` / `a > b ? a : b` reproduce _mm256_min_ps/_max_ps
// including their NaN rule (second operand).  NaNs can only come from 0 * inf,
// i.e. from a ray with a zero direction component or a non-finite origin; when
// no lane of the wave carries such a ray the test runs on v_min_f32/v_max_f32
// (v_min3/v_max3), which give the same values for non-NaN inputs at half the
// VALU cost.
//
// Stack: the first LDS_STACK entries of each lane live in LDS (entry-major,
// lane-minor: every lane owns one bank column, so divergent stack pointers
// never conflict); deeper entries spill to a per-lane strip in HBM.  The
// reference's arrays hold 64 entries per level and panic beyond that; here the
// same condition raises an error flag.
#pragma once
#include "pt_types.h"

namespace ptd {

constexpr int LDS_STACK = 16;
constexpr int LEVEL_STACK_LIMIT = 64;                 // accel/bvh.rs:320,379
constexpr int OVF_STACK = 2 * LEVEL_STACK_LIMIT - LDS_STACK;
constexpr int TRACE_BLOCK = 64;                       // one wave per block: traversal needs no barrier
constexpr uint32_t REFILL_MIN_IDLE = 16;              // refill a wave once this many lanes are idle

// File-scope LDS array (address space known at every use: plain ds_read/ds_write,
// no flat pointers into LDS).  Every kernel that traverses gets its own copy.
static __shared__ uint2 s_stack[LDS_STACK * TRACE_BLOCK];

struct StackRef {
    uint2* ovf;            // &ovf[0][global_lane], entry stride = ovf_stride
    uint32_t ovf_stride;
};
PT_DEV void st_write(const StackRef& s, int i, uint32_t idx, float d) {
    const uint2 v = make_uint2(idx, __float_as_uint(d));
    if (i < LDS_STACK) s_stack[i * TRACE_BLOCK + (int)threadIdx.x] = v;
    else s.ovf[(size_t)(i - LDS_STACK) * s.ovf_stride] = v;
}
PT_DEV uint2 st_read(const StackRef& s, int i) {
    if (i < LDS_STACK) return s_stack[i * TRACE_BLOCK + (int)threadIdx.x];
    return s.ovf[(size_t)(i - LDS_STACK) * s.ovf_stride];
}

struct Hit { uint32_t inst, tri; float t, u, v; };

struct LaneCnt { uint32_t node_visits, node_children, prim_reads, tri_tests, inst_entries, inst_hits; };

PT_DEV V3 xf_point(const float4 a, const float4 b, const float4 c, V3 p) {
    // Mat4::transform_point3: ((x_axis*x + y_axis*y) + z_axis*z) + w_axis, packed columns
    return v3(((a.x * p.x + a.w * p.y) + b.z * p.z) + c.y,
              ((a.y * p.x + b.x * p.y) + b.w * p.z) + c.z,
              ((a.z * p.x + b.y * p.y) + c.x * p.z) + c.w);
}
PT_DEV V3 xf_vector(const float4 a, const float4 b, const float4 c, V3 p) {
    return v3((a.x * p.x + a.w * p.y) + b.z * p.z,
              (a.y * p.x + b.x * p.y) + b.w * p.z,
              (a.z * p.x + b.y * p.y) + c.x * p.z);
}

// Bvh8Node::intersect for one child (accel/bvh.rs:198-215).
// EXACT: compare-select with the x86 NaN rule.  !EXACT: hardware min/max, valid when no NaN can occur.
template <bool EXACT>
PT_DEV bool box_test(const float4 lo4, const float4 hi4, V3 o, V3 inv, float tmin, float tmax, float* t_enter_out) {
    const float t1x = (lo4.x - o.x) * inv.x, t2x = (lo4.w - o.x) * inv.x;
    const float t1y = (lo4.y - o.y) * inv.y, t2y = (hi4.x - o.y) * inv.y;
    const float t1z = (lo4.z - o.z) * inv.z, t2z = (hi4.y - o.z) * inv.z;
    float t_enter, t_exit;
    if (EXACT) {
        const float nx = x86_min(t1x, t2x), fx = x86_max(t1x, t2x);
        const float ny = x86_min(t1y, t2y), fy = x86_max(t1y, t2y);
        const float nz = x86_min(t1z, t2z), fz = x86_max(t1z, t2z);
        t_enter = x86_max(tmin, x86_max(nx, x86_max(ny, nz)));
        t_exit = x86_min(tmax, x86_min(fx, x86_min(fy, fz)));
    } else {
        const float nx = __builtin_fminf(t1x, t2x), fx = __builtin_fmaxf(t1x, t2x);
        const float ny = __builtin_fminf(t1y, t2y), fy = __builtin_fmaxf(t1y, t2y);
        const float nz = __builtin_fminf(t1z, t2z), fz = __builtin_fmaxf(t1z, t2z);
        t_enter = __builtin_fmaxf(tmin, __builtin_fmaxf(nx, __builtin_fmaxf(ny, nz)));
        t_exit = __builtin_fminf(tmax, __builtin_fminf(fx, __builtin_fminf(fy, fz)));
    }
    *t_enter_out = t_enter;
    return t_enter <= t_exit;   // _CMP_LE_OQ: false on NaN
}

// a ray whose node tests can produce NaN (0 * inf): zero direction component or non-finite origin
PT_DEV bool ray_needs_exact(V3 o, V3 inv) {
    return !(finitef(inv.x) && finitef(inv.y) && finitef(inv.z) && finitef(o.x) && finitef(o.y) && finitef(o.z));
}

// Per-lane traversal state.  `begin` loads a ray, `step` visits one node record (or leaves an instance),
// `done` says the ray has finished; ANY rays also finish on the first triangle hit.
template <bool ANY, bool CNT>
struct Traverser {
    V3 wo, wd;                 // world ray
    V3 o, d, inv;              // ray of the level being traversed
    float ct;                  // closest_t of that level (ANY: its ray.t_max)
    float ct_world;            // world closest_t, parked while inside an instance
    int sp, base;
    uint32_t node_base, tri_base, cur_inst;
    uint32_t c_rec, c_kj;      // TLAS continuation: record, (slot << 8 | next primitive of that slot)
    uint32_t m_tri;            // best hit inside the current instance
    float m_t, m_u, m_v;
    Hit best;
    bool in_mesh, found, exact, finished;

    PT_DEV void begin(const SceneDev& S, const StackRef& st, V3 origin, V3 direction, float tmax) {
        wo = origin; wd = direction;
        o = origin; d = direction;
        inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);          // Ray::new (ray.rs:13-21)
        exact = ray_needs_exact(o, inv);
        ct = tmax; ct_world = tmax;
        sp = 0; base = 0; node_base = 0; tri_base = 0; cur_inst = 0; c_rec = 0; c_kj = 0;
        m_tri = PTX_MISS; m_t = 0.0f; m_u = 0.0f; m_v = 0.0f;
        best.inst = PTX_MISS; best.tri = PTX_MISS; best.t = PT_INF; best.u = 0.0f; best.v = 0.0f;
        in_mesh = false; found = false;
        finished = S.n_tlas_nodes == 0;                         // Bvh8::traverse_*: empty => None/false
        if (!finished) st_write(st, sp++, 0u, 0.0f);            // stack[0] = (root, 0.0)
    }

    // wave_exact: some lane of the wave needs the NaN-exact node test (wave-uniform)
    PT_DEV void step(const SceneDev& S, const StackRef& st, const bool wave_exact, LaneCnt& lc, uint32_t& err) {
        const float tmin = RAY_EPSILON;
        uint32_t rec = 0, j0 = 0;
        int k0 = 0;
        bool fresh = false, have = false;
        if (sp == base) {
            // ---- mesh level drained (a lane with sp == base at TLAS level is already finished):
            //      tail of Instance::hit (accel/mod.rs:128-137)
            if (!ANY && m_tri != PTX_MISS) {
                if (CNT) lc.inst_hits++;
                const InstRec& ir = S.inst[cur_inst];
                const float4 ma = ir.m[5], mb = ir.m[6], mc = ir.m[7];
                const V3 local_hit_point = o + m_t * d;
                const V3 world_hit_point = xf_point(ma, mb, mc, local_hit_point);
                const float t_world = dot(world_hit_point - wo, wd) / dot(wd, wd);
                if (t_world < ct_world) {                       // traverse_closest accept (accel/bvh.rs:353-357)
                    ct_world = t_world;
                    best.inst = cur_inst; best.tri = m_tri; best.t = t_world; best.u = m_u; best.v = m_v;
                    found = true;
                }
            }
            in_mesh = false;
            o = wo; d = wd;
            inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            exact = ray_needs_exact(o, inv);
            ct = ct_world;
            node_base = 0; base = 0;
            rec = c_rec; k0 = (int)(c_kj >> 8); j0 = c_kj & 0xFFu;
            have = true;
        } else {
            // pop until an entry survives the distance cull (accel/bvh.rs:324-329)
            while (sp > base) {
                --sp;
                const uint2 e = st_read(st, sp);
                if (ANY || !(__uint_as_float(e.y) >= ct)) { rec = e.x; fresh = true; have = true; break; }
            }
        }

        while (have) {
            const float4* p = reinterpret_cast<const float4*>(&S.nodes[node_base + rec]);
            float4 c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = p[i];
            const uint32_t meta = __float_as_uint(c[1].w);
            const int nchild = (int)(meta & 7u);
            const bool more = (meta & 8u) != 0;
            if (CNT && fresh) { lc.node_visits++; lc.node_children += (meta >> 8) & 15u; }
            // node.intersect(ray, closest_t): all children against the closest_t of this visit
            float te[4];
            uint32_t hits = 0;
            if (wave_exact) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { te[k] = 0.0f; if (k < nchild && box_test<true>(c[2 * k], c[2 * k + 1], o, inv, tmin, ct, &te[k])) hits |= 1u << k; }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) { te[k] = 0.0f; if (k < nchild && box_test<false>(c[2 * k], c[2 * k + 1], o, inv, tmin, ct, &te[k])) hits |= 1u << k; }
            }
            hits &= (1u << nchild) - 1u;                        // & valid_mask
            hits &= ~((1u << k0) - 1u);                         // slots before the continuation point are done
            if (j0 > 0) hits |= 1u << k0;                       // back from an instance: this leaf continues
            bool entered = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!((hits >> k) & 1u) || entered) continue;
                const uint32_t word = __float_as_uint(c[2 * k + 1].z);
                uint32_t jstart = 0;
                if (k == k0 && j0 > 0) jstart = j0;
                else if (!ANY && te[k] >= ct) continue;         // accel/bvh.rs:344-347 (closest_t may have shrunk)
                if (!(word & LEAF_FLAG)) {
                    if (sp - base >= LEVEL_STACK_LIMIT) { err = 1u; finished = true; found = false; return; }
                    st_write(st, sp++, word, te[k]);            // accel/bvh.rs:360-363
                    continue;
                }
                const uint32_t start = word & PRIM_START_MASK;
                const uint32_t count = ((word >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1u;
                if (in_mesh) {
                    // ---- leaf of a mesh BVH: intersect_triangle on each primitive, in order
                    for (uint32_t j = 0; j < count; ++j) {
                        const TriRec& tr = S.tris[tri_base + start + j];
                        const float4 ta = tr.a, tb = tr.b, tc = tr.c;
                        if (CNT) { lc.prim_reads++; lc.tri_tests++; }
                        const V3 v0 = v3(ta.x, ta.y, ta.z), edge1 = v3(ta.w, tb.x, tb.y), edge2 = v3(tb.z, tb.w, tc.x);
                        const V3 h = cross(d, edge2);
                        const float a = dot(edge1, h);
                        if (fabsf(a) < 1e-8f) continue;
                        const float f = 1.0f / a;
                        const V3 s = o - v0;
                        const float u = f * dot(s, h);
                        if (!(u >= 0.0f && u <= 1.0f)) continue;
                        const V3 q = cross(s, edge1);
                        const float v = f * dot(d, q);
                        if (v < 0.0f || u + v > 1.0f) continue;
                        const float t = f * dot(edge2, q);
                        if (!(t >= tmin && t <= PT_INF)) continue;  // local ray: t_max = +inf
                        if (ANY) { found = true; finished = true; return; }
                        if (t < ct) { ct = t; m_tri = __float_as_uint(tc.y); m_t = t; m_u = u; m_v = v; }
                    }
                } else {
                    // ---- leaf of the TLAS: Instance::hit / hit_any on each primitive, in order
                    for (uint32_t j = jstart; j < count; ++j) {
                        const uint32_t inst_idx = S.tlas_prims[start + j];
                        const InstRec& ir = S.inst[inst_idx];
                        const float4 ia = ir.m[0], ib = ir.m[1], ic = ir.m[2], b0 = ir.m[3], b1 = ir.m[4];
                        if (CNT) { lc.prim_reads++; lc.inst_entries++; lc.node_visits++; lc.node_children++; }
                        const V3 lo = xf_point(ia, ib, ic, wo);
                        const V3 ld = xf_vector(ia, ib, ic, wd);
                        const V3 linv = v3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
                        // Blas level: one node, slot 0 = leaf(0,1) boxed by the mesh bounds; t_max = +inf
                        float tb_enter;
                        const bool bhit = box_test<true>(b0, make_float4(b1.x, b1.y, 0.0f, 0.0f), lo, linv, tmin, PT_INF, &tb_enter);
                        if (!bhit) continue;
                        if (!ANY && tb_enter >= PT_INF) continue;
                        if (CNT) lc.prim_reads++;   // the Blas leaf's own primitive index
                        // descend into the mesh BVH (TriangleMesh::hit: closest_t = local t_max = +inf)
                        c_rec = rec; c_kj = ((uint32_t)k << 8) | (j + 1u);
                        in_mesh = true; entered = true;
                        cur_inst = inst_idx;
                        ct_world = ct; ct = PT_INF;
                        o = lo; d = ld; inv = linv;
                        exact = ray_needs_exact(o, inv);
                        node_base = __float_as_uint(b1.z); tri_base = __float_as_uint(b1.w);
                        m_tri = PTX_MISS;
                        base = sp;
                        st_write(st, sp++, 0u, 0.0f);
                        break;
                    }
                }
            }
            if (entered || !more) break;
            fresh = false; k0 = 0; j0 = 0; rec += 1u;           // a node with 5-8 children continues in the next record
        }
        if (sp == base && !in_mesh) finished = true;
    }
};

// Wave-persistent driver.  `Src::fetch(i, o, d, tmax)` loads queue entry i; `Sink::store(i, trav)` consumes a
// finished ray.  `work` is the queue's shared cursor, `n` its length.
template <bool ANY, bool CNT, class Src, class Sink>
PT_DEV void trace_queue(const SceneDev& S, const StackRef& st, uint32_t* work, const uint32_t n, Src& src, Sink& sink,
                        LaneCnt& lc, uint32_t& nrays, uint32_t& err) {
    Traverser<ANY, CNT> tv;
    tv.finished = true;
    tv.exact = false;
    bool active = false;
    uint32_t my = 0;
    bool exhausted = false;   // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (!exhausted && (uint32_t)__popcll(idle) >= REFILL_MIN_IDLE) {
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const int leader = __ffsll((long long)idle) - 1;
            uint32_t basei = 0;
            if ((int)lane == leader) basei = atomicAdd(work, n_idle);
            basei = __shfl(basei, leader);
            if (!active) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t i = basei + rank;
                if (i < n) {
                    V3 o, d;
                    float tmax;
                    src.fetch(i, &o, &d, &tmax);
                    tv.begin(S, st, o, d, tmax);
                    my = i;
                    active = true;
                    nrays++;
                }
            }
            if (basei + n_idle >= n) exhausted = true;
        }
        // retire rays that are done (an empty TLAS finishes inside begin)
        if (active && tv.finished) { sink.store(my, tv); active = false; }
        const unsigned long long live = __ballot(active);
        if (live == 0ull) {
            if (exhausted) break;
            continue;
        }
        const bool wave_exact = __ballot(active && tv.exact) != 0ull;
        if (active) {
            tv.step(S, st, wave_exact, lc, err);
            if (tv.finished) { sink.store(my, tv); active = false; }
        }
    }
}

}  // namespace ptd
