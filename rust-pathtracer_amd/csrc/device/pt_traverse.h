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
// inside an instance, so the three levels are flattened into ONE loop over
// node records with one unified stack: TLAS entries stay at the bottom while
// a lane is inside an instance, mesh entries sit above `base`.  When the mesh
// level drains, the lane finishes `Instance::hit` and resumes the TLAS record
// it came from at the slot/primitive after the instance (the box tests of the
// remaining slots are redone with the current closest_t; a slot that passed
// with the older, larger closest_t but fails now would have been skipped by
// the reference's `dist >= closest_t` check anyway, so the visit order and the
// accepted hits are unchanged).
//
// Stack: the first LDS_STACK entries of each lane live in LDS
// (entry-major, lane-minor: every lane owns one bank column, so divergent
// stack pointers never conflict); deeper entries spill to a per-lane strip in
// HBM.  The reference's arrays hold 64 entries per level and panic beyond
// that; here the same condition raises an error flag.
#pragma once
#include "pt_types.h"

namespace ptd {

constexpr int LDS_STACK = 16;
constexpr int LEVEL_STACK_LIMIT = 64;                 // accel/bvh.rs:320,379
constexpr int OVF_STACK = 2 * LEVEL_STACK_LIMIT - LDS_STACK;

constexpr int TRACE_BLOCK = 64;   // one wave per block: traversal needs no barrier

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

// Bvh8Node::intersect for one child (accel/bvh.rs:198-215)
PT_DEV bool box_test(const float4 lo4, const float4 hi4, V3 o, V3 inv, float tmin, float tmax, float* t_enter_out) {
    const float t1x = (lo4.x - o.x) * inv.x, t2x = (lo4.w - o.x) * inv.x;
    const float t1y = (lo4.y - o.y) * inv.y, t2y = (hi4.x - o.y) * inv.y;
    const float t1z = (lo4.z - o.z) * inv.z, t2z = (hi4.y - o.z) * inv.z;
    const float nx = x86_min(t1x, t2x), fx = x86_max(t1x, t2x);
    const float ny = x86_min(t1y, t2y), fy = x86_max(t1y, t2y);
    const float nz = x86_min(t1z, t2z), fz = x86_max(t1z, t2z);
    const float t_enter = x86_max(tmin, x86_max(nx, x86_max(ny, nz)));
    const float t_exit = x86_min(tmax, x86_min(fx, x86_min(fy, fz)));
    *t_enter_out = t_enter;
    return t_enter <= t_exit;   // _CMP_LE_OQ: false on NaN
}

// ANY:  returns true when some triangle is hit (Scene::hit_any)
// !ANY: returns true and fills `out` when a closest hit exists (Scene::hit)
template <bool ANY, bool CNT>
PT_DEV bool trace_ray(const SceneDev& S, const V3 wo, const V3 wd, const float tmax, const StackRef& st,
                      Hit& out, LaneCnt& lc, uint32_t& err) {
    if (S.n_tlas_nodes == 0) return false;   // Bvh8::traverse_*: empty => None/false

    V3 o = wo, d = wd;
    V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);        // Ray::new (ray.rs:13-21)
    const float tmin = RAY_EPSILON;
    float ct = tmax;           // closest_t of the level being traversed (ANY: the level's ray.t_max)
    float ct_world = tmax;     // world closest_t, parked while inside an instance
    bool in_mesh = false;
    int sp = 0, base = 0;
    uint32_t node_base = 0, tri_base = 0, cur_inst = 0;
    uint32_t c_rec = 0, c_j = 0;   // TLAS continuation: record, slot, next primitive of that slot
    int c_k = 0;
    uint32_t m_tri = PTX_MISS;     // best hit inside the current instance
    float m_t = 0.0f, m_u = 0.0f, m_v = 0.0f;
    bool found = false, resume = false;

    st_write(st, sp++, 0u, 0.0f);  // stack[0] = (root, 0.0)

    for (;;) {
        uint32_t rec, j0;
        int k0;
        bool fresh;
        if (resume) {
            rec = c_rec; k0 = c_k; j0 = c_j; resume = false; fresh = false;
        } else {
            if (sp == base) {
                if (!in_mesh) break;
                // ---- mesh level drained: tail of Instance::hit (accel/mod.rs:128-137)
                if (!ANY && m_tri != PTX_MISS) {
                    if (CNT) lc.inst_hits++;
                    const InstRec& ir = S.inst[cur_inst];
                    const float4 ma = ir.m[5], mb = ir.m[6], mc = ir.m[7];
                    const V3 local_hit_point = o + m_t * d;
                    const V3 world_hit_point = xf_point(ma, mb, mc, local_hit_point);
                    const float t_world = dot(world_hit_point - wo, wd) / dot(wd, wd);
                    if (t_world < ct_world) {       // traverse_closest accept (accel/bvh.rs:353-357)
                        ct_world = t_world;
                        out.inst = cur_inst; out.tri = m_tri; out.t = t_world; out.u = m_u; out.v = m_v;
                        found = true;
                    }
                }
                in_mesh = false;
                o = wo; d = wd;
                inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                ct = ct_world;
                node_base = 0; base = 0;
                resume = true;
                continue;
            }
            --sp;
            const uint2 e = st_read(st, sp);
            if (!ANY && __uint_as_float(e.y) >= ct) continue;      // accel/bvh.rs:327-329
            rec = e.x; k0 = 0; j0 = 0; fresh = true;
        }

        bool more;
        do {
            const float4* p = reinterpret_cast<const float4*>(&S.nodes[node_base + rec]);
            float4 c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = p[i];
            const uint32_t meta = __float_as_uint(c[1].w);
            const int nchild = (int)(meta & 7u);
            more = (meta & 8u) != 0;
            if (CNT && fresh) { lc.node_visits++; lc.node_children += (meta >> 8) & 15u; }
            const float ct0 = ct;   // node.intersect(ray, closest_t): t_max of this visit
            bool entered = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < k0 || k >= nchild || entered) continue;
                const float4 lo4 = c[2 * k], hi4 = c[2 * k + 1];
                const uint32_t word = __float_as_uint(hi4.z);
                float t_enter = 0.0f;
                uint32_t jstart = 0;
                if (k == k0 && j0 > 0) {
                    jstart = j0;                    // back from an instance: continue this leaf's primitives
                } else {
                    const bool hit = box_test(lo4, hi4, o, inv, tmin, ct0, &t_enter);
                    if (!hit) continue;
                    if (!ANY && t_enter >= ct) continue;            // accel/bvh.rs:344-347
                }
                if (!(word & LEAF_FLAG)) {
                    if (sp - base >= LEVEL_STACK_LIMIT) { err = 1u; return false; }
                    st_write(st, sp++, word, t_enter);              // accel/bvh.rs:360-363
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
                        if (ANY) return true;
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
                        float te;
                        const float4 blo = b0, bhi = make_float4(b1.x, b1.y, 0.0f, 0.0f);
                        const bool bhit = box_test(blo, bhi, lo, linv, tmin, PT_INF, &te);
                        if (!bhit) continue;
                        if (!ANY && te >= PT_INF) continue;
                        if (CNT) lc.prim_reads++;   // the Blas leaf's own primitive index
                        // descend into the mesh BVH (TriangleMesh::hit: closest_t = local t_max = +inf)
                        c_rec = rec; c_k = k; c_j = j + 1u;
                        in_mesh = true; entered = true;
                        cur_inst = inst_idx;
                        ct_world = ct; ct = PT_INF;
                        o = lo; d = ld; inv = linv;
                        node_base = __float_as_uint(b1.z); tri_base = __float_as_uint(b1.w);
                        m_tri = PTX_MISS;
                        base = sp;
                        st_write(st, sp++, 0u, 0.0f);
                        break;
                    }
                }
            }
            if (entered) break;
            fresh = false; k0 = 0; j0 = 0; rec += 1u;
        } while (more);
    }
    return found;
}

}  // namespace ptd
