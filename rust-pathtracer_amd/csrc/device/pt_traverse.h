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
// One lane = one ray; a wave is persistent and pulls rays from the queue as
// lanes finish (one wave-aggregated atomic).
//
// The reference nests three traversals (closures inside closures) and tests a
// leaf's primitives in the middle of a node's slot loop.  Run as written on a
// 64-wide wave, every iteration would execute the node test, the triangle
// loop, the instance entry and the instance exit one after the other, each for
// the few lanes that need it (measured: 23 % of lanes active per VALU
// instruction on coherent primary rays, 14 % on bounce rays).  Here the work
// is cut into four ACTIONS — visit a node record, test one triangle, enter an
// instance, leave an instance — driven by ONE typed stack per lane.  Each
// iteration every lane that finished its action pops its next entry (shared
// code), then the wave runs the action most lanes are waiting for, plus any
// other action with at least RUN_THRESHOLD lanes waiting; the remaining lanes
// sit the iteration out.  Each action body loads its own record, so a lane
// carries no prefetch registers and the kernel fits more waves per SIMD.
//
// Two ways to walk.  The reference-order walk below reproduces the reference's visit order exactly; it is what the
// counted kernels, the any-hit kernels, the packets and every second attempt run.  The uncounted closest-hit kernels
// walk nearest child first ("FAST mode" further down) and fall back to the reference order, ray by ray, whenever the
// order could decide the result.
//
// Order is preserved exactly through the unified stack:
//   * a record lists its children in the order the reference's walk takes them
//     (leaf slots ascending, then internal slots descending: pt_types.h); a node
//     visit pushes the hit children in REVERSE list order, so they pop in list
//     order: the leaves first, in ascending slot order, before any deeper node —
//     the reference's "test leaf slots immediately, push internal slots" — then
//     the internal children in descending slot order, as the reference's LIFO
//     stack yields them.  Rays without NaN slabs walk the WIDE trees (up to 8
//     children per record, same leaf order: pt_types.h says why that cannot
//     change a hit); rays with NaN slabs and the counted kernels walk the
//     faithful trees, one record per reference node;
//   * the reference's per-slot cull `dist >= closest_t` (accel/bvh.rs:344-347)
//     is applied when an entry is popped.  An internal child that an earlier
//     leaf of the same node makes cullable is pushed anyway and dropped by the
//     identical pop-time test (accel/bvh.rs:327-329): same nodes visited;
//   * entering an instance pushes a three-entry MARKER (world 1/d, world
//     closest_t, instance id) under the mesh level; popping it is the end of
//     the mesh traversal.  A TLAS leaf with several primitives re-pushes its
//     remainder (uncullable, like the reference's inner `for prim_idx` loop)
//     below the marker before entering the first.
//
// Node test: `a < b ? a : b` / `a > b ? a : b` reproduce _mm256_min_ps/_max_ps
// including their NaN rule (second operand).  NaNs can only come from 0 * inf,
// i.e. from a ray with a zero direction component or a non-finite origin; when
// no lane in the node action carries such a ray the test runs on
// v_min_f32/v_max_f32, which give the same values for non-NaN inputs at half
// the VALU cost.
//
// Stack: the first LDS_STACK entries of each lane live in LDS (entry-major,
// lane-minor: every lane owns one bank column, so divergent stack pointers
// never conflict); deeper entries spill to a per-lane strip in HBM.  The
// reference's arrays hold 64 entries per level and panic beyond that; here the
// same condition (on internal pushes) raises an error flag.
#pragma once
#include "pt_types.h"

namespace ptd {

#ifndef PT_LDS_STACK
#define PT_LDS_STACK 12
#endif
#ifndef PT_REFILL_MIN_IDLE
#define PT_REFILL_MIN_IDLE 15
#endif
#ifndef PT_REFILL_MIN_IDLE_ANY
#define PT_REFILL_MIN_IDLE_ANY 48
#endif
#ifndef PT_WAVES_PER_SIMD
#define PT_WAVES_PER_SIMD 6
#endif
constexpr int LDS_STACK = PT_LDS_STACK;
constexpr int LEVEL_STACK_LIMIT = 64;                 // accel/bvh.rs:320,379
// two levels of internal entries + per level: pending leaves (<= 4), TLAS remainder (1), marker (3)
constexpr int OVF_STACK = 2 * LEVEL_STACK_LIMIT + 16 - LDS_STACK;
constexpr int STACK_CAP = LDS_STACK + OVF_STACK;      // entries one lane's stack can hold (LDS part + HBM strip)
constexpr int TRACE_BLOCK = 64;                       // one wave per block: traversal needs no barrier
constexpr uint32_t REFILL_MIN_IDLE = PT_REFILL_MIN_IDLE, REFILL_MIN_IDLE_ANY = PT_REFILL_MIN_IDLE_ANY;   // refill a wave once this many lanes are idle
#ifndef PT_RUN_THRESHOLD
#define PT_RUN_THRESHOLD 10
#endif
#ifndef PT_TH_NODE
#define PT_TH_NODE PT_RUN_THRESHOLD
#endif
#ifndef PT_TH_TRI
#define PT_TH_TRI PT_RUN_THRESHOLD
#endif
#ifndef PT_TH_ENTER
#define PT_TH_ENTER PT_RUN_THRESHOLD
#endif
#ifndef PT_TH_EXIT
#define PT_TH_EXIT PT_RUN_THRESHOLD
#endif
constexpr uint32_t TH_NODE = PT_TH_NODE, TH_TRI = PT_TH_TRI, TH_ENTER = PT_TH_ENTER, TH_EXIT = PT_TH_EXIT;
#ifndef PT_RUN_THRESHOLD_ANY
#define PT_RUN_THRESHOLD_ANY 6
#endif
constexpr uint32_t TH_ANY = PT_RUN_THRESHOLD_ANY;   // any-hit traversals: one threshold for all four actions
constexpr uint32_t RUN_THRESHOLD = PT_RUN_THRESHOLD;  // a non-majority action also runs with this many lanes waiting
constexpr uint32_t NEG_INF_BITS = 0xFF800000u;
constexpr uint32_t MARK_INST_MASK = 0x00FFFFFFu;      // marker word: instance id | MARK_EXACT
constexpr uint32_t MARK_EXACT = 0x01000000u;

// File-scope LDS array (address space known at every use: plain ds_read/ds_write,
// no flat pointers into LDS).  Every kernel that traverses gets its own copy.
static __shared__ uint2 s_stack[LDS_STACK * TRACE_BLOCK];

// A queue is cut into QUEUE_SEGMENTS contiguous stretches with one cursor each.  Blocks b and b + 8 share an XCD (and
// its L2), so a block starts on segment b % 8 and the rays an XCD traces come from one stretch of the queue — one
// region of the image while the queue is still close to pixel order — instead of from all over it; a wave that finds
// its segment drained moves on to the next one, so placement only matters for speed.
#ifndef PT_QUEUE_SEGMENT_SHIFT
#define PT_QUEUE_SEGMENT_SHIFT 3
#endif
#ifndef PT_SEGMENT_MIN_QUEUE
#define PT_SEGMENT_MIN_QUEUE (8u << 20)
#endif
constexpr uint32_t QUEUE_SEGMENT_SHIFT = PT_QUEUE_SEGMENT_SHIFT;
constexpr uint32_t QUEUE_SEGMENTS = 1u << QUEUE_SEGMENT_SHIFT;
constexpr uint32_t SEGMENT_MIN_QUEUE = PT_SEGMENT_MIN_QUEUE;   // queues shorter than this use one cursor

struct StackRef {
    uint2* ovf;            // the strips of the whole grid (wave-uniform); a lane's entry k is ovf[k * ovf_stride + strip_lane()]
    uint32_t ovf_stride;
};
// A lane's column in the spill strips.  Recomputed at every use (the asm keeps the compiler from hoisting the 64-bit
// lane address out of the traversal loop, where it would occupy two registers for the sake of a rare access).
PT_DEV uint32_t strip_lane() {
    uint32_t t = blockIdx.x * 64u + threadIdx.x;      // TRACE_BLOCK == 64
    asm volatile("" : "+v"(t));
    return t;
}
// Nearly every access stays in the LDS part (0.7 % of the pops reach the strips), and a per-lane "LDS or strip" branch
// costs five or six scalar instructions of exec-mask bookkeeping per access on a CU whose one scalar unit is the
// fullest unit under these kernels.  So the question is asked once per wave: if no active lane is in the strips (one
// compare, one scalar branch) the access is a plain ds_read / ds_write; only otherwise the per-lane branches run.
PT_DEV void st_write(const StackRef& s, int i, uint32_t word, uint32_t dist_bits) {
    const uint2 v = make_uint2(word, dist_bits);
    if (__ballot(i >= LDS_STACK) == 0ull) { s_stack[i * TRACE_BLOCK + (int)threadIdx.x] = v; return; }
    if (i < LDS_STACK) s_stack[i * TRACE_BLOCK + (int)threadIdx.x] = v;
    else s.ovf[(size_t)(i - LDS_STACK) * s.ovf_stride + strip_lane()] = v;
}
PT_DEV uint2 st_read(const StackRef& s, int i) {
    // Two real branches, one load each.  Left alone, the compiler turns "LDS load or HBM load" into one FLAT load of a
    // selected address (slower than ds_read, and it ties LDS traffic to vmcnt); the empty asm is a side effect that
    // cannot be speculated, which keeps the branches and with them the address spaces apart.
    uint2 v;
    if (__ballot(i >= LDS_STACK) == 0ull) {
        v = s_stack[i * TRACE_BLOCK + (int)threadIdx.x];
        asm volatile("" : "+v"(v.x), "+v"(v.y));
        return v;
    }
    if (i < LDS_STACK) {
        v = s_stack[i * TRACE_BLOCK + (int)threadIdx.x];
        asm volatile("" : "+v"(v.x), "+v"(v.y));
    } else {
        v = s.ovf[(size_t)(i - LDS_STACK) * s.ovf_stride + strip_lane()];
    }
    return v;
}

// the distance half of an entry alone (do_tri's check of a hit against its leaf's entry distance)
PT_DEV float st_read_dist(const StackRef& s, int i) {
    uint32_t v;
    if (__ballot(i >= LDS_STACK) == 0ull) {
        v = s_stack[i * TRACE_BLOCK + (int)threadIdx.x].y;
        asm volatile("" : "+v"(v));
        return __uint_as_float(v);
    }
    if (i < LDS_STACK) {
        v = s_stack[i * TRACE_BLOCK + (int)threadIdx.x].y;
        asm volatile("" : "+v"(v));
    } else {
        v = s.ovf[(size_t)(i - LDS_STACK) * s.ovf_stride + strip_lane()].y;
    }
    return __uint_as_float(v);
}

// stack_hw: deepest stack this lane reached; undershoot: largest (t_enter(leaf box) - t) / t over the triangle hits this
// lane computed (how far a hit can lie in front of the boxes above it: forwarded_bound tolerates 2^-5) — counted kernels
// vote[0] = loop iterations of this wave, vote[1..4] = iterations in which the NODE / TRI / ENTER / EXIT action ran,
// vote[5..8] = lanes that took part in it (wave-uniform values; lane 0's copy is flushed)
struct LaneCnt { uint32_t node_visits, node_children, prim_reads, tri_tests, inst_entries, inst_hits, stack_hw; float undershoot; uint32_t vote[9];
                 uint32_t stk_reads, stk_reads_hbm, stk_writes, stk_writes_hbm; float undershoot_x; };   // per lane: stack entries read / written, and those in the HBM strip

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

// The bound a closest-hit ray takes into an instance (see "Forwarding the world bound" in DESIGN.md).  The reference
// searches an instance's mesh with t_max = +inf and compares the winner's re-derived world t with the world closest_t
// afterwards (accel/mod.rs:123-137).  Local t and world t are the same parameter up to rounding, so a local hit at
// t >= B, B = closest_t * (1 + 2^-5) + the certified rounding bound of the round trip, is rejected by that comparison
// whatever it is: the mesh level may start from closest_t = B instead of +inf.  Not for any-hit rays (hit_any ignores
// the world t_max inside an instance: accel/mod.rs:141-146), NaN-slab rays, the counted kernels, or directions far
// from unit length (the bound's constants assume 0.5 <= |d|^2 <= 2).
PT_DEV float forwarded_bound(const SceneDev& S, V3 o, V3 d, float ct) {
    const float dd = dot(d, d);
    const float b = ct * 1.03125f + 1.5f * (S.fwd_lin * ((fabsf(o.x) + fabsf(o.y)) + fabsf(o.z)) + S.fwd_t);
    return (S.fwd_on != 0u && dd >= 0.5f && dd <= 2.0f) ? b : PT_INF;
}

// ---- FAST mode: nearest child first, with a check that the order could not have mattered ---------------------------
// The reference's walk order (slot order, LIFO) decides nothing but WHICH of several near-tied hits wins and how much
// work the walk does.  A closest-hit kernel compiled with FAST pushes a node's hit children far to near, so that the
// nearest is popped first and closest_t shrinks early, culls an entry only when its distance exceeds
// closest_t * FAST_CULL + slack, and keeps a flag per level: "another hit lies within closest_t * FAST_BAND + slack of
// the best one".  A ray that ends with the flag clear has exactly one hit in that band — the reference's walk, whatever
// its order, can only return that hit (DESIGN.md "Nearest first" gives the argument; it rests on the same measured
// assumption as forwarded_bound: a computed hit distance does not lie in front of the computed entry distance of a box
// around it by more than a factor 1 + 2^-10 plus the certified rounding slack).  A ray that ends with the flag set, a
// NaN-slab ray, a ray outside the certified range of the slack (origin far outside the scene, direction far from unit
// length) is traced again by the same lane in reference order (`slow`): same kernel, no second launch.
constexpr float FAST_CULL = 1.00390625f;     // 1 + 4m, m = 2^-10
constexpr float FAST_BAND = 1.001953125f;    // 1 + 2m
constexpr float FAST_UNDER = 1.0009765625f;  // 1 + m: the undershoot every hit a nearest-first lane accepts is checked against (do_tri)
// The absolute parts: the band carries `slack` once, the cull four times (three are needed; four is the multiple an
// instance entry already keeps in a register, a fifth constant spills).  A hit Z in a box culled at
// entry >= t(Y) (1 + 4m) + 4 slack has, under the presumption entry <= t(Z) (1 + m) + slack,
// t(Z) (1 + m) >= t(Y) (1 + 4m) + 3 slack, i.e. t(Z) > t(Y) (1 + 2m) + slack for every t(Y) >= 0 — beyond the band also at
// short range, where slack > m t(Y) (with the same absolute term on both sides the slack cancelled and the margin shrank
// to ~3m t(Y): ADVICE r2).
constexpr float FAST_CULL_SLACKS = 4.0f;

// A lane that tests a leaf's triangles has act = ACT_TRI + the triangles left (1..128): the count rides in the action word
// instead of a register of its own.
enum : uint32_t { ACT_IDLE = 0, ACT_NODE = 1, ACT_ENTER = 3, ACT_EXIT = 4, ACT_POP = 5, ACT_DONE = 6, ACT_TRI = 16 };

// Per-lane traversal state.  The closest hit is written through the sink when it is accepted, not carried.
// UNC (FAST kernels): the scene holds meshes that failed the shape certificate; the TLAS visit looks for their flags
#ifndef PT_ANY_ORDER
#define PT_ANY_ORDER 2
#endif
template <bool ANY, bool CNT, bool FAST = false, bool UNC = false>
struct Traverser {
    // An any-hit walk returns "some triangle along the ray accepts" (accel/bvh.rs:367-414): nothing it decides depends on a
    // bound that shrinks, so the set of boxes it can visit is fixed by the ray and ANY visiting order gives the reference's
    // answer; only the work until the first accepted triangle changes.  0 = the reference's slot order, 1 = nearest child
    // first, 2 = farthest child first.  Measured on C3 / C2 / C4 (any-hit ms per frame): 113.2 / 91.7 / 52.9 in slot order,
    // 214 / 96.8 / 56.5 nearest first (a shadow ray leaves its own surroundings without hitting them; what occludes it is
    // the large thing further out), 110.3 / 90.0 / 50.9 farthest first — the default.  Longest or shortest chord first and
    // largest or smallest box first were measured too: 193 / 126 / 120.5 / 194 ms on C3.  The counted kernels keep the
    // reference's order: their events are its events.
    static constexpr int ANY_ORDER = (ANY && !CNT) ? PT_ANY_ORDER : 0;
    V3 o, d, inv;              // ray of the level being traversed
    // The world ray stays in registers while the lane is inside an instance (round 3: the build without the SLP
    // vectorizer left room for it; re-reading it from the queue entry at every exit cost k_extend 4.7 ms per frame).  The
    // instantiation for scenes with uncertified meshes has no room and re-reads it.
#ifndef PT_KEEP_WORLD
#define PT_KEEP_WORLD 1
#endif
    static constexpr bool KEEP_W = PT_KEEP_WORLD != 0 && !UNC;
    V3 w_o, w_d;
    float ct;                  // closest_t of that level (ANY: its ray.t_max)
    int sp;                    // stack pointer (one typed stack for the three levels)
    uint32_t tri_base;         // first TriRec of the mesh being traversed
    uint32_t m_tri;            // best hit inside the current instance (its t is `ct`)
    float m_u, m_v;
    uint32_t act;
    uint32_t pw;               // NODE: child word; TRI: next TriRec; ENTER: TLAS leaf word; EXIT: marker word
    PT_DEV bool is_tri() const { return act > ACT_TRI; }
    // The lane's flags live in ONE register.  As separate bools the compiler keeps each as a 64-bit lane mask in scalar
    // registers and merges it with three scalar instructions behind every divergent region that may change it — six
    // flags, eight regions per iteration — and a CU has one scalar unit for its four SIMDs: it ran at 0.70 instructions
    // per cycle under k_extend, fuller than the vector units (0.48).  In a vector register the merges are free (exec-masked
    // writes); a test costs two vector instructions.
    //   F_IN_MESH  inside an instance      F_FOUND  a world-level hit has been accepted      F_EXACT  NaN-slab ray at this level
    //   F_AMB_W / F_AMB_L (FAST)  another hit within the band of the best one, at the world level / inside the current instance
    //   F_SLOW (FAST kernels)  this lane walks its ray in reference order (second attempt, or a ray outside the margins' range)
    enum : uint32_t { F_IN_MESH = 1u, F_FOUND = 2u, F_EXACT = 4u, F_AMB_W = 8u, F_AMB_L = 16u, F_SLOW = 32u };
    uint32_t fl;
    PT_DEV bool get(uint32_t b) const { return (fl & b) != 0u; }
    PT_DEV void put(uint32_t b, bool v) { fl = v ? (fl | b) : (fl & ~b); }
    PT_DEV bool fast() const { return FAST && !get(F_SLOW); }
    float leaf_dist;           // counted kernels: entry distance of the leaf whose triangles are being tested
    LaneCnt* lcp;              // counted kernels: the lane's counters (visit_child has no other way to them)

    // which tree the current level is walked in: the faithful one for NaN-slab rays and for the counted kernels
    PT_DEV bool faithful() const { return CNT || get(F_EXACT); }
    PT_DEV void begin(const SceneDev& S, V3 origin, V3 direction, float tmax, bool again = false) {
        o = origin; d = direction;
        if (KEEP_W) { w_o = origin; w_d = direction; }
        inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);          // Ray::new (ray.rs:13-21)
        put(F_EXACT, ray_needs_exact(o, inv));
        ct = tmax;
        sp = 0; tri_base = 0;
        m_tri = PTX_MISS; m_u = 0.0f; m_v = 0.0f;
        put(F_IN_MESH, false); put(F_FOUND, false);
        pw = faithful() ? S.tlas_root_word : S.tlas_root_word_ray;
        // Bvh8::traverse_*: empty => None/false; the root is popped with dist 0.0: culled only if closest_t <= 0
        // (accel/bvh.rs:322-329)
        act = (S.n_tlas_nodes == 0 || (!ANY && 0.0f >= ct)) ? ACT_DONE : ACT_NODE;
        if (FAST) {
            put(F_AMB_L, false); put(F_AMB_W, false);
            // outside what the margins are certified for (or a second attempt): reference order
            const float dd = dot(d, d);
            put(F_SLOW, again || get(F_EXACT) || !(dd >= 0.5f && dd <= 2.0f) || !(((fabsf(o.x) + fabsf(o.y)) + fabsf(o.z)) <= S.fast_omax));
        }
    }
    // FAST: what an entry distance is compared with (pop-time cull, the box test's t_max)
    PT_DEV float cull_bound(const SceneDev& S) const { return fast() ? ct * FAST_CULL + FAST_CULL_SLACKS * S.fast_slack : ct; }
    PT_DEV float band_bound(const SceneDev& S, float best) const { return best * FAST_BAND + S.fast_slack; }

    // ---- pop entries until one needs work; sets the next action (accel/bvh.rs:322-329) ---------------
    PT_DEV void pop(const SceneDev& S, const StackRef& st, LaneCnt& lc) {
        const float cb = cull_bound(S);      // closest_t does not change inside pop
        for (;;) {
            if (sp == 0) { act = ACT_DONE; return; }
            --sp;
            if (CNT) { lc.stk_reads++; if (sp >= LDS_STACK) lc.stk_reads_hbm++; }
            const uint2 e = st_read(st, sp);
            if (!ANY && __uint_as_float(e.y) >= cb) continue;      // cb == closest_t for a reference-order walk (accel/bvh.rs:327-329)
            pw = e.x;
            if (get(F_IN_MESH)) {
                if (e.y == NEG_INF_BITS) act = ACT_EXIT;       // the marker: mesh level drained
                else if (e.x & LEAF_FLAG) {
                    act = ACT_TRI + ((e.x >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1u;
                    if (CNT) leaf_dist = __uint_as_float(e.y);
                    pw = tri_base + (e.x & PRIM_START_MASK);
                } else act = ACT_NODE;
            } else act = (e.x & LEAF_FLAG) ? ACT_ENTER : ACT_NODE;
            return;
        }
    }

    // ---- ACT_NODE: node.intersect + the slot loop's pushes -----------------------------
    // One child: the box test (accel/bvh.rs:198-215 for that slot) and, on a hit, the push (do_node has made sure
    // that the stack has room for all of the record's children).
    template <bool EXACT>
    PT_DEV void visit_child(const StackRef& st, const float4 lo, const float4 hi, bool valid, float tmax) {
        float te;
        const bool h = box_test<EXACT>(lo, hi, o, inv, RAY_EPSILON, tmax, &te) && valid;
        if (h) {
            if (CNT) { lcp->stk_writes++; if (sp >= LDS_STACK) lcp->stk_writes_hbm++; }
            const uint32_t w = __float_as_uint(hi.z);
            st_write(st, sp, w, __float_as_uint(te));
            ++sp;
        }
    }
    // up to four children of one 128-B unit, last first; q0..q5 = children 0..2 of the unit, already in registers
    template <bool EXACT, bool WIDE>
    PT_DEV void visit_unit(const StackRef& st, const float4* q, uint32_t nb, const float4 q0, const float4 q1, const float4 q2,
                           const float4 q3, const float4 q4, const float4 q5, float tmax) {
        if ((FAST || ANY_ORDER != 0) && !WIDE) {
            // One code path for both kinds of lane.  The three children start in the reference's push order (last child
            // first); a nearest-first lane then sorts them far to near, a reference-order lane (`slow`) leaves them
            // alone.  A miss gets key -1 and is skipped.
            float t0, t1, t2;
            const bool h0 = box_test<EXACT>(q0, q1, o, inv, RAY_EPSILON, tmax, &t0);
            const bool h1 = box_test<EXACT>(q2, q3, o, inv, RAY_EPSILON, tmax, &t1) && nb > 1u;
            const bool h2 = box_test<EXACT>(q4, q5, o, inv, RAY_EPSILON, tmax, &t2) && nb > 2u;
            float ka = h2 ? t2 : -1.0f, kb = h1 ? t1 : -1.0f, kc = h0 ? t0 : -1.0f;
            uint32_t wa = __float_as_uint(q5.z), wb = __float_as_uint(q3.z), wc = __float_as_uint(q1.z);
            const bool srt = ANY_ORDER != 0 || fast();
            // three compare-swaps as selects (no branches: exec-mask bookkeeping would cost more than the moves)
            {
                const bool s1 = srt && ka < kb;
                const float xa = s1 ? kb : ka, xb = s1 ? ka : kb; const uint32_t ya = s1 ? wb : wa, yb = s1 ? wa : wb;
                ka = xa; kb = xb; wa = ya; wb = yb;
                const bool s2 = srt && kb < kc;
                const float xb2 = s2 ? kc : kb, xc2 = s2 ? kb : kc; const uint32_t yb2 = s2 ? wc : wb, yc2 = s2 ? wb : wc;
                kb = xb2; kc = xc2; wb = yb2; wc = yc2;
                const bool s3 = srt && ka < kb;
                const float xa3 = s3 ? kb : ka, xb3 = s3 ? ka : kb; const uint32_t ya3 = s3 ? wb : wa, yb3 = s3 ? wa : wb;
                ka = xa3; kb = xb3; wa = ya3; wb = yb3;
            }
            if (ANY_ORDER == 2) {       // the same sorted triple pushed near to far: the farthest child is popped first
                if (kc >= 0.0f) { st_write(st, sp, wc, __float_as_uint(kc)); ++sp; }
                if (kb >= 0.0f) { st_write(st, sp, wb, __float_as_uint(kb)); ++sp; }
                if (ka >= 0.0f) { st_write(st, sp, wa, __float_as_uint(ka)); ++sp; }
                return;
            }
            if (ka >= 0.0f) { if (CNT) { lcp->stk_writes++; if (sp >= LDS_STACK) lcp->stk_writes_hbm++; } st_write(st, sp, wa, __float_as_uint(ka)); ++sp; }
            if (kb >= 0.0f) { if (CNT) { lcp->stk_writes++; if (sp >= LDS_STACK) lcp->stk_writes_hbm++; } st_write(st, sp, wb, __float_as_uint(kb)); ++sp; }
            if (kc >= 0.0f) { if (CNT) { lcp->stk_writes++; if (sp >= LDS_STACK) lcp->stk_writes_hbm++; } st_write(st, sp, wc, __float_as_uint(kc)); ++sp; }
            return;
        }
        if (WIDE && nb > 3u) { const float4 d0 = q[6], d1 = q[7]; visit_child<EXACT>(st, d0, d1, true, tmax); }
        visit_child<EXACT>(st, q4, q5, nb > 2u, tmax);
        visit_child<EXACT>(st, q2, q3, nb > 1u, tmax);
        visit_child<EXACT>(st, q0, q1, true, tmax);
    }
    // WIDE = false: no lane of this run has a record with more than 3 children (the faithful tree of the reference's
    // builder never has one), so the fourth child and the second unit are compiled out.
    // r0..r5 = children 0..2 of the unit visited FIRST: the second unit when the record has one (children 4..6), else the first.
    template <bool EXACT, bool WIDE>
    PT_DEV void do_node(const SceneDev& S, const StackRef& st, LaneCnt& lc, uint32_t* err, const float4 r0, const float4 r1,
                        const float4 r2, const float4 r3, const float4 r4, const float4 r5) {
        const uint32_t nchild = ((pw >> NODE_NCHILD_SHIFT) & NODE_NCHILD_MASK) + 1u;
        const float4* p = reinterpret_cast<const float4*>(&S.nodes[pw & NODE_INDEX_MASK]);
        // the device stack (all levels of one ray share it) holds STACK_CAP entries; the reference's fixed arrays hold
        // 64 per level and panic beyond (accel/bvh.rs:320,361,379,408)
        if (sp + (int)nchild > STACK_CAP) { atomicOr(err, 1u); put(F_FOUND, false); act = ACT_DONE; return; }
        act = ACT_POP;
        if (FAST && UNC && !ANY) {      // compiled in only for scenes that hold an uncertified mesh (k_extend<.., true, true>)
            // A nearest-first lane culls by distance what the reference's walk may still visit.  That is only sound for
            // meshes that passed the shape certificate: a lane whose ray meets the box of a TLAS child holding an
            // uncertified instance AT ANY DISTANCE (the reference's closest_t at that point is unknown, so none is
            // assumed) starts over in reference order.  Mesh records carry no such flag; scenes without an uncertified
            // mesh run the instantiation without this block.
            uint32_t unc = (__float_as_uint(r1.w) | __float_as_uint(r3.w) | __float_as_uint(r5.w)) & META_UNCERT;
            if (WIDE) unc |= META_UNCERT & (nchild > 3u ? 0xFFFFFFFFu : 0u);      // records of 4..8 children (host trees): look at every child
            if (unc != 0u && fast() && !get(F_IN_MESH)) {
                bool touch = false;
                float te;
                if (!WIDE) {      // <= 3 children, already in registers
                    touch = ((__float_as_uint(r1.w) & META_UNCERT) != 0u && box_test<true>(r0, r1, o, inv, RAY_EPSILON, PT_INF, &te)) ||
                            ((__float_as_uint(r3.w) & META_UNCERT) != 0u && box_test<true>(r2, r3, o, inv, RAY_EPSILON, PT_INF, &te)) ||
                            ((__float_as_uint(r5.w) & META_UNCERT) != 0u && box_test<true>(r4, r5, o, inv, RAY_EPSILON, PT_INF, &te));
                } else {
                    for (uint32_t k = 0; k < nchild; ++k) {
                        const float4 lo = p[2u * k], hi = p[2u * k + 1u];
                        if ((__float_as_uint(hi.w) & META_UNCERT) != 0u && box_test<true>(lo, hi, o, inv, RAY_EPSILON, PT_INF, &te)) touch = true;
                    }
                }
                if (touch) { put(F_AMB_W, true); put(F_FOUND, false); act = ACT_DONE; return; }
            }
        }
        if (WIDE && nchild > 4u) {
            // a record of two units: its children 7..4 now (they are pushed first), its children 3..0 in the lane's next
            // node action — the lane stays in ACT_NODE with the record's first unit as a 4-child record.  Two short
            // visits instead of one long one keep the kernel at 80 VGPRs; only hosts with truly wide BVH8 nodes (the
            // reference's builder stops at 3 children) and the PTX_WIDE_RAY_BVH experiment reach this.
            visit_unit<EXACT, true>(st, p + 8, nchild - 4u, r0, r1, r2, r3, r4, r5, cull_bound(S));
            pw = (pw & NODE_INDEX_MASK) | (3u << NODE_NCHILD_SHIFT);
            act = ACT_NODE;
        } else {
            if (CNT) { const uint32_t meta = __float_as_uint(r1.w); lc.node_visits += (meta >> 4) & 1u; lc.node_children += meta & 15u; }
            visit_unit<EXACT, WIDE>(st, p, nchild, r0, r1, r2, r3, r4, r5, cull_bound(S));
        }
        if (CNT) lc.stack_hw = (uint32_t)sp > lc.stack_hw ? (uint32_t)sp : lc.stack_hw;
    }

    // ---- ACT_TRI: intersect_triangle on the next primitive of the leaf ------------------
    PT_DEV void do_tri(const SceneDev& S, const StackRef& st, LaneCnt& lc, const float4 ta, const float4 tb, const float4 tc) {
        const float tmin = RAY_EPSILON;
        if (CNT) { lc.prim_reads++; lc.tri_tests++; }
        pw += 1u; act -= 1u;
        if (act == ACT_TRI) act = ACT_POP;
        // intersect_triangle with its early returns turned into one predicate: the same operations in the same
        // order, so a lane that reaches the end has the reference's t, u, v bit for bit; a lane the reference
        // would have returned early from computes values nobody reads
        const V3 v0 = v3(ta.x, ta.y, ta.z), edge1 = v3(ta.w, tb.x, tb.y), edge2 = v3(tb.z, tb.w, tc.x);
        const V3 h = cross(d, edge2);
        const float a = dot(edge1, h);
        const float f = 1.0f / a;
        const V3 s = o - v0;
        const float u = f * dot(s, h);
        const V3 q = cross(s, edge1);
        const float v = f * dot(d, q);
        const float t = f * dot(edge2, q);
        const bool hit = !(fabsf(a) < 1e-8f) && (u >= 0.0f && u <= 1.0f) && !(v < 0.0f || u + v > 1.0f) &&
                         (t >= tmin && t <= PT_INF);   // local ray: t_max = +inf
        if (hit) {
            if (CNT) {
                const float us = (leaf_dist - t) / t; lc.undershoot = us > lc.undershoot ? us : lc.undershoot;
                // ... and what of it the absolute slack of the margins does not cover (the relative part they must)
                const float ux = ((leaf_dist - t) - S.fast_slack) / t; lc.undershoot_x = ux > lc.undershoot_x ? ux : lc.undershoot_x;
            }
            if (ANY) { put(F_FOUND, true); act = ACT_DONE; return; }
            if (FAST) {
                // The presumption the margins rest on, checked on every hit a nearest-first lane is about to accept: the
                // hit does not lie in front of its leaf box's entry distance by more than (1 + m) and the slack.  The
                // leaf's stack entry still sits in the slot it was popped from (a triangle action pushes nothing), so
                // this costs one stack read per HIT, no register.  A hit that fails (a sliver triangle met at a grazing
                // angle: Moeller-Trumbore's t is then mostly rounding) sends the ray to the reference order.
                if (fast() && t < ct) {
                    // (read here, inside the hit branch: hoisted above the arithmetic the read costs 9 ms per frame)
                    const float ld = st_read_dist(st, sp);
                    if (ld > t * FAST_UNDER + S.fast_slack) { put(F_AMB_W, true); put(F_FOUND, false); act = ACT_DONE; return; }
                }
                // a second hit within the band of the instance's best one: which of them the reference's walk returns depends on its order
                const bool have = m_tri != PTX_MISS;
                if (t < ct) put(F_AMB_L, have && ct <= band_bound(S, t));
                else if (have && t <= band_bound(S, ct)) put(F_AMB_L, true);
            }
            if (t < ct) { ct = t; m_tri = __float_as_uint(tc.y); m_u = u; m_v = v; }
        }
    }

    // ---- ACT_ENTER: Instance::hit / hit_any head + the Blas-level box ---------------------
    PT_DEV void do_enter(const SceneDev& S, const StackRef& st, LaneCnt& lc, uint32_t* err) {
        const float tmin = RAY_EPSILON;
        const uint32_t start = pw & PRIM_START_MASK;
        const uint32_t count = ((pw >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1u;
        const float4* p = reinterpret_cast<const float4*>(&S.inst_tlas[start]);
        const float4 ia = p[0], ib = p[1], ic = p[2], b0 = p[3], b1 = p[4];
        const uint32_t inst_idx = S.tlas_prims[start];
        // TLAS leaf: this primitive now, the rest after the instance has been left
        if (sp + 4 > STACK_CAP) { atomicOr(err, 1u); put(F_FOUND, false); act = ACT_DONE; return; }     // remainder + 3-entry marker
        if (count > 1u) st_write(st, sp++, LEAF_FLAG | ((count - 2u) << PRIM_COUNT_SHIFT) | (start + 1u), NEG_INF_BITS);
        if (CNT) { lc.prim_reads++; lc.inst_entries++; lc.node_visits++; lc.node_children++; }
        const V3 lo = xf_point(ia, ib, ic, o);
        const V3 ld = xf_vector(ia, ib, ic, d);
        const V3 linv = v3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
        // Blas level: one node, slot 0 = leaf(0,1) boxed by the mesh bounds; t_max = +inf
        const bool lexact = ray_needs_exact(lo, linv);
        float tb_enter;
        bool bhit;
        if (__ballot(lexact) != 0ull) bhit = box_test<true>(b0, make_float4(b1.x, b1.y, 0.0f, 0.0f), lo, linv, tmin, PT_INF, &tb_enter);
        else bhit = box_test<false>(b0, make_float4(b1.x, b1.y, 0.0f, 0.0f), lo, linv, tmin, PT_INF, &tb_enter);
        // an instance of a mesh that failed the shape certificate (pt_types.h): the reference's walk, operation for operation
        const bool uncert = (__float_as_uint(b1.z) & INST_UNCERT) != 0u;
        if (fast() && (lexact || uncert)) { put(F_AMB_W, true); put(F_FOUND, false); act = ACT_DONE; return; }      // NaN slabs in the local frame / uncertified: start over in reference order
        if (FAST) put(F_AMB_L, false);
        // a nearest-first lane must SEE every local hit that could tie with the best one: its bound carries the scene-wide
        // slack (>= this ray's rounding bound) four times over instead of the per-ray term
        const float bound = (ANY || get(F_EXACT) || lexact || uncert || (CNT && S.fwd_on < 2u)) ? PT_INF
                            : fast() ? ct * 1.03125f + 4.0f * S.fast_slack : forwarded_bound(S, o, d, ct);
        if (!bhit || (!ANY && tb_enter >= bound)) { act = ACT_POP; return; }
        if (CNT) lc.prim_reads++;   // the Blas leaf's own primitive index
        // park the world level under a marker and descend into the mesh BVH
        // (TriangleMesh::hit: closest_t = local t_max = +inf; root popped with dist 0)
        st_write(st, sp++, __float_as_uint(inv.x), __float_as_uint(inv.y));
        st_write(st, sp++, __float_as_uint(inv.z), __float_as_uint(ct));
        st_write(st, sp++, inst_idx | (get(F_EXACT) ? MARK_EXACT : 0u), NEG_INF_BITS);
        if (CNT) lc.stack_hw = (uint32_t)sp > lc.stack_hw ? (uint32_t)sp : lc.stack_hw;
        put(F_IN_MESH, true);
        ct = bound;
        o = lo; d = ld; inv = linv;
        put(F_EXACT, lexact);
        tri_base = __float_as_uint(b1.w);
        m_tri = PTX_MISS;
        // root record of the mesh BVH (absolute), child count in the top bits: wide tree, or the faithful one
        pw = faithful() ? S.inst_root_faithful[start] : (__float_as_uint(b1.z) & ~INST_UNCERT);
        act = ACT_NODE;
    }

    // ---- ACT_EXIT: tail of Instance::hit (accel/mod.rs:128-137); pw = marker word --------------------
    template <class Src, class Sink>
    PT_DEV void do_exit(const SceneDev& S, const StackRef& st, LaneCnt& lc, const Src& src, const Sink& sink, uint32_t my) {
        const uint32_t cur_inst = pw & MARK_INST_MASK;
        const uint2 e1 = st_read(st, sp - 1), e0 = st_read(st, sp - 2);
        sp -= 2;
        V3 wo = w_o, wd = w_d;
        if (!KEEP_W) { float unused_tmax; src.fetch(my, &wo, &wd, &unused_tmax); }
        float ct_world = __uint_as_float(e1.y);
        if (!ANY && m_tri != PTX_MISS) {
            if (CNT) lc.inst_hits++;
            const float4* p = reinterpret_cast<const float4*>(&S.inst[cur_inst]);
            const float4 ma = p[5], mb = p[6], mc = p[7];
            const V3 local_hit_point = o + ct * d;
            const V3 world_hit_point = xf_point(ma, mb, mc, local_hit_point);
            const float t_world = dot(world_hit_point - wo, wd) / dot(wd, wd);
            if (FAST) {
                // this instance's winner against the best so far; a winner that is itself in doubt (F_AMB_L) stands for
                // hits down to one band below it
                if (get(F_AMB_L) && t_world <= band_bound(S, band_bound(S, ct_world))) put(F_AMB_W, true);
                // (the ray's own finite t_max stands for a hit as well: the reference skips an instance whose box is
                // entered at or beyond it, accel/bvh.rs:327,345, where this walk would still enter and accept — ADVICE r2)
                const bool rival = get(F_FOUND) || ct_world < PT_INF;
                if (t_world < ct_world) { if (!get(F_AMB_L)) put(F_AMB_W, rival && ct_world <= band_bound(S, t_world)); }
                else if (get(F_FOUND) && t_world <= band_bound(S, ct_world)) put(F_AMB_W, true);
            }
            if (t_world < ct_world) {                               // traverse_closest accept (accel/bvh.rs:353-357)
                ct_world = t_world;
                sink.hit(my, cur_inst, m_tri, t_world, m_u, m_v);
                put(F_FOUND, true);
            }
        }
        m_tri = PTX_MISS;
        put(F_IN_MESH, false);
        o = wo; d = wd;
        inv = v3(__uint_as_float(e0.x), __uint_as_float(e0.y), __uint_as_float(e1.x));
        put(F_EXACT, (pw & MARK_EXACT) != 0u);
        ct = ct_world;
        act = ACT_POP;
    }
};

// Wave-persistent driver.  `Src::key(i)` turns queue position i into the entry's key k (read once, kept in a register);
// `Src::fetch(k, o, d, tmax)` loads its ray; `Sink::hit(k, inst, tri, t, u, v)` records an accepted closest hit (the last
// call for an entry is its result); `Sink::done(k, found)` ends the entry.
// `work` is the queue's shared cursor, `n` its length.
// FAST (closest-hit only): see above.
// LOCAL: the queue and its cursor belong to this wave alone (k_tail): every wave of the grid has entries of its own.
template <bool ANY, bool CNT, bool FAST = false, bool UNC = false, bool LOCAL = false, class Src, class Sink>
PT_DEV void trace_queue(const SceneDev& S, const StackRef& st, uint32_t* work, const uint32_t n, const Src& src, const Sink& sink,
                        LaneCnt& lc, uint32_t& nrays, uint32_t* err,       // err: the device error word (set on a stack overflow)
                        unsigned long long* again_count = nullptr) {            // FAST: statistics, rays traced a second time
    // a short queue cannot feed every wave: the surplus ones leave without touching the cursors (6144 waves bumping
    // a few addresses made an empty launch cost 70 us)
    if (!LOCAL && (unsigned long long)blockIdx.x * 16ull >= (unsigned long long)n + 128ull) return;
    using TV = Traverser<ANY, CNT, FAST, UNC>;
    TV tv;
    tv.act = ACT_IDLE;
    tv.fl = 0u;
    uint32_t n_again = 0;
    tv.sp = 0; tv.pw = 0; tv.tri_base = 0; tv.m_tri = PTX_MISS; tv.m_u = 0.0f; tv.m_v = 0.0f;
    tv.ct = 0.0f; tv.o = v3(0.0f, 0.0f, 0.0f); tv.d = tv.o; tv.inv = tv.o;
    tv.w_o = tv.o; tv.w_d = tv.o;
    tv.leaf_dist = 0.0f; tv.lcp = nullptr;
    uint32_t my = 0;
    // Idle lanes claim queue entries with one wave-aggregated atomic as soon as REFILL_MIN_IDLE of them have
    // gathered.  (Claiming a chunk ahead of need was measured and rejected: a launch lasts only ~16 ray lifetimes,
    // and entries parked in 6144 waves' prefetch slots lengthen its tail by more than the hidden atomic saves.)
    bool exhausted = false;   // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    // short queues keep one cursor: with few entries per wave the extra cursor traffic at the end of every segment
    // costs more than the L2 affinity gains (Cornell 640x360: 7.8 ms per frame with one cursor, 9.3 with eight)
    const uint32_t seg_shift = !LOCAL && n >= SEGMENT_MIN_QUEUE ? QUEUE_SEGMENT_SHIFT : 0u, nseg = 1u << seg_shift;
    const uint32_t seg_size = (n + nseg - 1u) >> seg_shift;   // n < 2^31
    uint32_t seg = blockIdx.x & (nseg - 1u), seg_left = nseg;   // wave-uniform
    for (;;) {
        const unsigned long long idle = __ballot(tv.act == ACT_IDLE);
        if (!exhausted && (uint32_t)__popcll(idle) >= (ANY ? REFILL_MIN_IDLE_ANY : REFILL_MIN_IDLE)) {
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const int leader = __ffsll((long long)idle) - 1;
            const uint32_t lo = seg * seg_size;
            const uint32_t hi = lo + seg_size < n ? lo + seg_size : n;
            uint32_t basei = 0;
            if ((int)lane == leader) basei = atomicAdd(&work[seg], n_idle);
            // v_readlane, not a shuffle: the compiler then knows that the claim, and the segment bookkeeping that depends
            // on it, are wave-uniform and keeps them in scalar registers
            basei = lo + (uint32_t)__builtin_amdgcn_readlane((int)basei, leader);
            if (tv.act == ACT_IDLE) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t i = basei + rank;
                if (i < hi) {
                    V3 o, d;
                    float tmax;
                    my = src.key(i);             // what fetch / hit / done address the entry by (e.g. its path slot)
                    src.fetch(my, &o, &d, &tmax);
                    tv.begin(S, o, d, tmax);
                    nrays++;
                }
            }
            if (basei + n_idle >= hi) {          // this segment is drained: on to the next, or done after all of them
                seg = (seg + 1u) & (nseg - 1u);
                if (--seg_left == 0u) exhausted = true;
            }
        }
        if (CNT) tv.lcp = &lc;
        if (tv.act == ACT_POP) tv.pop(S, st, lc);
        if (FAST) {
            const bool redo = tv.act == ACT_DONE && tv.get(TV::F_AMB_W) && !tv.get(TV::F_SLOW);
            if (again_count) n_again += (uint32_t)__popcll(__ballot(redo));      // statistics: a wave-uniform count (a scalar register, not one per lane)
            if (redo) {
                // the result could depend on the walk order: the lane starts its ray over, in reference order (whatever the
                // first attempt wrote through the sink is overwritten)
                V3 o, d;
                float tmax;
                src.fetch(my, &o, &d, &tmax);
                tv.begin(S, o, d, tmax, true);
            }
        }
        if (tv.act == ACT_DONE) { sink.done(my, tv.get(TV::F_FOUND)); tv.act = ACT_IDLE; }

        // vote: the action most lanes are waiting for runs, and so does any other with RUN_THRESHOLD lanes waiting
        const uint32_t cn = (uint32_t)__popcll(__ballot(tv.act == ACT_NODE));
        const uint32_t ctn = (uint32_t)__popcll(__ballot(tv.is_tri()));
        const uint32_t ce = (uint32_t)__popcll(__ballot(tv.act == ACT_ENTER));
        const uint32_t cx = (uint32_t)__popcll(__ballot(tv.act == ACT_EXIT));
        const uint32_t a = cn > ctn ? cn : ctn, b = ce > cx ? ce : cx;
        const uint32_t mx = a > b ? a : b;
        if (mx == 0u) {
            if (exhausted) break;
            continue;
        }
        if (CNT) {
            lc.vote[0]++;
            if (cn == mx || cn >= (ANY ? TH_ANY : TH_NODE)) { lc.vote[1]++; lc.vote[5] += cn; }
            if (ctn == mx || ctn >= (ANY ? TH_ANY : TH_TRI)) { lc.vote[2]++; lc.vote[6] += ctn; }
            if (ce == mx || ce >= (ANY ? TH_ANY : TH_ENTER)) { lc.vote[3]++; lc.vote[7] += ce; }
            if (cx == mx || cx >= (ANY ? TH_ANY : TH_EXIT)) { lc.vote[4]++; lc.vote[8] += cx; }
        }
        if (cx == mx || cx >= (ANY ? TH_ANY : TH_EXIT)) {
            if (tv.act == ACT_EXIT) tv.do_exit(S, st, lc, src, sink, my);
        }
        if (ce == mx || ce >= (ANY ? TH_ANY : TH_ENTER)) {
            if (tv.act == ACT_ENTER) tv.do_enter(S, st, lc, err);
        }
        // Triangle and node lanes fetch their records together and wait once: the two sections used to pay one
        // memory round trip each, and a wave spends more time waiting on these loads than issuing.
        const bool f_tri = (ctn == mx || ctn >= (ANY ? TH_ANY : TH_TRI)) && tv.is_tri();
        const bool f_node = (cn == mx || cn >= (ANY ? TH_ANY : TH_NODE)) && tv.act == ACT_NODE;
        // wave-uniform choice of the node-visit variant: x86 NaN rule only if some lane can produce NaNs,
        // fourth child only if some lane's record has one
        const uint32_t nch = (tv.pw >> NODE_NCHILD_SHIFT) & NODE_NCHILD_MASK;      // children - 1 (node lanes)
        const bool wave_exact = __ballot(f_node && tv.get(TV::F_EXACT)) != 0ull;
        const bool wave_wide = __ballot(f_node && nch >= 3u) != 0ull;
        if (f_tri || f_node) {
            // node lanes fetch the first three children of the unit they visit first (the second unit of a record that
            // has one: its children are pushed first).  The records live in this block only: declared outside it they
            // cost the loop ~45 moves per iteration (zeros for the lanes that load nothing, copies where the paths merge).
            const float4* p = f_node ? reinterpret_cast<const float4*>(&S.nodes[(tv.pw & NODE_INDEX_MASK) + (nch >= 4u ? 1u : 0u)])
                                     : reinterpret_cast<const float4*>(&S.tris[tv.pw]);
            const float4 r0 = p[0], r1 = p[1], r2 = p[2];
            float4 r3, r4, r5;      // node lanes only
            if (f_node) { r3 = p[3]; r4 = p[4]; r5 = p[5]; }
            if (f_tri) tv.do_tri(S, st, lc, r0, r1, r2);
            if (f_node) {
                if (wave_exact) tv.template do_node<true, true>(S, st, lc, err, r0, r1, r2, r3, r4, r5);
                else if (wave_wide) tv.template do_node<false, true>(S, st, lc, err, r0, r1, r2, r3, r4, r5);
                else tv.template do_node<false, false>(S, st, lc, err, r0, r1, r2, r3, r4, r5);
            }
        }
    }
    if (FAST && again_count) {
        if (lane == 0u && n_again) atomicAdd(again_count, (unsigned long long)n_again);
    }
}

// ---- packet traversal (closest hit) ---------------------------------------------------------------
// For queues whose neighbouring entries are coherent rays (the primary rays of a pixel-major batch: 64 samples of one
// pixel per wave).  The 64 rays of a wave walk the tree TOGETHER: one stack (every lane keeps its own copy of each
// entry's word and its own entry distance, +inf where the ray did not hit that child), one record fetch per step
// from a wave-uniform address, no vote and no idle action.  A ray takes part in an entry exactly when its own
// traversal would have popped and kept it (`dist < closest_t`), the entries a ray takes part in appear in the order
// its own stack would have produced (children are pushed in the same slot order, a superset stays LIFO), and all
// arithmetic is per lane, so hits are bit-identical to trace_queue's; only nodes no ray needs are skipped.
// The packet stack: one entry = a wave-uniform word + one entry distance per lane.  23 entries live in LDS (5.9 KiB
// per wave, the per-ray kernels' budget, but twice their depth because a lane stores 4 bytes per entry, not 8);
// deeper entries go to the per-lane HBM strips.
#ifndef PT_PACKET_LDS
#define PT_PACKET_LDS 23
#endif
constexpr int PACKET_LDS = PT_PACKET_LDS;
static __shared__ uint32_t s_pdist[PACKET_LDS * TRACE_BLOCK];
static __shared__ uint32_t s_pword[PACKET_LDS];
PT_DEV void pk_write(const StackRef& s, int i, uint32_t word, uint32_t dist_bits) {      // i and word are wave-uniform
    if (i < PACKET_LDS) {
        s_pdist[i * TRACE_BLOCK + (int)threadIdx.x] = dist_bits;
        s_pword[i] = word;
    } else {
        s.ovf[(size_t)(i - PACKET_LDS) * s.ovf_stride + strip_lane()] = make_uint2(word, dist_bits);
    }
}
PT_DEV uint2 pk_read(const StackRef& s, int i) {
    uint2 v;
    if (i < PACKET_LDS) {
        v.x = s_pword[i];
        v.y = s_pdist[i * TRACE_BLOCK + (int)threadIdx.x];
        asm volatile("" : "+v"(v.x), "+v"(v.y));
    } else {
        v = s.ovf[(size_t)(i - PACKET_LDS) * s.ovf_stride + strip_lane()];
    }
    return v;
}

// one 128-B unit of a record for a packet: test its nb children for every ray that takes part, push the ones some ray
// hit, last child first (every lane stores its own entry distance, +inf where its ray missed)
PT_DEV void pk_unit(const StackRef& st, int& sp, bool act, bool ex, uint32_t nb, V3 o, V3 inv, float ct, const float4 a0, const float4 a1,
                    const float4 b0, const float4 b1, const float4 c0, const float4 c1, const float4 d0, const float4 d1) {
    constexpr uint32_t POS_INF_BITS = 0x7F800000u;
    const float tmin = RAY_EPSILON;
    float te0, te1, te2, te3 = 0.0f;
    bool h0, h1, h2, h3 = false;
    if (ex) {
        h0 = box_test<true>(a0, a1, o, inv, tmin, ct, &te0);
        h1 = box_test<true>(b0, b1, o, inv, tmin, ct, &te1);
        h2 = box_test<true>(c0, c1, o, inv, tmin, ct, &te2);
        if (nb > 3u) h3 = box_test<true>(d0, d1, o, inv, tmin, ct, &te3);
    } else {
        h0 = box_test<false>(a0, a1, o, inv, tmin, ct, &te0);
        h1 = box_test<false>(b0, b1, o, inv, tmin, ct, &te1);
        h2 = box_test<false>(c0, c1, o, inv, tmin, ct, &te2);
        if (nb > 3u) h3 = box_test<false>(d0, d1, o, inv, tmin, ct, &te3);
    }
    h0 = h0 && act; h1 = h1 && act && nb > 1u; h2 = h2 && act && nb > 2u; h3 = h3 && act;
    if (__ballot(h3) != 0ull) pk_write(st, sp++, __float_as_uint(d1.z), h3 ? __float_as_uint(te3) : POS_INF_BITS);
    if (__ballot(h2) != 0ull) pk_write(st, sp++, __float_as_uint(c1.z), h2 ? __float_as_uint(te2) : POS_INF_BITS);
    if (__ballot(h1) != 0ull) pk_write(st, sp++, __float_as_uint(b1.z), h1 ? __float_as_uint(te1) : POS_INF_BITS);
    if (__ballot(h0) != 0ull) pk_write(st, sp++, __float_as_uint(a1.z), h0 ? __float_as_uint(te0) : POS_INF_BITS);
}

#ifndef PT_PACKET_CHUNK
#define PT_PACKET_CHUNK 8
#endif
constexpr uint32_t PACKET_CHUNK = PT_PACKET_CHUNK;
template <bool ANY, class Src, class Sink>
PT_DEV void trace_packets(const SceneDev& S, const NodeRec* __restrict__ g_nodes, const TriRec* __restrict__ g_tris,
                          const InstRec* __restrict__ g_inst_tlas, const InstRec* __restrict__ g_inst, const uint32_t* __restrict__ g_tlas_prims,
                          const uint32_t* __restrict__ g_root_faithful, const uint32_t* __restrict__ g_root_packet, const StackRef& st, uint32_t* work, const uint32_t n, const Src& src, const Sink& sink, uint32_t& err,
                          const int capacity = PACKET_LDS + OVF_STACK, uint32_t* redo_keys = nullptr, uint32_t* redo_count = nullptr,
                          uint32_t* stack_hw = nullptr) {
    constexpr uint32_t POS_INF_BITS = 0x7F800000u;
    const float tmin = RAY_EPSILON;
    const uint32_t lane = threadIdx.x & 63u;
    // PACKET_CHUNK packets per claim: one cursor bumped once per packet would serialise 2 M same-address atomics
    // (~17 ns each) in a 133 M-ray launch
    uint32_t base = 0, chunk_left = 0;     // wave-uniform
    int hw = 0;                            // deepest packet stack of this wave (wave-uniform)
    for (;;) {
        if (chunk_left == 0u) {
            uint32_t b = 0;
            if (lane == 0u) b = atomicAdd(work, 64u * PACKET_CHUNK);
            base = __builtin_amdgcn_readfirstlane(b);
            chunk_left = PACKET_CHUNK;
        } else {
            base += 64u;
        }
        chunk_left -= 1u;
        if (base >= n) break;
        const uint32_t i = base + lane;
        const bool valid = i < n;
        uint32_t key = 0;
        V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(1.0f, 1.0f, 1.0f);
        float ct = 0.0f;
        if (valid) { key = src.key(i); src.fetch(key, &o, &d, &ct); }
        V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);          // Ray::new (ray.rs:13-21)
        bool exact = ray_needs_exact(o, inv);
        bool found = false;
        const V3 w_o = o, w_d = d;      // world level, parked while the packet is inside an instance
        V3 w_inv = inv;
        float w_ct = ct;
        bool w_exact = exact;
        uint32_t m_tri = PTX_MISS;
        float m_u = 0.0f, m_v = 0.0f;
        // wave-uniform state
        unsigned long long entered = 0ull;
        uint32_t cur_inst = 0u, tri_base = 0u;
        bool in_mesh = false;
        int sp = 0, marker_sp = -1;
        bool overflow = false;
        // Bvh8::traverse_closest: (root, 0.0) is popped first; a lane without a ray never takes part
        // the packet walks ONE tree: the wide one unless one of its rays can produce NaN slabs (pt_types.h)
        if (S.n_tlas_nodes != 0u) pk_write(st, sp++, __ballot(valid && exact) != 0ull ? S.tlas_root_word : S.tlas_root_word_packet, valid ? 0u : POS_INF_BITS);
        while (sp > 0) {
            hw = sp > hw ? sp : hw;
            --sp;
            if (in_mesh && sp == marker_sp) {
                // ---- tail of Instance::hit (accel/mod.rs:128-137) for the lanes that entered
                const bool was = ((entered >> lane) & 1ull) != 0ull;
                if (was) {
                    const V3 wo = w_o, wd = w_d;
                    if (!ANY && m_tri != PTX_MISS) {
                        const float4* p = reinterpret_cast<const float4*>(&g_inst[cur_inst]);
                        const float4 ma = p[5], mb = p[6], mc = p[7];
                        const V3 local_hit_point = o + ct * d;
                        const V3 world_hit_point = xf_point(ma, mb, mc, local_hit_point);
                        const float t_world = dot(world_hit_point - wo, wd) / dot(wd, wd);
                        if (t_world < w_ct) {                       // traverse_closest accept (accel/bvh.rs:353-357)
                            w_ct = t_world;
                            sink.hit(key, cur_inst, m_tri, t_world, m_u, m_v);
                            found = true;
                        }
                    }
                    o = wo; d = wd; inv = w_inv; ct = w_ct; exact = w_exact;
                    m_tri = PTX_MISS;
                }
                in_mesh = false;
                entered = 0ull;
                continue;
            }
            const uint2 e = pk_read(st, sp);
            const uint32_t word = __builtin_amdgcn_readfirstlane(e.x);
            // closest: the pop-time cull (accel/bvh.rs:327-329), per ray; any-hit never culls (accel/bvh.rs:387-389): a ray takes
            // part in the entries it pushed (distance below +inf) until it has found its hit
            const bool act = ANY ? (!found && e.y != POS_INF_BITS) : !(__uint_as_float(e.y) >= ct);
            if (__ballot(act) == 0ull) continue;
            if (sp + 10 > capacity) {
                // The packet's stack is the union of 64 rays' stacks and can outgrow what any one of them needs.  With a
                // redo queue the packet's rays are handed to the per-ray kernel (which starts them from scratch and
                // overwrites whatever this packet wrote); without one it is an error.
                overflow = true;
                break;
            }
            if (word & LEAF_FLAG) {
                const uint32_t start = word & PRIM_START_MASK;
                const uint32_t count = ((word >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1u;
                if (in_mesh) {
                    // the leaf's triangles in order (triangle_mesh.rs:64-100 under accel/bvh.rs:348-357)
                    for (uint32_t j = 0; j < count; ++j) {
                        const float4* p = reinterpret_cast<const float4*>(&g_tris[tri_base + start + j]);
                        const float4 ta = p[0], tb = p[1], tc = p[2];
                        const V3 v0 = v3(ta.x, ta.y, ta.z), edge1 = v3(ta.w, tb.x, tb.y), edge2 = v3(tb.z, tb.w, tc.x);
                        const V3 h = cross(d, edge2);
                        const float a = dot(edge1, h);
                        const float f = 1.0f / a;
                        const V3 sv = o - v0;
                        const float u = f * dot(sv, h);
                        const V3 q = cross(sv, edge1);
                        const float v = f * dot(d, q);
                        const float t = f * dot(edge2, q);
                        const bool hit = !(fabsf(a) < 1e-8f) && (u >= 0.0f && u <= 1.0f) && !(v < 0.0f || u + v > 1.0f) &&
                                         (t >= tmin && t <= PT_INF);
                        if (ANY) { if (act && hit) found = true; }
                        else if (act && hit && t < ct) { ct = t; m_tri = __float_as_uint(tc.y); m_u = u; m_v = v; }
                    }
                    if (ANY && __ballot(valid && !found) == 0ull) { sp = 0; break; }     // every ray of the packet is occluded
                } else {
                    // TLAS leaf: this primitive now, the rest afterwards and uncullable (the reference's inner loop)
                    if (count > 1u) pk_write(st, sp++, LEAF_FLAG | ((count - 2u) << PRIM_COUNT_SHIFT) | (start + 1u), act ? NEG_INF_BITS : POS_INF_BITS);
                    const float4* p = reinterpret_cast<const float4*>(&g_inst_tlas[start]);
                    const float4 ia = p[0], ib = p[1], ic = p[2], b0 = p[3], b1 = p[4];
                    const uint32_t inst_idx = g_tlas_prims[start];
                    const V3 lo = xf_point(ia, ib, ic, o);
                    const V3 ld = xf_vector(ia, ib, ic, d);
                    const V3 linv = v3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
                    const bool lexact = ray_needs_exact(lo, linv);
                    float tb_enter;
                    bool bhit;
                    if (__ballot(act && lexact) != 0ull) bhit = box_test<true>(b0, make_float4(b1.x, b1.y, 0.0f, 0.0f), lo, linv, tmin, PT_INF, &tb_enter);
                    else bhit = box_test<false>(b0, make_float4(b1.x, b1.y, 0.0f, 0.0f), lo, linv, tmin, PT_INF, &tb_enter);
                    const bool uncert = (__float_as_uint(b1.z) & INST_UNCERT) != 0u;      // wave-uniform: shape certificate failed, no forwarding
                    const float bound = (ANY || exact || lexact || uncert) ? PT_INF : forwarded_bound(S, o, d, ct);
                    const bool enter = act && bhit && (ANY || !(tb_enter >= bound));
                    const unsigned long long em = __ballot(enter);
                    if (em == 0ull) continue;
                    // park the world level under a marker entry and descend into the mesh BVH
                    // (TriangleMesh::hit: closest_t = local t_max = +inf; root popped with dist 0)
                    if (enter) { w_inv = inv; w_ct = ct; w_exact = exact; }
                    entered = em;
                    cur_inst = inst_idx;
                    tri_base = __float_as_uint(b1.w);
                    marker_sp = sp;
                    pk_write(st, sp++, 0u, 0u);
                    pk_write(st, sp++, (__ballot(enter && lexact) != 0ull ? g_root_faithful : g_root_packet)[start], enter ? 0u : POS_INF_BITS);
                    if (enter) { o = lo; d = ld; inv = linv; exact = lexact; ct = bound; m_tri = PTX_MISS; }
                    in_mesh = true;
                }
            } else {
                // ---- node.intersect for every ray that takes part + the slot loop's pushes: the hit children in
                // reverse list order, so that they pop in list order (leaf slots ascending, internal slots descending)
                const uint32_t nchild = ((word >> NODE_NCHILD_SHIFT) & NODE_NCHILD_MASK) + 1u;      // wave-uniform
                const float4* p = reinterpret_cast<const float4*>(&g_nodes[word & NODE_INDEX_MASK]);
                const bool ex = __ballot(act && exact) != 0ull;
                if (nchild > 4u) {
                    const float4* q = p + 8;
                    const float4 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3], c0 = q[4], c1 = q[5], d0 = q[6], d1 = q[7];
                    pk_unit(st, sp, act, ex, nchild - 4u, o, inv, ct, a0, a1, b0, b1, c0, c1, d0, d1);
                }
                {
                    const float4 a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3], c0 = p[4], c1 = p[5];
                    float4 d0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), d1 = d0;
                    if (nchild > 3u) { d0 = p[6]; d1 = p[7]; }
                    pk_unit(st, sp, act, ex, nchild > 4u ? 4u : nchild, o, inv, ct, a0, a1, b0, b1, c0, c1, d0, d1);
                }
            }
        }
        if (overflow) {
            if (redo_keys) {
                const unsigned long long vm = __ballot(valid);
                uint32_t rb = 0;
                if (lane == 0u) rb = atomicAdd(redo_count, (uint32_t)__popcll(vm));
                rb = __builtin_amdgcn_readfirstlane(rb);
                if (valid) redo_keys[rb + __builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0u))] = key;
            } else {
                err = 1u;
                if (valid) sink.done(key, false);
            }
        } else if (valid) sink.done(key, found);
    }
    // measurement only: most waves find a value at least as large already there and skip the atomic
    if (stack_hw && lane == 0u && (uint32_t)hw > *stack_hw) atomicMax(stack_hw, (uint32_t)hw);
}

}  // namespace ptd
