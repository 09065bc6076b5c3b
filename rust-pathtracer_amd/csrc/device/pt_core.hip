// pt_core.hip — gfx950 wavefront kernels + the C ABI of include/ptx_core.h.
//
// Kernel pipeline of one pass (one sample for every path slot):
//   k_raygen    blue-noise jitter -> camera ray -> RNG seed      (R1-R3)
//   loop over path-tracing iterations:
//     k_extend    closest-hit traversal for every live path      (T1-T13)
//     k_shade     pending adds, hit data, material, MIS integrator step, SSS
//                 state machine; pushes shadow rays / next / finished queues
//                 with wave64 ballot + mbcnt prefix compaction     (I1-I13)
//     k_shadow    any-hit traversal for the NEE shadow-ray queue   (T7)
//     k_finalize  finished paths: apply pending adds, clamp_radiance,
//                 accumulate into sum / sum_sq / count             (A1/A2)
// Path state is AoS-of-float4 in HBM (one 16-byte access per field group per
// lane, so sparse survivors touch few lines); queues hold path-slot indices.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "pt_shade.h"
#include "pt_traverse.h"
#include "pt_traverse_group.h"

using namespace ptd;

namespace {

// ============================================================== path state =====
constexpr uint32_t ST_NORMAL = 0, ST_THICK = 1, ST_WALK = 2;   // ST_THICK / ST_WALK paths live on the walk queue (k_sss_step)
constexpr uint32_t FL_BOUNCE_MASK = 0xFFu, FL_SPECULAR = 1u << 8, FL_STATE_SHIFT = 9, FL_STATE_MASK = 3u << 9,
                   FL_FIRST_HIT = 1u << 11, FL_STEP_SHIFT = 16, FL_STEP_MASK = 0xFFu << 16;

struct PathArrays {
    float4* ray_o;     // origin.xyz, t_max
    float4* ray_d;     // direction.xyz, last_brdf_pdf
    float4* thr;       // throughput.xyz, flags
    float4* rad;       // radiance.xyz, pixel index
    uint2* rng;        // PCG state
    float4* hit;       // t, u, v, inst
    uint32_t* hit_tri;
    float4* sh_o;      // shadow-ray origin
    float4* sh_d;      // sky shadow-ray direction
    float4* c_sun;     // pending contribution behind the sun shadow ray
    float4* c_sky;     // pending contribution behind the sky shadow ray
    float4* c_post;    // BSDF-sampled sun hit (added after the NEE terms, pathtracer.rs:319-322)
    float4* w_t;       // SSS walk throughput.xyz, thickness_scale
    float4* w_b;       // base_color.xyz of the entry vertex, material id
    float4* aov_a;     // first-hit albedo (AOV pass only)
    float4* aov_n;
    uint32_t* sample;  // sample_index of the slot
};

// per-iteration counter block
// (the queue cursors are arrays: one per queue segment, see trace_queue)
struct IterCtr { uint32_t n_active, n_shadow, n_final, n_walk; uint32_t w_extend[QUEUE_SEGMENTS], w_shadow[QUEUE_SEGMENTS], w_walk[QUEUE_SEGMENTS]; };
constexpr uint32_t MAX_ITERS = 4096;
// counter blocks past the per-iteration range [0, MAX_ITERS): one per helper, so none aliases an iteration's block or
// another helper's
constexpr uint32_t CTR_REDO = MAX_ITERS, CTR_STATS = MAX_ITERS + 1, CTR_BATCH = MAX_ITERS + 2, CTR_ADAPTIVE = MAX_ITERS + 3,
                   CTR_TAIL = MAX_ITERS + 4, CTR_BLOCKS = MAX_ITERS + 5;
// P.rad[].w = pixel index | which pending contributions of the previous step are stored (an absent one is +0.0: it is
// neither written nor read back, which is most of them — the BSDF-sampled sun term almost always, the NEE terms whenever
// the light is below the surface)
constexpr uint32_t PEND_SUN = 1u << 29, PEND_SKY = 1u << 30, PEND_POST = 1u << 31, PIXEL_MASK = (1u << 29) - 1u;

#ifndef PT_PACKET_GRID_WAVES
#define PT_PACKET_GRID_WAVES PT_WAVES_PER_SIMD
#endif
constexpr int PACKET_GRID = 256 * 4 * PT_PACKET_GRID_WAVES;   // the packet kernel needs fewer registers: more resident waves
#ifndef PT_SHADOW_WAVES
#define PT_SHADOW_WAVES PT_WAVES_PER_SIMD
#endif
// persistent waves per CU of the closest-hit / any-hit launches: what the VGPR budget (PT_WAVES_PER_SIMD per SIMD) and the
// LDS stacks (6 KiB per wave: 26 waves in a CU's 160 KiB) let be resident together
#ifndef PT_TRACE_GRID_PER_CU
#define PT_TRACE_GRID_PER_CU (4 * PT_WAVES_PER_SIMD)
#endif
#ifndef PT_SHADOW_GRID_PER_CU
#define PT_SHADOW_GRID_PER_CU (4 * PT_SHADOW_WAVES)
#endif
constexpr int SHADOW_GRID = 256 * PT_SHADOW_GRID_PER_CU;    // the any-hit kernel carries less state per lane
constexpr int TRACE_GRID = 256 * PT_TRACE_GRID_PER_CU;
constexpr int MAX_TRACE_GRID = SHADOW_GRID > TRACE_GRID ? SHADOW_GRID : TRACE_GRID;
#ifndef PT_SHADE_GRID_PER_CU
#define PT_SHADE_GRID_PER_CU (1 << 20)
#endif
#ifndef PT_SHADE_BLOCK
#define PT_SHADE_BLOCK 256
#endif
constexpr int SHADE_BLOCK = PT_SHADE_BLOCK;
// the uncounted k_shade / k_sss_step have no grid-stride loop: their grids must cover any queue in one stride
static_assert((unsigned long long)256 * PT_SHADE_GRID_PER_CU * PT_SHADE_BLOCK >= (1ull << 32), "k_shade<false, ..> visits one entry per thread: its grid may not be capped");
// k_shade is a chain of dependent gathers: it runs at the speed its resident waves allow.  The attribute only sets the
// floor (3 waves/SIMD = 168 VGPRs, what the kernel needed while a grid-stride loop kept every kernel argument live across
// its body); the uncounted kernels have no such loop any more and take 91-102 VGPRs: 4 waves/SIMD, no scratch (5 waves
// were measured with the 36 B of scratch that costs: slower).  DESIGN.md 4, "What the compiler was doing".
#ifndef PT_SHADE_ATTR
#define PT_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#endif
// traversal kernels are compiled for PT_WAVES_PER_SIMD resident waves per SIMD (80 VGPRs at 6)
#ifndef PT_PACKET_WAVES
#define PT_PACKET_WAVES 5
#endif
#ifndef PT_PACKET_ATTR
#define PT_PACKET_ATTR __attribute__((amdgpu_waves_per_eu(PT_PACKET_WAVES)))
#endif
#ifndef PT_TRACE_ATTR
#define PT_TRACE_ATTR __attribute__((amdgpu_waves_per_eu(PT_WAVES_PER_SIMD)))
#endif
#ifndef PT_SHADOW_ATTR
#define PT_SHADOW_ATTR __attribute__((amdgpu_waves_per_eu(PT_SHADOW_WAVES)))
#endif

struct Target {
    float* out_rgb;        // by pixel (or by slot) *3, overwritten          (render_pass / render_pixels)
    float* out_albedo;     // AOV, by pixel
    float* out_normal;
    float* sum;            // framebuffer accumulate (+=)
    float* sum_sq;
    uint32_t* counts;
    uint32_t by_slot;
    uint32_t defer;        // several samples per pixel in flight: park the clamped radiance in P.rad, k_accumulate_batch adds it in sample order
};

// ---- wave64 ballot / prefix queue push ------------------------------------------
PT_DEV void queue_push(uint32_t* q, uint32_t* counter, bool pred, uint32_t value) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (pred) q[base + rank] = value;
}

// rank of this lane among the set bits of a wave64 ballot
PT_DEV uint32_t wave_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

PT_DEV void wave_add(unsigned long long* dst, uint32_t v) {
    // wave reduction then one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63u) == 0u && v) atomicAdd(dst, (unsigned long long)v);
}

PT_DEV StackRef make_stack(uint2* ovf) {
    StackRef st;
    st.ovf = ovf;
    st.ovf_stride = (uint32_t)(gridDim.x * TRACE_BLOCK);
    return st;
}

PT_DEV void flush_lane_cnt(DevCounters* dc, const LaneCnt& lc, uint32_t rays, bool any) {
    wave_add(&dc->v[any ? C_RAYS_ANY : C_RAYS_CLOSEST], rays);
    wave_add(&dc->v[C_NODE_VISITS], lc.node_visits);
    wave_add(&dc->v[C_NODE_CHILDREN], lc.node_children);
    wave_add(&dc->v[C_PRIM_INDEX_READS], lc.prim_reads);
    wave_add(&dc->v[C_TRI_TESTS], lc.tri_tests);
    wave_add(&dc->v[C_INST_ENTRIES], lc.inst_entries);
    wave_add(&dc->v[C_INST_HITS], lc.inst_hits);
    { uint32_t hw = lc.stack_hw;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor(hw, off); hw = o > hw ? o : hw; }
      if ((threadIdx.x & 63u) == 0u && hw) atomicMax(&dc->dbg[0], hw); }
    { float us = lc.undershoot;      // >= 0: non-negative floats order like their bit patterns
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(us, off); us = o > us ? o : us; }
      if ((threadIdx.x & 63u) == 0u && us > 0.0f) atomicMax(&dc->dbg[4], __float_as_uint(us)); }
    { float us = lc.undershoot_x;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(us, off); us = o > us ? o : us; }
      if ((threadIdx.x & 63u) == 0u && us > 0.0f) atomicMax(&dc->dbg[5], __float_as_uint(us)); }
    if ((threadIdx.x & 63u) == 0u)
        for (int k = 0; k < 9; ++k) atomicAdd(&dc->vote[(any ? 9 : 0) + k], (unsigned long long)lc.vote[k]);
    wave_add(&dc->stk[0], lc.stk_reads); wave_add(&dc->stk[1], lc.stk_reads_hbm);
    wave_add(&dc->stk[2], lc.stk_writes); wave_add(&dc->stk[3], lc.stk_writes_hbm);
    if (any) {
        wave_add(&dc->v[C_ANY_NODE_VISITS], lc.node_visits);
        wave_add(&dc->v[C_ANY_NODE_CHILDREN], lc.node_children);
        wave_add(&dc->v[C_ANY_PRIM_INDEX_READS], lc.prim_reads);
        wave_add(&dc->v[C_ANY_TRI_TESTS], lc.tri_tests);
        wave_add(&dc->v[C_ANY_INST_ENTRIES], lc.inst_entries);
    }
}

// ================================================================ kernels ======

// ---- ray batches (ptx_trace_batch) ------------------------------------------------
struct BatchSrc {
    const PtxRay* rays;
    PT_DEV uint32_t key(uint32_t i) const { return i; }
    PT_DEV void fetch(uint32_t i, V3* o, V3* d, float* tmax) const {
        const PtxRay r = rays[i];
        *o = v3(r.origin[0], r.origin[1], r.origin[2]);
        *d = v3(r.direction[0], r.direction[1], r.direction[2]);
        *tmax = r.t_max;
    }
};
template <bool ANY>
struct BatchSink {
    PtxHitRecord* out_hits;
    uint8_t* out_any;
    PT_DEV void hit(uint32_t i, uint32_t inst, uint32_t tri, float t, float u, float v) const {
        PtxHitRecord o; o.inst_idx = inst; o.tri_idx = tri; o.t = t; o.u = u; o.v = v; out_hits[i] = o;
    }
    PT_DEV void done(uint32_t i, bool found) const {
        if (ANY) out_any[i] = found ? 1 : 0;
        else if (!found) hit(i, PTX_MISS, PTX_MISS, PT_INF, 0.0f, 0.0f);
    }
};
template <bool ANY, bool CNT, bool FAST = false, bool UNC = false>
__global__ __launch_bounds__(TRACE_BLOCK) PT_TRACE_ATTR void k_trace_batch(SceneDev S, const PtxRay* rays, uint32_t n, PtxHitRecord* out_hits,
                                                             uint8_t* out_any, uint32_t* work, uint2* ovf, DevCounters* dc, uint32_t* err) {
    const StackRef st = make_stack(ovf);
    LaneCnt lc = {0, 0, 0, 0, 0, 0, 0, 0.0f, {0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0, 0, 0.0f};
    uint32_t nrays = 0;
    BatchSrc src{rays};
    BatchSink<ANY> sink{out_hits, out_any};
    trace_queue<ANY, CNT, FAST, UNC>(S, st, work, n, src, sink, lc, nrays, err, reinterpret_cast<unsigned long long*>(&dc->dbg[6]));
    if (CNT) flush_lane_cnt(dc, lc, nrays, ANY);
}

// ---- ray-gen: render_pixel up to renderer.render (renderloop/parallel.rs:31-56) ----
__global__ void k_raygen(SceneDev S, CameraDev cam, PathArrays P, const uint32_t* wl_pixel, const uint32_t* wl_sample,
                         const uint32_t* fb_counts, uint32_t uniform_sample, uint32_t n, uint32_t width, uint32_t height,
                         uint32_t* q, IterCtr* ctr0, uint32_t aov, PtxRay* out_rays, uint32_t batch) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && ctr0) ctr0->n_active = n;
    if (i >= n) return;
    // batch > 1: slots [j*batch, (j+1)*batch) are `batch` consecutive samples of work-list pixel j (pixel-major, so a
    // stretch of the queue is a region of the image)
    const uint32_t j = batch > 1u ? i / batch : i, k = batch > 1u ? i - j * batch : 0u;
    const uint32_t pixel = wl_pixel[j];
    // sample index: explicit list (render_pixels), the pixel's own count (adaptive), or the pass's; + k within a batch
    const uint32_t sample_index = wl_sample ? wl_sample[i] : (fb_counts ? fb_counts[pixel] : uniform_sample) + k;
    const uint32_t x = pixel % width, y = pixel / width;
    float px, py, lx, ly;
    bluenoise_sample(S.bn_pixel, S.bn_size, x, y, sample_index, &px, &py);
    bluenoise_sample(S.bn_lens, S.bn_size, x, y, sample_index, &lx, &ly);
    const float u = ((float)x + px) / (float)width;
    const float v = ((float)y + py) / (float)height;
    V3 o, d;
    generate_ray_dof(cam, u, v, lx, ly, &o, &d);
    if (out_rays) {
        PtxRay r;
        r.origin[0] = o.x; r.origin[1] = o.y; r.origin[2] = o.z;
        r.direction[0] = d.x; r.direction[1] = d.y; r.direction[2] = d.z;
        r.t_max = PT_INF;
        out_rays[i] = r;
        return;
    }
    const Rng rng = rng_new(path_seed(o, d, sample_index));
    P.ray_o[i] = make_float4(o.x, o.y, o.z, PT_INF);
    P.ray_d[i] = make_float4(d.x, d.y, d.z, 0.0f);
    P.thr[i] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(aov ? FL_FIRST_HIT : 0u));
    P.rad[i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(pixel));
    P.rng[i] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    P.sample[i] = sample_index;
    if (aov) { P.aov_a[i] = z; P.aov_n[i] = z; }
    q[i] = i;
}

// ---- extend: Scene::hit for every live path ----------------------------------------
struct ExtendSrc {
    const PathArrays& P;
    const uint32_t* q;
    // the launch's queue is the iteration's active queue followed by its walk queue (subsurface step rays): one launch,
    // one set of persistent waves, no tail between the two
    uint32_t n_first = 0xFFFFFFFFu;
    const uint32_t* q_second = nullptr;
    PT_DEV uint32_t key(uint32_t i) const { return i < n_first ? q[i] : q_second[i - n_first]; }     // the path slot
    PT_DEV void fetch(uint32_t p, V3* o, V3* d, float* tmax) const {
        const float4 ro = P.ray_o[p], rd = P.ray_d[p];
        *o = v3(ro.x, ro.y, ro.z); *d = v3(rd.x, rd.y, rd.z); *tmax = ro.w;
    }
};
struct ExtendSink {
    const PathArrays& P;
    PT_DEV void hit(uint32_t p, uint32_t inst, uint32_t tri, float t, float u, float v) const {
        P.hit[p] = make_float4(t, u, v, __uint_as_float(inst));
        P.hit_tri[p] = tri;
    }
    PT_DEV void done(uint32_t i, bool found) const {
        if (!found) hit(i, PTX_MISS, PTX_MISS, PT_INF, 0.0f, 0.0f);
    }
};
// FAST: nearest child first + the order-independence check (pt_traverse.h); a ray that fails the check is traced again, in reference order, by its lane
// UNC: the scene holds meshes that failed the shape certificate (the FAST walk then looks for their flags in the TLAS)
template <bool CNT, bool FAST = false, bool UNC = false>
__global__ __launch_bounds__(TRACE_BLOCK) PT_TRACE_ATTR void k_extend(SceneDev S, PathArrays P, const uint32_t* q, const uint32_t* q_walk, IterCtr* ctr, uint2* ovf,
                                                        DevCounters* dc, uint32_t* err) {
    const StackRef st = make_stack(ovf);
    LaneCnt lc = {0, 0, 0, 0, 0, 0, 0, 0.0f, {0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0, 0, 0.0f};
    uint32_t nrays = 0;
    const uint32_t n_active = ctr->n_active, n_walk = ctr->n_walk;
    ExtendSrc src{P, q, n_active, q_walk};
    ExtendSink sink{P};
    trace_queue<false, CNT, FAST, UNC>(S, st, ctr->w_extend, n_active + n_walk, src, sink, lc, nrays, err, reinterpret_cast<unsigned long long*>(&dc->dbg[6]));
    if (CNT) flush_lane_cnt(dc, lc, nrays, false);
}

// Experiment (PTX_GROUP_LANES=4|8, DESIGN.md "G lanes per ray"): the closest-hit launches with G lanes per ray on the G-wide tree.
template <int G>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(5))) void k_extend_group(SceneDev S, PathArrays P, const uint32_t* q, const uint32_t* q_walk, IterCtr* ctr, uint32_t* err) {
    const uint32_t n_active = ctr->n_active, n_walk = ctr->n_walk;
    ExtendSrc src{P, q, n_active, q_walk};
    ExtendSink sink{P};
    trace_groups<G>(S, ctr->w_extend, n_active + n_walk, src, sink, err);
}
template <int G>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(5))) void k_trace_group(SceneDev S, const PtxRay* rays, uint32_t n, PtxHitRecord* out_hits, uint32_t* work, uint32_t* err) {
    BatchSrc src{rays};
    BatchSink<false> sink{out_hits, nullptr};
    trace_groups<G>(S, work, n, src, sink, err);
}

// Coherent queues (the primary rays of a wavefront batch): the wave's 64 rays walk the tree as one packet.
__global__ __launch_bounds__(TRACE_BLOCK) PT_PACKET_ATTR void k_extend_packet(SceneDev S, PathArrays P, const uint32_t* q, IterCtr* ctr, uint2* ovf,
                                                               uint32_t* err, const NodeRec* __restrict__ g_nodes, const TriRec* __restrict__ g_tris,
                                                               const InstRec* __restrict__ g_inst_tlas, const InstRec* __restrict__ g_inst,
                                                               const uint32_t* __restrict__ g_tlas_prims, int capacity, uint32_t* redo_keys,
                                                               IterCtr* redo_ctr, DevCounters* dc) {
    const StackRef st = make_stack(ovf);
    uint32_t e = 0;
    ExtendSrc src{P, q};
    ExtendSink sink{P};
    trace_packets<false>(S, g_nodes, g_tris, g_inst_tlas, g_inst, g_tlas_prims, S.inst_root_faithful, S.inst_root_packet, st, ctr->w_extend, ctr->n_active, src, sink, e, capacity,
                         redo_keys, &redo_ctr->n_active, &dc->dbg[1]);
    if (e) atomicOr(err, e);
}
// The rays of packets whose shared stack overflowed, traced one by one (normally none: the launch is a few waves
// that find an empty queue).  Its own name so that profiles and timings keep k_extend's launches apart.
__global__ __launch_bounds__(TRACE_BLOCK) PT_TRACE_ATTR void k_extend_redo(SceneDev S, PathArrays P, const uint32_t* q, IterCtr* ctr, uint2* ovf,
                                                             DevCounters* dc, uint32_t* err) {
    const StackRef st = make_stack(ovf);
    LaneCnt lc = {0, 0, 0, 0, 0, 0, 0, 0.0f, {0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0, 0, 0.0f};
    uint32_t nrays = 0;
    ExtendSrc src{P, q};
    ExtendSink sink{P};
    trace_queue<false, false>(S, st, ctr->w_extend, ctr->n_active, src, sink, lc, nrays, err);
    if (blockIdx.x == 0 && threadIdx.x == 0 && ctr->n_active)     // measurement: rays that took the redo road
        atomicAdd(reinterpret_cast<unsigned long long*>(&dc->dbg[2]), (unsigned long long)ctr->n_active);
}
template <bool ANY>
__global__ __launch_bounds__(TRACE_BLOCK) PT_PACKET_ATTR void k_trace_batch_packet(SceneDev S, const PtxRay* rays, uint32_t n, PtxHitRecord* out_hits,
                                                                    uint8_t* out_any, uint32_t* work, uint2* ovf, uint32_t* err, DevCounters* dc) {
    const StackRef st = make_stack(ovf);
    uint32_t e = 0;
    BatchSrc src{rays};
    BatchSink<ANY> sink{out_hits, out_any};
    trace_packets<ANY>(S, S.nodes, S.tris, S.inst_tlas, S.inst, S.tlas_prims, S.inst_root_faithful, S.inst_root_packet, st, work, n, src, sink, e, PACKET_LDS + OVF_STACK, nullptr, nullptr, &dc->dbg[1]);
    if (e) atomicOr(err, e);
}

// ---- shadow: Scene::hit_any for the NEE queue ---------------------------------------
struct ShadowSrc {
    const SceneDev& S;
    const PathArrays& P;
    const uint32_t* qs;
    PT_DEV uint32_t key(uint32_t i) const { return qs[i]; }    // path slot * 2 + (0 sun, 1 sky)
    PT_DEV void fetch(uint32_t entry, V3* o, V3* d, float* tmax) const {
        const uint32_t p = entry >> 1;
        const float4 so = P.sh_o[p];
        *o = v3(so.x, so.y, so.z);
        if ((entry & 1u) == 0u) *d = v3(S.sun_dir[0], S.sun_dir[1], S.sun_dir[2]);
        else { const float4 sd = P.sh_d[p]; *d = v3(sd.x, sd.y, sd.z); }
        *tmax = PT_INF;   // every shadow ray of the integrator is Ray::new: infinite
    }
};
struct ShadowSink {
    const PathArrays& P;
    PT_DEV void hit(uint32_t, uint32_t, uint32_t, float, float, float) const {}
    PT_DEV void done(uint32_t entry, bool found) const {
        if (!found) return;   // unoccluded: the pending contribution stands
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if ((entry & 1u) == 0u) P.c_sun[entry >> 1] = z; else P.c_sky[entry >> 1] = z;
    }
};
template <bool CNT>
__global__ __launch_bounds__(TRACE_BLOCK) PT_SHADOW_ATTR void k_shadow(SceneDev S, PathArrays P, const uint32_t* qs, IterCtr* ctr, uint2* ovf,
                                                        DevCounters* dc, uint32_t* err) {
    const StackRef st = make_stack(ovf);
    LaneCnt lc = {0, 0, 0, 0, 0, 0, 0, 0.0f, {0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0, 0, 0.0f};
    uint32_t nrays = 0;
    ShadowSrc src{S, P, qs};
    ShadowSink sink{P};
    trace_queue<true, CNT>(S, st, ctr->w_shadow, ctr->n_shadow, src, sink, lc, nrays, err);
    if (CNT) flush_lane_cnt(dc, lc, nrays, true);
}

// The shadow rays of the primary hits: a pixel's 64 samples start within its footprint and aim at the sun or at the
// bright part of the sky, so they too walk the tree as packets.
__global__ __launch_bounds__(TRACE_BLOCK) PT_PACKET_ATTR void k_shadow_packet(SceneDev S, PathArrays P, const uint32_t* qs, IterCtr* ctr, uint2* ovf,
                                                               uint32_t* err, const NodeRec* __restrict__ g_nodes, const TriRec* __restrict__ g_tris,
                                                               const InstRec* __restrict__ g_inst_tlas, const InstRec* __restrict__ g_inst,
                                                               const uint32_t* __restrict__ g_tlas_prims) {
    const StackRef st = make_stack(ovf);
    uint32_t e = 0;
    ShadowSrc src{S, P, qs};
    ShadowSink sink{P};
    trace_packets<true>(S, g_nodes, g_tris, g_inst_tlas, g_inst, g_tlas_prims, S.inst_root_faithful, S.inst_root_packet, st, ctr->w_shadow, ctr->n_shadow, src, sink, e);
    if (e) atomicOr(err, e);
}

// ---- shade: one step of trace_inner's state machine ---------------------------------
struct WalkCtx { V3 sigma_t, sigma_s; };
PT_DEV WalkCtx walk_ctx(const PtxMaterial& m, float scale) {
    // scaled_radius = radius * thickness_scale (pathtracer.rs:218-219); sigma (subsurface.rs:67-72)
    const V3 radius = v3(m.subsurface_radius[0], m.subsurface_radius[1], m.subsurface_radius[2]) * scale;
    WalkCtx w;
    w.sigma_t = v3(1.0f / f_max(radius.x, 0.0001f), 1.0f / f_max(radius.y, 0.0001f), 1.0f / f_max(radius.z, 0.0001f));
    w.sigma_s = w.sigma_t * v3(m.subsurface_albedo[0], m.subsurface_albedo[1], m.subsurface_albedo[2]);
    return w;
}

// head of one `for _ in 0..SSS_MAX_STEPS` iteration (subsurface.rs:78-98): pick the channel, sample the free-flight
// distance; false = `sum_weights < 1e-10` (the walk returns None)
PT_DEV bool walk_begin_step(const PtxMaterial& material, V3 wt, float w_scale, Rng& rng, V3 w_pos, V3 w_dir, V3* o, V3* d, float* dist) {
    const WalkCtx wc = walk_ctx(material, w_scale);
    const V3 weights = wt * wc.sigma_s;
    const float sum_weights = weights.x + weights.y + weights.z;
    if (sum_weights < 1e-10f) return false;
    const float r = rng.next() * sum_weights;
    const int channel = (r < weights.x) ? 0 : ((r < weights.x + weights.y) ? 1 : 2);
    const float sigma_t_c = comp(wc.sigma_t, channel);
    *dist = -d_logf(f_max(rng.next(), 1e-10f)) / sigma_t_c;
    *o = w_pos; *d = w_dir;
    return true;
}

// ---- the subsurface states of trace_inner (pathtracer.rs:203-271, subsurface.rs:58-160) ---------------------------
// A path that refracts into a subsurface medium leaves k_shade for the WALK QUEUE and stays there until the event is
// over; its rays (thickness ray, then one step ray per walk step, t_max = the sampled distance) are traced by k_extend
// together with the active queue, and this kernel does one state transition per iteration for every path on the queue:
//   ST_THICK  thickness ray traced -> thickness scale, head of the first walk step
//   ST_WALK   step ray traced -> leave through an exiting surface / hop through an entering one / scatter, then the
//             head of the next step.  At an exiting surface the event ends here as well: exit tint, Russian roulette,
//             sun and sky light sampling at the exit point (shadow rays go to this iteration's shadow queue), cosine
//             bounce — and the path returns to the next iteration's ACTIVE queue in state ST_NORMAL.
// A walk that returns None retires the path.  The integrator kernel never sees a subsurface state, so it carries none of
// this code or its registers, and the walk code runs on full waves instead of on the few walk lanes of a mixed wave.
#ifndef PT_WALK_BLOCK
#define PT_WALK_BLOCK 256
#endif
constexpr int WALK_BLOCK = PT_WALK_BLOCK;
// The body of k_sss_step over queue entries first, first + stride, ... < n, for a block of BLOCK threads: the kernel runs
// it grid-wide, the tail kernel (k_tail) on a wave's own paths.  c_* = the counters of the four queues it pushes to.
// ONCE: as in shade_range (the caller's grid covers the range in one stride)
template <bool CNT, int BLOCK, bool ONCE = false>
PT_DEV void sss_range(const SceneDev& S, const PathArrays& P, const uint32_t* q_walk, const uint32_t n, const uint32_t first, const uint32_t stride,
                      uint32_t* q_walk_next, uint32_t* q_next, uint32_t* q_shadow, uint32_t* q_final,
                      uint32_t* c_shadow, uint32_t* c_next, uint32_t* c_final, uint32_t* c_walk, const uint32_t max_bounces, ShadeCnt& sc) {
    __shared__ uint32_t s_cnt[4][BLOCK / 64];
    __shared__ uint32_t s_base[4];
    for (uint32_t base = first; base < n; base += stride) {
        const uint32_t i = base + threadIdx.x;
        const bool active = i < n;
        bool alive = false, exited = false, push_sun = false, push_sky = false;
        uint32_t p = 0;
        if (active) {
            p = q_walk[i];
            const float4 ro4 = P.ray_o[p], rd4 = P.ray_d[p], wb4 = P.w_b[p], hi4 = P.hit[p];
            const uint2 rs = P.rng[p];
            Rng rng;
            rng.state = (uint64_t)rs.x | ((uint64_t)rs.y << 32);
            uint32_t* flags_word = reinterpret_cast<uint32_t*>(&P.thr[p]) + 3;
            uint32_t flags = *flags_word;
            uint32_t step = (flags & FL_STEP_MASK) >> FL_STEP_SHIFT;
            const uint32_t state = (flags & FL_STATE_MASK) >> FL_STATE_SHIFT;
            V3 w_pos = v3(ro4.x, ro4.y, ro4.z), w_dir = v3(rd4.x, rd4.y, rd4.z);
            const float dist = ro4.w;
            float last_pdf = rd4.w;
            const bool has_hit = __float_as_uint(hi4.w) != PTX_MISS;
            const PtxMaterial& material = S.materials[__float_as_uint(wb4.w)];
            V3 wt = splat(1.0f);
            float w_scale = 1.0f;
            alive = true;
            V3 exit_point = splat(0.0f), exit_n = splat(0.0f);
            if (state == ST_THICK) {
                // thickness ray (pathtracer.rs:207-219): the walk starts at the entry point along the entry direction
                float thickness = 1.0f;
                if (has_hit) {
                    const HitData bh = compute_hit_data(S, __float_as_uint(hi4.w), P.hit_tri[p], hi4.x, hi4.y, hi4.z);
                    if (CNT) sc.hitdata++;
                    if (dot(w_dir, bh.normal) > 0.0f) thickness = bh.t;
                }
                w_scale = f_clamp(thickness / 1.0f, 0.1f, 2.0f);
                step = 0;
            } else {
                const float4 wt4 = P.w_t[p];
                wt = v3(wt4.x, wt4.y, wt4.z);
                w_scale = wt4.w;
                const WalkCtx wc = walk_ctx(material, w_scale);
                const V3 weights = wt * wc.sigma_s;
                const float sum_weights = weights.x + weights.y + weights.z;
                if (has_hit) {
                    const HitData wh = compute_hit_data(S, __float_as_uint(hi4.w), P.hit_tri[p], hi4.x, hi4.y, hi4.z);
                    if (CNT) sc.hitdata++;
                    const V3 hit_point = w_pos + w_dir * wh.t;
                    if (dot(w_dir, wh.normal) > 0.0f) {
                        // exiting surface (subsurface.rs:104-117)
                        const float t = wh.t;
                        const V3 tr = v3(d_expf(-wc.sigma_t.x * t), d_expf(-wc.sigma_t.y * t), d_expf(-wc.sigma_t.z * t));
                        const float pdf_sum = (tr.x * (weights.x / sum_weights) + tr.y * (weights.y / sum_weights)) + tr.z * (weights.z / sum_weights);
                        wt = wt * (tr / f_max(pdf_sum, 1e-10f));
                        exit_point = hit_point; exit_n = wh.normal;
                        exited = true;
                    } else {
                        w_pos = hit_point + w_dir * RAY_EPSILON;       // entering surface: hop over it, the step is spent
                    }
                } else {
                    // no surface within the sampled distance: scatter there (subsurface.rs:125-157)
                    w_pos = w_pos + w_dir * dist;
                    const V3 tr = v3(d_expf(-wc.sigma_t.x * dist), d_expf(-wc.sigma_t.y * dist), d_expf(-wc.sigma_t.z * dist));
                    const float pdf_sum = (wc.sigma_t.x * tr.x * (weights.x / sum_weights) + wc.sigma_t.y * tr.y * (weights.y / sum_weights)) +
                                          wc.sigma_t.z * tr.z * (weights.z / sum_weights);
                    wt = wt * (tr * wc.sigma_s / f_max(pdf_sum, 1e-10f));
                    const float max_tp = max_element(wt);
                    if (max_tp < 0.1f) {
                        const float survive = f_min(max_tp * 10.0f, 1.0f);
                        if (rng.next() > survive) alive = false;
                        else wt = wt / survive;
                    }
                    if (alive && (!is_finite(wt) || max_tp > 1000.0f)) alive = false;
                    if (alive) {
                        // sample_hg_phase (subsurface.rs:41-56)
                        const float g = material.subsurface_anisotropy;
                        float cos_theta;
                        if (fabsf(g) < 0.001f) cos_theta = 1.0f - 2.0f * rng.next();
                        else {
                            const float g2 = g * g;
                            const float sqr = (1.0f - g2) / (1.0f - g + 2.0f * g * rng.next());
                            cos_theta = f_clamp((1.0f + g2 - sqr * sqr) / (2.0f * g), -1.0f, 1.0f);
                        }
                        const float sin_theta = sqrtf(f_max(1.0f - cos_theta * cos_theta, 0.0f));
                        const float phi = 2.0f * PT_PI * rng.next();
                        const V3 nn = -w_dir;
                        V3 tb, bb;
                        build_basis(nn, &tb, &bb);
                        float sp, cp;
                        d_sincosf(phi, &sp, &cp);
                        w_dir = to_world(v3(sin_theta * cp, sin_theta * sp, cos_theta), tb, bb, nn);
                    }
                }
            }
            if (alive && !exited) {
                // head of the next `for _ in 0..SSS_MAX_STEPS` iteration (subsurface.rs:78-98)
                float next_dist = 0.0f;
                if (step >= 64u) alive = false;
                else {
                    if (CNT) sc.sss_steps++;
                    V3 so, sd;
                    if (!walk_begin_step(material, wt, w_scale, rng, w_pos, w_dir, &so, &sd, &next_dist)) alive = false;
                    else step++;
                }
                if (alive) {
                    flags = (flags & ~(FL_STEP_MASK | FL_STATE_MASK)) | ((step << FL_STEP_SHIFT) & FL_STEP_MASK) | (ST_WALK << FL_STATE_SHIFT);
                    *flags_word = flags;
                    P.w_t[p] = make_float4(wt.x, wt.y, wt.z, w_scale);
                    P.rng[p] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                    P.ray_o[p] = make_float4(w_pos.x, w_pos.y, w_pos.z, next_dist);
                    P.ray_d[p] = make_float4(w_dir.x, w_dir.y, w_dir.z, last_pdf);
                }
            }
            if (exited) {
                // success branch of trace_inner (pathtracer.rs:230-271)
                const float4 th4 = P.thr[p];
                V3 T = v3(th4.x, th4.y, th4.z);
                uint32_t bounce = flags & FL_BOUNCE_MASK;
                const V3 sss_base = v3(wb4.x, wb4.y, wb4.z);
                const V3 sun_dir = v3(S.sun_dir[0], S.sun_dir[1], S.sun_dir[2]);
                const V3 sun_color = v3(S.sun_color[0], S.sun_color[1], S.sun_color[2]);
                T = T * (sss_base * wt * v3(material.subsurface_exit_tint[0], material.subsurface_exit_tint[1], material.subsurface_exit_tint[2]));
                if (!russian_roulette(rng, T, bounce)) alive = false;
                else {
                    const V3 diffuse_brdf = sss_base / PT_PI;
                    const V3 sh_o = exit_point + exit_n * RAY_EPSILON;
                    P.sh_o[p] = make_float4(sh_o.x, sh_o.y, sh_o.z, 0.0f);
                    if (S.sun_enabled) {
                        const float n_dot_l = dot(exit_n, sun_dir);
                        if (n_dot_l > 0.0f) { const V3 c_sun = T * diffuse_brdf * sun_color * n_dot_l; push_sun = true; P.c_sun[p] = make_float4(c_sun.x, c_sun.y, c_sun.z, 0.0f); }
                    }
                    {
                        const float u1 = rng.next();
                        const float u2 = rng.next();
                        const SkySample ss = sky_sample_direction<!CNT>(S, u1, u2, sc);
                        const float n_dot_l = dot(exit_n, ss.direction);
                        if (n_dot_l > 0.0f && ss.pdf > 0.0f) {
                            const V3 sky_contribution = clamp_radiance(ss.radiance * safe_inv_pdf(ss.pdf));
                            const V3 c_sky = T * diffuse_brdf * sky_contribution * n_dot_l;
                            push_sky = true;
                            P.c_sky[p] = make_float4(c_sky.x, c_sky.y, c_sky.z, 0.0f);
                            P.sh_d[p] = make_float4(ss.direction.x, ss.direction.y, ss.direction.z, 0.0f);
                        }
                    }
                    V3 et, eb;
                    build_basis(exit_n, &et, &eb);
                    const float u1 = rng.next();
                    const float u2 = rng.next();
                    const V3 exit_dir = to_world(sample_cosine_hemisphere(u1, u2), et, eb, exit_n);
                    const V3 ro = exit_point + exit_n * RAY_EPSILON;
                    last_pdf = pdf_cosine(f_max(dot(exit_n, exit_dir), 0.001f));
                    bounce++;
                    // `for bounce in 0..MAX_BOUNCES` ran out (pathtracer.rs:117)
                    if (bounce >= max_bounces) alive = false;
                    // back to the integrator: ST_NORMAL, not specular, step count cleared; FL_FIRST_HIT is long gone
                    flags = bounce & FL_BOUNCE_MASK;
                    P.ray_o[p] = make_float4(ro.x, ro.y, ro.z, PT_INF);
                    P.ray_d[p] = make_float4(exit_dir.x, exit_dir.y, exit_dir.z, last_pdf);
                    P.thr[p] = make_float4(T.x, T.y, T.z, __uint_as_float(flags));
                    P.rng[p] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
                    // the NEE terms wait behind their shadow rays: flag them for the next k_shade / k_finalize
                    uint32_t* pix = reinterpret_cast<uint32_t*>(&P.rad[p]) + 3;
                    *pix = (*pix & PIXEL_MASK) | (push_sun ? PEND_SUN : 0u) | (push_sky ? PEND_SKY : 0u);
                }
            }
        }
        // four queues, block totals through LDS, one atomic per queue and block iteration (as in k_shade)
        {
            const uint32_t wave = threadIdx.x >> 6;
            const unsigned long long m_sun = __ballot(active && push_sun), m_sky = __ballot(active && push_sky);
            const unsigned long long m_walk = __ballot(active && alive && !exited), m_exit = __ballot(active && alive && exited);
            const unsigned long long m_fin = __ballot(active && !alive);
            const uint32_t n_sun = (uint32_t)__popcll(m_sun), n_sky = (uint32_t)__popcll(m_sky);
            if ((threadIdx.x & 63u) == 0u) {
                s_cnt[0][wave] = n_sun + n_sky;
                s_cnt[1][wave] = (uint32_t)__popcll(m_exit);
                s_cnt[2][wave] = (uint32_t)__popcll(m_fin);
                s_cnt[3][wave] = (uint32_t)__popcll(m_walk);
            }
            __syncthreads();
            if (threadIdx.x < 4u) {
                uint32_t total = 0;
                for (uint32_t k = 0; k < BLOCK / 64; ++k) total += s_cnt[threadIdx.x][k];
                uint32_t* counter = threadIdx.x == 0u ? c_shadow : (threadIdx.x == 1u ? c_next : (threadIdx.x == 2u ? c_final : c_walk));
                s_base[threadIdx.x] = total ? atomicAdd(counter, total) : 0u;
            }
            __syncthreads();
            uint32_t o_sh = s_base[0], o_ex = s_base[1], o_fi = s_base[2], o_wk = s_base[3];
            for (uint32_t k = 0; k < wave; ++k) { o_sh += s_cnt[0][k]; o_ex += s_cnt[1][k]; o_fi += s_cnt[2][k]; o_wk += s_cnt[3][k]; }
            if (active && push_sun) q_shadow[o_sh + wave_rank(m_sun)] = p * 2u;
            if (active && push_sky) q_shadow[o_sh + n_sun + wave_rank(m_sky)] = p * 2u + 1u;
            if (active && alive && !exited) q_walk_next[o_wk + wave_rank(m_walk)] = p;
            if (active && alive && exited) q_next[o_ex + wave_rank(m_exit)] = p;
            if (active && !alive) q_final[o_fi + wave_rank(m_fin)] = p;      // walk returned None, roulette, or out of bounces
            if (ONCE) break;
            __syncthreads();
        }
    }
}
template <bool CNT>
__global__ __launch_bounds__(WALK_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_sss_step(SceneDev S, PathArrays P, const uint32_t* q_walk, uint32_t* q_walk_next, uint32_t* q_next,
                                                         uint32_t* q_shadow, uint32_t* q_final, IterCtr* ctr, IterCtr* ctr_next, uint32_t max_bounces,
                                                         DevCounters* dc) {
    ShadeCnt sc = {0, 0, 0, 0, 0, 0, 0};
    sss_range<CNT, WALK_BLOCK, !CNT>(S, P, q_walk, ctr->n_walk, blockIdx.x * blockDim.x, gridDim.x * blockDim.x, q_walk_next, q_next, q_shadow, q_final,
                               &ctr->n_shadow, &ctr_next->n_active, &ctr->n_final, &ctr_next->n_walk, max_bounces, sc);
    if (CNT) {
        wave_add(&dc->v[C_HITDATA], sc.hitdata);
        wave_add(&dc->v[C_SSS_STEPS], sc.sss_steps);
        wave_add(&dc->v[C_CDF_PROBES], sc.cdf_probes);
        wave_add(&dc->v[C_SKY_PDF], sc.sky_pdf);
        wave_add(&dc->v[C_SKY_TEXEL], sc.sky_texel);
        wave_add(&dc->v[C_TEXTURE], sc.texture);
    }
}

// The body of k_shade over queue entries first, first + stride, ... < n, for a block of BLOCK threads (see sss_range).
// SPEC ("Sky search ahead"): the two CDF searches of sample_sky_light start ahead of the hit-data and
// material gathers, on the numbers the sky sample will draw unless the material draws first (transmission, subsurface).
// ONCE: the caller's grid covers the range in one stride (k_shade<false, ..>): no loop, so nothing is hoisted out of one
// and kept in scalar registers across the whole body.
template <bool CNT, int BLOCK, bool SPEC = false, bool ONCE = false>
PT_DEV void shade_range(const SceneDev& S, const PathArrays& P, const uint32_t* q, const uint32_t n, const uint32_t first, const uint32_t stride,
                        uint32_t* q_next, uint32_t* q_shadow, uint32_t* q_final, uint32_t* q_walk,
                        uint32_t* c_shadow, uint32_t* c_next, uint32_t* c_final, uint32_t* c_walk, const uint32_t max_bounces, ShadeCnt& sc) {
    __shared__ uint32_t s_cnt[4][BLOCK / 64];
    __shared__ uint32_t s_base[4];
    for (uint32_t base = first; base < n; base += stride) {
        const uint32_t i = base + threadIdx.x;
        const bool active = i < n;
        bool alive = false, push_sun = false, push_sky = false, walk = false;
        uint32_t p = 0;
        if (active) {
            p = q[i];
            const float4 ro4 = P.ray_o[p], rd4 = P.ray_d[p], th4 = P.thr[p], ra4 = P.rad[p], hi4 = P.hit[p];
            const uint2 rs = P.rng[p];
            Rng rng;
            rng.state = (uint64_t)rs.x | ((uint64_t)rs.y << 32);
            V3 ro = v3(ro4.x, ro4.y, ro4.z), rd = v3(rd4.x, rd4.y, rd4.z);
            float ray_tmax = ro4.w, last_pdf = rd4.w;
            V3 T = v3(th4.x, th4.y, th4.z), L = v3(ra4.x, ra4.y, ra4.z);
            uint32_t flags = __float_as_uint(th4.w);
            uint32_t bounce = flags & FL_BOUNCE_MASK;
            bool last_spec = (flags & FL_SPECULAR) != 0;
            uint32_t state = (flags & FL_STATE_MASK) >> FL_STATE_SHIFT;
            bool first_hit = (flags & FL_FIRST_HIT) != 0;
            uint32_t step = (flags & FL_STEP_MASK) >> FL_STEP_SHIFT;
            const uint32_t h_inst = __float_as_uint(hi4.w);
            const bool has_hit = h_inst != PTX_MISS;

            // pending adds of the previous step, in the reference's order (NEE sun, NEE sky, BSDF-sampled sun)
            { const uint32_t pend = __float_as_uint(ra4.w);
              float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a, c = a;
              if (pend & PEND_SUN) a = P.c_sun[p];
              if (pend & PEND_SKY) b = P.c_sky[p];
              if (pend & PEND_POST) c = P.c_post[p];
              L = L + v3(a.x, a.y, a.z); L = L + v3(b.x, b.y, b.z); L = L + v3(c.x, c.y, c.z); }
            V3 c_sun = splat(0.0f), c_sky = splat(0.0f), c_post = splat(0.0f);
            bool has_post = false, l_stored = false;
            V3 sh_o = splat(0.0f), sh_d = splat(0.0f);
            alive = true;
            const V3 sun_dir = v3(S.sun_dir[0], S.sun_dir[1], S.sun_dir[2]);
            const V3 sun_color = v3(S.sun_color[0], S.sun_color[1], S.sun_color[2]);

            {
                if (!has_hit) {
                    // miss (pathtracer.rs:118-134)
                    const V3 sky_radiance = sky_sample(S, rd, sc);
                    if (bounce == 0u) {
                        L = L + T * sky_radiance;
                        if (first_hit) {
                            const V3 a = clamp01(sky_radiance), nn = -rd;
                            P.aov_a[p] = make_float4(a.x, a.y, a.z, 0.0f);
                            P.aov_n[p] = make_float4(nn.x, nn.y, nn.z, 0.0f);
                        }
                    } else if (last_spec) {
                        L = L + T * sky_radiance;
                    } else {
                        const float spdf = sky_pdf(S, rd, sc);
                        const float mis = mis_weight(last_pdf, spdf);
                        L = L + T * sky_radiance * mis;
                    }
                    alive = false;
                } else {
                    // The sky-light sample's two numbers are this vertex's first two draws unless the material draws
                    // first; its six dependent table reads do not depend on the hit, so they are issued here, beside the
                    // hit-data -> material -> texel chain instead of behind it.  A vertex whose material did draw first
                    // (drew) searches again with its own numbers below.
                    SkyUV suv = {0.0f, 0.0f};
                    bool drew = false;
                    HitData hit;
                    if (SPEC) {
                        Rng r2 = rng;
                        const float a1 = r2.next();
                        const float a2 = r2.next();
                        float4 i0, i1, i2, qa, qb, qc, qd, qe;
                        suv = gather_hit_and_sky(S, h_inst, &P.hit_tri[p], a1, a2, i0, i1, i2, qa, qb, qc, qd, qe);
                        hit = hit_data_from(i0, i1, i2, qa, qb, qc, qd, qe, hi4.x, hi4.y, hi4.z);
                    } else {
                        hit = compute_hit_data(S, h_inst, P.hit_tri[p], hi4.x, hi4.y, hi4.z);
                    }
                    if (CNT) { sc.hitdata++; sc.material++; }
                    const PtxMaterial& material = S.materials[hit.material_id];
                    const SampledMaterial mat = sample_material(S, material, hit, sc);
                    if (first_hit) {
                        const V3 nn = mat.normal * 0.5f + splat(0.5f);
                        P.aov_a[p] = make_float4(mat.base_color.x, mat.base_color.y, mat.base_color.z, 0.0f);
                        P.aov_n[p] = make_float4(nn.x, nn.y, nn.z, 0.0f);
                        first_hit = false;
                    }
                    L = L + T * mat.emissive;
                    // L is final for this step: its three words go out now, the pixel/pending word at the end
                    { float* lr = reinterpret_cast<float*>(&P.rad[p]); lr[0] = L.x; lr[1] = L.y; lr[2] = L.z; }
                    l_stored = true;

                    const bool front_face = dot(rd, hit.geo_normal) < 0.0f;
                    const V3 geo_n = front_face ? hit.geo_normal : -hit.geo_normal;
                    const V3 v = -normalize(rd);
                    const V3 shading_n = (dot(mat.normal, geo_n) > 0.0f) ? mat.normal : -mat.normal;
                    const V3 nrm = fix_shading_normal(shading_n, geo_n, v);
                    const V3 hit_point = ro + rd * hit.t;
                    const float transmission = material.transmission;
                    bool handled = false;

                    if (SPEC && transmission > 0.0f) drew = true;
                    if (transmission > 0.0f && rng.next() < transmission) {
                        // sample_transmission (pathtracer.rs:328-348)
                        const float ior = material.ior;
                        const float eta = front_face ? 1.0f / ior : ior;
                        const float cos_theta = f_min(dot(-rd, nrm), 1.0f);
                        const float fresnel = fresnel_dielectric(cos_theta, ior);
                        V3 new_dir, offset_normal;
                        if (rng.next() < fresnel) { new_dir = reflect(rd, nrm); offset_normal = nrm; }
                        else {
                            V3 refr;
                            if (refract(rd, nrm, eta, &refr)) { new_dir = normalize(refr); offset_normal = -nrm; }
                            else { new_dir = reflect(rd, nrm); offset_normal = nrm; }
                        }
                        T = T * mat.base_color;
                        if (!russian_roulette(rng, T, bounce)) alive = false;
                        else { ro = hit_point + offset_normal * RAY_EPSILON; rd = new_dir; ray_tmax = PT_INF; last_spec = true; bounce++; }
                        handled = true;
                    }
                    if (!handled) {
                        const float sss_prob = (material.subsurface > 0.0f && front_face) ? material.subsurface : 0.0f;
                        if (SPEC && sss_prob > 0.0f) drew = true;
                        if (sss_prob > 0.0f && rng.next() < sss_prob) {
                            handled = true;
                            T = T / sss_prob;
                            const float cos_theta = f_min(dot(v, nrm), 1.0f);
                            const float fresnel = fresnel_dielectric(cos_theta, material.ior);
                            bool do_reflect = false;
                            V3 refr = splat(0.0f);
                            if (rng.next() < fresnel) do_reflect = true;
                            else if (!refract(-v, nrm, 1.0f / material.ior, &refr)) do_reflect = true;
                            if (do_reflect) {
                                rd = reflect(-v, nrm);
                                ro = hit_point + nrm * RAY_EPSILON; ray_tmax = PT_INF;
                                last_spec = true; bounce++;
                            } else {
                                const V3 entry_dir = normalize(refr);
                                T = T * v3(material.subsurface_entry_tint[0], material.subsurface_entry_tint[1], material.subsurface_entry_tint[2]);
                                ro = hit_point - nrm * RAY_EPSILON;   // entry_point; thickness ray (pathtracer.rs:205-207)
                                rd = entry_dir; ray_tmax = PT_INF;
                                state = ST_THICK;
                                P.w_b[p] = make_float4(mat.base_color.x, mat.base_color.y, mat.base_color.z, __uint_as_float(hit.material_id));
                            }
                        } else if (sss_prob > 0.0f) {
                            T = T / (1.0f - sss_prob);
                        }
                    }
                    if (!handled) {
                        last_spec = false;
                        const float alpha = mat.roughness * mat.roughness;
                        const V3 f0 = lerp(splat(0.04f), mat.base_color, mat.metallic);
                        const V3 diffuse_color = mat.base_color * (1.0f - mat.metallic);
                        const float n_dot_v = f_max(dot(v, nrm), 0.001f);
                        const float spec_prob = compute_spec_prob(n_dot_v, f0, diffuse_color);
                        sh_o = hit_point + geo_n * RAY_EPSILON;
                        P.sh_o[p] = make_float4(sh_o.x, sh_o.y, sh_o.z, 0.0f);   // stored here: three registers fewer across the sampling code

                        // sample_direct_light (pathtracer.rs:372-400): visibility is resolved by k_shadow
                        if (S.sun_enabled) {
                            const float n_dot_l = dot(nrm, sun_dir);
                            if (n_dot_l > 0.0f) {
                                const V3 brdf = evaluate_brdf(v, sun_dir, nrm, alpha, f0, diffuse_color);
                                float n_dot_h, v_dot_h;
                                half_vector_dots(v, sun_dir, nrm, &n_dot_h, &v_dot_h);
                                const float brdf_pdf = pdf_brdf(n_dot_l, n_dot_h, v_dot_h, alpha, spec_prob);
                                const float mis = mis_weight(S.sun_pdf, brdf_pdf);
                                c_sun = T * brdf * sun_color * mis;
                                push_sun = true;
                                P.c_sun[p] = make_float4(c_sun.x, c_sun.y, c_sun.z, 0.0f);
                            }
                        }
                        // sample_sky_light (pathtracer.rs:402-435)
                        {
                            const float u1 = rng.next();
                            const float u2 = rng.next();
                            SkySample ss;
                            if (SPEC) ss = sky_sample_at(S, drew ? sky_search_uv<true>(S, u1, u2, sc) : suv, sc);
                            else ss = sky_sample_direction<!CNT>(S, u1, u2, sc);
                            const V3 l = ss.direction;
                            const float n_dot_l = dot(nrm, l);
                            if (!(n_dot_l <= 0.0f || ss.pdf <= 0.0f)) {
                                const V3 brdf = evaluate_brdf(v, l, nrm, alpha, f0, diffuse_color);
                                float n_dot_h, v_dot_h;
                                half_vector_dots(v, l, nrm, &n_dot_h, &v_dot_h);
                                const float brdf_pdf = pdf_brdf(n_dot_l, n_dot_h, v_dot_h, alpha, spec_prob);
                                const float mis = mis_weight(ss.pdf, brdf_pdf);
                                const V3 sky_contribution = clamp_radiance(ss.radiance * safe_inv_pdf(ss.pdf));
                                c_sky = T * brdf * sky_contribution * mis;
                                sh_d = l;
                                push_sky = true;
                                P.c_sky[p] = make_float4(c_sky.x, c_sky.y, c_sky.z, 0.0f);
                                P.sh_d[p] = make_float4(sh_d.x, sh_d.y, sh_d.z, 0.0f);
                            }
                        }
                        // sample_brdf (pathtracer.rs:437-479)
                        V3 tb, bb;
                        build_basis(nrm, &tb, &bb);      // only the BSDF sample needs the frame: built here, not before the NEE code
                        V3 new_dir = splat(0.0f), weight = splat(0.0f);
                        float pdf = 0.0f;
                        bool ok = true;
                        if (rng.next() < spec_prob) {
                            const float u1 = rng.next();
                            const float u2 = rng.next();
                            const V3 h = to_world(sample_ggx(u1, u2, alpha), tb, bb, nrm);
                            const V3 l = normalize(2.0f * dot(v, h) * h - v);
                            const float n_dot_l = dot(nrm, l);
                            if (n_dot_l <= 0.0f) ok = false;
                            else {
                                const float n_dot_h = f_max(dot(nrm, h), 0.001f);
                                const float v_dot_h = f_max(dot(v, h), 0.001f);
                                const V3 f = fresnel_schlick(v_dot_h, f0);
                                const float g = ggx_g(n_dot_l, n_dot_v, alpha);
                                weight = f * g * v_dot_h / (n_dot_v * n_dot_h * spec_prob);
                                pdf = pdf_brdf(n_dot_l, n_dot_h, v_dot_h, alpha, spec_prob);
                                new_dir = l;
                            }
                        } else {
                            const float u1 = rng.next();
                            const float u2 = rng.next();
                            const V3 l = to_world(sample_cosine_hemisphere(u1, u2), tb, bb, nrm);
                            const float n_dot_l = f_max(dot(nrm, l), 0.001f);
                            float n_dot_h, v_dot_h;
                            half_vector_dots(v, l, nrm, &n_dot_h, &v_dot_h);
                            const V3 f = fresnel_schlick(v_dot_h, f0);
                            weight = diffuse_color * (splat(1.0f) - f) / (1.0f - spec_prob);
                            pdf = pdf_brdf(n_dot_l, n_dot_h, v_dot_h, alpha, spec_prob);
                            new_dir = l;
                        }
                        if (!ok) alive = false;
                        else {
                            T = T * weight;
                            if (!russian_roulette(rng, T, bounce)) alive = false;
                            else {
                                ro = hit_point + geo_n * RAY_EPSILON; rd = new_dir; ray_tmax = PT_INF;
                                last_pdf = pdf;
                                if (S.sun_enabled && dot(new_dir, sun_dir) >= S.sun_cos_angle) {
                                    const float mis = mis_weight(pdf, S.sun_pdf);
                                    c_post = T * sun_color * mis;
                                    has_post = true;
                                    P.c_post[p] = make_float4(c_post.x, c_post.y, c_post.z, 0.0f);
                                }
                                bounce++;
                            }
                        }
                    }
                }
            }
            // (paths in a subsurface state never come here: they live on the walk queue, k_sss_step)

            // `for bounce in 0..MAX_BOUNCES` ran out (pathtracer.rs:117)
            if (alive && state == ST_NORMAL && bounce >= max_bounces) alive = false;
            walk = alive && state == ST_THICK;       // refracted into a subsurface medium: the thickness ray is traced with the walk queue

            flags = (bounce & FL_BOUNCE_MASK) | (last_spec ? FL_SPECULAR : 0u) | (state << FL_STATE_SHIFT) |
                    (first_hit ? FL_FIRST_HIT : 0u) | ((step << FL_STEP_SHIFT) & FL_STEP_MASK);
            const uint32_t pend_next = (push_sun ? PEND_SUN : 0u) | (push_sky ? PEND_SKY : 0u) | (has_post ? PEND_POST : 0u);
            if (l_stored) reinterpret_cast<uint32_t*>(&P.rad[p])[3] = (__float_as_uint(ra4.w) & PIXEL_MASK) | pend_next;
            else P.rad[p] = make_float4(L.x, L.y, L.z, __uint_as_float((__float_as_uint(ra4.w) & PIXEL_MASK) | pend_next));
            // (pending contributions and the shadow rays were stored where they were computed)
            if (alive) {
                P.ray_o[p] = make_float4(ro.x, ro.y, ro.z, ray_tmax);
                P.ray_d[p] = make_float4(rd.x, rd.y, rd.z, last_pdf);
                P.thr[p] = make_float4(T.x, T.y, T.z, __uint_as_float(flags));
                P.rng[p] = make_uint2((uint32_t)rng.state, (uint32_t)(rng.state >> 32));
            }
        }
        // Compaction into the three queues: wave64 ballots, block-level totals through LDS, ONE atomic per queue and
        // block iteration (every wave bumping the same three counters itself made their L2 atomics the kernel's
        // bottleneck).  The loop bounds are block-uniform, so every thread reaches the barriers.
        {
            const uint32_t wave = threadIdx.x >> 6;
            const unsigned long long m_sun = __ballot(active && push_sun), m_sky = __ballot(active && push_sky);
            // a path that starts a subsurface walk goes to the next iteration's walk queue (k_walk_step takes it from there
            // until the walk ends), every other live path to the next iteration's active queue
            const unsigned long long m_next = __ballot(active && alive && !walk), m_fin = __ballot(active && !alive);
            const unsigned long long m_walk = __ballot(active && walk);
            const uint32_t n_sun = (uint32_t)__popcll(m_sun), n_sky = (uint32_t)__popcll(m_sky);
            if ((threadIdx.x & 63u) == 0u) {
                s_cnt[0][wave] = n_sun + n_sky;
                s_cnt[1][wave] = (uint32_t)__popcll(m_next);
                s_cnt[2][wave] = (uint32_t)__popcll(m_fin);
                s_cnt[3][wave] = (uint32_t)__popcll(m_walk);
            }
            __syncthreads();
            if (threadIdx.x < 4u) {
                uint32_t total = 0;
                for (uint32_t k = 0; k < BLOCK / 64; ++k) total += s_cnt[threadIdx.x][k];
                uint32_t* counter = threadIdx.x == 0u ? c_shadow : (threadIdx.x == 1u ? c_next : (threadIdx.x == 2u ? c_final : c_walk));
                s_base[threadIdx.x] = total ? atomicAdd(counter, total) : 0u;
            }
            __syncthreads();
            uint32_t o_sh = s_base[0], o_nx = s_base[1], o_fi = s_base[2], o_wk = s_base[3];
            for (uint32_t k = 0; k < wave; ++k) { o_sh += s_cnt[0][k]; o_nx += s_cnt[1][k]; o_fi += s_cnt[2][k]; o_wk += s_cnt[3][k]; }
            if (active && push_sun) q_shadow[o_sh + wave_rank(m_sun)] = p * 2u;
            if (active && push_sky) q_shadow[o_sh + n_sun + wave_rank(m_sky)] = p * 2u + 1u;
            if (active && alive && !walk) q_next[o_nx + wave_rank(m_next)] = p;
            if (active && !alive) q_final[o_fi + wave_rank(m_fin)] = p;
            if (active && walk) q_walk[o_wk + wave_rank(m_walk)] = p;
            if (ONCE) break;
            __syncthreads();   // s_cnt / s_base are rewritten by the next iteration
        }
    }
}
template <bool CNT, bool SPEC = false>
__global__ __launch_bounds__(SHADE_BLOCK) PT_SHADE_ATTR void k_shade(SceneDev S, PathArrays P, const uint32_t* q, uint32_t* q_next, uint32_t* q_shadow,
                                                       uint32_t* q_final, uint32_t* q_walk, IterCtr* ctr, IterCtr* ctr_next, uint32_t max_bounces,
                                                       DevCounters* dc) {
    ShadeCnt sc = {0, 0, 0, 0, 0, 0, 0};
    shade_range<CNT, SHADE_BLOCK, SPEC, !CNT>(S, P, q, ctr->n_active, blockIdx.x * blockDim.x, gridDim.x * blockDim.x, q_next, q_shadow, q_final, q_walk,
                                  &ctr->n_shadow, &ctr_next->n_active, &ctr->n_final, &ctr_next->n_walk, max_bounces, sc);
    if (CNT) {
        wave_add(&dc->v[C_HITDATA], sc.hitdata);
        wave_add(&dc->v[C_MATERIAL], sc.material);
        wave_add(&dc->v[C_TEXTURE], sc.texture);
        wave_add(&dc->v[C_CDF_PROBES], sc.cdf_probes);
        wave_add(&dc->v[C_SKY_PDF], sc.sky_pdf);
        wave_add(&dc->v[C_SKY_TEXEL], sc.sky_texel);
        wave_add(&dc->v[C_SSS_STEPS], sc.sss_steps);
    }
}

// ---- finalize: finished paths -> clamp_radiance -> framebuffer ------------------------
PT_DEV uint32_t finalize_range(const PathArrays& P, const uint32_t* q_final, const uint32_t n, const uint32_t first, const uint32_t stride, const Target& tg) {
    uint32_t written = 0;
    for (uint32_t i = first; i < n; i += stride) {
        const uint32_t p = q_final[i];
        const float4 ra4 = P.rad[p];
        V3 L = v3(ra4.x, ra4.y, ra4.z);
        { const uint32_t pend = __float_as_uint(ra4.w);
          float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a, c = a;
          if (pend & PEND_SUN) a = P.c_sun[p];
          if (pend & PEND_SKY) b = P.c_sky[p];
          if (pend & PEND_POST) c = P.c_post[p];
          L = L + v3(a.x, a.y, a.z); L = L + v3(b.x, b.y, b.z); L = L + v3(c.x, c.y, c.z); }
        const V3 c = clamp_radiance(L);     // pathtracer.rs:325
        const uint32_t pixel = __float_as_uint(ra4.w) & PIXEL_MASK;
        const size_t o = (size_t)(tg.by_slot ? p : pixel) * 3;
        if (tg.out_rgb) { tg.out_rgb[o] = c.x; tg.out_rgb[o + 1] = c.y; tg.out_rgb[o + 2] = c.z; }
        if (tg.out_albedo) {
            const float4 a = P.aov_a[p], nn = P.aov_n[p];
            tg.out_albedo[o] = a.x; tg.out_albedo[o + 1] = a.y; tg.out_albedo[o + 2] = a.z;
            tg.out_normal[o] = nn.x; tg.out_normal[o + 1] = nn.y; tg.out_normal[o + 2] = nn.z;
        }
        if (tg.defer) P.rad[p] = make_float4(c.x, c.y, c.z, ra4.w);
        else if (tg.sum) {
            // Framebuffer::accumulate / AdaptiveFramebuffer::add_sample (framebuffer.rs:44-50, 110-125)
            const size_t f = (size_t)pixel * 3;
            tg.sum[f] += c.x; tg.sum[f + 1] += c.y; tg.sum[f + 2] += c.z;
            if (tg.sum_sq) { tg.sum_sq[f] += c.x * c.x; tg.sum_sq[f + 1] += c.y * c.y; tg.sum_sq[f + 2] += c.z * c.z; }
            tg.counts[pixel] += 1u;
        }
        written++;
    }
    return written;
}
__global__ void k_finalize(PathArrays P, const uint32_t* q_final, const IterCtr* ctr, Target tg, DevCounters* dc, uint32_t counting) {
    const uint32_t written = finalize_range(P, q_final, ctr->n_final, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, tg);
    if (counting) wave_add(&dc->v[C_PIXEL_SAMPLES], written);
}

// ---- tail: the late iterations of a batch, wave by wave -----------------------------------------------
// Once few paths are left, a grid-wide iteration lasts as long as the longest ray of the whole launch, four launches in
// a row.  Here every wave takes an even slice of the active and walk queues and carries its own paths to their end:
// extend -> shade -> walk step -> shadow -> finalize on queue stretches of its own, with no other wave to wait for, so a
// phase lasts as long as the longest ray among the wave's own.  The phases are the kernels' own bodies (trace_queue,
// shade_range, sss_range, finalize_range) and every path still draws from its own PCG stream: same bits
// (test_tail_kernel_changes_nothing).
#ifndef PT_TAIL_WAVES
#define PT_TAIL_WAVES 3                             // k_shade's register budget
#endif
#ifndef PT_TAIL_PER_WAVE
#define PT_TAIL_PER_WAVE 256
#endif
constexpr uint32_t TAIL_PER_WAVE = PT_TAIL_PER_WAVE;    // paths one wave can own
constexpr int TAIL_GRID = 256 * 4 * PT_TAIL_WAVES;      // the waves that are resident together
constexpr uint32_t TAIL_REGION = 7 * TAIL_PER_WAVE; // u32 of queue space per wave: active x2, walk x2, shadow (2 per path), final
constexpr uint32_t PTX_DEV_ERR_TAIL = 0x100u;       // device error word: a wave was handed more than TAIL_PER_WAVE paths
template <bool FAST, bool UNC>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_TAIL_WAVES))) void k_tail(SceneDev S, PathArrays P, const uint32_t* q, const uint32_t* q_walk, const IterCtr* ctr,
                                                     uint32_t* tail_q, uint32_t* left_alive, uint2* ovf, DevCounters* dc, uint32_t* err, Target tg,
                                                     uint32_t max_bounces, uint32_t iters, uint32_t first_extend_done) {
    __shared__ uint32_t t_c[8];     // 0 closest-hit cursor, 1 any-hit cursor, 2 shadow entries, 3 next active, 4 final, 5 next walk
    const StackRef st = make_stack(ovf);
    const uint32_t W = gridDim.x, w = blockIdx.x;
    const uint32_t tot_a = ctr->n_active, tot_w = ctr->n_walk;
    const uint32_t a_lo = (uint32_t)((unsigned long long)tot_a * w / W), a_hi = (uint32_t)((unsigned long long)tot_a * (w + 1u) / W);
    const uint32_t w_lo = (uint32_t)((unsigned long long)tot_w * w / W), w_hi = (uint32_t)((unsigned long long)tot_w * (w + 1u) / W);
    uint32_t nA = a_hi - a_lo, nW = w_hi - w_lo;
    if (nA + nW > TAIL_PER_WAVE) {                  // the host sizes the grid from an upper bound of the live count
        if (threadIdx.x == 0u) atomicOr(err, PTX_DEV_ERR_TAIL);
        return;
    }
    uint32_t* const mine = tail_q + (size_t)w * TAIL_REGION;
    uint32_t* const q_sh = mine + 4u * TAIL_PER_WAVE;
    uint32_t* const q_fin = mine + 6u * TAIL_PER_WAVE;
    const uint32_t* qa = q + a_lo;
    const uint32_t* qw = q_walk + w_lo;
    LaneCnt lc = {0, 0, 0, 0, 0, 0, 0, 0.0f, {0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0, 0, 0.0f};
    ShadeCnt sc = {0, 0, 0, 0, 0, 0, 0};
    uint32_t nrays = 0;
    for (uint32_t it = 0; it < iters && (nA | nW) != 0u; ++it) {
        uint32_t* const qa_next = mine + (it & 1u) * TAIL_PER_WAVE;
        uint32_t* const qw_next = mine + (2u + (it & 1u)) * TAIL_PER_WAVE;
        if (threadIdx.x < 8u) t_c[threadIdx.x] = 0u;
        __syncthreads();
        if (it != 0u || !first_extend_done) {     // the first closest-hit pass was an ordinary launch (it ran beside the any-hit launch before it)
            ExtendSrc src{P, qa, nA, qw};
            ExtendSink sink{P};
            trace_queue<false, false, FAST, UNC, true>(S, st, &t_c[0], nA + nW, src, sink, lc, nrays, err, reinterpret_cast<unsigned long long*>(&dc->dbg[6]));
        }
        __syncthreads();    // one wave: no barrier, but the hit records the lanes wrote are read by other lanes below
        shade_range<false, TRACE_BLOCK>(S, P, qa, nA, 0u, (uint32_t)TRACE_BLOCK, qa_next, q_sh, q_fin, qw_next, &t_c[2], &t_c[3], &t_c[4], &t_c[5], max_bounces, sc);
        if (S.has_sss) sss_range<false, TRACE_BLOCK>(S, P, qw, nW, 0u, (uint32_t)TRACE_BLOCK, qw_next, qa_next, q_sh, q_fin, &t_c[2], &t_c[3], &t_c[4], &t_c[5], max_bounces, sc);
        __syncthreads();
        const uint32_t n_sh = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_c[2]), n_fin = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_c[4]);
        nA = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_c[3]);
        nW = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_c[5]);
        if (n_sh) {
            ShadowSrc src{S, P, q_sh};
            ShadowSink sink{P};
            trace_queue<true, false, false, false, true>(S, st, &t_c[1], n_sh, src, sink, lc, nrays, err);
            __syncthreads();
        }
        (void)finalize_range(P, q_fin, n_fin, threadIdx.x, (uint32_t)TRACE_BLOCK, tg);
        __syncthreads();
        qa = qa_next;
        qw = qw_next;
    }
    if ((nA | nW) != 0u && threadIdx.x == 0u) atomicAdd(left_alive, nA + nW);     // out of iterations (subsurface walks): run_paths refuses the frame
}

// ---- several passes in flight: add each pixel's samples in pass order (main.rs:131-137 / framebuffer.rs:44-50) ----
__global__ void k_accumulate_batch(PathArrays P, const uint32_t* wl_pixel, uint32_t n_pixels, uint32_t batch, Target tg) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_pixels) return;
    const uint32_t pixel = wl_pixel[slot];
    const size_t f = (size_t)pixel * 3;
    float sx = tg.sum[f], sy = tg.sum[f + 1], sz = tg.sum[f + 2];
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (tg.sum_sq) { qx = tg.sum_sq[f]; qy = tg.sum_sq[f + 1]; qz = tg.sum_sq[f + 2]; }
    for (uint32_t k = 0; k < batch; ++k) {
        const float4 c = P.rad[(size_t)slot * batch + k];
        sx += c.x; sy += c.y; sz += c.z;
        qx += c.x * c.x; qy += c.y * c.y; qz += c.z * c.z;
    }
    tg.sum[f] = sx; tg.sum[f + 1] = sy; tg.sum[f + 2] = sz;
    if (tg.sum_sq) { tg.sum_sq[f] = qx; tg.sum_sq[f + 1] = qy; tg.sum_sq[f + 2] = qz; }
    tg.counts[pixel] += batch;
}

// ---- UnshadedRenderer (renderer/unshaded.rs:7-23) after one extend ---------------------
__global__ void k_unshaded(SceneDev S, PathArrays P, uint32_t n, float* out_rgb) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float4 rd4 = P.ray_d[p], hi4 = P.hit[p], ra4 = P.rad[p];
    const V3 rd = v3(rd4.x, rd4.y, rd4.z);
    const uint32_t inst = __float_as_uint(hi4.w);
    ShadeCnt sc = {0, 0, 0, 0, 0, 0, 0};
    V3 c;
    if (inst != PTX_MISS) {
        const HitData hit = compute_hit_data(S, inst, P.hit_tri[p], hi4.x, hi4.y, hi4.z);
        const PtxMaterial& m = S.materials[hit.material_id];
        const V3 bcf = v3(m.base_color_factor[0], m.base_color_factor[1], m.base_color_factor[2]);
        const V3 base = m.base_color_texture >= 0 ? texture_sample(S.textures[m.base_color_texture], S.texels, hit.uvx, hit.uvy, sc.texture) * bcf : bcf;
        const float n_dot_l = f_max(dot(hit.normal, -rd), 0.0f);
        c = base * (0.2f + 0.8f * n_dot_l);
    } else {
        c = sky_sample(S, rd, sc) * 0.5f;
    }
    const size_t o = (size_t)(__float_as_uint(ra4.w) & PIXEL_MASK) * 3;
    out_rgb[o] = c.x; out_rgb[o + 1] = c.y; out_rgb[o + 2] = c.z;
}

// ---- adaptive: get_noisy_pixels (framebuffer.rs:181-214) ------------------------------
PT_DEV bool pixel_noisy(uint32_t count, V3 sum, V3 sum_sq, uint32_t min_samples, float threshold, float* err) {
    if (count < min_samples) { *err = PT_INF; return true; }
    const float nf = (float)count;
    const V3 mean = sum / nf;
    const V3 mean_sq = sum_sq / nf;
    const V3 variance = mean_sq - mean * mean;
    const float var_scalar = (f_max(variance.x, 0.0f) + f_max(variance.y, 0.0f) + f_max(variance.z, 0.0f)) / 3.0f;
    *err = sqrtf(var_scalar / nf);
    return *err > threshold;
}
__global__ void k_noise_keys(const uint32_t* owned, uint32_t n, const float* sum, const float* sum_sq, const uint32_t* counts,
                             uint32_t min_samples, float threshold, uint32_t* keys, uint32_t* n_noisy) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint32_t i = base + threadIdx.x;
        bool noisy = false;
        if (i < n) {
            const uint32_t pixel = owned[i];
            const size_t f = (size_t)pixel * 3;
            float err;
            noisy = pixel_noisy(counts[pixel], v3(sum[f], sum[f + 1], sum[f + 2]), v3(sum_sq[f], sum_sq[f + 1], sum_sq[f + 2]),
                                min_samples, threshold, &err);
            // key 0 = not selected; selected keys keep the f32 order of err (err >= 0), +1 so that err == 0 stays selected
            keys[i] = noisy ? (__float_as_uint(err) + 1u) : 0u;
        }
        const unsigned long long m = __ballot(noisy);
        if ((threadIdx.x & 63u) == 0u && m) atomicAdd(n_noisy, (uint32_t)__popcll(m));
    }
}
// `batch` adaptive passes traced together for the `n` pixels of `work`: slot i*batch + k holds the sample the pixel would
// get in pass k IF it is still noisy then (sample_index = its count, parallel.rs:109-114).  A pixel's sums change only
// when it is sampled, so replaying the passes per pixel — add sample k, stop at the first pass that finds the pixel
// converged, drop the rest — leaves sum / sum_sq / count exactly as `batch` separate passes would.
// out[0] = passes that rendered something (max samples applied to one pixel), out[1..2] = samples applied (u64).
__global__ void k_accumulate_adaptive(PathArrays P, const uint32_t* work, uint32_t n, uint32_t batch, uint32_t min_samples,
                                      float threshold, Target tg, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t applied = 0;
    if (i < n) {
        const uint32_t pixel = work[i];
        const size_t f = (size_t)pixel * 3;
        V3 s1 = v3(tg.sum[f], tg.sum[f + 1], tg.sum[f + 2]);
        V3 s2 = v3(tg.sum_sq[f], tg.sum_sq[f + 1], tg.sum_sq[f + 2]);
        uint32_t count = tg.counts[pixel];
        for (uint32_t k = 0; k < batch; ++k) {
            float err;
            if (!pixel_noisy(count, s1, s2, min_samples, threshold, &err)) break;
            const float4 c = P.rad[(size_t)i * batch + k];
            s1 = v3(s1.x + c.x, s1.y + c.y, s1.z + c.z);
            s2 = v3(s2.x + c.x * c.x, s2.y + c.y * c.y, s2.z + c.z * c.z);
            count += 1u;
            applied += 1u;
        }
        tg.sum[f] = s1.x; tg.sum[f + 1] = s1.y; tg.sum[f + 2] = s1.z;
        tg.sum_sq[f] = s2.x; tg.sum_sq[f + 1] = s2.y; tg.sum_sq[f + 2] = s2.z;
        tg.counts[pixel] = count;
    }
    uint32_t mx = applied, total = applied;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (uint32_t)__shfl_xor(mx, off)); total += __shfl_xor(total, off); }
    if ((threadIdx.x & 63u) == 0u && total) {
        atomicMax(out, mx);
        atomicAdd(reinterpret_cast<unsigned long long*>(out + 2), (unsigned long long)total);
    }
}
__global__ void k_compact_noisy(const uint32_t* owned, const uint32_t* keys, uint32_t n, uint32_t* out, uint32_t* out_count) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint32_t i = base + threadIdx.x;
        const bool sel = i < n && keys[i] != 0u;
        queue_push(out, out_count, sel, sel ? owned[i] : 0u);
    }
}
// the same selection with the key beside the pixel (ptx_fb_noisy_pairs: the list several ranks sort together)
__global__ void k_compact_noisy_pairs(const uint32_t* owned, const uint32_t* keys, uint32_t n, uint32_t* out_pixels, uint32_t* out_keys, uint32_t* out_count) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {
        const uint32_t i = base + threadIdx.x;
        const bool sel = i < n && keys[i] != 0u;
        const unsigned long long m = __ballot(sel);
        if (m == 0ull) continue;
        uint32_t b = 0;
        const int leader = __ffsll((long long)m) - 1;
        if ((int)(threadIdx.x & 63u) == leader) b = atomicAdd(out_count, (uint32_t)__popcll(m));
        b = (uint32_t)__shfl((int)b, leader);
        if (sel) { const uint32_t at = b + wave_rank(m); out_pixels[at] = owned[i]; out_keys[at] = keys[i]; }
    }
}
// ptx_fb_adaptive_pass_pixels: every pixel of a caller's list must lie inside the framebuffer
__global__ void k_check_pixels(const uint32_t* pixels, uint32_t n, uint32_t n_pixels, uint32_t* bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && pixels[i] >= n_pixels) atomicOr(bad, 1u);
}
__global__ void k_iota(uint32_t* a, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = i; }
__global__ void k_gather(const uint32_t* src, const uint32_t* idx, uint32_t n, uint32_t* dst) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[idx[i]]; }

__global__ void k_fb_stats(const uint32_t* owned, uint32_t n, const uint32_t* counts, uint32_t* out_min, uint32_t* out_max, unsigned long long* out_total) {
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    unsigned long long tot = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t c = counts[owned[i]];
        mn = c < mn ? c : mn; mx = c > mx ? c : mx; tot += c;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t a = __shfl_xor(mn, off), b = __shfl_xor(mx, off);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        tot += __shfl_xor(tot, off);
    }
    if ((threadIdx.x & 63u) == 0u) { atomicMin(out_min, mn); atomicMax(out_max, mx); atomicAdd(out_total, tot); }
}
__global__ void k_mean(const float* sum, const uint32_t* counts, uint32_t n_pixels, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const uint32_t c = counts[i];
    const size_t f = (size_t)i * 3;
    if (c == 0u) { out[f] = 0.0f; out[f + 1] = 0.0f; out[f + 2] = 0.0f; }
    else { const float cf = (float)c; out[f] = sum[f] / cf; out[f + 1] = sum[f + 1] / cf; out[f + 2] = sum[f + 2] / cf; }
}

// ---- multi-GPU exchange: a rank's own 8x8 tiles as packed records, and back -------------------------------------
// Record = 7 words (sum.xyz, sum_sq.xyz, count); tile k of a rank (global tile index rank + k * world) occupies records
// [64 k, 64 k + 64) in row-major order inside the tile; pixels outside the image are zero records.
__global__ void k_export_tiles(const float* sum, const float* sum_sq, const uint32_t* counts, uint32_t w, uint32_t h, uint32_t rank,
                               uint32_t world, uint32_t n_tiles_rank, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles_rank * 64u) return;
    const uint32_t tx_n = (w + 7u) / 8u;
    const uint32_t t = rank + (i >> 6) * world, l = i & 63u;
    const uint32_t x = (t % tx_n) * 8u + (l & 7u), y = (t / tx_n) * 8u + (l >> 3);
    uint32_t r[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (x < w && y < h) {
        const size_t p = (size_t)y * w + x;
        r[0] = __float_as_uint(sum[3 * p]); r[1] = __float_as_uint(sum[3 * p + 1]); r[2] = __float_as_uint(sum[3 * p + 2]);
        if (sum_sq) { r[3] = __float_as_uint(sum_sq[3 * p]); r[4] = __float_as_uint(sum_sq[3 * p + 1]); r[5] = __float_as_uint(sum_sq[3 * p + 2]); }
        r[6] = counts[p];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) out[(size_t)i * 7 + k] = r[k];
}
__global__ void k_import_tiles(float* sum, float* sum_sq, uint32_t* counts, uint32_t w, uint32_t h, uint32_t rank, uint32_t world,
                               uint32_t n_tiles_rank, const uint32_t* in) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles_rank * 64u) return;
    const uint32_t tx_n = (w + 7u) / 8u;
    const uint32_t t = rank + (i >> 6) * world, l = i & 63u;
    const uint32_t x = (t % tx_n) * 8u + (l & 7u), y = (t / tx_n) * 8u + (l >> 3);
    if (x >= w || y >= h) return;
    const size_t p = (size_t)y * w + x;
    const uint32_t* r = in + (size_t)i * 7;
    sum[3 * p] = __uint_as_float(r[0]); sum[3 * p + 1] = __uint_as_float(r[1]); sum[3 * p + 2] = __uint_as_float(r[2]);
    if (sum_sq) { sum_sq[3 * p] = __uint_as_float(r[3]); sum_sq[3 * p + 1] = __uint_as_float(r[4]); sum_sq[3 * p + 2] = __uint_as_float(r[5]); }
    counts[p] = r[6];
}

// ---- upload time: pack one mesh's shading triangles (pt_types.h ShadeTri) -------------------------
__global__ void k_build_shade_tris(SceneDev S, MeshRec m, ShadeTri* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m.n_triangles) out[t] = build_shade_tri(S, m, t);
}

// ---- tone mapping + sRGB8 on the device (exporter/tonemapping.rs:12-115) ---------------------
// mode: 0 None, 1 Aces, 2 Reinhard, 3 Agx.  Cuts the readback from 12 B to 3 B per pixel for display / export.
PT_DEV V3 tm_aces(V3 c) {
    const V3 a0 = v3((c.x * 0.59719f + c.y * 0.35458f) + c.z * 0.04823f,
                     (c.x * 0.07600f + c.y * 0.90834f) + c.z * 0.01566f,
                     (c.x * 0.02840f + c.y * 0.13383f) + c.z * 0.83777f);
    const V3 a = a0 * (a0 + splat(0.0245786f)) - splat(0.000090537f);
    const V3 b = a0 * (a0 * 0.983729f + splat(0.432951f)) + splat(0.238081f);
    const V3 r = a / b;
    const V3 o = v3((r.x * 1.60475f + r.y * -0.53108f) + r.z * -0.07367f,
                    (r.x * -0.10208f + r.y * 1.10813f) + r.z * -0.00605f,
                    (r.x * -0.00327f + r.y * -0.07276f) + r.z * 1.07602f);
    return clamp01(o);
}
PT_DEV float tm_agx_contrast(float x) {
    const float x2 = x * x;
    const float x4 = x2 * x2;
    return (((((x4 * x2 * 15.5f - x4 * x * 40.14f) + x4 * 31.96f) - x2 * x * 6.868f) + x2 * 0.4298f) + x * 0.1191f) - 0.00232f;
}
PT_DEV float tm_agx_encode(float c) {
    const float MIN_EV = -12.47393f, MAX_EV = 4.026069f;
    return (f_clamp(log2f_det(f_max(c, 1e-10f)), MIN_EV, MAX_EV) - MIN_EV) / (MAX_EV - MIN_EV);
}
PT_DEV V3 tm_agx(V3 v) {
    const V3 m = v3((0.842479062253094f * v.x + 0.0423282422610123f * v.y) + 0.0423756549057051f * v.z,
                    (0.0784335999999992f * v.x + 0.878468636469772f * v.y) + 0.0784336f * v.z,
                    (0.0792237451477643f * v.x + 0.0791661274605434f * v.y) + 0.879142973793104f * v.z);
    const V3 e = v3(tm_agx_contrast(tm_agx_encode(m.x)), tm_agx_contrast(tm_agx_encode(m.y)), tm_agx_contrast(tm_agx_encode(m.z)));
    const V3 o = v3((1.19687900512017f * e.x + -0.0528968517574562f * e.y) + -0.0529716355144438f * e.z,
                    (-0.0980208811401368f * e.x + 1.15190312990417f * e.y) + -0.0980434501171241f * e.z,
                    (-0.0990297440797205f * e.x + -0.0989611768448433f * e.y) + 1.15107367264116f * e.z);
    return v3(powf_det(o.x, 2.2f), powf_det(o.y, 2.2f), powf_det(o.z, 2.2f));
}
PT_DEV V3 tm_apply(int mode, V3 c, float exposure) {
    c = c * exposure;
    if (mode == 1) return tm_aces(c);
    if (mode == 2) return c / (c + splat(1.0f));
    if (mode == 3) return tm_agx(c);
    return c;
}
PT_DEV uint32_t tm_srgb_u8(float c) {   // linear_to_srgb_u8; Rust `as u8` saturates and maps NaN to 0
    c = f_clamp(c, 0.0f, 1.0f);
    const float s = (c <= 0.0031308f) ? 12.92f * c : 1.055f * powf_det(c, 1.0f / 2.4f) - 0.055f;
    const float v = s * 255.0f;
    if (!(v > 0.0f)) return 0u;
    if (v >= 255.0f) return 255u;
    return (uint32_t)v;
}
__global__ void k_tonemap_srgb8(const float* sum, const uint32_t* counts, uint32_t n_pixels, int mode, float exposure, uint8_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const uint32_t c = counts[i];
    const size_t f = (size_t)i * 3;
    V3 mean = splat(0.0f);
    if (c != 0u) { const float cf = (float)c; mean = v3(sum[f] / cf, sum[f + 1] / cf, sum[f + 2] / cf); }
    const V3 t = tm_apply(mode, mean, exposure);
    out[f] = (uint8_t)tm_srgb_u8(t.x); out[f + 1] = (uint8_t)tm_srgb_u8(t.y); out[f + 2] = (uint8_t)tm_srgb_u8(t.z);
}

// ---- test hook: the kernels' transcendental implementations on a caller's inputs --------
__global__ void k_debug_math(int fn, const float* x, const float* y, uint64_t n, float* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (fn) {
        case 0: r = d_sinf(x[i]); break;
        case 1: r = d_cosf(x[i]); break;
        case 2: r = d_expf(x[i]); break;
        case 3: r = d_logf(x[i]); break;
        case 4: r = d_atan2f(x[i], y[i]); break;
        case 5: r = d_asinf(x[i]); break;
        case 6: { float s, c; d_sincosf(x[i], &s, &c); r = s; break; }
        case 8: r = powf_det(x[i], y[i]); break;
        case 9: r = log2f_det(x[i]); break;
        default: { float s, c; d_sincosf(x[i], &s, &c); r = c; break; }
    }
    out[i] = r;
}

}  // namespace

// ================================================================ host side =====
struct PtxContext {
    uint64_t max_paths_in_flight = 128ull << 20;  // ptx_fb_render_passes traces up to this many path slots (260 B each: 35 GB of the 288) per wavefront batch (PTX_PATHS_IN_FLIGHT)
    int device = 0;
    hipStream_t stream = nullptr;
    // the any-hit launch of iteration i and its k_finalize run here, beside iteration i + 1's closest-hit launch on
    // `stream` (run_paths); ev_main / ev_side order the two streams
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    hipEvent_t ev_ext_in = nullptr, ev_ext_out = nullptr;     // ptx_order_after_stream / ptx_order_stream_after
    hipEvent_t ev_tail = nullptr;       // recorded once a frame's heavy iterations are enqueued (ptx_order_after_frame_tail)
    uint32_t tail_gate_iter = 3;        // PTX_TAIL_GATE_ITER
    // k_tail (run_paths): the late iterations of a batch as one launch of independent waves
    bool tail_kernel = true;            // PTX_TAIL_KERNEL=0: every iteration as its own launches
    bool tail_poll = true;              // read the live count back to find the takeover point (off inside ptx_fb_render_passes_async, which must not wait)
    uint32_t tail_first_poll = 1;       // PTX_TAIL_FIRST_POLL: first iteration whose live count is read back
    hipEvent_t ev_poll = nullptr;
    uint64_t tail_cap = ~0ull;          // PTX_TAIL_CAP: take over at this many live paths or fewer (at most what the waves can hold)
    uint32_t tail_min_per_wave = 32;    // PTX_TAIL_MIN_PER_WAVE: fewer paths than this per wave are not worth another wave
    uint32_t* tail_q = nullptr;         // TAIL_GRID x TAIL_REGION: the waves' own queue stretches
    bool overlap_shadow = true;
    bool sky_ahead_env = true;          // PTX_SKY_AHEAD=0: k_shade searches the sky tables where the reference does
    bool sky_ahead = false;             // this scene's sky has the three-level tables k_shade<false, true> reads (set by ptx_upload_scene)
    std::string err;

    bool has_scene = false, has_camera = false;
    uint32_t n_uncertified = 0;     // meshes of the scene that failed the shape certificate (mesh_shapes_certified)
    SceneDev S{};
    std::vector<void*> scene_allocs;
    CameraDev cam{};
    uint32_t rank = 0, world = 1;

    // path state
    uint64_t capacity = 0;
    PathArrays P{};
    std::vector<void*> path_allocs;
    uint32_t *q[2] = {nullptr, nullptr}, *q_shadow = nullptr, *q_final = nullptr, *q_redo = nullptr, *q_walk[2] = {nullptr, nullptr};
    int packet_capacity = PACKET_LDS + OVF_STACK;   // PTX_PACKET_CAPACITY (tests): smaller packet stacks, to exercise the redo path
    IterCtr* ctrs = nullptr;
    uint2* ovf = nullptr;
    uint2* ovf_side = nullptr;      // stack strips of the launches on side_stream
    uint32_t* d_err = nullptr;
    DevCounters* d_counters = nullptr;
    uint32_t* h_pinned = nullptr;    // 64 u32 of pinned readback scratch

    // owned-pixel work list (tile order)
    uint32_t wl_w = 0, wl_h = 0, wl_rank = 0, wl_world = 0, wl_n = 0;
    uint32_t* wl_pixel = nullptr;

    // staging
    float* img = nullptr;  uint64_t img_cap = 0;      // 3 images of w*h*3
    uint32_t* list_a = nullptr; uint32_t* list_b = nullptr; uint32_t* list_c = nullptr; uint32_t* list_d = nullptr; uint64_t list_cap = 0;
    void* cub_tmp = nullptr; size_t cub_tmp_bytes = 0;

    // framebuffer
    uint32_t fb_w = 0, fb_h = 0; bool fb_adaptive = false;
    float *fb_sum = nullptr, *fb_sum_sq = nullptr; uint32_t* fb_counts = nullptr;

    bool packet_primary = true, packet_shadow = false, packet_batch = false;   // PTX_PACKET_SHADOW=1: the first shadow launch as packets too (measured: 132 -> 137 ms, its queue interleaves sun and sky rays)   // PTX_PACKET_PRIMARY (0 disables), PTX_PACKET_BATCH (tests: ptx_trace_batch through the packet kernel)
    // children per record of the collapsed trees (0 / 1 = walk the faithful tree): the packet kernels gain from wide records
    // (one record fetch serves 64 rays: primary launch 66 -> 60 ms at 8), the per-ray kernels lose (they are VALU-issue
    // bound and a wider record means more box tests per ray: extend 368 -> 402 ms at 4, 550 ms at 8 with 5 waves/SIMD)
    uint32_t packet_width = 8, ray_width = 0;     // PTX_WIDE_BVH, PTX_WIDE_RAY_BVH
    uint32_t group_lanes = 0;       // PTX_GROUP_LANES = 4 | 8 (experiment): closest-hit launches with that many lanes per ray, on a tree that wide
    uint32_t adaptive_batch = 32;   // adaptive passes traced together by ptx_fb_adaptive_passes (PTX_ADAPTIVE_BATCH)
    bool adaptive_batch_auto = true;   // no PTX_ADAPTIVE_BATCH: start at 32 and double (up to 256) while >= 90 % of a batch's speculative samples were kept
    bool counting = false, profiling = false;
    bool count_shipped = false;     // ptx_set_counting(ctx, 2): the counted closest-hit launches of a frame walk as the timed ones do (forwarded bound, nearest first)
    struct Ev { hipEvent_t a, b; int cls; hipStream_t stream; };
    std::vector<Ev> events;
    PtxTimings timings{};
};

static std::string g_last_error;

// ---- ptx_trace_batch_noalloc: the one entry point that allocates nothing -------------------------------------
// Its kernels are ptx_trace_batch's; what that call takes from the context's allocations (queue cursors, error word,
// per-lane stack spill strips) lives in this file-scope device array instead, one copy per GPU, held until the process
// ends.  Host code never uses the array's name as an address: ptx_create resolves it once per device with
// hipGetSymbolAddress and everything below goes through that pointer.
constexpr int NOALLOC_WAVES = 512;                       // persistent waves of a no-alloc launch (2 per CU)
constexpr size_t NOALLOC_HEAD = 256;                     // [0, 32) queue cursors, [64, 68) error word
constexpr size_t NOALLOC_ERR_OFFSET = 64;
constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_HEAD + sizeof(uint2) * (size_t)NOALLOC_WAVES * TRACE_BLOCK * OVF_STACK;   // 34.6 MB
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
constexpr int NOALLOC_MAX_DEVICES = 64;
static unsigned char* g_scratch_ptr[NOALLOC_MAX_DEVICES];   // per device: each GPU holds its own copy at its own address
static std::mutex g_scratch_mutex;

#define HIPCHK(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            char buf_[512];                                                                                   \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            if (ctx) ctx->err = buf_; else g_last_error = buf_;                                               \
            return PTX_ERR_HIP;                                                                               \
        }                                                                                                     \
    } while (0)

static int32_t fail(PtxContext* ctx, int32_t code, const char* msg) {
    if (ctx) ctx->err = msg; else g_last_error = msg;
    return code;
}

template <class T>
static hipError_t dev_alloc(std::vector<void*>& pool, T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) { pool.push_back(p); *out = static_cast<T*>(p); }
    return e;
}
template <class T>
static hipError_t dev_upload(std::vector<void*>& pool, const T** out, const std::vector<T>& v) {
    T* p = nullptr;
    hipError_t e = dev_alloc(pool, &p, v.size());
    if (e != hipSuccess) return e;
    if (!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    *out = p;
    return e;
}
static void free_pool(std::vector<void*>& pool) {
    for (void* p : pool) (void)hipFree(p);
    pool.clear();
}

// ---- Bvh8Node[] (256 B, 8 slots) -> NodeRec units: the faithful tree and its order-preserving wide collapse ----
// (layout and the argument why the collapse cannot change a result: pt_types.h)
struct RepackItem { float box[6]; uint32_t word; };   // min.xyz, max.xyz; word = the reference child word (internal: node index)

// a node's children in the order the reference's walk takes them: leaf slots ascending, then internal slots descending
static void ordered_children(const PtxBvh8Node& nd, std::vector<RepackItem>& out) {
    out.clear();
    auto item = [&nd](int s) {
        RepackItem it;
        it.box[0] = nd.min_x[s]; it.box[1] = nd.min_y[s]; it.box[2] = nd.min_z[s];
        it.box[3] = nd.max_x[s]; it.box[4] = nd.max_y[s]; it.box[5] = nd.max_z[s];
        it.word = nd.children[s];
        return it;
    };
    for (int s = 0; s < 8; ++s) if ((nd.valid_mask & (1u << s)) && (nd.children[s] & LEAF_FLAG)) out.push_back(item(s));
    for (int s = 7; s >= 0; --s) if ((nd.valid_mask & (1u << s)) && !(nd.children[s] & LEAF_FLAG)) out.push_back(item(s));
}
static void fill_record(NodeRec* rec, const std::vector<RepackItem>& list, const std::vector<uint32_t>& words, uint32_t meta) {
    auto as_f = [](uint32_t w) { float f; std::memcpy(&f, &w, 4); return f; };
    const size_t units = list.size() > 4 ? 2 : 1;
    std::memset(rec, 0, sizeof(NodeRec) * units);
    for (size_t k = 0; k < list.size(); ++k) {
        NodeRec& r = rec[k / 4];
        const float* b = list[k].box;
        r.c[2 * (k % 4)] = make_float4(b[0], b[1], b[2], b[3]);
        r.c[2 * (k % 4) + 1] = make_float4(b[4], b[5], as_f(words[k]), 0.0f);
    }
    rec[0].c[1].w = as_f(meta);
}
constexpr uint32_t REPACK_MAX_DEPTH = 512;
// Returns false on an out-of-range child word, an empty node, or a node graph that is not a tree.
// *root_faithful / *root_ray / *root_packet = root child words (absolute unit index | (children - 1) << 28) of the faithful tree and
// of the collapsed trees the per-ray and the packet kernels walk (width <= 1: no collapse, the faithful root).
static bool repack_wide(const PtxBvh8& bvh, const std::vector<uint32_t>& count, std::vector<NodeRec>& out, uint32_t width, uint32_t* root_wide);
static bool repack_bvh(const PtxBvh8& bvh, std::vector<NodeRec>& out, uint32_t* root_faithful, uint32_t ray_width, uint32_t* root_ray,
                       uint32_t packet_width, uint32_t* root_packet, bool* nests = nullptr) {
    const size_t first = out.size();
    if (nests) *nests = false;
    const uint32_t n = bvh.n_nodes;
    if (n == 0 || n > NODE_INDEX_MASK / 2) return false;
    std::vector<uint32_t> unit(n), count(n);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const int nv = __builtin_popcount(bvh.nodes[i].valid_mask);
        if (nv == 0) return false;
        unit[i] = (uint32_t)total;
        count[i] = (uint32_t)nv;
        total += nv <= 4 ? 1u : 2u;
        for (int s = 0; s < 8; ++s) {
            if (!(bvh.nodes[i].valid_mask & (1u << s))) continue;
            const uint32_t word = bvh.nodes[i].children[s];
            if (word & LEAF_FLAG) {
                const uint32_t start = word & PRIM_START_MASK, cnt = ((word >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1u;
                if ((uint64_t)start + cnt > bvh.n_primitive_indices) return false;
            } else if (word >= n) return false;
        }
    }
    if (first + 2 * total > NODE_INDEX_MASK) return false;
    const uint32_t abs0 = (uint32_t)first;   // child words carry absolute unit indices
    auto internal_word = [](uint32_t abs_unit, uint32_t cnt) { return abs_unit | ((cnt - 1u) << NODE_NCHILD_SHIFT); };
    // ---- faithful tree: one record per reference node
    out.resize(first + total);
    std::vector<RepackItem> list;
    std::vector<uint32_t> words;
    for (uint32_t i = 0; i < n; ++i) {
        ordered_children(bvh.nodes[i], list);
        words.resize(list.size());
        for (size_t k = 0; k < list.size(); ++k)
            words[k] = (list[k].word & LEAF_FLAG) ? list[k].word : internal_word(abs0 + unit[list[k].word], count[list[k].word]);
        fill_record(&out[first + unit[i]], list, words, count[i] | 16u);
    }
    *root_faithful = internal_word(abs0 + unit[0], count[0]);
    *root_ray = *root_packet = *root_faithful;
    // ---- the walk from the root must be a tree (a cycle would never end on the GPU either)
    {
        std::vector<uint8_t> seen(n, 0);
        std::vector<std::pair<uint32_t, uint32_t>> stack{{0u, 0u}};
        seen[0] = 1;
        while (!stack.empty()) {
            const auto [i, depth] = stack.back();
            stack.pop_back();
            if (depth > REPACK_MAX_DEPTH) return false;
            for (int s = 0; s < 8; ++s) {
                if (!(bvh.nodes[i].valid_mask & (1u << s)) || (bvh.nodes[i].children[s] & LEAF_FLAG)) continue;
                const uint32_t c = bvh.nodes[i].children[s];
                if (seen[c]) return false;
                seen[c] = 1;
                stack.push_back({c, depth + 1u});
            }
        }
    }
    // The collapse is only equivalent when every internal child's box contains the boxes of that child's own children
    // (exactly, as min/max unions do: the reference's builder, any sane builder).  A host tree that breaks this keeps
    // its faithful records for every kernel: the reference's walk of it is then the only definition of a hit.
    for (uint32_t i = 0; i < n; ++i) {
        const PtxBvh8Node& nd = bvh.nodes[i];
        for (int sl = 0; sl < 8; ++sl) {
            if (!(nd.valid_mask & (1u << sl)) || (nd.children[sl] & LEAF_FLAG)) continue;
            const PtxBvh8Node& cn = bvh.nodes[nd.children[sl]];
            for (int t = 0; t < 8; ++t) {
                if (!(cn.valid_mask & (1u << t))) continue;
                const bool inside = nd.min_x[sl] <= cn.min_x[t] && nd.min_y[sl] <= cn.min_y[t] && nd.min_z[sl] <= cn.min_z[t] &&
                                    nd.max_x[sl] >= cn.max_x[t] && nd.max_y[sl] >= cn.max_y[t] && nd.max_z[sl] >= cn.max_z[t];
                if (!inside) return true;      // faithful roots everywhere (set above)
            }
        }
    }
    if (nests) *nests = true;
    if (ray_width > 1u && !repack_wide(bvh, count, out, ray_width, root_ray)) return false;
    if (packet_width > 1u) {
        if (packet_width == ray_width) *root_packet = *root_ray;
        else if (!repack_wide(bvh, count, out, packet_width, root_packet)) return false;
    }
    return true;
}
static bool repack_wide(const PtxBvh8& bvh, const std::vector<uint32_t>& count, std::vector<NodeRec>& out, uint32_t wide_width, uint32_t* root_wide) {
    auto internal_word = [](uint32_t abs_unit, uint32_t cnt) { return abs_unit | ((cnt - 1u) << NODE_NCHILD_SHIFT); };
    std::vector<RepackItem> list;
    std::vector<uint32_t> words;
    // ---- wide tree: replace an internal child by its own ordered list, largest surface area first, while the record
    // stays within wide_width children; the depth-first order of the leaves is unchanged
    const uint32_t width = std::min<uint32_t>(wide_width, 8u);
    // iterative (explicit stack): build a record for `node`, then one for each of its remaining internal children
    std::vector<std::pair<uint32_t, size_t>> todo;       // (reference node, position of the child word to patch: unit index * 64 + child k, or ~0 for the root)
    todo.push_back({0u, ~(size_t)0});
    std::vector<RepackItem> sub;
    while (!todo.empty()) {
        const auto [node, patch] = todo.back();
        todo.pop_back();
        ordered_children(bvh.nodes[node], list);
        for (;;) {
            int best = -1;
            float best_area = -1.0f;
            for (size_t k = 0; k < list.size(); ++k) {
                if (list[k].word & LEAF_FLAG) continue;
                if (list.size() - 1 + count[list[k].word] > width) continue;
                const float* b = list[k].box;
                const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
                const float area = (dx * dy + dy * dz) + dz * dx;
                if (!(area <= best_area)) { best_area = area; best = (int)k; }     // NaN areas are taken too: any choice is valid
            }
            if (best < 0) break;
            ordered_children(bvh.nodes[list[best].word], sub);
            list.erase(list.begin() + best);
            list.insert(list.begin() + best, sub.begin(), sub.end());
        }
        const size_t rec = out.size();
        if (rec + 2 > NODE_INDEX_MASK) return false;
        out.resize(rec + (list.size() > 4 ? 2 : 1));
        words.assign(list.size(), 0u);
        for (size_t k = 0; k < list.size(); ++k) {
            if (list[k].word & LEAF_FLAG) words[k] = list[k].word;
            else todo.push_back({list[k].word, rec * 64 + k});      // patched when the child's record exists
        }
        fill_record(&out[rec], list, words, 0u);
        const uint32_t my_word = internal_word((uint32_t)rec, (uint32_t)list.size());
        if (patch == ~(size_t)0) *root_wide = my_word;
        else {
            const size_t prec = patch / 64, k = patch % 64;
            float f; std::memcpy(&f, &my_word, 4);
            out[prec + k / 4].c[2 * (k % 4) + 1].z = f;
        }
    }
    return true;
}

// forwarded_bound (pt_traverse.h) culls a mesh node by its entry distance before any of its triangles has been seen,
// which presumes that a triangle lies inside the boxes above it: every leaf box must contain its triangles' vertices
// and so must the Blas box (the mesh bounds: the unpadded vertex extent, shape/triangle_mesh.rs:23,54-59).  True of the
// reference's builder (leaf boxes are unions of triangle boxes padded by 1e-5: shape/triangle_mesh.rs:25-38); a host
// tree that breaks it is walked without forwarding.
static bool mesh_boxes_hold_geometry(const PtxMesh& m) {
    const PtxBvh8& bvh = m.bvh;
    for (uint32_t i = 0; i < bvh.n_nodes; ++i) {
        const PtxBvh8Node& nd = bvh.nodes[i];
        for (int s = 0; s < 8; ++s) {
            if (!(nd.valid_mask & (1u << s))) continue;
            const float lo[3] = {nd.min_x[s], nd.min_y[s], nd.min_z[s]}, hi[3] = {nd.max_x[s], nd.max_y[s], nd.max_z[s]};
            if (!(nd.children[s] & LEAF_FLAG)) continue;
            const uint32_t start = nd.children[s] & PRIM_START_MASK, cnt = ((nd.children[s] >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1u;
            for (uint32_t k = start; k < start + cnt; ++k) {
                const uint32_t* tri = &m.indices[3 * (size_t)bvh.primitive_indices[k]];
                for (int v = 0; v < 3; ++v) {
                    const float* p = &m.positions[3 * (size_t)tri[v]];
                    for (int a = 0; a < 3; ++a) if (!(lo[a] <= p[a] && p[a] <= hi[a] && m.bounds_min[a] <= p[a] && p[a] <= m.bounds_max[a])) return false;
                }
            }
        }
    }
    return true;
}

// Shape certificate.  intersect_triangle (shape/triangle_mesh.rs:63-100) divides by a = e1 . (d x e2) = |d| |e1| |e2|
// sin(phi) cos(theta) (phi = the angle between the two edges at v0, theta = between the ray and the normal); its t, u, v
// carry a relative rounding error of about eps / (sin(phi) |cos(theta)|).  The forwarded bound tolerates 3 %, the
// nearest-first walk 0.4 % (pt_traverse.h) before a computed hit can sit in front of a box that walk has culled while
// the reference's walk still visits it.  Measured (tests/probe_adversarial.py, 2 M random + 320 k aimed rays per scene
// against clouds of needle triangles with well-shaped walls among them): random rays never differ from the reference,
// whatever the aspect (up to 10^6); rays AIMED at a triangle hidden behind a nearer surface, within an angle theta_c of
// its plane, do: theta_c ~ 10^-6 rad for aspect 1-4, 10^-5 for 10-110, 10^-3 for 300-1000, any angle from 10^4 on.
// The certificate draws the line at sin(phi) >= 2^-9 (aspect ~500, phi ~0.11 degrees): a mesh with a sharper triangle is
// walked as the reference walks it, by every ray that could reach it (the procedural scenes' sharpest: 0.29 degrees).
// What stays presumed for certified meshes is the statement above for aimed rays within ~10^-3 rad; PTX_EXACT_ORDER=1
// PTX_NO_FORWARD=1 drops both shortcuts for anyone who needs the reference's garbage hits as well.
// Edges of length zero are fine: a is then exactly 0 and the triangle is always rejected (|a| < 1e-8).
constexpr double SHAPE_MIN_SIN = 1.0 / 512.0;
static bool mesh_shapes_certified(const PtxMesh& m) {
    for (uint32_t t = 0; t < m.n_triangles; ++t) {
        const uint32_t* tri = &m.indices[3 * (size_t)t];
        const float* p0 = &m.positions[3 * (size_t)tri[0]]; const float* p1 = &m.positions[3 * (size_t)tri[1]]; const float* p2 = &m.positions[3 * (size_t)tri[2]];
        // the f32 edges the kernels use (TriRec)
        const double e1[3] = {(double)(p1[0] - p0[0]), (double)(p1[1] - p0[1]), (double)(p1[2] - p0[2])};
        const double e2[3] = {(double)(p2[0] - p0[0]), (double)(p2[1] - p0[1]), (double)(p2[2] - p0[2])};
        const double n1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), n2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        if (n1 == 0.0 || n2 == 0.0) continue;
        const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double area2 = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        if (!(area2 >= SHAPE_MIN_SIN * n1 * n2)) return false;      // also NaN / inf coordinates
    }
    return true;
}

// META_UNCERT on every TLAS child whose subtree holds an instance of an uncertified mesh (all TLAS trees: records
// [0, n_records) of `nodes`).  Returns whether the record itself holds one.
static bool flag_uncertified(std::vector<NodeRec>& nodes, uint32_t word, const std::vector<uint32_t>& tlas_prims,
                             const std::vector<uint8_t>& inst_uncert, uint32_t depth = 0) {
    const uint32_t unit = word & NODE_INDEX_MASK, nchild = ((word >> NODE_NCHILD_SHIFT) & NODE_NCHILD_MASK) + 1u;
    bool any = false;
    if (depth > REPACK_MAX_DEPTH + 8) return false;
    for (uint32_t k = 0; k < nchild; ++k) {
        float4& hi = nodes[unit + k / 4].c[2 * (k % 4) + 1];
        uint32_t cw, meta;
        std::memcpy(&cw, &hi.z, 4); std::memcpy(&meta, &hi.w, 4);
        bool f = false;
        if (cw & LEAF_FLAG) {
            const uint32_t start = cw & PRIM_START_MASK, cnt = ((cw >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1u;
            for (uint32_t j = start; j < start + cnt && j < tlas_prims.size(); ++j) f = f || inst_uncert[tlas_prims[j]] != 0;
        } else f = flag_uncertified(nodes, cw, tlas_prims, inst_uncert, depth + 1);
        if (f) { meta |= META_UNCERT; std::memcpy(&hi.w, &meta, 4); any = true; }
    }
    return any;
}

static void pack_cols(const float* m, float4* dst) {
    // columns x,y,z,w (xyz of each) -> 3 float4
    dst[0] = make_float4(m[0], m[1], m[2], m[4]);
    dst[1] = make_float4(m[5], m[6], m[8], m[9]);
    dst[2] = make_float4(m[10], m[12], m[13], m[14]);
}

static int32_t ensure_lists(PtxContext* ctx, uint64_t n) {
    if (n <= ctx->list_cap) return PTX_OK;
    for (uint32_t** p : {&ctx->list_a, &ctx->list_b, &ctx->list_c, &ctx->list_d}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    HIPCHK(hipMalloc((void**)&ctx->list_a, n * 3 * 4));
    HIPCHK(hipMalloc((void**)&ctx->list_b, n * 4));
    HIPCHK(hipMalloc((void**)&ctx->list_c, n * 4));
    HIPCHK(hipMalloc((void**)&ctx->list_d, n * 4));
    ctx->list_cap = n;
    return PTX_OK;
}

static int32_t ensure_capacity(PtxContext* ctx, uint64_t n) {
    if (n <= ctx->capacity) return PTX_OK;
    free_pool(ctx->path_allocs);
    ctx->capacity = 0;
    PathArrays& P = ctx->P;
    auto& pool = ctx->path_allocs;
    HIPCHK(dev_alloc(pool, &P.ray_o, n)); HIPCHK(dev_alloc(pool, &P.ray_d, n)); HIPCHK(dev_alloc(pool, &P.thr, n));
    HIPCHK(dev_alloc(pool, &P.rad, n)); HIPCHK(dev_alloc(pool, &P.rng, n)); HIPCHK(dev_alloc(pool, &P.hit, n));
    HIPCHK(dev_alloc(pool, &P.hit_tri, n)); HIPCHK(dev_alloc(pool, &P.sh_o, n)); HIPCHK(dev_alloc(pool, &P.sh_d, n));
    HIPCHK(dev_alloc(pool, &P.c_sun, n)); HIPCHK(dev_alloc(pool, &P.c_sky, n)); HIPCHK(dev_alloc(pool, &P.c_post, n));
    HIPCHK(dev_alloc(pool, &P.w_t, n)); HIPCHK(dev_alloc(pool, &P.w_b, n)); HIPCHK(dev_alloc(pool, &P.aov_a, n));
    HIPCHK(dev_alloc(pool, &P.aov_n, n)); HIPCHK(dev_alloc(pool, &P.sample, n));
    HIPCHK(dev_alloc(pool, &ctx->q[0], n)); HIPCHK(dev_alloc(pool, &ctx->q[1], n));
    HIPCHK(dev_alloc(pool, &ctx->q_shadow, 2 * n)); HIPCHK(dev_alloc(pool, &ctx->q_final, n));
    HIPCHK(dev_alloc(pool, &ctx->q_redo, n));
    HIPCHK(dev_alloc(pool, &ctx->q_walk[0], n)); HIPCHK(dev_alloc(pool, &ctx->q_walk[1], n));
    ctx->capacity = n;
    return PTX_OK;
}

static int32_t ensure_images(PtxContext* ctx, uint64_t n_pixels) {
    if (n_pixels <= ctx->img_cap) return PTX_OK;
    if (ctx->img) (void)hipFree(ctx->img);
    ctx->img = nullptr; ctx->img_cap = 0;
    HIPCHK(hipMalloc((void**)&ctx->img, n_pixels * 9 * sizeof(float)));
    ctx->img_cap = n_pixels;
    return PTX_OK;
}

// owned pixels in 8x8-tile order: one wave of primary rays == one tile
static int32_t ensure_worklist(PtxContext* ctx, uint32_t w, uint32_t h) {
    if (ctx->wl_pixel && ctx->wl_w == w && ctx->wl_h == h && ctx->wl_rank == ctx->rank && ctx->wl_world == ctx->world) return PTX_OK;
    std::vector<uint32_t> list;
    list.reserve((size_t)w * h / ctx->world + 64);
    const uint32_t tx_n = (w + 7) / 8, ty_n = (h + 7) / 8;
    for (uint32_t ty = 0; ty < ty_n; ++ty)
        for (uint32_t tx = 0; tx < tx_n; ++tx) {
            const uint32_t t = ty * tx_n + tx;
            if (t % ctx->world != ctx->rank) continue;
            for (uint32_t ly = 0; ly < 8; ++ly)
                for (uint32_t lx = 0; lx < 8; ++lx) {
                    const uint32_t x = tx * 8 + lx, y = ty * 8 + ly;
                    if (x < w && y < h) list.push_back(y * w + x);
                }
        }
    if (ctx->wl_pixel) (void)hipFree(ctx->wl_pixel);
    ctx->wl_pixel = nullptr;
    HIPCHK(hipMalloc((void**)&ctx->wl_pixel, std::max<size_t>(list.size(), 1) * 4));
    if (!list.empty()) HIPCHK(hipMemcpy(ctx->wl_pixel, list.data(), list.size() * 4, hipMemcpyHostToDevice));
    ctx->wl_w = w; ctx->wl_h = h; ctx->wl_rank = ctx->rank; ctx->wl_world = ctx->world; ctx->wl_n = (uint32_t)list.size();
    return PTX_OK;
}

constexpr uint32_t SSS_POLL_EVERY = 8;
enum { CLS_RAYGEN = 0, CLS_EXTEND, CLS_SHADE, CLS_SHADOW, CLS_ACCUM, CLS_OTHER, CLS_PACKET, CLS_WALK, CLS_TAIL };
struct ScopedEvent {
    PtxContext* ctx; int idx = -1;
    ScopedEvent(PtxContext* c, int cls, hipStream_t on = nullptr) : ctx(c) {
        if (!c->profiling) return;
        PtxContext::Ev e; e.cls = cls; e.stream = on ? on : c->stream;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
        (void)hipEventRecord(e.a, e.stream);
        c->events.push_back(e); idx = (int)c->events.size() - 1;
    }
    ~ScopedEvent() { if (idx >= 0) (void)hipEventRecord(ctx->events[idx].b, ctx->events[idx].stream); }
};
static void drain_events(PtxContext* ctx) {
    if (ctx->events.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    for (auto& e : ctx->events) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            switch (e.cls) {
                case CLS_RAYGEN: ctx->timings.raygen_ms += ms; break;
                case CLS_EXTEND: ctx->timings.extend_ms += ms; ctx->timings.extend_launches++; break;
                case CLS_PACKET: ctx->timings.packet_ms += ms; ctx->timings.packet_launches++; break;
                case CLS_SHADE: ctx->timings.shade_ms += ms; ctx->timings.shade_launches++; break;
                case CLS_SHADOW: ctx->timings.shadow_ms += ms; ctx->timings.shadow_launches++; break;
                case CLS_ACCUM: ctx->timings.accumulate_ms += ms; break;
                case CLS_WALK: ctx->timings.walk_ms += ms; ctx->timings.walk_launches++; break;
                case CLS_TAIL: ctx->timings.tail_ms += ms; ctx->timings.tail_launches++; break;
                default: ctx->timings.other_ms += ms; break;
            }
        }
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    ctx->events.clear();
}

static int32_t check_device_error(PtxContext* ctx) {
    // d_err is read back through pinned memory on the library stream
    HIPCHK(hipMemcpyAsync(ctx->h_pinned, ctx->d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->h_pinned[0] != 0) {
        HIPCHK(hipMemsetAsync(ctx->d_err, 0, 4, ctx->stream));
        if (ctx->h_pinned[0] & PTX_DEV_ERR_TAIL) return fail(ctx, PTX_ERR_HIP, "internal: the tail kernel was launched with too few waves for the live paths");
        return fail(ctx, PTX_ERR_STACK_OVERFLOW, "a traversal needed more than the 144 stack entries a ray has on the device (the reference panics beyond 64 per BVH level)");
    }
    return PTX_OK;
}

// Primary rays as packets (+ the per-ray pass over the rays of packets whose shared stack overflowed: normally none).
static int32_t launch_primary_packets(PtxContext* ctx, const uint32_t* q, IterCtr* c, uint64_t n) {
    IterCtr* redo = &ctx->ctrs[CTR_REDO];
    HIPCHK(hipMemsetAsync(redo, 0, sizeof(IterCtr), ctx->stream));
    const int grid = (int)std::min<uint64_t>(PACKET_GRID, (n + TRACE_BLOCK - 1) / TRACE_BLOCK);
    {
        ScopedEvent ev(ctx, CLS_PACKET);
        hipLaunchKernelGGL(k_extend_packet, dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, q, c, ctx->ovf, ctx->d_err,
                           ctx->S.nodes, ctx->S.tris, ctx->S.inst_tlas, ctx->S.inst, ctx->S.tlas_prims, ctx->packet_capacity, ctx->q_redo, redo, ctx->d_counters);
    }
    {
        ScopedEvent ev(ctx, CLS_OTHER);
        hipLaunchKernelGGL(k_extend_redo, dim3((int)std::min<uint64_t>(TRACE_GRID, (n + TRACE_BLOCK - 1) / TRACE_BLOCK)), dim3(TRACE_BLOCK), 0, ctx->stream,
                           ctx->S, ctx->P, ctx->q_redo, redo, ctx->ovf, ctx->d_counters, ctx->d_err);
    }
    return PTX_OK;
}

// Trace n path slots that k_raygen is about to initialise from (wl_pixel, wl_sample | fb counts | uniform).
static int32_t run_paths(PtxContext* ctx, const uint32_t* wl_pixel, const uint32_t* wl_sample, const uint32_t* per_pixel_counts,
                         uint32_t uniform_sample, uint32_t n, uint32_t w, uint32_t h, uint32_t max_bounces, const Target& tg, bool aov,
                         uint32_t batch = 0) {
    if (n == 0) return PTX_OK;
    if ((uint64_t)w * h > PIXEL_MASK) return fail(ctx, PTX_ERR_UNSUPPORTED, "frames above 2^29 pixels are not supported");
    // the bounce counter of a path is 8 bits wide (FL_BOUNCE_MASK); the reference's MAX_BOUNCES is 8 or 16
    if (max_bounces > FL_BOUNCE_MASK) return fail(ctx, PTX_ERR_INVALID, "max_bounces above 255 is not supported");
    int32_t st = ensure_capacity(ctx, n);
    if (st != PTX_OK) return st;
    if (max_bounces == 0) {
        // `for bounce in 0..0` (pathtracer.rs:117): no ray is traced; every sample is clamp_radiance(0) = 0 and still counts
        HIPCHK(hipMemsetAsync(ctx->ctrs, 0, sizeof(IterCtr), ctx->stream));
        hipLaunchKernelGGL(k_raygen, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->S, ctx->cam, ctx->P, wl_pixel, wl_sample,
                           per_pixel_counts, uniform_sample, n, w, h, ctx->q[0], ctx->ctrs, aov ? 1u : 0u, (PtxRay*)nullptr, batch);
        HIPCHK(hipMemcpyAsync(&ctx->ctrs[0].n_final, &ctx->ctrs[0].n_active, 4, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(k_finalize, dim3((int)std::min<uint64_t>(256 * 8, (n + SHADE_BLOCK - 1) / SHADE_BLOCK)), dim3(SHADE_BLOCK), 0, ctx->stream,
                           ctx->P, ctx->q[0], ctx->ctrs, tg, ctx->d_counters, ctx->counting ? 1u : 0u);
        HIPCHK(hipGetLastError());
        return PTX_OK;
    }
    const bool sss = ctx->S.has_sss != 0;
    // A subsurface walk takes one iteration per step (up to 64 per event) on top of the path's bounces: the iteration
    // count is data dependent, the loop below polls the live counts.
    const uint32_t iter_cap = sss ? MAX_ITERS - 1 : std::min<uint32_t>(max_bounces, MAX_ITERS - 1);
    HIPCHK(hipMemsetAsync(ctx->ctrs, 0, sizeof(IterCtr) * (size_t)(sss ? MAX_ITERS : iter_cap + 1), ctx->stream));
    {
        ScopedEvent ev(ctx, CLS_RAYGEN);
        hipLaunchKernelGGL(k_raygen, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->S, ctx->cam, ctx->P, wl_pixel, wl_sample,
                           per_pixel_counts, uniform_sample, n, w, h, ctx->q[0], ctx->ctrs, aov ? 1u : 0u, (PtxRay*)nullptr, batch);
    }
    const bool coherent = batch >= 16u || wl_pixel == ctx->wl_pixel;   // a wave's slots are samples of one or two pixels, or a tile of the owned-pixel list
    const int trace_grid = (int)std::min<uint64_t>(TRACE_GRID, (n + TRACE_BLOCK - 1) / TRACE_BLOCK);
    // k_shade: one block per 256 queue entries, not a capped grid-stride grid: the queues it writes then keep the order of
    // the queue it reads more closely (blocks claim their output ranges roughly in order), which the next extend
    // launch feels (383 -> 375 ms), and the kernel itself runs 5 % faster
    const int shade_grid = (int)std::min<uint64_t>((uint64_t)256 * PT_SHADE_GRID_PER_CU, (n + SHADE_BLOCK - 1) / SHADE_BLOCK);
    const int final_grid = (int)std::min<uint64_t>(256 * 8, (n + SHADE_BLOCK - 1) / SHADE_BLOCK);
    bool drained = false;
    const bool side = ctx->overlap_shadow && !ctx->counting && ctx->side_stream != nullptr;
    bool side_pending = false;      // work on the side stream the main stream has not been ordered behind yet
    bool tail_recorded = false;     // ev_tail: "the heavy iterations of this batch are enqueued" (ptx_order_after_frame_tail)
    // An early return below (a failed HIP call, the subsurface poll, a packet launch) must not leave side-stream kernels
    // running behind the caller's next main-stream call: they still write P and the framebuffer.
    struct SideJoin {
        PtxContext* c; bool armed = false;
        ~SideJoin() { if (armed && c->side_stream) (void)hipStreamSynchronize(c->side_stream); }
    } side_join{ctx};
    // k_tail takes the batch over at the first iteration whose live paths fit its waves (that iteration's closest-hit
    // pass is still an ordinary launch).  `live_ub` is what the host knows: the batch size, then the counts it reads back.
    const uint64_t tail_cap = std::min<uint64_t>(ctx->tail_cap, (uint64_t)TAIL_GRID * (TAIL_PER_WAVE - 2));      // - 2: a wave's two slices are rounded up
    const bool tail = ctx->tail_kernel && ctx->tail_q != nullptr && !ctx->counting && !ctx->group_lanes;
    uint64_t live_ub = n;
    bool poll_pending = false;
    for (uint32_t it = 0; it < iter_cap; ++it) {
        uint32_t* qa = ctx->q[it & 1];
        uint32_t* qb = ctx->q[(it + 1) & 1];
        uint32_t* qwa = ctx->q_walk[it & 1];          // subsurface walks in flight: their step rays are traced with qa
        uint32_t* qwb = ctx->q_walk[(it + 1) & 1];
        IterCtr* c = ctx->ctrs + it;
        const bool poll = poll_pending;     // this iteration's live count is on its way to the host (enqueued behind k_shade(it - 1), below)
        poll_pending = false;
        {
            // primary rays of neighbouring queue entries are coherent (a pixel's samples, or an 8x8 tile): packets
            const bool packets = it == 0 && coherent && ctx->packet_primary && !ctx->counting;
            if (packets) {
                st = launch_primary_packets(ctx, qa, c, n);
                if (st != PTX_OK) return st;
            } else {
                const bool fast = ctx->S.fast_on != 0u && !ctx->counting;
                ScopedEvent ev(ctx, CLS_EXTEND);
                if (ctx->group_lanes && !ctx->counting) {
                    // 64 / G rays per wave: G times as many waves for the same queue
                    const int gg = (int)std::min<uint64_t>(TRACE_GRID, ((uint64_t)n * ctx->group_lanes + TRACE_BLOCK - 1) / TRACE_BLOCK);
                    if (ctx->group_lanes == 8) hipLaunchKernelGGL(k_extend_group<8>, dim3(gg), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qwa, c, ctx->d_err);
                    else hipLaunchKernelGGL(k_extend_group<4>, dim3(gg), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qwa, c, ctx->d_err);
                } else
                if (ctx->counting && (ctx->S.fwd_on == 2u || (ctx->count_shipped && ctx->S.fwd_on)) && ctx->S.fast_on) {
                    // ptx_set_counting(ctx, 2) / PTX_COUNT_FORWARD=1: count what the shipped per-ray walk does (its node visits,
                    // triangle tests and entries, second attempts included) instead of what the reference's walk does
                    SceneDev Sc = ctx->S;
                    Sc.fwd_on = 2u;
                    hipLaunchKernelGGL((k_extend<true, true, true>), dim3(trace_grid), dim3(TRACE_BLOCK), 0, ctx->stream, Sc, ctx->P, qa, qwa, c, ctx->ovf, ctx->d_counters, ctx->d_err);
                }
                else if (ctx->counting) hipLaunchKernelGGL((k_extend<true, false>), dim3(trace_grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qwa, c, ctx->ovf, ctx->d_counters, ctx->d_err);
                else if (fast && ctx->n_uncertified) hipLaunchKernelGGL((k_extend<false, true, true>), dim3(trace_grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qwa, c, ctx->ovf, ctx->d_counters, ctx->d_err);
                else if (fast) hipLaunchKernelGGL((k_extend<false, true, false>), dim3(trace_grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qwa, c, ctx->ovf, ctx->d_counters, ctx->d_err);
                else hipLaunchKernelGGL((k_extend<false, false>), dim3(trace_grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qwa, c, ctx->ovf, ctx->d_counters, ctx->d_err);
            }
        }
        if (tail) {
            if (poll) {
                HIPCHK(hipEventSynchronize(ctx->ev_poll));
                live_ub = (uint64_t)ctx->h_pinned[20] + ctx->h_pinned[23];
                if (live_ub == 0) { drained = true; break; }        // (the launch above found an empty queue)
            }
            if (live_ub <= tail_cap) {
                // as few paths per wave as the resident waves allow (a wave's phase lasts as long as its longest ray)
                const uint64_t per_wave = std::min<uint64_t>(TAIL_PER_WAVE - 2, std::max<uint64_t>(ctx->tail_min_per_wave, (live_ub + TAIL_GRID - 1) / TAIL_GRID));
                const int tg_grid = (int)std::max<uint64_t>(1, (live_ub + per_wave - 1) / per_wave);
                uint32_t* left = &ctx->ctrs[CTR_TAIL].n_active;
                HIPCHK(hipMemsetAsync(left, 0, 4, ctx->stream));
                // iteration it - 1's any-hit launch and k_finalize (side stream) read and zero the pending terms k_tail's
                // shading adds; they ran beside the closest-hit launch above
                if (side_pending) { HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0)); side_pending = false; }
                {
                    ScopedEvent ev(ctx, CLS_TAIL);
                    const bool fast = ctx->S.fast_on != 0u;
                    const uint32_t iters = iter_cap - it;
                    if (fast && ctx->n_uncertified) hipLaunchKernelGGL((k_tail<true, true>), dim3(tg_grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qwa, c, ctx->tail_q, left, ctx->ovf, ctx->d_counters, ctx->d_err, tg, max_bounces, iters, 1u);
                    else if (fast) hipLaunchKernelGGL((k_tail<true, false>), dim3(tg_grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qwa, c, ctx->tail_q, left, ctx->ovf, ctx->d_counters, ctx->d_err, tg, max_bounces, iters, 1u);
                    else hipLaunchKernelGGL((k_tail<false, false>), dim3(tg_grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qwa, c, ctx->tail_q, left, ctx->ovf, ctx->d_counters, ctx->d_err, tg, max_bounces, iters, 1u);
                }
                if (sss) {
                    HIPCHK(hipMemcpyAsync(ctx->h_pinned + 24, left, 4, hipMemcpyDeviceToHost, ctx->stream));
                    HIPCHK(hipStreamSynchronize(ctx->stream));
                    drained = ctx->h_pinned[24] == 0;
                }
                break;
            }
        }
        if (it == ctx->tail_gate_iter) { HIPCHK(hipEventRecord(ctx->ev_tail, ctx->stream)); tail_recorded = true; }
        // iteration it - 1's any-hit launch (side stream) zeroes the pending terms this k_shade adds, and reads the shadow
        // and final queues this k_shade overwrites
        if (side && it > 0) { HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0)); side_pending = false; }
        {
            ScopedEvent ev(ctx, CLS_SHADE);
            if (ctx->counting) hipLaunchKernelGGL(k_shade<true>, dim3(final_grid), dim3(SHADE_BLOCK)   /* counted: few blocks, each ends with 7 counter atomics per wave */, 0, ctx->stream, ctx->S, ctx->P, qa, qb, ctx->q_shadow, ctx->q_final, qwb, c, c + 1, max_bounces, ctx->d_counters);
            else if (ctx->sky_ahead) hipLaunchKernelGGL((k_shade<false, true>), dim3(shade_grid), dim3(SHADE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qb, ctx->q_shadow, ctx->q_final, qwb, c, c + 1, max_bounces, ctx->d_counters);
            else hipLaunchKernelGGL(k_shade<false>, dim3(shade_grid), dim3(SHADE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qa, qb, ctx->q_shadow, ctx->q_final, qwb, c, c + 1, max_bounces, ctx->d_counters);
        }
        if (sss && it > 0) {
            // one state transition for every path in a subsurface event: continuing walks join qwb, finished events qb
            // (and this iteration's shadow queue), dead paths q_final
            ScopedEvent ev(ctx, CLS_WALK);
            // (one block per 256 slots the queue could hold, like k_shade: the uncounted kernel has no grid-stride loop)
            const int wg = (int)((n + WALK_BLOCK - 1) / WALK_BLOCK);
            if (ctx->counting) hipLaunchKernelGGL(k_sss_step<true>, dim3(std::min(wg, 256 * 8)), dim3(WALK_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qwa, qwb, qb, ctx->q_shadow, ctx->q_final, c, c + 1, max_bounces, ctx->d_counters);
            else hipLaunchKernelGGL(k_sss_step<false>, dim3(wg), dim3(WALK_BLOCK), 0, ctx->stream, ctx->S, ctx->P, qwa, qwb, qb, ctx->q_shadow, ctx->q_final, c, c + 1, max_bounces, ctx->d_counters);
        }
        // The any-hit launch and k_finalize of this iteration depend on this iteration's k_shade / k_sss_step only, and the
        // next iteration's closest-hit launch does not depend on them: they go to the side stream and share the GPU with
        // that launch, so neither kernel's tail (waves waiting for their last long ray) leaves the machine idle.
        hipStream_t ss = side ? ctx->side_stream : ctx->stream;
        uint2* sovf = side ? ctx->ovf_side : ctx->ovf;
        if (side) {
            side_join.armed = true;
            HIPCHK(hipEventRecord(ctx->ev_main, ctx->stream));
            HIPCHK(hipStreamWaitEvent(ss, ctx->ev_main, 0));
        }
        if (tail && live_ub > tail_cap && ctx->tail_poll && it + 1 >= ctx->tail_first_poll && it + 1 < iter_cap) {
            // The next iteration's live count goes to the host from the side stream: the next closest-hit launch (main
            // stream) is then the first kernel ready when k_shade ends, as without the copy (the any-hit launch taking the
            // wave slots first costs 1 ms per heavy iteration), and the host's wait ends while that launch has the GPU busy.
            HIPCHK(hipMemcpyAsync(ctx->h_pinned + 20, c + 1, 16, hipMemcpyDeviceToHost, ss));      // n_active .. n_walk
            HIPCHK(hipEventRecord(ctx->ev_poll, ss));
            poll_pending = true;
        }
        {
            ScopedEvent ev(ctx, CLS_SHADOW, ss);
            const int sg = (int)std::min<uint64_t>(SHADOW_GRID, (2ull * n + TRACE_BLOCK - 1) / TRACE_BLOCK);
            if (it == 0 && coherent && ctx->packet_shadow && !ctx->counting)
                hipLaunchKernelGGL(k_shadow_packet, dim3(std::min(sg, PACKET_GRID)), dim3(TRACE_BLOCK), 0, ss, ctx->S, ctx->P, ctx->q_shadow, c, sovf, ctx->d_err,
                                   ctx->S.nodes, ctx->S.tris, ctx->S.inst_tlas, ctx->S.inst, ctx->S.tlas_prims);
            else if (ctx->counting) hipLaunchKernelGGL(k_shadow<true>, dim3(sg), dim3(TRACE_BLOCK), 0, ss, ctx->S, ctx->P, ctx->q_shadow, c, sovf, ctx->d_counters, ctx->d_err);
            else hipLaunchKernelGGL(k_shadow<false>, dim3(sg), dim3(TRACE_BLOCK), 0, ss, ctx->S, ctx->P, ctx->q_shadow, c, sovf, ctx->d_counters, ctx->d_err);
        }
        {
            ScopedEvent ev(ctx, CLS_ACCUM, ss);
            hipLaunchKernelGGL(k_finalize, dim3(final_grid), dim3(SHADE_BLOCK), 0, ss, ctx->P, ctx->q_final, c, tg, ctx->d_counters, ctx->counting ? 1u : 0u);
        }
        if (side) { HIPCHK(hipEventRecord(ctx->ev_side, ss)); side_pending = true; }
        if (sss && it + 1 >= max_bounces && ((it + 1 - max_bounces) % SSS_POLL_EVERY == 0u || it + 1 == iter_cap)) {
            // subsurface walks make the iteration count data dependent: poll the live count every few iterations (the
            // launches of an iteration whose queues are empty leave in microseconds; a host round trip per iteration
            // does not)
            HIPCHK(hipMemcpyAsync(ctx->h_pinned + 16, &ctx->ctrs[it + 1], 16, hipMemcpyDeviceToHost, ctx->stream));     // n_active .. n_walk
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (ctx->h_pinned[16] == 0 && ctx->h_pinned[19] == 0) { drained = true; break; }
        }
    }
    // whatever follows on the main stream (accumulation, readbacks, the next batch's k_raygen) comes after the side stream's work
    if (side_pending) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
    side_join.armed = false;        // joined through the event: no host wait on the normal road
    if (!tail_recorded) HIPCHK(hipEventRecord(ctx->ev_tail, ctx->stream));      // a batch with fewer iterations than the gate
    HIPCHK(hipGetLastError());
    // Without SSS every path ends by iteration max_bounces (k_shade retires it at bounce >= max_bounces).  With SSS a
    // path that is still walking after MAX_ITERS - 1 iterations would never reach k_finalize: refuse instead of
    // returning a frame with samples missing.
    if (sss && !drained) return fail(ctx, PTX_ERR_UNSUPPORTED, "paths still alive after 4095 wavefront iterations (subsurface walks x bounces)");
    if (ctx->counting) {
        // primary rays are counted on the host side of the launch: one per slot
        ctx->timings.other_ms += 0.0;
    }
    return PTX_OK;
}

#define REQUIRE_CTX() do { if (!ctx) return fail(nullptr, PTX_ERR_INVALID, "null context"); (void)hipSetDevice(ctx->device); } while (0)
#define REQUIRE_SCENE() do { REQUIRE_CTX(); if (!ctx->has_scene) return fail(ctx, PTX_ERR_NO_SCENE, "no scene uploaded"); } while (0)
#define REQUIRE_CAMERA() do { REQUIRE_SCENE(); if (!ctx->has_camera) return fail(ctx, PTX_ERR_INVALID, "no camera set"); } while (0)

static void add_primary_count(PtxContext* ctx, uint64_t n) {
    if (!ctx->counting) return;
    unsigned long long v = n;
    // host-side bump of one device counter (rare, tiny)
    unsigned long long cur = 0;
    (void)hipMemcpyAsync(&cur, &ctx->d_counters->v[C_PRIMARY], 8, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    cur += v;
    (void)hipMemcpyAsync(&ctx->d_counters->v[C_PRIMARY], &cur, 8, hipMemcpyHostToDevice, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
}

extern "C" {

const char* ptx_last_error(const PtxContext* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int32_t ptx_create(int32_t device_index, PtxContext** out_ctx) {
    PtxContext* ctx = nullptr;
    if (!out_ctx) return fail(nullptr, PTX_ERR_INVALID, "out_ctx is null");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(nullptr, PTX_ERR_NO_DEVICE, "no HIP device visible");
    if (device_index < 0 || device_index >= count) return fail(nullptr, PTX_ERR_NO_DEVICE, "device index out of range");
    HIPCHK(hipSetDevice(device_index));
    PtxContext* c = new PtxContext();
    c->device = device_index;
    ctx = c;
    if (const char* e = std::getenv("PTX_PATHS_IN_FLIGHT")) {
        const unsigned long long v = std::strtoull(e, nullptr, 10);
        if (v) c->max_paths_in_flight = v;
    }
    if (const char* e = std::getenv("PTX_PACKET_PRIMARY")) c->packet_primary = std::strtoul(e, nullptr, 10) != 0;
    if (const char* e = std::getenv("PTX_PACKET_SHADOW")) c->packet_shadow = std::strtoul(e, nullptr, 10) != 0;
    if (const char* e = std::getenv("PTX_PACKET_CAPACITY")) c->packet_capacity = std::max(8, std::min((int)std::strtol(e, nullptr, 10), PACKET_LDS + OVF_STACK));
    if (const char* e = std::getenv("PTX_PACKET_BATCH")) c->packet_batch = std::strtoul(e, nullptr, 10) != 0;
    if (const char* e = std::getenv("PTX_WIDE_BVH")) c->packet_width = (uint32_t)std::min<unsigned long>(std::strtoul(e, nullptr, 10), 8);
    if (const char* e = std::getenv("PTX_WIDE_RAY_BVH")) c->ray_width = (uint32_t)std::min<unsigned long>(std::strtoul(e, nullptr, 10), 8);
    if (const char* e = std::getenv("PTX_GROUP_LANES")) {
        const unsigned long v = std::strtoul(e, nullptr, 10);
        if (v == 4 || v == 8) { c->group_lanes = (uint32_t)v; c->ray_width = (uint32_t)v; }
    }
    if (const char* e = std::getenv("PTX_ADAPTIVE_BATCH")) {
        const unsigned long v = std::strtoul(e, nullptr, 10);
        if (v) { c->adaptive_batch = (uint32_t)std::min<unsigned long>(v, 256); c->adaptive_batch_auto = false; }
    }
    if (device_index < NOALLOC_MAX_DEVICES) {
        // once per device, with that device current (set above)
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device_index]) {
            void* p = nullptr;
            HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            g_scratch_ptr[device_index] = static_cast<unsigned char*>(p);
        }
    }
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (const char* e = std::getenv("PTX_OVERLAP_SHADOW")) c->overlap_shadow = std::strtoul(e, nullptr, 10) != 0;
    if (const char* e = std::getenv("PTX_SKY_AHEAD")) c->sky_ahead_env = std::strtoul(e, nullptr, 10) != 0;
    if (c->overlap_shadow) {
        HIPCHK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
        HIPCHK(hipMalloc((void**)&c->ovf_side, sizeof(uint2) * (size_t)std::max(MAX_TRACE_GRID, PACKET_GRID) * TRACE_BLOCK * OVF_STACK));
    }
    HIPCHK(hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_poll, hipEventDisableTiming));
    if (const char* e = std::getenv("PTX_TAIL_GATE_ITER")) c->tail_gate_iter = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = std::getenv("PTX_TAIL_KERNEL")) c->tail_kernel = std::strtoul(e, nullptr, 10) != 0;
    if (const char* e = std::getenv("PTX_TAIL_CAP")) c->tail_cap = std::strtoull(e, nullptr, 10);
    if (const char* e = std::getenv("PTX_TAIL_FIRST_POLL")) c->tail_first_poll = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = std::getenv("PTX_TAIL_MIN_PER_WAVE")) c->tail_min_per_wave = (uint32_t)std::max(1ul, std::min<unsigned long>(std::strtoul(e, nullptr, 10), TAIL_PER_WAVE - 2));
    if (c->tail_kernel) HIPCHK(hipMalloc((void**)&c->tail_q, sizeof(uint32_t) * (size_t)TAIL_GRID * TAIL_REGION));
    HIPCHK(hipEventCreateWithFlags(&c->ev_ext_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_ext_out, hipEventDisableTiming));
    HIPCHK(hipMalloc((void**)&c->ctrs, sizeof(IterCtr) * CTR_BLOCKS));
    HIPCHK(hipMalloc((void**)&c->ovf, sizeof(uint2) * (size_t)std::max(MAX_TRACE_GRID, PACKET_GRID) * TRACE_BLOCK * OVF_STACK));
    HIPCHK(hipMalloc((void**)&c->d_err, 4));
    HIPCHK(hipMemset(c->d_err, 0, 4));
    HIPCHK(hipMalloc((void**)&c->d_counters, sizeof(DevCounters)));
    HIPCHK(hipMemset(c->d_counters, 0, sizeof(DevCounters)));
    HIPCHK(hipHostMalloc((void**)&c->h_pinned, 64 * 4, hipHostMallocDefault));
    *out_ctx = c;
    return PTX_OK;
}

int32_t ptx_destroy(PtxContext* ctx) {
    if (!ctx) return PTX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    drain_events(ctx);
    free_pool(ctx->scene_allocs);
    free_pool(ctx->path_allocs);
    for (void* p : {(void*)ctx->ctrs, (void*)ctx->ovf, (void*)ctx->d_err, (void*)ctx->d_counters, (void*)ctx->wl_pixel, (void*)ctx->img,
                    (void*)ctx->list_a, (void*)ctx->list_b, (void*)ctx->list_c, (void*)ctx->list_d, (void*)ctx->cub_tmp,
                    (void*)ctx->fb_sum, (void*)ctx->fb_sum_sq, (void*)ctx->fb_counts})
        if (p) (void)hipFree(p);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->ev_main) (void)hipEventDestroy(ctx->ev_main);
    if (ctx->ev_side) (void)hipEventDestroy(ctx->ev_side);
    if (ctx->ev_tail) (void)hipEventDestroy(ctx->ev_tail);
    if (ctx->ev_poll) (void)hipEventDestroy(ctx->ev_poll);
    if (ctx->ev_ext_in) (void)hipEventDestroy(ctx->ev_ext_in);
    if (ctx->ev_ext_out) (void)hipEventDestroy(ctx->ev_ext_out);
    if (ctx->ovf_side) (void)hipFree(ctx->ovf_side);
    if (ctx->tail_q) (void)hipFree(ctx->tail_q);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PTX_OK;
}

int32_t ptx_upload_scene(PtxContext* ctx, const PtxSceneDesc* sc) {
    REQUIRE_CTX();
    if (!sc || sc->abi_version != PTX_ABI_VERSION) return fail(ctx, PTX_ERR_INVALID, "scene descriptor missing or ABI version mismatch");
    if (sc->n_instances && (!sc->instances || !sc->tlas.nodes)) return fail(ctx, PTX_ERR_INVALID, "instances without a TLAS");
    if (sc->n_instances > MARK_INST_MASK) return fail(ctx, PTX_ERR_UNSUPPORTED, "more than 2^24 - 1 instances");
    if (!sc->n_materials || !sc->materials) return fail(ctx, PTX_ERR_INVALID, "scene needs at least one material");
    if (!sc->blue_noise_pixel.samples || !sc->blue_noise_lens.samples || sc->blue_noise_pixel.size != sc->blue_noise_lens.size || !sc->blue_noise_pixel.size)
        return fail(ctx, PTX_ERR_INVALID, "blue-noise tables missing");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_pool(ctx->scene_allocs);
    ctx->has_scene = false;

    std::vector<NodeRec> nodes;
    std::vector<TriRec> tris;
    std::vector<MeshRec> meshes(sc->n_meshes);
    std::vector<float> positions, normals, uvs;
    std::vector<uint32_t> indices;
    uint32_t tlas_root_word = 0, tlas_root_word_ray = 0, tlas_root_word_packet = 0;
    std::vector<uint32_t> mesh_root_word(sc->n_meshes, 0u), mesh_root_word_ray(sc->n_meshes, 0u), mesh_root_word_packet(sc->n_meshes, 0u);
    if (sc->n_instances) {
        if (!repack_bvh(sc->tlas, nodes, &tlas_root_word, ctx->ray_width, &tlas_root_word_ray, ctx->packet_width, &tlas_root_word_packet))
            return fail(ctx, PTX_ERR_INVALID, "TLAS is empty, too large, not a tree or has a child word out of range");
        for (uint32_t i = 0; i < sc->tlas.n_primitive_indices; ++i)
            if (sc->tlas.primitive_indices[i] >= sc->n_instances) return fail(ctx, PTX_ERR_INVALID, "TLAS primitive index out of range");
    }
    const uint32_t n_tlas_records = (uint32_t)nodes.size();
    bool forward_ok = true;      // forwarded_bound needs triangles inside the boxes above them, in every mesh
    std::vector<uint8_t> mesh_uncert(sc->n_meshes, 0);      // meshes that fail the shape certificate
    const char* no_cert = std::getenv("PTX_NO_SHAPE_CERT");     // measurement only: what the shortcuts do to ill-shaped meshes without it
    const bool shape_cert = !(no_cert && no_cert[0] == '1');
    for (uint32_t mi = 0; mi < sc->n_meshes; ++mi) {
        const PtxMesh& m = sc->meshes[mi];
        if (!m.positions || !m.indices || !m.n_triangles) return fail(ctx, PTX_ERR_INVALID, "mesh without geometry");
        if (m.material_id >= sc->n_materials) return fail(ctx, PTX_ERR_INVALID, "mesh material id out of range");
        MeshRec& r = meshes[mi];
        r.pos_base = (uint32_t)positions.size();
        positions.insert(positions.end(), m.positions, m.positions + (size_t)m.n_positions * 3);
        if (m.n_normals) { r.nrm_base = (uint32_t)normals.size(); normals.insert(normals.end(), m.normals, m.normals + (size_t)m.n_normals * 3); } else r.nrm_base = 0xFFFFFFFFu;
        if (m.n_uvs) { r.uv_base = (uint32_t)uvs.size(); uvs.insert(uvs.end(), m.uvs, m.uvs + (size_t)m.n_uvs * 2); } else r.uv_base = 0xFFFFFFFFu;
        r.idx_base = (uint32_t)indices.size();
        for (size_t k = 0; k < (size_t)m.n_triangles * 3; ++k) {
            if (m.indices[k] >= m.n_positions || (m.n_normals && m.indices[k] >= m.n_normals) || (m.n_uvs && m.indices[k] >= m.n_uvs))
                return fail(ctx, PTX_ERR_INVALID, "vertex index out of range");
            indices.push_back(m.indices[k]);
        }
        r.material_id = m.material_id;
        r.n_triangles = m.n_triangles;
        r.node_base = (uint32_t)nodes.size();
        r.tri_base = (uint32_t)tris.size();
        if (r.node_base > NODE_INDEX_MASK) return fail(ctx, PTX_ERR_UNSUPPORTED, "more than 2^29 node records");
        bool nests = false;
        if (!repack_bvh(m.bvh, nodes, &mesh_root_word[mi], ctx->ray_width, &mesh_root_word_ray[mi], ctx->packet_width, &mesh_root_word_packet[mi], &nests))
            return fail(ctx, PTX_ERR_INVALID, "mesh BVH is empty, not a tree or has a child word out of range");
        for (uint32_t s = 0; s < m.bvh.n_primitive_indices; ++s)
            if (m.bvh.primitive_indices[s] >= m.n_triangles) return fail(ctx, PTX_ERR_INVALID, "mesh BVH primitive index out of range");
        if (!nests || !mesh_boxes_hold_geometry(m)) forward_ok = false;
        mesh_uncert[mi] = (shape_cert && !mesh_shapes_certified(m)) ? 1 : 0;
        for (uint32_t s = 0; s < m.bvh.n_primitive_indices; ++s) {
            const uint32_t t = m.bvh.primitive_indices[s];
            const uint32_t* tri = &m.indices[3 * (size_t)t];
            const float* p0 = &m.positions[3 * (size_t)tri[0]]; const float* p1 = &m.positions[3 * (size_t)tri[1]]; const float* p2 = &m.positions[3 * (size_t)tri[2]];
            TriRec tr;
            float tf; std::memcpy(&tf, &t, 4);
            tr.a = make_float4(p0[0], p0[1], p0[2], p1[0] - p0[0]);
            tr.b = make_float4(p1[1] - p0[1], p1[2] - p0[2], p2[0] - p0[0], p2[1] - p0[1]);
            tr.c = make_float4(p2[2] - p0[2], tf, 0.0f, 0.0f);
            tris.push_back(tr);
        }
    }
    std::vector<uint32_t> shade_base(sc->n_meshes, 0u);
    uint64_t n_shade_tris = 0;
    for (uint32_t mi = 0; mi < sc->n_meshes; ++mi) { shade_base[mi] = (uint32_t)n_shade_tris; n_shade_tris += sc->meshes[mi].n_triangles; }
    if (n_shade_tris > 0xFFFFFFFFull || sc->n_materials > SHADE_MATERIAL_MASK) return fail(ctx, PTX_ERR_UNSUPPORTED, "more than 2^32 triangles or 2^30 materials");
    std::vector<InstRec> inst(sc->n_instances);
    std::vector<InstShade> inst_shade(sc->n_instances);
    for (uint32_t i = 0; i < sc->n_instances; ++i) {
        const PtxInstance& in = sc->instances[i];
        if (in.blas_index >= sc->n_meshes) return fail(ctx, PTX_ERR_INVALID, "instance blas index out of range");
        const PtxMesh& m = sc->meshes[in.blas_index];
        InstRec& r = inst[i];
        pack_cols(in.inverse_transform, &r.m[0]);
        r.m[3] = make_float4(m.bounds_min[0], m.bounds_min[1], m.bounds_min[2], m.bounds_max[0]);
        float nb, tb;
        const uint32_t nb_word = mesh_root_word_ray[in.blas_index] | (mesh_uncert[in.blas_index] ? INST_UNCERT : 0u);
        std::memcpy(&nb, &nb_word, 4);
        std::memcpy(&tb, &meshes[in.blas_index].tri_base, 4);
        r.m[4] = make_float4(m.bounds_max[1], m.bounds_max[2], nb, tb);
        pack_cols(in.transform, &r.m[5]);
        const float* nm = in.normal_matrix;
        InstShade& s = inst_shade[i];
        s.n[0] = nm[0]; s.n[1] = nm[1]; s.n[2] = nm[2]; s.n[3] = nm[4]; s.n[4] = nm[5]; s.n[5] = nm[6]; s.n[6] = nm[8]; s.n[7] = nm[9]; s.n[8] = nm[10];
        s.mesh_id = in.blas_index;
        s.shade_base = shade_base[in.blas_index];
        s.mat_flags = m.material_id | (m.n_normals ? SHADE_HAS_NORMALS : 0u) | (m.n_uvs ? SHADE_HAS_UVS : 0u);
    }
    std::vector<PtxMaterial> materials(sc->materials, sc->materials + sc->n_materials);
    bool has_sss = false;
    for (auto& m : materials) {
        for (int32_t t : {m.base_color_texture, m.metallic_roughness_texture, m.normal_texture, m.emissive_texture})
            if (t >= (int32_t)sc->n_textures) return fail(ctx, PTX_ERR_INVALID, "material texture id out of range");
        if (m.subsurface > 0.0f) has_sss = true;
    }
    std::vector<TexRec> textures(sc->n_textures);
    std::vector<float> texels;
    auto add_tex = [&](const PtxTexture& t) {
        TexRec r;
        r.width = t.width; r.height = t.height; r.pad = 0;
        r.solid[0] = t.solid[0]; r.solid[1] = t.solid[1]; r.solid[2] = t.solid[2];
        if (t.texels && t.width && t.height) { r.offset = texels.size(); texels.insert(texels.end(), t.texels, t.texels + (size_t)t.width * t.height * 3); }
        else r.offset = ~0ull;
        return r;
    };
    for (uint32_t i = 0; i < sc->n_textures; ++i) textures[i] = add_tex(sc->textures[i]);
    SceneDev S{};
    S.sky_tex = add_tex(sc->sky.texture);
    S.sky_w = sc->sky.width; S.sky_h = sc->sky.height;
    S.sky_total_weight = sc->sky.total_weight; S.sky_rotation = sc->sky.rotation;
    if (!sc->sky.marginal_cdf || !sc->sky.conditional_cdfs || !S.sky_w || !S.sky_h) return fail(ctx, PTX_ERR_INVALID, "sky CDFs missing");
    std::vector<float> marginal(sc->sky.marginal_cdf, sc->sky.marginal_cdf + S.sky_h + 1);
    std::vector<float> conditional(sc->sky.conditional_cdfs, sc->sky.conditional_cdfs + (size_t)S.sky_h * (S.sky_w + 1));
    std::vector<uint32_t> tlas_prims(sc->tlas.primitive_indices, sc->tlas.primitive_indices + (sc->n_instances ? sc->tlas.n_primitive_indices : 0));
    const size_t bn = (size_t)sc->blue_noise_pixel.size * sc->blue_noise_pixel.size * 2;
    std::vector<float> bn_pixel(sc->blue_noise_pixel.samples, sc->blue_noise_pixel.samples + bn);
    std::vector<float> bn_lens(sc->blue_noise_lens.samples, sc->blue_noise_lens.samples + bn);

    ctx->n_uncertified = 0;
    for (uint8_t f : mesh_uncert) ctx->n_uncertified += f;
    if (ctx->n_uncertified && sc->n_instances) {
        std::vector<uint8_t> inst_uncert(sc->n_instances);
        for (uint32_t i = 0; i < sc->n_instances; ++i) inst_uncert[i] = mesh_uncert[sc->instances[i].blas_index];
        for (uint32_t root : {tlas_root_word, tlas_root_word_ray, tlas_root_word_packet}) flag_uncertified(nodes, root, tlas_prims, inst_uncert);
    }
    auto& pool = ctx->scene_allocs;
    HIPCHK(dev_upload(pool, &S.nodes, nodes)); HIPCHK(dev_upload(pool, &S.tlas_prims, tlas_prims)); HIPCHK(dev_upload(pool, &S.tris, tris));
    std::vector<InstRec> inst_tlas(tlas_prims.size());
    std::vector<uint32_t> inst_root_faithful(tlas_prims.size()), inst_root_packet(tlas_prims.size());
    for (size_t j = 0; j < tlas_prims.size(); ++j) {
        inst_tlas[j] = inst[tlas_prims[j]];
        inst_root_faithful[j] = mesh_root_word[sc->instances[tlas_prims[j]].blas_index];
        inst_root_packet[j] = mesh_root_word_packet[sc->instances[tlas_prims[j]].blas_index];
    }
    HIPCHK(dev_upload(pool, &S.inst_tlas, inst_tlas));
    HIPCHK(dev_upload(pool, &S.inst_root_faithful, inst_root_faithful));
    HIPCHK(dev_upload(pool, &S.inst_root_packet, inst_root_packet));
    HIPCHK(dev_upload(pool, &S.inst, inst)); HIPCHK(dev_upload(pool, &S.inst_shade, inst_shade)); HIPCHK(dev_upload(pool, &S.meshes, meshes));
    HIPCHK(dev_upload(pool, &S.positions, positions)); HIPCHK(dev_upload(pool, &S.normals, normals)); HIPCHK(dev_upload(pool, &S.uvs, uvs));
    HIPCHK(dev_upload(pool, &S.indices, indices)); HIPCHK(dev_upload(pool, &S.materials, materials)); HIPCHK(dev_upload(pool, &S.textures, textures));
    HIPCHK(dev_upload(pool, &S.texels, texels)); HIPCHK(dev_upload(pool, &S.sky_marginal, marginal)); HIPCHK(dev_upload(pool, &S.sky_conditional, conditional));
    {
        // 16-ary search tables (pt_types.h): only for non-decreasing CDFs, where counting equals the reference's bisection
        bool monotone = true;
        for (uint32_t i = 0; i + 1 < S.sky_h + 1 && monotone; ++i) monotone = marginal[i] <= marginal[i + 1];
        for (size_t r = 0; r < S.sky_h && monotone; ++r)
            for (uint32_t i = 0; i < S.sky_w && monotone; ++i) monotone = conditional[r * (S.sky_w + 1) + i] <= conditional[r * (S.sky_w + 1) + i + 1];
        std::vector<float> search;
        bool sky_tables3 = false;
        const float inf = __builtin_inff();
        auto level_sizes = [](uint32_t n_real, uint32_t* sizes) {   // padded sizes per level; returns the level count (0 = too large)
            uint32_t levels = 0, n = n_real;
            for (;;) {
                if (levels == 4) return 0u;
                const uint32_t padded = (n + 1u + 15u) / 16u * 16u;
                sizes[levels++] = padded;
                if (padded == 16u && levels >= 3u) return levels;      // at least three levels (a level above the last real one holds +inf only: its count is 0), so that k_shade's straight-line search serves every sky up to 4095 entries a side
                n = padded / 16u;
            }
        };
        uint32_t sm[4] = {0, 0, 0, 0}, scs[4] = {0, 0, 0, 0};
        const uint32_t lm = level_sizes(S.sky_h, sm), lc = level_sizes(S.sky_w, scs);
        if (monotone && lm && lc && S.sky_total_weight > 0.0f) {
            auto build = [&](const float* cdf, uint32_t n_real, const uint32_t* sizes, uint32_t levels, const size_t* offs, size_t row, const uint32_t* strides) {
                // level 0
                float* l0 = &search[offs[0] + row * strides[0]];
                for (uint32_t i = 0; i < sizes[0]; ++i) l0[i] = i < n_real ? cdf[i] : inf;
                for (uint32_t l = 1; l < levels; ++l) {
                    const float* below = &search[offs[l - 1] + row * strides[l - 1]];
                    float* cur = &search[offs[l] + row * strides[l]];
                    const uint32_t groups = sizes[l - 1] / 16u;
                    for (uint32_t i = 0; i < sizes[l]; ++i) cur[i] = i < groups ? below[16u * i + 15u] : inf;
                }
            };
            size_t total = 0, off_m[4] = {0, 0, 0, 0}, off_c[4] = {0, 0, 0, 0};
            for (uint32_t l = 0; l < lm; ++l) { off_m[l] = total; total += sm[l]; }
            for (uint32_t l = 0; l < lc; ++l) { off_c[l] = total; total += (size_t)scs[l] * S.sky_h; }
            if (total < 0xFFFFFFFFull) {
                search.assign(total, inf);
                const uint32_t zero[4] = {0, 0, 0, 0};
                build(marginal.data(), S.sky_h, sm, lm, off_m, 0, zero);
                for (size_t r = 0; r < S.sky_h; ++r) build(&conditional[r * (S.sky_w + 1)], S.sky_w, scs, lc, off_c, r, scs);
                S.cdf_levels_m = lm; S.cdf_levels_c = lc;
                sky_tables3 = lm == 3u && lc == 3u;
                for (uint32_t l = 0; l < 4; ++l) { S.cdf_off_m[l] = (uint32_t)off_m[l]; S.cdf_off_c[l] = (uint32_t)off_c[l]; S.cdf_stride_c[l] = scs[l]; }
            }
        }
        HIPCHK(dev_upload(pool, &S.sky_search, search));
        ctx->sky_ahead = ctx->sky_ahead_env && sky_tables3;
    }
    HIPCHK(dev_upload(pool, &S.bn_pixel, bn_pixel)); HIPCHK(dev_upload(pool, &S.bn_lens, bn_lens));
    S.bn_size = sc->blue_noise_pixel.size;
    S.sun_enabled = sc->sun.enabled;
    for (int k = 0; k < 3; ++k) { S.sun_dir[k] = sc->sun.direction[k]; S.sun_color[k] = sc->sun.color[k]; }
    S.sun_cos_angle = sc->sun.cos_angle; S.sun_pdf = sc->sun.pdf;
    S.n_tlas_nodes = sc->n_instances ? n_tlas_records : 0;
    S.tlas_root_word = tlas_root_word;
    S.tlas_root_word_ray = tlas_root_word_ray;
    S.tlas_root_word_packet = tlas_root_word_packet;
    S.has_sss = has_sss ? 1u : 0u;
    {
        // forwarded_bound's constants (pt_traverse.h).  For a ray o + t d and an instance with matrices M, V (= the f32
        // inverse the host supplies), the reference computes o_l = fl(V o), d_l = fl(V d) and, for a local hit at t,
        // t_w = fl(((M (o_l + t d_l)) - o) . d / d.d).  In exact arithmetic on those f32 values
        //   t_w - t = ((R o_1 + M do) . d + t (R d + M dd) . d) / d.d,   R = M V - I,  |do| <= g |V| |o_1|,  |dd| <= g |V| |d|,
        // (o_1 = (o, 1)) and evaluating t_w in f32 adds a few more ulps of the same magnitudes.  With the abs-matrix
        // norms below, |t_w - t| <= (lin |o|_1 + tr) / |d| + t lin, lin = 64 eps |M||V| + 4 |R|: 64 ulps cover the
        // ~20 roundings on any path with room to spare.  The relative part must stay far below the 2^-5 the bound adds.
        const double eps = 5.9604644775390625e-08;
        double lin = 0.0, tr = 0.0;
        bool ok = forward_ok;
        for (uint32_t i = 0; i < sc->n_instances; ++i) {
            const float* M = sc->instances[i].transform;            // column-major
            const float* V = sc->instances[i].inverse_transform;
            for (int r = 0; r < 3; ++r) {
                double a_lin = 0.0, a_t = std::fabs((double)M[12 + r]), r_lin = 0.0, r_t = (double)M[12 + r];
                for (int c = 0; c < 3; ++c) {
                    double prod = 0.0;
                    for (int k = 0; k < 3; ++k) {
                        a_lin += std::fabs((double)M[4 * k + r]) * std::fabs((double)V[4 * c + k]);
                        prod += (double)M[4 * k + r] * (double)V[4 * c + k];
                    }
                    r_lin += std::fabs(prod - (r == c ? 1.0 : 0.0));
                }
                for (int k = 0; k < 3; ++k) {
                    a_t += std::fabs((double)M[4 * k + r]) * std::fabs((double)V[12 + k]);
                    r_t += (double)M[4 * k + r] * (double)V[12 + k];
                }
                const double l = 64.0 * eps * a_lin + 4.0 * r_lin, t = 64.0 * eps * a_t + 4.0 * std::fabs(r_t);
                if (!(l == l) || !(t == t)) ok = false;
                lin = l > lin ? l : lin;
                tr = t > tr ? t : tr;
            }
            // the fourth row must be (0,0,0,1): transform_point3 ignores it
        }
        const char* off = std::getenv("PTX_NO_FORWARD");
        S.fwd_on = (ok && lin <= 1.0 / 256.0 && tr < 1e30 && !(off && off[0] == '1')) ? 1u : 0u;
        // measurement only: PTX_COUNT_FORWARD=1 lets the counted kernels forward too (their counts are then this library's, not the reference's)
        if (S.fwd_on) { const char* cf = std::getenv("PTX_COUNT_FORWARD"); if (cf && cf[0] == '1') S.fwd_on = 2u; }
        S.fwd_lin = (float)(lin * 1.0000002); S.fwd_t = (float)(tr * 1.0000002);
        // FAST traversal: on with forwarding (same presumptions), for rays that start within the scene's extent
        {
            double omax = 0.0;      // max |p|_1 over the world boxes of the instances, doubled: rays that start on or near a surface
            for (uint32_t i = 0; i < sc->n_instances; ++i) {
                const float* M = sc->instances[i].transform;
                const PtxMesh& m = sc->meshes[sc->instances[i].blas_index];
                for (int corner = 0; corner < 8; ++corner) {
                    const double x = (corner & 1) ? m.bounds_max[0] : m.bounds_min[0], y = (corner & 2) ? m.bounds_max[1] : m.bounds_min[1],
                                 z = (corner & 4) ? m.bounds_max[2] : m.bounds_min[2];
                    double l1 = 0.0;
                    for (int r = 0; r < 3; ++r) l1 += std::fabs((double)M[r] * x + (double)M[4 + r] * y + (double)M[8 + r] * z + (double)M[12 + r]);
                    if (l1 == l1 && l1 > omax) omax = l1;
                }
            }
            omax = 2.0 * omax + 1.0;
            const double slack = 1.5 * (lin * omax + tr) * 1.0000002;
            const char* exact_order = std::getenv("PTX_EXACT_ORDER");
            S.fast_on = (S.fwd_on != 0u && omax < 1e30 && slack < 1e30 && !(exact_order && exact_order[0] == '1')) ? 1u : 0u;
            S.fast_omax = (float)omax; S.fast_slack = (float)slack;
            if (std::getenv("PTX_VERBOSE")) fprintf(stderr, "[ptx] forwarding %u (lin %.3g, tr %.3g); nearest-first %u (|o|_1 <= %.4g, slack %.3g); %u of %u meshes fail the shape certificate\n", S.fwd_on, lin, tr, S.fast_on, omax, slack, ctx->n_uncertified, sc->n_meshes);
        }
    }
    // shading triangles: built on the device from the pools just uploaded, one launch per mesh
    {
        ShadeTri* d_shade = nullptr;
        HIPCHK(dev_alloc(pool, &d_shade, (size_t)n_shade_tris));
        S.shade_tris = d_shade;
        for (uint32_t mi = 0; mi < sc->n_meshes; ++mi) {
            const MeshRec& m = meshes[mi];
            hipLaunchKernelGGL(k_build_shade_tris, dim3((m.n_triangles + 255) / 256), dim3(256), 0, ctx->stream, S, m, d_shade + shade_base[mi]);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    ctx->S = S;
    ctx->has_scene = true;
    return PTX_OK;
}

int32_t ptx_set_camera(PtxContext* ctx, const PtxCamera* cam) {
    REQUIRE_CTX();
    if (!cam) return fail(ctx, PTX_ERR_INVALID, "camera is null");
    std::memcpy(ctx->cam.inv_vp, cam->inv_vp_matrix, sizeof(float) * 16);
    ctx->cam.aperture = cam->aperture; ctx->cam.focus_distance = cam->focus_distance;
    for (int k = 0; k < 3; ++k) { ctx->cam.right[k] = cam->right[k]; ctx->cam.up[k] = cam->up[k]; }
    ctx->has_camera = true;
    return PTX_OK;
}

int32_t ptx_set_partition(PtxContext* ctx, uint32_t rank, uint32_t world) {
    REQUIRE_CTX();
    if (world == 0 || rank >= world) return fail(ctx, PTX_ERR_INVALID, "partition needs rank < world");
    ctx->rank = rank; ctx->world = world;
    return PTX_OK;
}

int32_t ptx_sync(PtxContext* ctx) {
    REQUIRE_CTX();
    return check_device_error(ctx);   // synchronises the library stream and reports a traversal stack overflow since the last check
}

static int32_t render_to_images(PtxContext* ctx, uint32_t w, uint32_t h, uint32_t sample_index, uint32_t max_bounces, float* out_rgb,
                                float* out_albedo, float* out_normal, int mode /*0 pt, 1 aov, 2 unshaded*/) {
    if (!w || !h || !out_rgb || (uint64_t)w * h > 0x7FFFFFFFull) return fail(ctx, PTX_ERR_INVALID, "bad image size or null output");
    int32_t st = ensure_worklist(ctx, w, h);
    if (st != PTX_OK) return st;
    const uint64_t npx = (uint64_t)w * h;
    st = ensure_images(ctx, npx);
    if (st != PTX_OK) return st;
    float* d_rgb = ctx->img; float* d_alb = ctx->img + npx * 3; float* d_nrm = ctx->img + npx * 6;
    HIPCHK(hipMemsetAsync(ctx->img, 0, npx * 9 * sizeof(float), ctx->stream));   // pixels another rank owns stay 0
    const uint32_t n = ctx->wl_n;
    if (mode == 2) {
        st = ensure_capacity(ctx, n);
        if (st != PTX_OK) return st;
        if (n) {
            HIPCHK(hipMemsetAsync(ctx->ctrs, 0, sizeof(IterCtr) * 2, ctx->stream));
            hipLaunchKernelGGL(k_raygen, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->S, ctx->cam, ctx->P, ctx->wl_pixel, (const uint32_t*)nullptr,
                               (const uint32_t*)nullptr, sample_index, n, w, h, ctx->q[0], ctx->ctrs, 0u, (PtxRay*)nullptr, 0u);
            const int tg = (int)std::min<uint64_t>(TRACE_GRID, (n + TRACE_BLOCK - 1) / TRACE_BLOCK);
            // the preview is primary rays only, one 8x8 tile per wave: packets
            if (ctx->packet_primary) {
                const int32_t ps = launch_primary_packets(ctx, ctx->q[0], ctx->ctrs, n);
                if (ps != PTX_OK) return ps;
            } else
                hipLaunchKernelGGL((k_extend<false, false>), dim3(tg), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, ctx->P, ctx->q[0], (const uint32_t*)nullptr, ctx->ctrs, ctx->ovf, ctx->d_counters, ctx->d_err);
            hipLaunchKernelGGL(k_unshaded, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->S, ctx->P, n, d_rgb);
        }
    } else {
        Target tg{};
        tg.out_rgb = d_rgb;
        if (mode == 1) { tg.out_albedo = d_alb; tg.out_normal = d_nrm; }
        st = run_paths(ctx, ctx->wl_pixel, nullptr, nullptr, sample_index, n, w, h, max_bounces, tg, mode == 1);
        if (st != PTX_OK) return st;
        add_primary_count(ctx, n);
    }
    HIPCHK(hipMemcpyAsync(out_rgb, d_rgb, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (mode == 1) {
        HIPCHK(hipMemcpyAsync(out_albedo, d_alb, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(out_normal, d_nrm, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    return check_device_error(ctx);
}

int32_t ptx_render_pass(PtxContext* ctx, uint32_t w, uint32_t h, uint32_t sample_index, uint32_t max_bounces, float* out_rgb) {
    REQUIRE_CAMERA();
    return render_to_images(ctx, w, h, sample_index, max_bounces, out_rgb, nullptr, nullptr, 0);
}

int32_t ptx_render_first_pass_aov(PtxContext* ctx, uint32_t w, uint32_t h, uint32_t max_bounces, float* out_rgb, float* out_albedo, float* out_normal) {
    REQUIRE_CAMERA();
    if (!out_albedo || !out_normal) return fail(ctx, PTX_ERR_INVALID, "AOV outputs are null");
    return render_to_images(ctx, w, h, 0, max_bounces, out_rgb, out_albedo, out_normal, 1);
}

int32_t ptx_render_pass_unshaded(PtxContext* ctx, uint32_t w, uint32_t h, uint32_t sample_index, float* out_rgb) {
    REQUIRE_CAMERA();
    return render_to_images(ctx, w, h, sample_index, 0, out_rgb, nullptr, nullptr, 2);
}

int32_t ptx_render_pixels(PtxContext* ctx, uint32_t w, uint32_t h, const uint32_t* xys, uint64_t n, uint32_t max_bounces, float* out_rgb) {
    REQUIRE_CAMERA();
    if (n == 0) return PTX_OK;
    if (!xys || !out_rgb || !w || !h || n > 0x7FFFFFFFull) return fail(ctx, PTX_ERR_INVALID, "bad work list");
    std::vector<uint32_t> pix(n), smp(n);
    for (uint64_t i = 0; i < n; ++i) {
        if (xys[3 * i] >= w || xys[3 * i + 1] >= h) return fail(ctx, PTX_ERR_INVALID, "work-list pixel outside the image");
        pix[i] = xys[3 * i + 1] * w + xys[3 * i];
        smp[i] = xys[3 * i + 2];
    }
    int32_t st = ensure_lists(ctx, n);
    if (st != PTX_OK) return st;
    HIPCHK(hipMemcpyAsync(ctx->list_b, pix.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->list_c, smp.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // pix/smp are stack-lifetime host vectors
    Target tg{};
    tg.out_rgb = reinterpret_cast<float*>(ctx->list_a);   // n*3 floats, by slot
    tg.by_slot = 1;
    st = run_paths(ctx, ctx->list_b, ctx->list_c, nullptr, 0, (uint32_t)n, w, h, max_bounces, tg, false);
    if (st != PTX_OK) return st;
    add_primary_count(ctx, n);
    HIPCHK(hipMemcpyAsync(out_rgb, ctx->list_a, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    return check_device_error(ctx);
}

// ---- framebuffer -------------------------------------------------------------------------
int32_t ptx_fb_reset(PtxContext* ctx, uint32_t w, uint32_t h, int32_t adaptive) {
    REQUIRE_CTX();
    if (!w || !h || (uint64_t)w * h > 0x7FFFFFFFull) return fail(ctx, PTX_ERR_INVALID, "bad framebuffer size");
    const uint64_t npx = (uint64_t)w * h;
    if (ctx->fb_w != w || ctx->fb_h != h || ctx->fb_adaptive != (adaptive != 0) || !ctx->fb_sum) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (void* p : {(void*)ctx->fb_sum, (void*)ctx->fb_sum_sq, (void*)ctx->fb_counts}) if (p) (void)hipFree(p);
        ctx->fb_sum = ctx->fb_sum_sq = nullptr; ctx->fb_counts = nullptr;
        HIPCHK(hipMalloc((void**)&ctx->fb_sum, npx * 3 * sizeof(float)));
        if (adaptive) HIPCHK(hipMalloc((void**)&ctx->fb_sum_sq, npx * 3 * sizeof(float)));
        HIPCHK(hipMalloc((void**)&ctx->fb_counts, npx * 4));
        ctx->fb_w = w; ctx->fb_h = h; ctx->fb_adaptive = adaptive != 0;
    }
    HIPCHK(hipMemsetAsync(ctx->fb_sum, 0, npx * 3 * sizeof(float), ctx->stream));
    if (ctx->fb_sum_sq) HIPCHK(hipMemsetAsync(ctx->fb_sum_sq, 0, npx * 3 * sizeof(float), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->fb_counts, 0, npx * 4, ctx->stream));
    return PTX_OK;
}

#define REQUIRE_FB() do { REQUIRE_CAMERA(); if (!ctx->fb_sum) return fail(ctx, PTX_ERR_INVALID, "no framebuffer: call ptx_fb_reset"); } while (0)

static int32_t fb_render_passes(PtxContext* ctx, uint32_t first_sample, uint32_t count, uint32_t max_bounces, bool check) {
    REQUIRE_FB();
    int32_t st = ensure_worklist(ctx, ctx->fb_w, ctx->fb_h);
    if (st != PTX_OK) return st;
    Target tg{};
    tg.sum = ctx->fb_sum; tg.sum_sq = ctx->fb_sum_sq; tg.counts = ctx->fb_counts;
    // The passes of this call are independent (each path seeds its own RNG, parallel.rs:31-56), so several are
    // traced together: the wavefront queues stay long enough to fill the chip, and k_accumulate_batch adds every
    // pixel's samples in pass order, which keeps the sums bit-identical to one pass at a time.
    const uint32_t n1 = ctx->wl_n;
    uint32_t in_flight = n1 ? (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, ctx->max_paths_in_flight / n1), 64) : 1u;
    for (uint32_t k = 0; k < count && n1;) {
        const uint32_t b = std::min(in_flight, count - k);
        if (b == 1) {
            st = run_paths(ctx, ctx->wl_pixel, nullptr, nullptr, first_sample + k, n1, ctx->fb_w, ctx->fb_h, max_bounces, tg, false);
            if (st != PTX_OK) return st;
        } else {
            Target deferred = tg;
            deferred.defer = 1u;
            st = run_paths(ctx, ctx->wl_pixel, nullptr, nullptr, first_sample + k, n1 * b, ctx->fb_w, ctx->fb_h, max_bounces, deferred, false, b);
            if (st != PTX_OK) return st;
            ScopedEvent ev(ctx, CLS_ACCUM);
            hipLaunchKernelGGL(k_accumulate_batch, dim3((n1 + 255) / 256), dim3(256), 0, ctx->stream, ctx->P, ctx->wl_pixel, n1, b, tg);
        }
        k += b;
    }
    add_primary_count(ctx, (uint64_t)ctx->wl_n * count);
    // a traversal that ran out of stack ends its ray early (less work, a faster frame): never hand that back as OK
    // (the async form leaves that to the next synchronising call: ptx_sync, ptx_fb_stats, ptx_fb_read*)
    return check ? check_device_error(ctx) : PTX_OK;
}
int32_t ptx_fb_render_passes(PtxContext* ctx, uint32_t first_sample, uint32_t count, uint32_t max_bounces) {
    return fb_render_passes(ctx, first_sample, count, max_bounces, true);
}
int32_t ptx_fb_render_passes_async(PtxContext* ctx, uint32_t first_sample, uint32_t count, uint32_t max_bounces) {
    if (!ctx) return fail(nullptr, PTX_ERR_INVALID, "null context");
    // no host wait: k_tail takes over only where the batch size alone says the live paths fit (scenes with subsurface
    // materials still wait, as they always did: their iteration count is data dependent)
    const bool poll = ctx->tail_poll;
    ctx->tail_poll = false;
    const int32_t st = fb_render_passes(ctx, first_sample, count, max_bounces, false);
    ctx->tail_poll = poll;
    return st;
}

int32_t ptx_fb_adaptive_pass(PtxContext* ctx, const PtxAdaptiveConfig* cfg, uint32_t max_bounces, uint64_t* out_rendered) {
    REQUIRE_FB();
    if (!cfg || !out_rendered) return fail(ctx, PTX_ERR_INVALID, "null argument");
    if (!ctx->fb_adaptive) return fail(ctx, PTX_ERR_INVALID, "framebuffer was not created adaptive");
    *out_rendered = 0;
    int32_t st = ensure_worklist(ctx, ctx->fb_w, ctx->fb_h);
    if (st != PTX_OK) return st;
    const uint32_t n = ctx->wl_n;
    if (n == 0) return PTX_OK;
    st = ensure_lists(ctx, n);
    if (st != PTX_OK) return st;
    uint32_t* keys = ctx->list_b; uint32_t* work = ctx->list_c;
    uint32_t* d_count = reinterpret_cast<uint32_t*>(&ctx->ctrs[CTR_ADAPTIVE]);
    HIPCHK(hipMemsetAsync(d_count, 0, sizeof(IterCtr), ctx->stream));
    const int grid = (int)std::min<uint32_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(k_noise_keys, dim3(grid), dim3(256), 0, ctx->stream, ctx->wl_pixel, n, ctx->fb_sum, ctx->fb_sum_sq, ctx->fb_counts,
                       cfg->min_samples, cfg->variance_threshold, keys, d_count);
    HIPCHK(hipMemcpyAsync(ctx->h_pinned + 2, d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint32_t n_noisy = ctx->h_pinned[2];
    if (n_noisy == 0) return PTX_OK;           // renderloop/parallel.rs:98-100
    uint32_t n_work = n_noisy;
    if ((uint64_t)n_noisy <= cfg->max_samples_per_pass) {
        // nothing is truncated, so the descending sort (parallel.rs:102-106) cannot change any pixel's result
        hipLaunchKernelGGL(k_compact_noisy, dim3(grid), dim3(256), 0, ctx->stream, ctx->wl_pixel, keys, n, work, d_count + 1);
    } else {
        // truncate(max): stable descending sort by error, ties in row-major pixel order as on the CPU
        if (ctx->world != 1) return fail(ctx, PTX_ERR_UNSUPPORTED, "max_samples_per_pass below the noisy count needs world == 1");
        n_work = (uint32_t)cfg->max_samples_per_pass;
        if (n_work == 0) return PTX_OK;
        // keys are indexed in tile order; rebuild them in row-major order: owned == all pixels when world == 1
        uint32_t* rm_keys = ctx->list_d; uint32_t* rm_idx = reinterpret_cast<uint32_t*>(ctx->list_a);
        uint32_t* s_keys = rm_idx + n; uint32_t* s_idx = rm_idx + 2 * (size_t)n;
        hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, rm_idx, n);
        // scatter: rm_keys[pixel] = keys[i] — done as a gather through a one-off inverse built by sorting (pixel, i)
        {
            size_t tmp = 0;
            (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, ctx->wl_pixel, s_keys, keys, rm_keys, (int)n, 0, 32, ctx->stream);
            if (tmp > ctx->cub_tmp_bytes) { if (ctx->cub_tmp) (void)hipFree(ctx->cub_tmp); ctx->cub_tmp = nullptr; HIPCHK(hipMalloc(&ctx->cub_tmp, tmp)); ctx->cub_tmp_bytes = tmp; }
            HIPCHK(hipcub::DeviceRadixSort::SortPairs(ctx->cub_tmp, tmp, ctx->wl_pixel, s_keys, keys, rm_keys, (int)n, 0, 32, ctx->stream));
        }
        {
            size_t tmp = 0;
            (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp, rm_keys, s_keys, rm_idx, s_idx, (int)n, 0, 32, ctx->stream);
            if (tmp > ctx->cub_tmp_bytes) { if (ctx->cub_tmp) (void)hipFree(ctx->cub_tmp); ctx->cub_tmp = nullptr; HIPCHK(hipMalloc(&ctx->cub_tmp, tmp)); ctx->cub_tmp_bytes = tmp; }
            HIPCHK(hipcub::DeviceRadixSort::SortPairsDescending(ctx->cub_tmp, tmp, rm_keys, s_keys, rm_idx, s_idx, (int)n, 0, 32, ctx->stream));
        }
        HIPCHK(hipMemcpyAsync(work, s_idx, (size_t)n_work * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    Target tg{};
    tg.sum = ctx->fb_sum; tg.sum_sq = ctx->fb_sum_sq; tg.counts = ctx->fb_counts;
    // sample_index = the pixel's own sample count (parallel.rs:109-114)
    st = run_paths(ctx, work, nullptr, ctx->fb_counts, 0, n_work, ctx->fb_w, ctx->fb_h, max_bounces, tg, false);
    if (st != PTX_OK) return st;
    add_primary_count(ctx, n_work);
    *out_rendered = n_work;
    return check_device_error(ctx);
}

int32_t ptx_fb_noisy_pairs(PtxContext* ctx, const PtxAdaptiveConfig* cfg, uint32_t* out_keys, uint32_t* out_pixels, uint64_t capacity,
                           uint64_t* out_n) {
    REQUIRE_FB();
    if (!cfg || !out_n) return fail(ctx, PTX_ERR_INVALID, "null argument");
    if (!ctx->fb_adaptive) return fail(ctx, PTX_ERR_INVALID, "framebuffer was not created adaptive");
    *out_n = 0;
    int32_t st = ensure_worklist(ctx, ctx->fb_w, ctx->fb_h);
    if (st != PTX_OK) return st;
    const uint32_t n = ctx->wl_n;
    if (n == 0) return PTX_OK;
    st = ensure_lists(ctx, n);
    if (st != PTX_OK) return st;
    uint32_t* keys = ctx->list_b; uint32_t* pix = ctx->list_c; uint32_t* ckeys = ctx->list_d;
    uint32_t* d_count = reinterpret_cast<uint32_t*>(&ctx->ctrs[CTR_ADAPTIVE]);
    HIPCHK(hipMemsetAsync(d_count, 0, sizeof(IterCtr), ctx->stream));
    const int grid = (int)std::min<uint32_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(k_noise_keys, dim3(grid), dim3(256), 0, ctx->stream, ctx->wl_pixel, n, ctx->fb_sum, ctx->fb_sum_sq, ctx->fb_counts,
                       cfg->min_samples, cfg->variance_threshold, keys, d_count);
    hipLaunchKernelGGL(k_compact_noisy_pairs, dim3(grid), dim3(256), 0, ctx->stream, ctx->wl_pixel, keys, n, pix, ckeys, d_count + 1);
    HIPCHK(hipMemcpyAsync(ctx->h_pinned + 2, d_count + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint32_t n_noisy = ctx->h_pinned[2];
    *out_n = n_noisy;
    if (n_noisy == 0) return PTX_OK;
    if (!out_keys || !out_pixels || capacity < n_noisy) return fail(ctx, PTX_ERR_INVALID, "ptx_fb_noisy_pairs: buffers too small (*out_n holds the count)");
    HIPCHK(hipMemcpyAsync(out_keys, ckeys, (size_t)n_noisy * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(out_pixels, pix, (size_t)n_noisy * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PTX_OK;
}

int32_t ptx_fb_adaptive_pass_pixels(PtxContext* ctx, const uint32_t* pixels, uint64_t n64, uint32_t max_bounces, uint64_t* out_rendered) {
    REQUIRE_FB();
    if (!out_rendered || (n64 && !pixels)) return fail(ctx, PTX_ERR_INVALID, "null argument");
    if (!ctx->fb_adaptive) return fail(ctx, PTX_ERR_INVALID, "framebuffer was not created adaptive");
    *out_rendered = 0;
    if (n64 == 0) return PTX_OK;
    if (n64 > (uint64_t)ctx->fb_w * ctx->fb_h) return fail(ctx, PTX_ERR_INVALID, "more pixels than the framebuffer holds (a pixel may appear once per pass)");
    const uint32_t n = (uint32_t)n64;
    int32_t st = ensure_lists(ctx, std::max<uint64_t>(n, (uint64_t)ctx->fb_w * ctx->fb_h));
    if (st != PTX_OK) return st;
    uint32_t* work = ctx->list_c;
    uint32_t* d_bad = reinterpret_cast<uint32_t*>(&ctx->ctrs[CTR_ADAPTIVE]);
    HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(IterCtr), ctx->stream));
    HIPCHK(hipMemcpyAsync(work, pixels, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_check_pixels, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, work, n, ctx->fb_w * ctx->fb_h, d_bad);
    HIPCHK(hipMemcpyAsync(ctx->h_pinned + 2, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->h_pinned[2] != 0u) return fail(ctx, PTX_ERR_INVALID, "pixel index outside the framebuffer");
    Target tg{};
    tg.sum = ctx->fb_sum; tg.sum_sq = ctx->fb_sum_sq; tg.counts = ctx->fb_counts;
    // sample_index = the pixel's own sample count (parallel.rs:109-114)
    st = run_paths(ctx, work, nullptr, ctx->fb_counts, 0, n, ctx->fb_w, ctx->fb_h, max_bounces, tg, false);
    if (st != PTX_OK) return st;
    add_primary_count(ctx, n);
    *out_rendered = n;
    return check_device_error(ctx);
}

int32_t ptx_fb_adaptive_passes(PtxContext* ctx, const PtxAdaptiveConfig* cfg, uint32_t max_bounces, uint32_t max_passes,
                               uint32_t* out_passes, uint64_t* out_rendered) {
    REQUIRE_FB();
    if (!cfg || !out_passes || !out_rendered) return fail(ctx, PTX_ERR_INVALID, "null argument");
    if (!ctx->fb_adaptive) return fail(ctx, PTX_ERR_INVALID, "framebuffer was not created adaptive");
    *out_passes = 0; *out_rendered = 0;
    int32_t st = ensure_worklist(ctx, ctx->fb_w, ctx->fb_h);
    if (st != PTX_OK) return st;
    const uint32_t n = ctx->wl_n;
    if (n == 0) return PTX_OK;
    st = ensure_lists(ctx, n);
    if (st != PTX_OK) return st;
    uint32_t* keys = ctx->list_b; uint32_t* work = ctx->list_c;
    uint32_t* d_count = reinterpret_cast<uint32_t*>(&ctx->ctrs[CTR_ADAPTIVE]);
    const int grid = (int)std::min<uint32_t>(2048, (n + 255) / 256);
    uint32_t passes = 0; uint64_t rendered = 0;
    uint32_t batch = ctx->adaptive_batch;
    while (passes < max_passes) {
        HIPCHK(hipMemsetAsync(d_count, 0, sizeof(IterCtr), ctx->stream));
        hipLaunchKernelGGL(k_noise_keys, dim3(grid), dim3(256), 0, ctx->stream, ctx->wl_pixel, n, ctx->fb_sum, ctx->fb_sum_sq, ctx->fb_counts,
                           cfg->min_samples, cfg->variance_threshold, keys, d_count);
        HIPCHK(hipMemcpyAsync(ctx->h_pinned + 2, d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const uint32_t n_noisy = ctx->h_pinned[2];
        if (n_noisy == 0) break;
        // passes in flight: every noisy pixel gets `b` speculative samples; k_accumulate_adaptive keeps the ones the
        // pass-at-a-time loop would have taken.  Device counters would include the dropped samples, and a truncated
        // work list couples the pixels, so both cases go one pass at a time.
        const uint32_t b = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(batch, max_passes - passes),
                                                        std::max<uint64_t>(1, ctx->max_paths_in_flight / n_noisy));
        if (ctx->counting || b <= 1 || (uint64_t)n_noisy > cfg->max_samples_per_pass) {
            uint64_t r = 0;
            st = ptx_fb_adaptive_pass(ctx, cfg, max_bounces, &r);
            if (st != PTX_OK) return st;
            if (r == 0) break;
            passes += 1; rendered += r;
            continue;
        }
        hipLaunchKernelGGL(k_compact_noisy, dim3(grid), dim3(256), 0, ctx->stream, ctx->wl_pixel, keys, n, work, d_count + 1);
        Target tg{};
        tg.sum = ctx->fb_sum; tg.sum_sq = ctx->fb_sum_sq; tg.counts = ctx->fb_counts;
        Target deferred = tg;
        deferred.defer = 1u;
        st = run_paths(ctx, work, nullptr, ctx->fb_counts, 0, n_noisy * b, ctx->fb_w, ctx->fb_h, max_bounces, deferred, false, b);
        if (st != PTX_OK) return st;
        HIPCHK(hipMemsetAsync(d_count, 0, sizeof(IterCtr), ctx->stream));
        {
            ScopedEvent ev(ctx, CLS_ACCUM);
            hipLaunchKernelGGL(k_accumulate_adaptive, dim3((n_noisy + 255) / 256), dim3(256), 0, ctx->stream, ctx->P, work, n_noisy, b,
                               cfg->min_samples, cfg->variance_threshold, tg, d_count);
        }
        HIPCHK(hipMemcpyAsync(ctx->h_pinned + 8, d_count, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const uint32_t applied_max = ctx->h_pinned[8];
        const uint64_t applied = (uint64_t)ctx->h_pinned[10] | ((uint64_t)ctx->h_pinned[11] << 32);
        passes += applied_max; rendered += applied;
        if (applied_max < b) break;    // the next pass would find no noisy pixel
        // longer batches keep the queues long when few pixels are left (or a rank owns an eighth of them); they only
        // pay while pixels rarely converge inside a batch, i.e. while nearly every speculative sample is kept
        if (ctx->adaptive_batch_auto) {
            if (applied * 10 >= (uint64_t)n_noisy * b * 9) batch = std::min<uint32_t>(batch * 2u, 256u);
            else if (applied * 2 < (uint64_t)n_noisy * b) batch = std::max<uint32_t>(batch / 2u, 8u);
        }
    }
    *out_passes = passes; *out_rendered = rendered;
    return check_device_error(ctx);
}

int32_t ptx_fb_stats(PtxContext* ctx, uint32_t* out_min, uint32_t* out_max, uint64_t* out_total) {
    REQUIRE_CTX();
    if (!ctx->fb_sum) return fail(ctx, PTX_ERR_INVALID, "no framebuffer: call ptx_fb_reset");
    int32_t st = ensure_worklist(ctx, ctx->fb_w, ctx->fb_h);
    if (st != PTX_OK) return st;
    uint32_t* d = reinterpret_cast<uint32_t*>(&ctx->ctrs[CTR_STATS]);
    const uint32_t init[4] = {0xFFFFFFFFu, 0u, 0u, 0u};
    HIPCHK(hipMemcpyAsync(d, init, 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->wl_n) hipLaunchKernelGGL(k_fb_stats, dim3(std::min<uint32_t>(1024, (ctx->wl_n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->wl_pixel, ctx->wl_n,
                                      ctx->fb_counts, d, d + 1, reinterpret_cast<unsigned long long*>(d + 2));
    HIPCHK(hipMemcpyAsync(ctx->h_pinned + 4, d, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (out_min) *out_min = ctx->wl_n ? ctx->h_pinned[4] : 0;
    if (out_max) *out_max = ctx->h_pinned[5];
    if (out_total) std::memcpy(out_total, ctx->h_pinned + 6, 8);
    return check_device_error(ctx);
}

int32_t ptx_fb_read(PtxContext* ctx, float* out_sum, float* out_sum_sq, uint32_t* out_counts) {
    REQUIRE_CTX();
    if (!ctx->fb_sum) return fail(ctx, PTX_ERR_INVALID, "no framebuffer: call ptx_fb_reset");
    const uint64_t npx = (uint64_t)ctx->fb_w * ctx->fb_h;
    if (out_sum) HIPCHK(hipMemcpyAsync(out_sum, ctx->fb_sum, npx * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (out_sum_sq && ctx->fb_sum_sq) HIPCHK(hipMemcpyAsync(out_sum_sq, ctx->fb_sum_sq, npx * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (out_counts) HIPCHK(hipMemcpyAsync(out_counts, ctx->fb_counts, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    return check_device_error(ctx);
}

int32_t ptx_fb_read_mean(PtxContext* ctx, float* out_mean) {
    REQUIRE_CTX();
    if (!ctx->fb_sum || !out_mean) return fail(ctx, PTX_ERR_INVALID, "no framebuffer or null output");
    const uint64_t npx = (uint64_t)ctx->fb_w * ctx->fb_h;
    int32_t st = ensure_images(ctx, npx);
    if (st != PTX_OK) return st;
    hipLaunchKernelGGL(k_mean, dim3((uint32_t)((npx + 255) / 256)), dim3(256), 0, ctx->stream, ctx->fb_sum, ctx->fb_counts, (uint32_t)npx, ctx->img);
    HIPCHK(hipMemcpyAsync(out_mean, ctx->img, npx * 12, hipMemcpyDeviceToHost, ctx->stream));
    return check_device_error(ctx);
}

int32_t ptx_fb_tonemap_srgb8(PtxContext* ctx, int32_t tonemap, float exposure, uint8_t* out_rgb8) {
    REQUIRE_CTX();
    if (!ctx->fb_sum || !out_rgb8) return fail(ctx, PTX_ERR_INVALID, "no framebuffer or null output");
    if (tonemap < 0 || tonemap > 3) return fail(ctx, PTX_ERR_INVALID, "tonemap must be 0 (None), 1 (Aces), 2 (Reinhard) or 3 (Agx)");
    const uint64_t npx = (uint64_t)ctx->fb_w * ctx->fb_h;
    int32_t st = ensure_images(ctx, npx);
    if (st != PTX_OK) return st;
    uint8_t* d_out = reinterpret_cast<uint8_t*>(ctx->img);
    hipLaunchKernelGGL(k_tonemap_srgb8, dim3((uint32_t)((npx + 255) / 256)), dim3(256), 0, ctx->stream, ctx->fb_sum, ctx->fb_counts,
                       (uint32_t)npx, (int)tonemap, exposure, d_out);
    HIPCHK(hipMemcpyAsync(out_rgb8, d_out, npx * 3, hipMemcpyDeviceToHost, ctx->stream));
    return check_device_error(ctx);
}

int32_t ptx_fb_export_dev(PtxContext* ctx, float* d_sum, float* d_sum_sq, uint32_t* d_counts) {
    REQUIRE_CTX();
    if (!ctx->fb_sum) return fail(ctx, PTX_ERR_INVALID, "no framebuffer: call ptx_fb_reset");
    const uint64_t npx = (uint64_t)ctx->fb_w * ctx->fb_h;
    if (d_sum) HIPCHK(hipMemcpyAsync(d_sum, ctx->fb_sum, npx * 12, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_sum_sq && ctx->fb_sum_sq) HIPCHK(hipMemcpyAsync(d_sum_sq, ctx->fb_sum_sq, npx * 12, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_counts) HIPCHK(hipMemcpyAsync(d_counts, ctx->fb_counts, npx * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return PTX_OK;
}

// ---- multi-GPU exchange ---------------------------------------------------------------------
static uint32_t tiles_of_rank(uint32_t w, uint32_t h, uint32_t rank, uint32_t world) {
    const uint64_t n_tiles = (uint64_t)((w + 7) / 8) * ((h + 7) / 8);
    return rank < n_tiles ? (uint32_t)((n_tiles - rank + world - 1) / world) : 0u;
}
uint64_t ptx_fb_tiles_bytes(uint32_t width, uint32_t height, uint32_t world) {
    if (!world || !width || !height) return 0;
    return (uint64_t)tiles_of_rank(width, height, 0, world) * 64u * 28u;     // rank 0 owns the most tiles
}
int32_t ptx_fb_export_tiles_dev(PtxContext* ctx, void* d_out, uint64_t capacity_bytes) {
    REQUIRE_CTX();
    if (!ctx->fb_sum) return fail(ctx, PTX_ERR_INVALID, "no framebuffer: call ptx_fb_reset");
    const uint64_t need = ptx_fb_tiles_bytes(ctx->fb_w, ctx->fb_h, ctx->world);
    if (!d_out || capacity_bytes < need) return fail(ctx, PTX_ERR_INVALID, "tile export buffer is null or smaller than ptx_fb_tiles_bytes");
    const uint32_t nt = tiles_of_rank(ctx->fb_w, ctx->fb_h, ctx->rank, ctx->world);
    // ranks with one tile fewer than rank 0: the tail stays zero (count 0), so that the gathered block has one size
    if ((uint64_t)nt * 64u * 28u < need) HIPCHK(hipMemsetAsync((char*)d_out + (uint64_t)nt * 64u * 28u, 0, need - (uint64_t)nt * 64u * 28u, ctx->stream));
    if (nt) hipLaunchKernelGGL(k_export_tiles, dim3((nt * 64u + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->fb_sum, ctx->fb_sum_sq, ctx->fb_counts,
                               ctx->fb_w, ctx->fb_h, ctx->rank, ctx->world, nt, (uint32_t*)d_out);
    HIPCHK(hipGetLastError());
    return PTX_OK;
}
int32_t ptx_fb_import_tiles_dev(PtxContext* ctx, const void* d_in, uint32_t src_rank, uint32_t world) {
    REQUIRE_CTX();
    if (!ctx->fb_sum) return fail(ctx, PTX_ERR_INVALID, "no framebuffer: call ptx_fb_reset");
    if (!d_in || !world || src_rank >= world) return fail(ctx, PTX_ERR_INVALID, "bad tile import arguments");
    const uint32_t nt = tiles_of_rank(ctx->fb_w, ctx->fb_h, src_rank, world);
    if (nt) hipLaunchKernelGGL(k_import_tiles, dim3((nt * 64u + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->fb_sum, ctx->fb_sum_sq, ctx->fb_counts,
                               ctx->fb_w, ctx->fb_h, src_rank, world, nt, (const uint32_t*)d_in);
    HIPCHK(hipGetLastError());
    return PTX_OK;
}

// ---- parity / measurement ------------------------------------------------------------------
int32_t ptx_order_after_frame_tail(PtxContext* ctx, PtxContext* other) {
    REQUIRE_CTX();
    if (!other || other == ctx) return fail(ctx, PTX_ERR_INVALID, "needs another context");
    if (other->device != ctx->device) return fail(ctx, PTX_ERR_INVALID, "contexts are on different devices");
    HIPCHK(hipStreamWaitEvent(ctx->stream, other->ev_tail, 0));     // an event never recorded is complete: no-op
    return PTX_OK;
}
int32_t ptx_order_after_stream(PtxContext* ctx, void* stream) {
    REQUIRE_CTX();
    HIPCHK(hipEventRecord(ctx->ev_ext_in, static_cast<hipStream_t>(stream)));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_ext_in, 0));
    return PTX_OK;
}
int32_t ptx_order_stream_after(PtxContext* ctx, void* stream) {
    REQUIRE_CTX();
    HIPCHK(hipEventRecord(ctx->ev_ext_out, ctx->stream));
    HIPCHK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), ctx->ev_ext_out, 0));
    return PTX_OK;
}

int32_t ptx_render_frame(PtxContext* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_bounces,
                         const PtxAdaptiveConfig* adaptive, uint32_t max_spp, float* out_mean, uint32_t* out_counts,
                         PtxFrameStats* out_stats) {
    REQUIRE_CAMERA();
    int32_t st = ptx_fb_reset(ctx, width, height, adaptive ? 1 : 0);
    if (st != PTX_OK) return st;
    st = ptx_fb_render_passes(ctx, 0, spp, max_bounces);
    if (st != PTX_OK) return st;
    PtxFrameStats fs{};
    fs.uniform_passes = spp;
    if (adaptive && max_spp > spp) {
        // every pixel starts at `spp` samples and a noisy pixel is sampled in every pass, so "stop once a pixel holds
        // max_spp samples" (app.rs) is a cap of max_spp - spp passes
        uint32_t passes = 0; uint64_t rendered = 0;
        st = ptx_fb_adaptive_passes(ctx, adaptive, max_bounces, max_spp - spp, &passes, &rendered);
        if (st != PTX_OK) return st;
        fs.adaptive_passes = passes;
        fs.adaptive_samples = rendered;
    }
    st = ptx_fb_stats(ctx, &fs.min_samples, &fs.max_samples, &fs.total_samples);
    if (st != PTX_OK) return st;
    if (out_mean) { st = ptx_fb_read_mean(ctx, out_mean); if (st != PTX_OK) return st; }
    if (out_counts) { st = ptx_fb_read(ctx, nullptr, nullptr, out_counts); if (st != PTX_OK) return st; }
    if (out_stats) *out_stats = fs;
    return check_device_error(ctx);
}

int32_t ptx_trace_batch(PtxContext* ctx, const PtxRay* rays, uint64_t n, int32_t mode, PtxHitRecord* out_hits, uint8_t* out_any) {
    REQUIRE_SCENE();
    if (n == 0) return PTX_OK;
    if (n > 0x7FFFFFFFull) return fail(ctx, PTX_ERR_INVALID, "batch too large (max 2^31-1 rays per call)");
    if (!rays || (mode == PTX_TRACE_CLOSEST && !out_hits) || (mode == PTX_TRACE_ANY && !out_any) || (mode != PTX_TRACE_CLOSEST && mode != PTX_TRACE_ANY))
        return fail(ctx, PTX_ERR_INVALID, "bad trace_batch arguments");
    PtxRay* d_rays = nullptr; void* d_out = nullptr;
    const size_t out_bytes = mode == PTX_TRACE_ANY ? n : n * sizeof(PtxHitRecord);
    HIPCHK(hipMalloc((void**)&d_rays, n * sizeof(PtxRay)));
    if (hipMalloc(&d_out, out_bytes) != hipSuccess) { (void)hipFree(d_rays); return fail(ctx, PTX_ERR_HIP, "hipMalloc failed"); }
    int32_t st = PTX_OK;
    do {
        if (hipMemcpyAsync(d_rays, rays, n * sizeof(PtxRay), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { st = fail(ctx, PTX_ERR_HIP, "H2D copy failed"); break; }
        const int grid = (int)std::min<uint64_t>(TRACE_GRID, (n + TRACE_BLOCK - 1) / TRACE_BLOCK);
        uint32_t* d_work = ctx->ctrs[CTR_BATCH].w_extend;   // the batch cursor
        if (hipMemsetAsync(&ctx->ctrs[CTR_BATCH], 0, sizeof(IterCtr), ctx->stream) != hipSuccess) { st = fail(ctx, PTX_ERR_HIP, "memset failed"); break; }
        ScopedEvent* ev = new ScopedEvent(ctx, mode == PTX_TRACE_ANY ? CLS_SHADOW : CLS_EXTEND);
        if (mode == PTX_TRACE_ANY && ctx->packet_batch && !ctx->counting) {
            hipLaunchKernelGGL(k_trace_batch_packet<true>, dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)nullptr, (uint8_t*)d_out, d_work, ctx->ovf, ctx->d_err, ctx->d_counters);
        } else if (mode == PTX_TRACE_ANY) {
            if (ctx->counting) hipLaunchKernelGGL((k_trace_batch<true, true>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)nullptr, (uint8_t*)d_out, d_work, ctx->ovf, ctx->d_counters, ctx->d_err);
            else hipLaunchKernelGGL((k_trace_batch<true, false>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)nullptr, (uint8_t*)d_out, d_work, ctx->ovf, ctx->d_counters, ctx->d_err);
        } else if (ctx->group_lanes && !ctx->counting) {
            const int gg = (int)std::min<uint64_t>(TRACE_GRID, (n * ctx->group_lanes + TRACE_BLOCK - 1) / TRACE_BLOCK);
            if (ctx->group_lanes == 8) hipLaunchKernelGGL(k_trace_group<8>, dim3(gg), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)d_out, d_work, ctx->d_err);
            else hipLaunchKernelGGL(k_trace_group<4>, dim3(gg), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)d_out, d_work, ctx->d_err);
        } else if (ctx->packet_batch && !ctx->counting) {
            hipLaunchKernelGGL(k_trace_batch_packet<false>, dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)d_out, (uint8_t*)nullptr, d_work, ctx->ovf, ctx->d_err, ctx->d_counters);
        } else {
            if (ctx->counting) hipLaunchKernelGGL((k_trace_batch<false, true>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)d_out, (uint8_t*)nullptr, d_work, ctx->ovf, ctx->d_counters, ctx->d_err);
            else if (ctx->S.fast_on && ctx->n_uncertified) hipLaunchKernelGGL((k_trace_batch<false, false, true, true>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)d_out, (uint8_t*)nullptr, d_work, ctx->ovf, ctx->d_counters, ctx->d_err);
            else if (ctx->S.fast_on) hipLaunchKernelGGL((k_trace_batch<false, false, true, false>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)d_out, (uint8_t*)nullptr, d_work, ctx->ovf, ctx->d_counters, ctx->d_err);
            else hipLaunchKernelGGL((k_trace_batch<false, false>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)d_out, (uint8_t*)nullptr, d_work, ctx->ovf, ctx->d_counters, ctx->d_err);
        }
        delete ev;
        if (hipMemcpyAsync(mode == PTX_TRACE_ANY ? (void*)out_any : (void*)out_hits, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { st = fail(ctx, PTX_ERR_HIP, "D2H copy failed"); break; }
        st = check_device_error(ctx);
    } while (0);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_rays); (void)hipFree(d_out);
    return st;
}

int32_t ptx_trace_batch_noalloc(PtxContext* ctx, const PtxRay* d_rays, uint64_t n, int32_t mode, PtxHitRecord* d_out_hits,
                                uint8_t* d_out_any) {
    REQUIRE_SCENE();
    if (n == 0) return PTX_OK;
    if (n > PTX_NOALLOC_MAX_RAYS) return fail(ctx, PTX_ERR_INVALID, "ptx_trace_batch_noalloc: more than PTX_NOALLOC_MAX_RAYS (2^24) rays in one call");
    if (!d_rays || (mode == PTX_TRACE_CLOSEST && !d_out_hits) || (mode == PTX_TRACE_ANY && !d_out_any) || (mode != PTX_TRACE_CLOSEST && mode != PTX_TRACE_ANY))
        return fail(ctx, PTX_ERR_INVALID, "bad trace_batch arguments");
    unsigned char* base = ctx->device < NOALLOC_MAX_DEVICES ? g_scratch_ptr[ctx->device] : nullptr;
    if (!base) return fail(ctx, PTX_ERR_UNSUPPORTED, "no scratch array resolved for this device");
    uint32_t* d_work = reinterpret_cast<uint32_t*>(base);
    uint32_t* d_err = reinterpret_cast<uint32_t*>(base + NOALLOC_ERR_OFFSET);
    uint2* d_ovf = reinterpret_cast<uint2*>(base + NOALLOC_HEAD);
    // the array keeps the last call's contents: cursors and error word must start at zero (stack strips are written
    // before they are read)
    HIPCHK(hipMemsetAsync(base, 0, NOALLOC_HEAD, ctx->stream));
    const int grid = (int)std::min<uint64_t>(NOALLOC_WAVES, (n + TRACE_BLOCK - 1) / TRACE_BLOCK);
    if (mode == PTX_TRACE_ANY) {
        if (ctx->counting) hipLaunchKernelGGL((k_trace_batch<true, true>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)nullptr, d_out_any, d_work, d_ovf, ctx->d_counters, d_err);
        else hipLaunchKernelGGL((k_trace_batch<true, false>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, (PtxHitRecord*)nullptr, d_out_any, d_work, d_ovf, ctx->d_counters, d_err);
    } else {
        if (ctx->counting) hipLaunchKernelGGL((k_trace_batch<false, true>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, d_out_hits, (uint8_t*)nullptr, d_work, d_ovf, ctx->d_counters, d_err);
        else if (ctx->S.fast_on && ctx->n_uncertified) hipLaunchKernelGGL((k_trace_batch<false, false, true, true>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, d_out_hits, (uint8_t*)nullptr, d_work, d_ovf, ctx->d_counters, d_err);
        else if (ctx->S.fast_on) hipLaunchKernelGGL((k_trace_batch<false, false, true, false>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, d_out_hits, (uint8_t*)nullptr, d_work, d_ovf, ctx->d_counters, d_err);
        else hipLaunchKernelGGL((k_trace_batch<false, false>), dim3(grid), dim3(TRACE_BLOCK), 0, ctx->stream, ctx->S, d_rays, (uint32_t)n, d_out_hits, (uint8_t*)nullptr, d_work, d_ovf, ctx->d_counters, d_err);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ctx->h_pinned + 12, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->h_pinned[12] != 0)
        return fail(ctx, PTX_ERR_STACK_OVERFLOW, "a traversal needed more than the 144 stack entries a ray has on the device (the reference panics beyond 64 per BVH level)");
    return PTX_OK;
}

int32_t ptx_generate_primary_rays(PtxContext* ctx, uint32_t w, uint32_t h, uint32_t sample_index, PtxRay* out_rays) {
    REQUIRE_CAMERA();
    if (!w || !h || !out_rays) return fail(ctx, PTX_ERR_INVALID, "bad arguments");
    const uint32_t n = w * h;
    int32_t st = ensure_lists(ctx, (uint64_t)n * 3);   // list_a then holds n PtxRay (28 B <= 36 B)
    if (st != PTX_OK) return st;
    uint32_t* ident = ctx->list_b;
    hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ident, n);
    PtxRay* d_rays = reinterpret_cast<PtxRay*>(ctx->list_a);
    PathArrays none{};
    hipLaunchKernelGGL(k_raygen, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->S, ctx->cam, none, ident, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, sample_index, n, w, h, (uint32_t*)nullptr, (IterCtr*)nullptr, 0u, d_rays, 0u);
    HIPCHK(hipMemcpyAsync(out_rays, d_rays, (size_t)n * sizeof(PtxRay), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PTX_OK;
}

int32_t ptx_debug_math(PtxContext* ctx, int32_t fn, const float* x, const float* y, uint64_t n, float* out) {
    REQUIRE_CTX();
    if (!x || !y || !out || n == 0) return fail(ctx, PTX_ERR_INVALID, "bad arguments");
    float* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, n * 3 * sizeof(float)));
    int32_t st = PTX_OK;
    if (hipMemcpyAsync(d, x, n * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d + n, y, n * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = fail(ctx, PTX_ERR_HIP, "H2D copy failed");
    if (st == PTX_OK) {
        hipLaunchKernelGGL(k_debug_math, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, fn, d, d + n, n, d + 2 * n);
        if (hipMemcpyAsync(out, d + 2 * n, n * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = fail(ctx, PTX_ERR_HIP, "D2H copy failed");
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return st;
}

int32_t ptx_set_counting(PtxContext* ctx, int32_t enabled) { REQUIRE_CTX(); ctx->counting = enabled != 0; ctx->count_shipped = enabled == 2; return PTX_OK; }
int32_t ptx_reset_counters(PtxContext* ctx) {
    REQUIRE_CTX();
    HIPCHK(hipMemsetAsync(ctx->d_counters, 0, sizeof(DevCounters), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PTX_OK;
}
int32_t ptx_get_counters(PtxContext* ctx, PtxCounters* out) {
    REQUIRE_CTX();
    if (!out) return fail(ctx, PTX_ERR_INVALID, "null output");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out, ctx->d_counters, sizeof(PtxCounters), hipMemcpyDeviceToHost));
    return PTX_OK;
}
int32_t ptx_get_debug_stats(PtxContext* ctx, PtxDebugStats* out) {
    REQUIRE_CTX();
    if (!out) return fail(ctx, PTX_ERR_INVALID, "null output");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    unsigned int dbg[8];
    HIPCHK(hipMemcpy(dbg, ctx->d_counters->dbg, sizeof(dbg), hipMemcpyDeviceToHost));
    out->fast_redo_rays = (uint64_t)dbg[6] | ((uint64_t)dbg[7] << 32);
    out->fast_order = ctx->has_scene ? ctx->S.fast_on : 0u;
    out->uncertified_meshes = ctx->has_scene ? ctx->n_uncertified : 0u;
    std::memcpy(&out->max_leaf_undershoot, &dbg[4], 4);
    HIPCHK(hipMemcpy(out->vote_closest, ctx->d_counters->vote, sizeof(uint64_t) * 18, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out->stack_ops, ctx->d_counters->stk, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost));
    out->bound_forwarding = ctx->has_scene ? ctx->S.fwd_on : 0u;
    std::memcpy(&out->max_leaf_undershoot_excess, &dbg[5], 4);
    out->trace_stack_high_water = dbg[0];
    out->packet_stack_high_water = dbg[1];
    out->packet_redo_rays = (uint64_t)dbg[2] | ((uint64_t)dbg[3] << 32);
    out->lds_stack_entries = (uint32_t)LDS_STACK;
    out->packet_lds_entries = (uint32_t)PACKET_LDS;
    out->packet_capacity = (uint32_t)ctx->packet_capacity;
    return PTX_OK;
}
int32_t ptx_set_profiling(PtxContext* ctx, int32_t enabled) { REQUIRE_CTX(); drain_events(ctx); ctx->profiling = enabled != 0; return PTX_OK; }
int32_t ptx_reset_timings(PtxContext* ctx) { REQUIRE_CTX(); drain_events(ctx); ctx->timings = PtxTimings{}; return PTX_OK; }
int32_t ptx_get_timings(PtxContext* ctx, PtxTimings* out) {
    REQUIRE_CTX();
    if (!out) return fail(ctx, PTX_ERR_INVALID, "null output");
    drain_events(ctx);
    *out = ctx->timings;
    return PTX_OK;
}

}  // extern "C"
