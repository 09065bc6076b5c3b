/* ptx_core.h — C ABI of the MI355X render core (libmi355_pt.so).
 *
 * This is the one boundary between a host renderer (the reference's Rust host:
 * scene loading, SAH build + collapse, camera, tonemapping, OIDN) and the
 * hand-written gfx950 HIP kernels.  It sits where the reference's
 * `RenderLoop::render_pass` sits (src/raytracer/renderloop/mod.rs:9-22) and
 * where `run_benchmark`'s `for sample` loop sits (src/main.rs:123-141).
 *
 * Conventions
 *   - every call returns int32 status, 0 == PTX_OK; nothing throws or aborts
 *     across the boundary; `ptx_last_error` gives a message;
 *   - the caller owns every host buffer it passes in; the library owns all
 *     device memory (except explicit `*_dev` outputs, which are caller-owned
 *     device pointers);
 *   - matrices are column-major float[16], exactly glam's `Mat4::to_cols_array`;
 *   - one context drives one GPU; it is not re-entrant (the reference's render
 *     loop is single-caller too).
 *
 * All struct layouts below are plain `repr(C)`-compatible PODs so that a Rust
 * `-sys` crate, cgo or ctypes can bind them directly.
 */
#ifndef PTX_CORE_H
#define PTX_CORE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTX_ABI_VERSION 1

enum {
    PTX_OK = 0,
    PTX_ERR_INVALID = 1,      /* bad argument / scene fails validation          */
    PTX_ERR_NO_DEVICE = 2,    /* no HIP device / device index out of range      */
    PTX_ERR_HIP = 3,          /* a HIP runtime call failed                      */
    PTX_ERR_NO_SCENE = 4,     /* render/trace before ptx_upload_scene           */
    PTX_ERR_STACK_OVERFLOW = 5, /* a traversal needed more stack than a ray's device stack
                                   holds (144 entries shared by the TLAS, Blas and mesh
                                   levels).  The reference's arrays hold 64 entries per
                                   level and panic beyond (accel/bvh.rs:320,361,379,408);
                                   scenes it can trace need far less than 144              */
    PTX_ERR_UNSUPPORTED = 6
};

/* ---- acceleration structure, as the reference host builds it ------------- */

/* Bit-for-bit the reference's `Bvh8Node` (accel/bvh.rs:86-98):
 * #[repr(C, align(64))], 232 B payload, sizeof 256.  Child word
 * (accel/bvh.rs:116-169): internal = node index (bit31 = 0); leaf =
 * 1<<31 | (count-1)<<24 | prim_start. */
typedef struct __attribute__((aligned(64))) PtxBvh8Node {
    float min_x[8], max_x[8];
    float min_y[8], max_y[8];
    float min_z[8], max_z[8];
    uint32_t children[8];
    uint8_t valid_mask;
    uint8_t _padding[7];
} PtxBvh8Node;

/* `Bvh8` (accel/bvh.rs:266-269). */
typedef struct PtxBvh8 {
    const PtxBvh8Node* nodes;
    uint32_t n_nodes;
    const uint32_t* primitive_indices;
    uint32_t n_primitive_indices;
} PtxBvh8;

/* One `Blas<TriangleMesh>` holding exactly one `TriangleMesh`
 * (shape/triangle_mesh.rs:6-14; the loader never builds anything else:
 * loader/gltf.rs:300-302).  `bounds_*` is the mesh's unpadded position bounds
 * (triangle_mesh.rs:56-61) == slot 0 of the one-node item BVH of the Blas
 * (accel/mod.rs:27-52).
 * NOT SUPPORTED: a `Blas` built over several items (`Blas::build(vec![a, b, ..])`,
 * accel/mod.rs:27-52, which the type allows and nothing in the reference does).
 * Its item-level BVH8 would be a third tree between instance and mesh; this
 * descriptor has no place for it, so such a host must give every item its own
 * PtxMesh + PtxInstance (same image, different instance ids) or stay on the CPU. */
typedef struct PtxMesh {
    const float* positions;  uint32_t n_positions;   /* xyz triples            */
    const float* normals;    uint32_t n_normals;     /* xyz triples, 0 = none  */
    const float* uvs;        uint32_t n_uvs;         /* uv pairs,   0 = none   */
    const uint32_t* indices; uint32_t n_triangles;   /* [u32;3] per triangle   */
    uint32_t material_id;
    PtxBvh8 bvh;                                     /* over triangles, leaf 4 */
    float bounds_min[3];
    float bounds_max[3];
} PtxMesh;

/* `Instance` (accel/mod.rs:75-81). */
typedef struct PtxInstance {
    uint32_t blas_index;
    float transform[16];
    float inverse_transform[16];
    float normal_matrix[16];       /* (M^-1)^T */
} PtxInstance;

/* ---- shading inputs -------------------------------------------------------- */

/* `Material` (material.rs:12-36); texture ids are -1 for None.  Only the
 * fields the integrator reads are carried. */
typedef struct PtxMaterial {
    int32_t base_color_texture;
    int32_t metallic_roughness_texture;
    int32_t normal_texture;
    int32_t emissive_texture;
    float base_color_factor[4];
    float roughness_factor;
    float metallic_factor;
    float normal_scale;
    float emissive_factor[3];
    float ior;
    float transmission;
    float subsurface;
    float subsurface_radius[3];
    float subsurface_albedo[3];
    float subsurface_anisotropy;
    float subsurface_entry_tint[3];
    float subsurface_exit_tint[3];
} PtxMaterial;

/* `Texture` (texture.rs:4-12): f32 RGB texels, row-major, or a solid colour
 * (`texels == NULL`). */
typedef struct PtxTexture {
    uint32_t width, height;
    const float* texels;           /* width*height*3, NULL => SolidColor(solid) */
    float solid[3];
} PtxTexture;

/* `TextureSky` (sky/texture.rs:6-14) incl. the host-built CDFs (:67-112). */
typedef struct PtxSky {
    PtxTexture texture;
    const float* marginal_cdf;      /* height + 1                      */
    const float* conditional_cdfs;  /* height * (width + 1), row-major */
    uint32_t width, height;         /* (1,1) for a solid colour        */
    float total_weight;
    float rotation;
} PtxSky;

/* `Option<DirectionalLight>` (light.rs:4-10). */
typedef struct PtxSun {
    uint32_t enabled;
    float direction[3];
    float color[3];
    float cos_angle;
    float pdf;
} PtxSun;

/* `BlueNoiseTexture` (bluenoise.rs:3-6): size*size Vec2 samples. */
typedef struct PtxBlueNoise {
    uint32_t size;
    const float* samples;
} PtxBlueNoise;

/* What `Scene::new` holds after `Tlas::build` (scene.rs:22-40,
 * accel/mod.rs:166-192) plus the render loop's two blue-noise tables
 * (renderloop/parallel.rs:19-28). */
typedef struct PtxSceneDesc {
    uint32_t abi_version;
    const PtxMesh* meshes;          uint32_t n_meshes;      /* == blases */
    const PtxInstance* instances;   uint32_t n_instances;
    PtxBvh8 tlas;                                           /* over instances, leaf 1 */
    const PtxMaterial* materials;   uint32_t n_materials;
    const PtxTexture* textures;     uint32_t n_textures;
    PtxSky sky;
    PtxSun sun;
    PtxBlueNoise blue_noise_pixel;  /* seed 12345 */
    PtxBlueNoise blue_noise_lens;   /* seed 67890 */
} PtxSceneDesc;

/* The part of `Camera` that `generate_ray_dof` reads (camera.rs:66-95). */
typedef struct PtxCamera {
    float inv_vp_matrix[16];
    float aperture;
    float focus_distance;
    float right[3];                /* Camera::right()                  */
    float up[3];                   /* right.cross(forward)             */
} PtxCamera;

/* ---- ray batches (parity/debug entry) ------------------------ */

/* `Ray` as the caller builds it: `Ray::new(origin, direction)` gives
 * t_min = 0.001, t_max = +inf; `with_t_max` overrides t_max (ray.rs:12-27).
 * inv_direction is recomputed by the library exactly as Ray::new does. */
typedef struct PtxRay {
    float origin[3];
    float direction[3];
    float t_max;
} PtxRay;

#define PTX_MISS 0xFFFFFFFFu

/* What the reference's closures know and `HitData` drops (ray.rs:29-37):
 * the TLAS primitive index (instance), the mesh primitive index (triangle),
 * world t, and the barycentrics of `intersect_triangle`. */
typedef struct PtxHitRecord {
    uint32_t inst_idx;             /* PTX_MISS when nothing was hit    */
    uint32_t tri_idx;
    float t;
    float u, v;
} PtxHitRecord;

enum { PTX_TRACE_CLOSEST = 0, PTX_TRACE_ANY = 1 };

/* Event counts that define the algorithmic bytes of SURVEY.md §8(d). */
typedef struct PtxCounters {
    uint64_t rays_closest;          /* Scene::hit calls                           */
    uint64_t rays_any;              /* Scene::hit_any calls                       */
    uint64_t node_visits;           /* every pop that reaches node.intersect      */
    uint64_t node_children;         /* sum of popcount(valid_mask) over visits    */
    uint64_t prim_index_reads;
    uint64_t triangle_tests;
    uint64_t instance_entries;
    uint64_t instance_hits;         /* Instance::hit returned Some                */
    uint64_t hitdata_built;         /* compute_hit_data for a path vertex         */
    uint64_t material_fetches;
    uint64_t texture_lookups;
    uint64_t cdf_probes;
    uint64_t sky_pdf_evals;
    uint64_t sky_texel_lookups;
    uint64_t primary_rays;
    uint64_t pixel_samples_written;
    uint64_t sss_steps;
    /* the share of the five traversal counts above that hit_any rays caused
     * (already included in the totals); lets a caller price closest-hit
     * (extend) and any-hit (shadow) traversal separately */
    uint64_t any_node_visits;
    uint64_t any_node_children;
    uint64_t any_prim_index_reads;
    uint64_t any_triangle_tests;
    uint64_t any_instance_entries;
} PtxCounters;

/* `AdaptiveSamplingConfig` (renderloop/parallel.rs:13-17). */
typedef struct PtxAdaptiveConfig {
    uint32_t min_samples;
    float variance_threshold;
    uint64_t max_samples_per_pass;
} PtxAdaptiveConfig;

/* Per-kernel-class device time of the calls since the last reset (HIP events
 * on the library's own stream; only recorded while profiling is enabled). */
typedef struct PtxTimings {
    double raygen_ms, extend_ms, shade_ms, shadow_ms, accumulate_ms, other_ms;
    uint64_t extend_launches, shade_launches, shadow_launches;
    /* closest-hit launches traced as packets (the primary rays of a frame): their own kernel, so their own line;
     * extend_ms / extend_launches are the per-ray kernel's alone */
    double packet_ms;
    uint64_t packet_launches;
    /* the subsurface random-walk kernel (one launch per iteration in scenes with subsurface materials) */
    double walk_ms;
    uint64_t walk_launches;
    /* the tail kernel: the late iterations of a batch, all phases in one launch (closest hit, shading, walk steps,
     * any hit, accumulation of the paths it carries are in here, not in the lines above) */
    double tail_ms;
    uint64_t tail_launches;
} PtxTimings;

typedef struct PtxContext PtxContext;

/* ---- lifecycle ------------------------------------------------------------- */
int32_t ptx_create(int32_t device_index, PtxContext** out_ctx);
int32_t ptx_destroy(PtxContext* ctx);
const char* ptx_last_error(const PtxContext* ctx);   /* ctx may be NULL */

/* Mirrors the result of `Scene::new` (scene.rs:22-40): copies and repacks the
 * whole scene into HBM.  The descriptor may be freed afterwards. */
int32_t ptx_upload_scene(PtxContext* ctx, const PtxSceneDesc* scene);
/* Mirrors `Camera` after `update_matrix` (camera.rs:51-60). */
int32_t ptx_set_camera(PtxContext* ctx, const PtxCamera* camera);

/* Pixel-tile partition for multi-GPU frames: this context renders only 8x8
 * tiles with tile_index % world == rank.  Default (0, 1) = everything. */
int32_t ptx_set_partition(PtxContext* ctx, uint32_t rank, uint32_t world);

/* ---- the reference's render-loop entry points ------------------------------ */

/* `RenderLoop::render_pass` (renderloop/parallel.rs:171-194): one sample per
 * pixel with `PathTracer<max_bounces>`; out_rgb = width*height*3 floats
 * (row-major Vec<Vec3>). */
int32_t ptx_render_pass(PtxContext* ctx, uint32_t width, uint32_t height,
                        uint32_t sample_index, uint32_t max_bounces, float* out_rgb);

/* `render_first_pass_with_aov` (renderloop/parallel.rs:125-168): sample 0 plus
 * first-hit albedo / normal*0.5+0.5. */
int32_t ptx_render_first_pass_aov(PtxContext* ctx, uint32_t width, uint32_t height,
                                  uint32_t max_bounces, float* out_rgb,
                                  float* out_albedo, float* out_normal);

/* `render_pixels` (renderloop/parallel.rs:58-78): work list of (x, y,
 * sample_index) triplets; out_rgb = n*3 floats in list order. */
int32_t ptx_render_pixels(PtxContext* ctx, uint32_t width, uint32_t height,
                          const uint32_t* xys, uint64_t n, uint32_t max_bounces,
                          float* out_rgb);

/* `UnshadedRenderer` through `render_pass` (renderer/unshaded.rs:7-23,
 * app.rs:178-193): the interactive low-res preview. */
int32_t ptx_render_pass_unshaded(PtxContext* ctx, uint32_t width, uint32_t height,
                                 uint32_t sample_index, float* out_rgb);

/* ---- device-resident framebuffer: the `for sample` loop on the GPU --------- */

/* `Framebuffer::new` / `AdaptiveFramebuffer::new` + `reset`
 * (framebuffer.rs:17-42, 84-108).  adaptive != 0 keeps sum, sum_sq, counts. */
int32_t ptx_fb_reset(PtxContext* ctx, uint32_t width, uint32_t height, int32_t adaptive);

/* `for sample in first..first+count { render_pass; accumulate }`
 * (main.rs:124-133, framebuffer.rs:44-50, 118-125).  Nothing leaves the GPU. */
int32_t ptx_fb_render_passes(PtxContext* ctx, uint32_t first_sample, uint32_t count,
                             uint32_t max_bounces);
/* The same, returning as soon as the work is enqueued on the library's streams (no counterpart in the reference, which is
 * synchronous): a traversal stack overflow is then reported by the next call that synchronises (ptx_sync, ptx_fb_stats,
 * ptx_fb_read*, ptx_fb_tonemap_srgb8) instead of by this one.  Lets a host keep several frames in flight — e.g. two
 * contexts rendering alternate frames, so that the latency-bound last iterations of one frame share the GPU with the
 * first iterations of the next (bench.py --frames-in-flight).  Scenes with subsurface materials still poll the device
 * every few iterations (their iteration count is data dependent).  The synchronous calls read a 16-byte live count per
 * iteration to hand the last paths of a batch to the tail kernel (PTX_TAIL_KERNEL); this call does not, and uses the tail
 * kernel only for batches small enough to fit it from the start. */
int32_t ptx_fb_render_passes_async(PtxContext* ctx, uint32_t first_sample, uint32_t count, uint32_t max_bounces);
/* Two contexts on one GPU rendering alternate frames: everything `ctx` enqueues from now on starts after `other`'s most
 * recent ptx_fb_render_passes* call has got past its heavy iterations (the launches of its first three path-tracing
 * iterations per wavefront batch; PTX_TAIL_GATE_ITER).  The latency-bound last iterations of `other`'s frame then share
 * the GPU with the first iterations of `ctx`'s next frame instead of leaving it idle, without the two frames' heavy
 * launches competing.  No counterpart in the reference; a no-op if `other` has not rendered yet. */
int32_t ptx_order_after_frame_tail(PtxContext* ctx, PtxContext* other);

/* `render_adaptive_pass` (renderloop/parallel.rs:80-123) incl.
 * `get_noisy_pixels` (framebuffer.rs:181-214); returns the work-list length. */
int32_t ptx_fb_adaptive_pass(PtxContext* ctx, const PtxAdaptiveConfig* cfg,
                             uint32_t max_bounces, uint64_t* out_rendered);

/* The two halves of `render_adaptive_pass` for a frame split over several GPUs
 * when `max_samples_per_pass` truncates the noisy list (renderloop/parallel.rs:
 * 96-107: sort all noisy pixels by error, keep the first max): the cut is global,
 * so the ranks' lists have to meet.  ptx_fb_noisy_pairs is `get_noisy_pixels`
 * (framebuffer.rs:181-214) over the pixels this context owns, as host arrays:
 * out_keys[i] orders like the pixel's error (larger = noisier, +inf for a pixel
 * below min_samples), out_pixels[i] = y * width + x.  *out_n is always the
 * count; with buffers that are null or too small nothing else is written and
 * PTX_ERR_INVALID returned.  The caller gathers every rank's pairs, sorts them
 * by key descending and pixel ascending (the reference's stable sort over
 * row-major pixels), keeps the first max and hands each rank its own pixels:
 * ptx_fb_adaptive_pass_pixels traces one sample of each (host array, every
 * index < width * height, no pixel twice; sample_index = the pixel's count,
 * parallel.rs:109-114) and accumulates it.  `render.adaptive_pass_distributed`
 * is that loop; INTEGRATION.md shows it. */
int32_t ptx_fb_noisy_pairs(PtxContext* ctx, const PtxAdaptiveConfig* cfg,
                           uint32_t* out_keys, uint32_t* out_pixels,
                           uint64_t capacity, uint64_t* out_n);
int32_t ptx_fb_adaptive_pass_pixels(PtxContext* ctx, const uint32_t* pixels, uint64_t n,
                                    uint32_t max_bounces, uint64_t* out_rendered);

/* The adaptive loop of the app (app.rs:195-236: `render_adaptive_pass` once per
 * frame until it renders nothing): up to `max_passes` passes, stopping after
 * the first that finds no noisy pixel.  Same sum / sum_sq / counts as calling
 * ptx_fb_adaptive_pass that many times, bit for bit: a pixel's sums change only
 * when it is sampled and sample k of a pixel has `sample_index = its count`
 * (parallel.rs:109-114), so several passes are traced together (32 by default,
 * PTX_ADAPTIVE_BATCH) and replayed per pixel in pass order, dropping the
 * samples of passes the pixel would have sat out.  With counters on, or when
 * max_samples_per_pass truncates the list, it runs one pass at a time.
 * *out_passes = passes that rendered something, *out_rendered = samples added. */
int32_t ptx_fb_adaptive_passes(PtxContext* ctx, const PtxAdaptiveConfig* cfg,
                               uint32_t max_bounces, uint32_t max_passes,
                               uint32_t* out_passes, uint64_t* out_rendered);

/* `min_sample_count` / `max_sample_count` / `total_samples`
 * (framebuffer.rs:132-142) over the pixels this context owns. */
int32_t ptx_fb_stats(PtxContext* ctx, uint32_t* out_min, uint32_t* out_max,
                     uint64_t* out_total);

/* Raw accumulators to host (any pointer may be NULL): sum / sum_sq are
 * width*height*3 floats, counts width*height u32. */
int32_t ptx_fb_read(PtxContext* ctx, float* out_sum, float* out_sum_sq, uint32_t* out_counts);
/* `to_color_buffer` / `get_averaged` (framebuffer.rs:52-59, 216-228). */
int32_t ptx_fb_read_mean(PtxContext* ctx, float* out_mean);
/* `ToneMap::apply_with_exposure` + `linear_to_srgb_u8` (exporter/tonemapping.rs:12-115) over the averaged
 * framebuffer, on the device: width*height*3 bytes leave the GPU instead of 12 per pixel.  tonemap: 0 None, 1 Aces
 * (the window's default with exposure 1.5, window.rs:37-39), 2 Reinhard, 3 Agx (the PNG export's, app.rs:272). */
enum { PTX_TONEMAP_NONE = 0, PTX_TONEMAP_ACES = 1, PTX_TONEMAP_REINHARD = 2, PTX_TONEMAP_AGX = 3 };
int32_t ptx_fb_tonemap_srgb8(PtxContext* ctx, int32_t tonemap, float exposure, uint8_t* out_rgb8);
/* Same accumulators into caller-owned DEVICE memory (e.g. a torch tensor that
 * an RCCL all-gather then combines across GPUs).  Not synchronised: ordered
 * on the library stream; call ptx_sync before another stream reads it. */
int32_t ptx_fb_export_dev(PtxContext* ctx, float* d_sum, float* d_sum_sq, uint32_t* d_counts);

/* The one exchange step of a multi-GPU frame (no counterpart in the reference, which is one process): a rank's own
 * 8x8 tiles (ptx_set_partition) as packed 28-byte records (sum.xyz, sum_sq.xyz, count; 64 records per tile, zero
 * records for pixels outside the image and for the tail when this rank owns one tile fewer than rank 0), written to
 * caller-owned DEVICE memory of ptx_fb_tiles_bytes(width, height, world) bytes — the same size on every rank, so
 * one all-gather (RCCL over xGMI) moves them; ptx_fb_import_tiles_dev then scatters rank `src_rank`'s block into this
 * context's framebuffer, which ends up holding the whole frame, bit-identical to a single-GPU frame.  Both calls are
 * ordered on the library stream and not synchronised (ptx_sync before another stream touches the buffers). */
uint64_t ptx_fb_tiles_bytes(uint32_t width, uint32_t height, uint32_t world);
int32_t ptx_fb_export_tiles_dev(PtxContext* ctx, void* d_out, uint64_t capacity_bytes);
int32_t ptx_fb_import_tiles_dev(PtxContext* ctx, const void* d_in, uint32_t src_rank, uint32_t world);
/* Ordering between the library's own stream and a caller's HIP stream without stalling the host (no counterpart in the
 * reference: one process, no streams).  `stream` is a hipStream_t (NULL = the default stream).
 * ptx_order_after_stream: everything the library enqueues from now on runs after the work ALREADY enqueued on `stream`
 * (e.g. a collective that wrote the buffer ptx_fb_import_tiles_dev is about to read).
 * ptx_order_stream_after: work enqueued on `stream` from now on runs after everything the library has enqueued so far
 * (e.g. a collective that reads the buffer ptx_fb_export_tiles_dev just filled). */
int32_t ptx_order_after_stream(PtxContext* ctx, void* stream);
int32_t ptx_order_stream_after(PtxContext* ctx, void* stream);

/* One whole frame in one call: the `for sample` loop of `run_benchmark`
 * (src/main.rs:123-141) and, with `adaptive` non-NULL, the progressive loop of
 * `App::render_fullres_frame` (raytracer/app.rs:195-236): `spp` uniform passes
 * (sample indices 0..spp-1), then `render_adaptive_pass` until it selects no
 * pixel or any pixel has `max_spp` samples (max_spp = 0: no adaptive phase).
 * Accumulation and variance selection stay on the device; one readback at the
 * end.  out_mean (width*height*3), out_counts (width*height) and out_stats
 * may each be NULL. */
typedef struct PtxFrameStats {
    uint64_t total_samples;        /* framebuffer.rs:140-142 */
    uint32_t min_samples, max_samples;
    uint32_t uniform_passes, adaptive_passes;
    uint64_t adaptive_samples;     /* sum of the adaptive work-list lengths */
} PtxFrameStats;
int32_t ptx_render_frame(PtxContext* ctx, uint32_t width, uint32_t height, uint32_t spp,
                         uint32_t max_bounces, const PtxAdaptiveConfig* adaptive, uint32_t max_spp,
                         float* out_mean, uint32_t* out_counts, PtxFrameStats* out_stats);
int32_t ptx_sync(PtxContext* ctx);

/* ---- parity / measurement --------------------------------------------------- */

/* `Scene::hit` / `Scene::hit_any` (scene.rs:44-50) on a caller's ray batch.
 * mode CLOSEST fills out_hits[n]; mode ANY fills out_any[n] (0/1). */
int32_t ptx_trace_batch(PtxContext* ctx, const PtxRay* rays, uint64_t n, int32_t mode,
                        PtxHitRecord* out_hits, uint8_t* out_any);

/* ptx_trace_batch for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  It has no counterpart in the reference.  Same kernels and same results as ptx_trace_batch, but:
 * `d_rays` and the output are caller-owned DEVICE memory, and the scratch the kernels need (queue cursors, error word,
 * per-lane traversal-stack spill strips) is a file-scope device array of 34.6 MB per GPU inside the library instead of
 * a hipMalloc — the call makes no hipMalloc / hipFree / allocating library call.  At most PTX_NOALLOC_MAX_RAYS rays per
 * call (more is PTX_ERR_INVALID); the launch uses 512 persistent waves, so it is slower than ptx_trace_batch on large
 * batches.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two contexts, two threads)
 * race.  Returns after the results are complete (synchronises the library stream). */
#define PTX_NOALLOC_MAX_RAYS (1u << 24)
int32_t ptx_trace_batch_noalloc(PtxContext* ctx, const PtxRay* d_rays, uint64_t n, int32_t mode,
                                PtxHitRecord* d_out_hits, uint8_t* d_out_any);

/* The primary rays `render_pixel` would generate (renderloop/parallel.rs:31-56,
 * bluenoise.rs:156-172, camera.rs:66-95), for the RNG-seed parity check. */
int32_t ptx_generate_primary_rays(PtxContext* ctx, uint32_t width, uint32_t height,
                                  uint32_t sample_index, PtxRay* out_rays);

/* Test hook: evaluates the kernels' own f32 transcendentals (fn: 0 sin, 1 cos,
 * 2 exp, 3 ln, 4 atan2(x, y), 5 asin, 6/7 sin/cos of the fused sincos) on the
 * device, so tests can pin them against glibc and against the oracle. */
int32_t ptx_debug_math(PtxContext* ctx, int32_t fn, const float* x, const float* y, uint64_t n, float* out);

/* enabled: 0 = off; 1 = the counted kernels walk exactly as the reference does (no forwarded bound, reference order:
 * the events are the reference's own, accel/bvh.rs:308-415, and equal the oracle's); 2 = the closest-hit launches of a
 * frame walk as the uncounted kernels do (forwarded bound, nearest child first, second attempts included), so the events
 * are what the timed kernels execute — the numerator of bench.py's roofline.  Counted kernels are slower. */
int32_t ptx_set_counting(PtxContext* ctx, int32_t enabled);
int32_t ptx_reset_counters(PtxContext* ctx);
int32_t ptx_get_counters(PtxContext* ctx, PtxCounters* out);

/* Which traversal paths the launches since the last ptx_reset_counters exercised (no counterpart in the reference,
 * whose stacks are fixed 64-entry arrays, accel/bvh.rs:320,379): the deepest per-ray stack (all three BVH levels of
 * one ray share one typed stack; recorded by the counted kernels only), the deepest packet stack, and how many rays
 * were handed from a packet whose shared stack overflowed to the per-ray kernel.  A high-water mark above
 * lds_stack_entries / packet_lds_entries means the HBM spill strips were used. */
typedef struct PtxDebugStats {
    uint32_t trace_stack_high_water;
    uint32_t packet_stack_high_water;
    uint64_t packet_redo_rays;
    uint32_t lds_stack_entries, packet_lds_entries, packet_capacity;
    float max_leaf_undershoot;   /* counted kernels: largest (t_enter(leaf box) - t) / t over all triangle hits computed */
    uint32_t bound_forwarding;   /* 1 = closest-hit rays take the world closest_t into instances, 0 = off for this scene */
    float max_leaf_undershoot_excess;   /* the same with the scene's absolute slack taken off first: (t_enter - t - slack) / t, 0 if never positive; what the relative margins must cover */
    /* The wave vote of the counted per-ray kernels since ptx_reset_counters: [0] loop iterations summed over waves,
     * [1..4] iterations in which the NODE / TRI / ENTER / EXIT action ran, [5..8] lanes that took part in it. */
    uint64_t vote_closest[9];
    uint64_t vote_any[9];
    /* counted per-ray kernels: stack entries popped, popped from the per-lane HBM strips (index >= lds_stack_entries),
     * children pushed, pushed to the strips */
    uint64_t stack_ops[4];
    /* rays the nearest-first closest-hit kernels traced a second time, in reference order, because their result could
     * depend on the walk order (DESIGN.md "Nearest first"); fast_order: 1 = those kernels are in use for this scene */
    uint64_t fast_redo_rays;
    uint32_t fast_order;
    /* meshes of the uploaded scene whose triangles fail the shape certificate behind the two shortcuts (a triangle whose
     * edges at v0 meet under less than ~0.11 degrees: Moeller-Trumbore's t is then mostly rounding); every ray whose
     * line meets such a mesh's instance box is walked exactly as the reference walks it (DESIGN.md "Shape certificate") */
    uint32_t uncertified_meshes;
} PtxDebugStats;
int32_t ptx_get_debug_stats(PtxContext* ctx, PtxDebugStats* out);

int32_t ptx_set_profiling(PtxContext* ctx, int32_t enabled);
int32_t ptx_reset_timings(PtxContext* ctx);
int32_t ptx_get_timings(PtxContext* ctx, PtxTimings* out);

#ifdef __cplusplus
}
#endif
#endif /* PTX_CORE_H */
