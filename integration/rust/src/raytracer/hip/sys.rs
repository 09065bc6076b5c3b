//! FFI declarations for libmi355_pt.so — field for field `include/ptx_core.h` (ABI version 1).
//! Source only: not compiled in the repository that ships it (no Rust toolchain there).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::c_char;

use crate::raytracer::accel::bvh::Bvh8Node; // #[repr(C, align(64))], 256 B: passes as-is (accel/bvh.rs:86-98)

pub const PTX_ABI_VERSION: u32 = 1;
pub const PTX_OK: i32 = 0;
pub const PTX_MISS: u32 = 0xFFFF_FFFF;
pub const PTX_TRACE_CLOSEST: i32 = 0;
pub const PTX_TRACE_ANY: i32 = 1;

pub enum PtxContext {}

#[repr(C)]
pub struct PtxBvh8 {
    pub nodes: *const Bvh8Node,
    pub n_nodes: u32,
    pub primitive_indices: *const u32,
    pub n_primitive_indices: u32,
}

#[repr(C)]
pub struct PtxMesh {
    pub positions: *const f32, // Vec<Vec3>::as_ptr() as *const f32 (glam scalar Vec3 = 3 x f32)
    pub n_positions: u32,
    pub normals: *const f32,
    pub n_normals: u32,
    pub uvs: *const f32,
    pub n_uvs: u32,
    pub indices: *const u32, // Vec<[u32; 3]>
    pub n_triangles: u32,
    pub material_id: u32,
    pub bvh: PtxBvh8,
    pub bounds_min: [f32; 3],
    pub bounds_max: [f32; 3],
}

#[repr(C)]
pub struct PtxInstance {
    pub blas_index: u32,
    pub transform: [f32; 16], // Mat4::to_cols_array()
    pub inverse_transform: [f32; 16],
    pub normal_matrix: [f32; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct PtxMaterial {
    pub base_color_texture: i32, // Option<u32> -> -1 = None
    pub metallic_roughness_texture: i32,
    pub normal_texture: i32,
    pub emissive_texture: i32,
    pub base_color_factor: [f32; 4],
    pub roughness_factor: f32,
    pub metallic_factor: f32,
    pub normal_scale: f32,
    pub emissive_factor: [f32; 3],
    pub ior: f32,
    pub transmission: f32,
    pub subsurface: f32,
    pub subsurface_radius: [f32; 3],
    pub subsurface_albedo: [f32; 3],
    pub subsurface_anisotropy: f32,
    pub subsurface_entry_tint: [f32; 3],
    pub subsurface_exit_tint: [f32; 3],
}

#[repr(C)]
pub struct PtxTexture {
    pub width: u32,
    pub height: u32,
    pub texels: *const f32, // Texture::Image data; null => SolidColor(solid)
    pub solid: [f32; 3],
}

#[repr(C)]
pub struct PtxSky {
    pub texture: PtxTexture,
    pub marginal_cdf: *const f32,     // height + 1
    pub conditional_cdfs: *const f32, // height * (width + 1)
    pub width: u32,
    pub height: u32,
    pub total_weight: f32,
    pub rotation: f32,
}

#[repr(C)]
pub struct PtxSun {
    pub enabled: u32,
    pub direction: [f32; 3],
    pub color: [f32; 3],
    pub cos_angle: f32,
    pub pdf: f32,
}

#[repr(C)]
pub struct PtxBlueNoise {
    pub size: u32,
    pub samples: *const f32, // size*size Vec2
}

#[repr(C)]
pub struct PtxSceneDesc {
    pub abi_version: u32,
    pub meshes: *const PtxMesh,
    pub n_meshes: u32,
    pub instances: *const PtxInstance,
    pub n_instances: u32,
    pub tlas: PtxBvh8,
    pub materials: *const PtxMaterial,
    pub n_materials: u32,
    pub textures: *const PtxTexture,
    pub n_textures: u32,
    pub sky: PtxSky,
    pub sun: PtxSun,
    pub blue_noise_pixel: PtxBlueNoise,
    pub blue_noise_lens: PtxBlueNoise,
}

#[repr(C)]
pub struct PtxCamera {
    pub inv_vp_matrix: [f32; 16],
    pub aperture: f32,
    pub focus_distance: f32,
    pub right: [f32; 3],
    pub up: [f32; 3],
}

#[repr(C)]
pub struct PtxRay {
    pub origin: [f32; 3],
    pub direction: [f32; 3],
    pub t_max: f32,
}

#[repr(C)]
pub struct PtxHitRecord {
    pub inst_idx: u32,
    pub tri_idx: u32,
    pub t: f32,
    pub u: f32,
    pub v: f32,
}

#[repr(C)]
pub struct PtxAdaptiveConfig {
    pub min_samples: u32,
    pub variance_threshold: f32,
    pub max_samples_per_pass: u64,
}

#[repr(C)]
#[derive(Default)]
pub struct PtxFrameStats {
    pub total_samples: u64,
    pub min_samples: u32,
    pub max_samples: u32,
    pub uniform_passes: u32,
    pub adaptive_passes: u32,
    pub adaptive_samples: u64,
}

/// Event counts behind the algorithmic bytes: filled by the counted kernels.
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct PtxCounters {
    pub rays_closest: u64,
    pub rays_any: u64,
    pub node_visits: u64,
    pub node_children: u64,
    pub prim_index_reads: u64,
    pub triangle_tests: u64,
    pub instance_entries: u64,
    pub instance_hits: u64,
    pub hitdata_built: u64,
    pub material_fetches: u64,
    pub texture_lookups: u64,
    pub cdf_probes: u64,
    pub sky_pdf_evals: u64,
    pub sky_texel_lookups: u64,
    pub primary_rays: u64,
    pub pixel_samples_written: u64,
    pub sss_steps: u64,
    pub any_node_visits: u64,
    pub any_node_children: u64,
    pub any_prim_index_reads: u64,
    pub any_triangle_tests: u64,
    pub any_instance_entries: u64,
}

/// Per-kernel-class device time since the last reset (HIP events on the library's streams).
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct PtxTimings {
    pub raygen_ms: f64,
    pub extend_ms: f64,
    pub shade_ms: f64,
    pub shadow_ms: f64,
    pub accumulate_ms: f64,
    pub other_ms: f64,
    pub extend_launches: u64,
    pub shade_launches: u64,
    pub shadow_launches: u64,
    pub packet_ms: f64,
    pub packet_launches: u64,
    pub walk_ms: f64,
    pub walk_launches: u64,
    pub tail_ms: f64,
    pub tail_launches: u64,
}

/// Which traversal roads the launches since `ptx_reset_counters` took (stack high-water marks, packet redo, the
/// measured undershoot the shortcuts presume, the wave vote, second attempts, uncertified meshes).
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct PtxDebugStats {
    pub trace_stack_high_water: u32,
    pub packet_stack_high_water: u32,
    pub packet_redo_rays: u64,
    pub lds_stack_entries: u32,
    pub packet_lds_entries: u32,
    pub packet_capacity: u32,
    pub max_leaf_undershoot: f32,
    pub bound_forwarding: u32,
    pub max_leaf_undershoot_excess: f32,
    pub vote_closest: [u64; 9],
    pub vote_any: [u64; 9],
    pub stack_ops: [u64; 4],
    pub fast_redo_rays: u64,
    pub fast_order: u32,
    pub uncertified_meshes: u32,
}

pub const PTX_TONEMAP_NONE: i32 = 0;
pub const PTX_TONEMAP_ACES: i32 = 1;
pub const PTX_TONEMAP_REINHARD: i32 = 2;
pub const PTX_TONEMAP_AGX: i32 = 3;

#[link(name = "mi355_pt")]
extern "C" {
    pub fn ptx_create(device_index: i32, out_ctx: *mut *mut PtxContext) -> i32;
    pub fn ptx_destroy(ctx: *mut PtxContext) -> i32;
    pub fn ptx_last_error(ctx: *const PtxContext) -> *const c_char;
    pub fn ptx_upload_scene(ctx: *mut PtxContext, scene: *const PtxSceneDesc) -> i32;
    pub fn ptx_set_camera(ctx: *mut PtxContext, camera: *const PtxCamera) -> i32;
    pub fn ptx_set_partition(ctx: *mut PtxContext, rank: u32, world: u32) -> i32;
    pub fn ptx_render_pass(ctx: *mut PtxContext, w: u32, h: u32, sample_index: u32, max_bounces: u32, out_rgb: *mut f32) -> i32;
    pub fn ptx_render_first_pass_aov(ctx: *mut PtxContext, w: u32, h: u32, max_bounces: u32, out_rgb: *mut f32,
                                     out_albedo: *mut f32, out_normal: *mut f32) -> i32;
    pub fn ptx_render_pixels(ctx: *mut PtxContext, w: u32, h: u32, xys: *const u32, n: u64, max_bounces: u32, out_rgb: *mut f32) -> i32;
    pub fn ptx_render_pass_unshaded(ctx: *mut PtxContext, w: u32, h: u32, sample_index: u32, out_rgb: *mut f32) -> i32;
    pub fn ptx_fb_reset(ctx: *mut PtxContext, w: u32, h: u32, adaptive: i32) -> i32;
    pub fn ptx_fb_render_passes(ctx: *mut PtxContext, first_sample: u32, count: u32, max_bounces: u32) -> i32;
    /// The same without the final device-error readback: returns once the work is enqueued (errors surface at the next synchronising call).
    pub fn ptx_fb_render_passes_async(ctx: *mut PtxContext, first_sample: u32, count: u32, max_bounces: u32) -> i32;
    pub fn ptx_fb_adaptive_pass(ctx: *mut PtxContext, cfg: *const PtxAdaptiveConfig, max_bounces: u32, out_rendered: *mut u64) -> i32;
    pub fn ptx_fb_adaptive_passes(ctx: *mut PtxContext, cfg: *const PtxAdaptiveConfig, max_bounces: u32, max_passes: u32, out_passes: *mut u32, out_rendered: *mut u64) -> i32;
    /// The two halves of `render_adaptive_pass` for a frame split over several GPUs when `max_samples_per_pass` truncates:
    /// this rank's noisy pixels as (key, pixel) host arrays, and one adaptive sample for an explicit list of pixels.
    pub fn ptx_fb_noisy_pairs(ctx: *mut PtxContext, cfg: *const PtxAdaptiveConfig, out_keys: *mut u32, out_pixels: *mut u32, capacity: u64, out_n: *mut u64) -> i32;
    pub fn ptx_fb_adaptive_pass_pixels(ctx: *mut PtxContext, pixels: *const u32, n: u64, max_bounces: u32, out_rendered: *mut u64) -> i32;
    pub fn ptx_fb_stats(ctx: *mut PtxContext, out_min: *mut u32, out_max: *mut u32, out_total: *mut u64) -> i32;
    pub fn ptx_fb_read(ctx: *mut PtxContext, out_sum: *mut f32, out_sum_sq: *mut f32, out_counts: *mut u32) -> i32;
    pub fn ptx_fb_read_mean(ctx: *mut PtxContext, out_mean: *mut f32) -> i32;
    pub fn ptx_render_frame(ctx: *mut PtxContext, w: u32, h: u32, spp: u32, max_bounces: u32, adaptive: *const PtxAdaptiveConfig,
                            max_spp: u32, out_mean: *mut f32, out_counts: *mut u32, out_stats: *mut PtxFrameStats) -> i32;
    pub fn ptx_sync(ctx: *mut PtxContext) -> i32;
    pub fn ptx_trace_batch(ctx: *mut PtxContext, rays: *const PtxRay, n: u64, mode: i32, out_hits: *mut PtxHitRecord, out_any: *mut u8) -> i32;
    /// `ptx_trace_batch` on caller-owned DEVICE buffers, no allocation inside the call (at most 2^24 rays).
    pub fn ptx_trace_batch_noalloc(ctx: *mut PtxContext, d_rays: *const PtxRay, n: u64, mode: i32, d_out_hits: *mut PtxHitRecord, d_out_any: *mut u8) -> i32;
    /// Multi-GPU exchange: this rank's own tiles as packed records into device memory of `ptx_fb_tiles_bytes` bytes
    /// (one all-gather moves them), and the scatter of another rank's block into this context's framebuffer.
    pub fn ptx_fb_tiles_bytes(width: u32, height: u32, world: u32) -> u64;
    pub fn ptx_fb_export_tiles_dev(ctx: *mut PtxContext, d_out: *mut core::ffi::c_void, capacity_bytes: u64) -> i32;
    pub fn ptx_fb_import_tiles_dev(ctx: *mut PtxContext, d_in: *const core::ffi::c_void, src_rank: u32, world: u32) -> i32;
    /// Two contexts rendering alternate frames on one GPU: `ctx` starts after `other`'s frame is past its heavy iterations.
    pub fn ptx_order_after_frame_tail(ctx: *mut PtxContext, other: *mut PtxContext) -> i32;
    /// Stream ordering without a host stall: the library after `stream`'s enqueued work / `stream` after the library's.
    pub fn ptx_order_after_stream(ctx: *mut PtxContext, stream: *mut core::ffi::c_void) -> i32;
    pub fn ptx_order_stream_after(ctx: *mut PtxContext, stream: *mut core::ffi::c_void) -> i32;
    /// `ToneMap::apply_with_exposure` + `linear_to_srgb_u8` on the device: w*h*3 bytes leave the GPU (exporter/tonemapping.rs:12-115).
    pub fn ptx_fb_tonemap_srgb8(ctx: *mut PtxContext, tonemap: i32, exposure: f32, out_rgb8: *mut u8) -> i32;
    /// The accumulators into caller-owned DEVICE memory, ordered on the library stream.
    pub fn ptx_fb_export_dev(ctx: *mut PtxContext, d_sum: *mut f32, d_sum_sq: *mut f32, d_counts: *mut u32) -> i32;
    /// The primary rays `render_pixel` would generate (renderloop/parallel.rs:31-56).
    pub fn ptx_generate_primary_rays(ctx: *mut PtxContext, width: u32, height: u32, sample_index: u32, out_rays: *mut PtxRay) -> i32;
    pub fn ptx_debug_math(ctx: *mut PtxContext, func: i32, x: *const f32, y: *const f32, n: u64, out: *mut f32) -> i32;
    /// 0 off, 1 the reference's events, 2 the shipped walk's events.
    pub fn ptx_set_counting(ctx: *mut PtxContext, enabled: i32) -> i32;
    pub fn ptx_reset_counters(ctx: *mut PtxContext) -> i32;
    pub fn ptx_get_counters(ctx: *mut PtxContext, out: *mut PtxCounters) -> i32;
    pub fn ptx_get_debug_stats(ctx: *mut PtxContext, out: *mut PtxDebugStats) -> i32;
    pub fn ptx_set_profiling(ctx: *mut PtxContext, enabled: i32) -> i32;
    pub fn ptx_reset_timings(ctx: *mut PtxContext) -> i32;
    pub fn ptx_get_timings(ctx: *mut PtxContext, out: *mut PtxTimings) -> i32;
}
