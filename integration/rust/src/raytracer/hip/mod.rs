//! `HipRenderLoop`: the reference's `RenderLoop` (renderloop/mod.rs:9-22) and `ParallelRenderLoop`'s inherent methods
//! (renderloop/parallel.rs:58-168) on libmi355_pt.so.  Source only — see ../../../README.md.
//!
//! Accessors this file needs that are private today (add as `pub(crate) fn`, nothing else changes):
//!   TriangleMesh::{positions, normals, uvs, indices, material_id, bvh, bounds}   shape/triangle_mesh.rs:6-14
//!   Blas::items                                                                   accel/mod.rs:20-24
//!   Tlas::{bvh8, instances, blases}, Instance::normal_matrix                      accel/mod.rs:75-81,159-163
//!   TextureSky::{texture, marginal_cdf, conditional_cdfs, width, height, total_weight, rotation}   sky/texture.rs:6-14
//!   Camera::inv_vp_matrix                                                         camera.rs:12
//!   renderloop::parallel::{pixel_noise(), lens_noise()} (the two OnceLock<BlueNoiseTexture>)   parallel.rs:19-28
//!   PathTracer::<N>::MAX_BOUNCES as an associated const on `Renderer`             renderer/pathtracer.rs
pub mod sys;

use std::ffi::CStr;

use glam::Vec3;

use crate::raytracer::camera::Camera;
use crate::raytracer::framebuffer::AdaptiveFramebuffer;
use crate::raytracer::renderer::{AovBuffers, Renderer};
use crate::raytracer::renderloop::parallel::{lens_noise, pixel_noise, AdaptiveSamplingConfig};
use crate::raytracer::renderloop::RenderLoop;
use crate::raytracer::scene::{Scene, SceneAccess};
use crate::raytracer::shape::TriangleMesh;
use crate::raytracer::sky::TextureSky;
use crate::raytracer::texture::Texture;

pub struct HipRenderLoop {
    ctx: *mut sys::PtxContext,
}

// The context is driven by one caller at a time, like ParallelRenderLoop's callers do (main thread).
unsafe impl Send for HipRenderLoop {}
unsafe impl Sync for HipRenderLoop {}

fn last_error(ctx: *const sys::PtxContext) -> String {
    unsafe { CStr::from_ptr(sys::ptx_last_error(ctx)).to_string_lossy().into_owned() }
}

fn check(status: i32, ctx: *const sys::PtxContext) -> Result<(), String> {
    if status == sys::PTX_OK { Ok(()) } else { Err(last_error(ctx)) }
}

fn texture_view(t: &Texture) -> sys::PtxTexture {
    match t {
        Texture::SolidColor(c) => sys::PtxTexture { width: 0, height: 0, texels: std::ptr::null(), solid: c.to_array() },
        Texture::Image { data, width, height } => {
            sys::PtxTexture { width: *width, height: *height, texels: data.as_ptr() as *const f32, solid: [0.0; 3] }
        }
    }
}

impl HipRenderLoop {
    /// Copies the scene into HBM once (`Scene::new`'s result, scene.rs:22-40); every borrowed vector may drop afterwards.
    pub fn new(scene: &Scene<TriangleMesh, TextureSky>) -> Result<Self, String> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { sys::ptx_create(0, &mut ctx) }, std::ptr::null())?;
        let tlas = &scene.tlas;
        let meshes: Vec<sys::PtxMesh> = tlas.blases().iter().map(|blas| {
            let m = &blas.items()[0]; // one Blas = one TriangleMesh (loader/gltf.rs:300-302)
            sys::PtxMesh {
                positions: m.positions().as_ptr() as *const f32, n_positions: m.positions().len() as u32,
                normals: m.normals().as_ptr() as *const f32, n_normals: m.normals().len() as u32,
                uvs: m.uvs().as_ptr() as *const f32, n_uvs: m.uvs().len() as u32,
                indices: m.indices().as_ptr() as *const u32, n_triangles: m.indices().len() as u32,
                material_id: m.material_id(),
                bvh: sys::PtxBvh8 {
                    nodes: m.bvh().nodes.as_ptr(), n_nodes: m.bvh().nodes.len() as u32,
                    primitive_indices: m.bvh().primitive_indices.as_ptr(), n_primitive_indices: m.bvh().primitive_indices.len() as u32,
                },
                bounds_min: m.bounds().min.to_array(), bounds_max: m.bounds().max.to_array(),
            }
        }).collect();
        let instances: Vec<sys::PtxInstance> = tlas.instances().iter().map(|i| sys::PtxInstance {
            blas_index: i.blas_index,
            transform: i.transform.to_cols_array(),
            inverse_transform: i.inverse_transform.to_cols_array(),
            normal_matrix: i.normal_matrix().to_cols_array(),
        }).collect();
        let opt = |t: Option<u32>| t.map_or(-1, |v| v as i32);
        let materials: Vec<sys::PtxMaterial> = scene.materials.iter().map(|m| sys::PtxMaterial {
            base_color_texture: opt(m.base_color_texture), metallic_roughness_texture: opt(m.metallic_roughness_texture),
            normal_texture: opt(m.normal_texture), emissive_texture: opt(m.emissive_texture),
            base_color_factor: m.base_color_factor.to_array(), roughness_factor: m.roughness_factor,
            metallic_factor: m.metallic_factor, normal_scale: m.normal_scale, emissive_factor: m.emissive_factor.to_array(),
            ior: m.ior, transmission: m.transmission, subsurface: m.subsurface,
            subsurface_radius: m.subsurface_radius.to_array(), subsurface_albedo: m.subsurface_albedo.to_array(),
            subsurface_anisotropy: m.subsurface_anisotropy,
            subsurface_entry_tint: m.subsurface_entry_tint.to_array(), subsurface_exit_tint: m.subsurface_exit_tint.to_array(),
        }).collect();
        let textures: Vec<sys::PtxTexture> = scene.textures.iter().map(texture_view).collect();
        let sky = &scene.sky;
        let (pn, ln) = (pixel_noise(), lens_noise());
        let desc = sys::PtxSceneDesc {
            abi_version: sys::PTX_ABI_VERSION,
            meshes: meshes.as_ptr(), n_meshes: meshes.len() as u32,
            instances: instances.as_ptr(), n_instances: instances.len() as u32,
            tlas: sys::PtxBvh8 {
                nodes: tlas.bvh8().nodes.as_ptr(), n_nodes: tlas.bvh8().nodes.len() as u32,
                primitive_indices: tlas.bvh8().primitive_indices.as_ptr(),
                n_primitive_indices: tlas.bvh8().primitive_indices.len() as u32,
            },
            materials: materials.as_ptr(), n_materials: materials.len() as u32,
            textures: textures.as_ptr(), n_textures: textures.len() as u32,
            sky: sys::PtxSky {
                texture: texture_view(sky.texture()),
                marginal_cdf: sky.marginal_cdf().as_ptr(), conditional_cdfs: sky.conditional_cdfs().as_ptr(),
                width: sky.width(), height: sky.height(), total_weight: sky.total_weight(), rotation: sky.rotation(),
            },
            sun: match &scene.sun {
                Some(s) => sys::PtxSun { enabled: 1, direction: s.direction.to_array(), color: s.color.to_array(), cos_angle: s.cos_angle, pdf: s.pdf },
                None => sys::PtxSun { enabled: 0, direction: [0.0; 3], color: [0.0; 3], cos_angle: 0.0, pdf: 0.0 },
            },
            blue_noise_pixel: sys::PtxBlueNoise { size: pn.size as u32, samples: pn.samples.as_ptr() as *const f32 },
            blue_noise_lens: sys::PtxBlueNoise { size: ln.size as u32, samples: ln.samples.as_ptr() as *const f32 },
        };
        let status = unsafe { sys::ptx_upload_scene(ctx, &desc) };
        if let Err(e) = check(status, ctx) {
            unsafe { sys::ptx_destroy(ctx) };
            return Err(e);
        }
        Ok(Self { ctx })
    }

    fn set_camera(&self, c: &Camera) {
        let right = c.right();
        let cam = sys::PtxCamera {
            inv_vp_matrix: c.inv_vp_matrix().to_cols_array(),
            aperture: c.aperture, focus_distance: c.focus_distance,
            right: right.to_array(), up: right.cross(c.forward()).to_array(), // camera.rs:79-80
        };
        let st = unsafe { sys::ptx_set_camera(self.ctx, &cam) };
        assert_eq!(st, sys::PTX_OK, "{}", last_error(self.ctx));
    }

    /// `render_first_pass_with_aov` (parallel.rs:125-168).
    pub fn render_first_pass_with_aov<R: Renderer>(&self, camera: &Camera, width: usize, height: usize) -> (Vec<Vec3>, AovBuffers) {
        self.set_camera(camera);
        let n = width * height;
        let (mut rgb, mut albedo, mut normal) = (vec![Vec3::ZERO; n], vec![Vec3::ZERO; n], vec![Vec3::ZERO; n]);
        let st = unsafe {
            sys::ptx_render_first_pass_aov(self.ctx, width as u32, height as u32, R::MAX_BOUNCES, rgb.as_mut_ptr() as *mut f32,
                                           albedo.as_mut_ptr() as *mut f32, normal.as_mut_ptr() as *mut f32)
        };
        assert_eq!(st, sys::PTX_OK, "{}", last_error(self.ctx));
        (rgb, AovBuffers { albedo, normal })
    }

    /// `render_adaptive_pass` (parallel.rs:80-123) with the framebuffer kept on the device: the host
    /// `AdaptiveFramebuffer` is refreshed from `ptx_fb_read` only when the caller needs it (display, export).
    pub fn render_adaptive_pass<R: Renderer>(&self, camera: &Camera, config: &AdaptiveSamplingConfig, pixels: usize) -> usize {
        self.set_camera(camera);
        let cfg = sys::PtxAdaptiveConfig {
            min_samples: config.min_samples, variance_threshold: config.variance_threshold,
            max_samples_per_pass: config.max_samples_per_pass.unwrap_or(pixels) as u64,
        };
        let mut rendered = 0u64;
        let st = unsafe { sys::ptx_fb_adaptive_pass(self.ctx, &cfg, R::MAX_BOUNCES, &mut rendered) };
        assert_eq!(st, sys::PTX_OK, "{}", last_error(self.ctx));
        rendered as usize
    }

    /// Up to `max_passes` adaptive passes between two redraws of the app (app.rs:195-236), traced together on the
    /// device and replayed per pixel: same accumulators as that many `render_adaptive_pass` calls.
    /// Returns (passes that rendered something, samples added); (0, 0) means the frame has converged.
    pub fn render_adaptive_passes<R: Renderer>(&self, camera: &Camera, config: &AdaptiveSamplingConfig, pixels: usize,
                                               max_passes: u32) -> (u32, usize) {
        self.set_camera(camera);
        let cfg = sys::PtxAdaptiveConfig {
            min_samples: config.min_samples, variance_threshold: config.variance_threshold,
            max_samples_per_pass: config.max_samples_per_pass.unwrap_or(pixels) as u64,
        };
        let (mut passes, mut rendered) = (0u32, 0u64);
        let st = unsafe { sys::ptx_fb_adaptive_passes(self.ctx, &cfg, R::MAX_BOUNCES, max_passes, &mut passes, &mut rendered) };
        assert_eq!(st, sys::PTX_OK, "{}", last_error(self.ctx));
        (passes, rendered as usize)
    }

    /// Pull the device accumulators into the host framebuffer (framebuffer.rs:84-142 fields).
    pub fn read_framebuffer(&self, fb: &mut AdaptiveFramebuffer) {
        let st = unsafe {
            sys::ptx_fb_read(self.ctx, fb.sum_mut().as_mut_ptr() as *mut f32, fb.sum_sq_mut().as_mut_ptr() as *mut f32,
                             fb.sample_counts_mut().as_mut_ptr())
        };
        assert_eq!(st, sys::PTX_OK, "{}", last_error(self.ctx));
    }

    /// The whole `for sample in 0..spp` loop of run_benchmark (main.rs:123-141) in one call.
    pub fn render_frame<R: Renderer>(&self, camera: &Camera, width: usize, height: usize, spp: u32) -> Vec<Vec3> {
        self.set_camera(camera);
        let mut mean = vec![Vec3::ZERO; width * height];
        let st = unsafe {
            sys::ptx_render_frame(self.ctx, width as u32, height as u32, spp, R::MAX_BOUNCES, std::ptr::null(), 0,
                                  mean.as_mut_ptr() as *mut f32, std::ptr::null_mut(), std::ptr::null_mut())
        };
        assert_eq!(st, sys::PTX_OK, "{}", last_error(self.ctx));
        mean
    }
}

impl RenderLoop for HipRenderLoop {
    /// The scene argument is the one uploaded in `new`; the trait is infallible (the reference panics on bugs), so a
    /// failing status panics with the library's message.
    fn render_pass<S, R>(&self, _scene: &S, camera: &Camera, _renderer: &R, width: usize, height: usize, sample_index: u32) -> Vec<Vec3>
    where
        S: SceneAccess + Sync,
        R: Renderer + Sync,
    {
        self.set_camera(camera);
        let mut out = vec![Vec3::ZERO; width * height];
        let st = unsafe {
            sys::ptx_render_pass(self.ctx, width as u32, height as u32, sample_index, R::MAX_BOUNCES, out.as_mut_ptr() as *mut f32)
        };
        assert_eq!(st, sys::PTX_OK, "{}", last_error(self.ctx));
        out
    }
}

impl Drop for HipRenderLoop {
    fn drop(&mut self) {
        unsafe { sys::ptx_destroy(self.ctx) };
    }
}
