/* pt_host.h — C ABI of libpt_host.so, the C++ stand-in for the part of the
 * reference that stays on the host (no Rust toolchain exists in this image):
 * TriangleMesh::new + Bvh2 SAH build + Bvh8 collapse (accel/bvh.rs:279-306,
 * 418-814, shape/triangle_mesh.rs:17-54), Blas::build, Instance::new,
 * Tlas::build (accel/mod.rs:27-52, 84-120, 166-192), TextureSky::build_cdfs
 * (sky/texture.rs:67-112), BlueNoiseTexture::generate (bluenoise.rs:9-154),
 * DirectionalLight::new (light.rs:13-26), Camera::new/update_matrix
 * (camera.rs:16-60), and `Scene::new` (scene.rs:22-40).
 *
 * It produces the PtxSceneDesc that ptx_upload_scene consumes.  A real Rust
 * host would fill the same descriptor from its own `Scene`.
 * Nothing here runs on the GPU and nothing here is part of the oracle.
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include "ptx_core.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PthBuilder PthBuilder;
typedef struct PthScene PthScene;

PthBuilder* pth_builder_new(void);
void pth_builder_free(PthBuilder* b);

/* TriangleMesh::new(positions, normals, uvs, indices, material_id) followed by
 * Blas::build(vec![mesh]) (loader/gltf.rs:300-302).  Returns the blas index,
 * or -1 on invalid input.  All arrays are copied. */
int32_t pth_builder_add_mesh(PthBuilder* b,
                             const float* positions, uint32_t n_positions,
                             const float* normals, uint32_t n_normals,
                             const float* uvs, uint32_t n_uvs,
                             const uint32_t* indices, uint32_t n_triangles,
                             uint32_t material_id);
/* instances.push((blas_index, transform)) — column-major Mat4. */
int32_t pth_builder_add_instance(PthBuilder* b, uint32_t blas_index, const float transform[16]);
int32_t pth_builder_add_material(PthBuilder* b, const PtxMaterial* m);
/* Texture::Image { data, width, height } with f32 RGB texels (copied). */
int32_t pth_builder_add_texture(PthBuilder* b, uint32_t width, uint32_t height, const float* texels);
/* TextureSky::new(Texture::solid(rgb)) */
void pth_builder_set_sky_solid(PthBuilder* b, float r, float g, float bl);
/* TextureSky::with_rotation(Texture::Image{..}, rotation) */
void pth_builder_set_sky_image(PthBuilder* b, uint32_t width, uint32_t height,
                               const float* texels, float rotation);
/* Some(DirectionalLight::new(direction, color)) */
void pth_builder_set_sun(PthBuilder* b, const float direction[3], const float color[3]);

/* Scene::new: Tlas::build, sky CDFs, blue-noise tables.  Consumes nothing;
 * the builder can be freed afterwards. */
PthScene* pth_builder_finish(PthBuilder* b);
const PtxSceneDesc* pth_scene_desc(const PthScene* s);
void pth_scene_free(PthScene* s);
/* {triangles, nodes over all BVHs, instances, texel bytes} */
void pth_scene_stats(const PthScene* s, uint64_t out[4]);

/* Material::default() (material.rs:38-65) */
void pth_material_default(PtxMaterial* out);
/* Camera::new(position, look_at, fov_degrees, aspect) (camera.rs:16-60) */
void pth_camera_new(const float position[3], const float look_at[3], float fov_degrees,
                    float aspect_ratio, PtxCamera* out);
/* Same from (position, yaw, pitch) — `update_matrix` after the app moved it. */
void pth_camera_from_angles(const float position[3], float yaw, float pitch, float fov_degrees,
                            float aspect_ratio, float aperture, float focus_distance,
                            PtxCamera* out);

/* BlueNoiseTexture::generate(size, seed) -> size*size*2 floats. */
void pth_bluenoise_generate(uint32_t size, uint64_t seed, float* out_samples);

/* Bvh8::build(primitive_bounds, max_prims_per_leaf) on its own, for tests:
 * bounds = n * {min xyz, max xyz}.  Returns a handle; query with pth_bvh8_view. */
typedef struct PthBvh8 PthBvh8;
PthBvh8* pth_bvh8_build(const float* bounds, uint32_t n, uint32_t max_prims_per_leaf);
void pth_bvh8_view(const PthBvh8* h, PtxBvh8* out);
void pth_bvh8_free(PthBvh8* h);

/* Seeded procedural stand-in scenes (no assets exist: SURVEY.md F4).
 * name: "cornell", "sponza_like", "bistro_like", "instanced_stress",
 *       "helmet_like", "sss_test", "glass_test".
 * detail scales the triangle budget (1.0 = the size SURVEY.md §8(d) names).
 * flags: bit0 = app lighting (procedural HDR sky + sun) instead of the
 *        --benchmark solid sky; bit1 = make ~10% of materials subsurface. */
PthScene* pth_scene_procedural(const char* name, uint64_t seed, float detail, uint32_t flags);
/* Default camera for that scene (eye, look_at, fov). */
int32_t pth_scene_default_camera(const char* name, float aspect_ratio, PtxCamera* out);

#ifdef __cplusplus
}
#endif
#endif
