// pt_host_internal.h — data holders behind libpt_host.so (see include/pt_host.h).
#pragma once
#include "pt_host.h"
#include "pt_hostmath.h"

#include <memory>
#include <utility>
#include <vector>

namespace pth {

struct Aabb {
    V3 mn, mx;
    static Aabb empty() { return Aabb{splat(INF), splat(-INF)}; }
};

// Bvh8 { nodes, primitive_indices } (accel/bvh.rs:266-269)
struct Bvh8Data {
    std::vector<PtxBvh8Node> nodes;
    std::vector<uint32_t> primitive_indices;
};
Bvh8Data bvh8_build(const std::vector<std::pair<uint32_t, Aabb>>& primitive_bounds, uint32_t max_prims_per_leaf);

// TriangleMesh (shape/triangle_mesh.rs:6-14)
struct MeshData {
    std::vector<float> positions, normals, uvs;
    std::vector<uint32_t> indices;
    uint32_t material_id = 0;
    Bvh8Data bvh;
    Aabb bounds = Aabb::empty();
    void build();
};

// Texture (texture.rs:4-12)
struct TextureData {
    uint32_t width = 0, height = 0;
    std::vector<float> texels;  // empty => SolidColor(solid)
    V3 solid{0.5f, 0.5f, 0.5f};
    V3 sample(float u, float v) const;
};

// TextureSky (sky/texture.rs:6-14)
struct SkyData {
    TextureData texture;
    std::vector<float> marginal_cdf, conditional_cdfs;
    uint32_t width = 1, height = 1;
    float total_weight = 0.0f;
    float rotation = 0.0f;
    void build_cdfs();
};

// Scene (scene.rs:12-19) before/after Scene::new
struct SceneData {
    std::vector<std::unique_ptr<MeshData>> meshes;            // blases
    std::vector<std::pair<uint32_t, M4>> instance_data;       // (blas_idx, transform)
    std::vector<PtxMaterial> materials;
    std::vector<std::unique_ptr<TextureData>> textures;
    SkyData sky;
    PtxSun sun{};

    // built by finish()
    std::vector<PtxInstance> instances;
    Bvh8Data tlas;
    std::vector<PtxMesh> mesh_descs;
    std::vector<PtxTexture> tex_descs;
    PtxSceneDesc desc{};

    int32_t add_mesh(std::vector<float> positions, std::vector<float> normals, std::vector<float> uvs,
                     std::vector<uint32_t> indices, uint32_t material_id);
    void set_sun(V3 direction, V3 color);
    void finish();
};

std::vector<float> bluenoise_generate(uint32_t size, uint64_t seed);
void camera_new(V3 position, V3 look_at, float fov, float aspect, PtxCamera* out);

}  // namespace pth

// opaque handle of include/pt_host.h
struct PthScene { std::unique_ptr<pth::SceneData> scene; };
