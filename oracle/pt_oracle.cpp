// pt_oracle.cpp — TEST INFRASTRUCTURE: CPU restatement of the reference's hot path.
//
// This file is the parity oracle for the HIP kernels and the "port" CPU
// baseline of bench.py.  Nothing in the product (rust-pathtracer_amd/,
// include/) links, loads or calls it; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg do.
//
// PARITY UNPINNED: the reference (Rust, /root/reference) has no tests, golden
// vectors or fixtures on this path, and no Rust toolchain
// exists here to run it.  Each function below restates the cited reference
// lines; hand-derived known answers pin pieces of it (tests/test_oracle_*.py).
// Third-party arithmetic restated from its published definition: glam 0.30
// (Vec3/Mat4 op order), Rust std f32 -> glibc libm.
//
// Build: g++ -O2 -mavx2 -ffp-contract=off (rustc never contracts to FMA);
//   -DPT_ORACLE_DET_MATH  -> transcendentals from det_math.h (bit-matches the GPU)
//   (default)             -> glibc libm, as a Rust build on this box would call.
#include "../include/ptx_core.h"

#include <immintrin.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <mutex>
#include <optional>
#include <thread>
#include <utility>
#include <vector>
#include <algorithm>

#ifdef PT_ORACLE_DET_MATH
#include "det_math.h"
#endif

namespace {

// ----------------------------------------------------------------- math -------
const float INF = std::numeric_limits<float>::infinity();
const float PI = 3.14159265358979323846264338327950288f;  // std::f32::consts::PI
const float TAU = 6.28318530717958647692528676655900577f;

#ifdef PT_ORACLE_DET_MATH
inline float m_sin(float x) { return detm::sinf_det(x); }
inline float m_cos(float x) { return detm::cosf_det(x); }
inline float m_exp(float x) { return detm::expf_det(x); }
inline float m_ln(float x) { return detm::logf_det(x); }
inline float m_atan2(float y, float x) { return detm::atan2f_det(y, x); }
inline float m_asin(float x) { return detm::asinf_det(x); }
inline float m_pow(float x, float y) { return detm::powf_det(x, y); }
inline float m_log2(float x) { return detm::log2f_det(x); }
#else
inline float m_sin(float x) { return ::sinf(x); }
inline float m_cos(float x) { return ::cosf(x); }
inline float m_exp(float x) { return ::expf(x); }
inline float m_ln(float x) { return ::logf(x); }
inline float m_atan2(float y, float x) { return ::atan2f(y, x); }
inline float m_asin(float x) { return ::asinf(x); }
inline float m_pow(float x, float y) { return ::powf(x, y); }
inline float m_log2(float x) { return ::log2f(x); }
#endif

// Rust f32::min/max/clamp, powi (compiler-rt __powisf2 / LLVM's expansion:
// x^5 = x * (x^2)^2, x^2 = x*x)
inline float f_min(float a, float b) { return (a != a) ? b : ((b < a) ? b : a); }
inline float f_max(float a, float b) { return (a != a) ? b : ((b > a) ? b : a); }
inline float f_clamp(float x, float lo, float hi) { if (x < lo) return lo; if (x > hi) return hi; return x; }
inline float powi5(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x * x4; }
inline float powi2(float x) { return x * x; }
inline float fract(float x) { return x - std::trunc(x); }
inline float rem_euclid1(float x) { float r = std::fmod(x, 1.0f); return (r < 0.0f) ? r + 1.0f : r; }
inline int rem_euclid_i(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }
inline uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline bool is_finite(float f) { return std::isfinite(f); }

struct Vec3 {
    float x, y, z;
    float operator[](size_t i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 v3(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 splat(float s) { return Vec3{s, s, s}; }
const Vec3 ZERO{0, 0, 0}, ONE{1, 1, 1}, AXIS_X{1, 0, 0}, AXIS_Y{0, 1, 0};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator/(Vec3 a, Vec3 b) { return Vec3{a.x / b.x, a.y / b.y, a.z / b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return Vec3{a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return Vec3{s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator/(Vec3 a, float s) { return Vec3{a.x / s, a.y / s, a.z / s}; }
inline Vec3 operator-(Vec3 a) { return Vec3{-a.x, -a.y, -a.z}; }
inline Vec3& operator+=(Vec3& a, Vec3 b) { a = a + b; return a; }
inline Vec3& operator*=(Vec3& a, Vec3 b) { a = a * b; return a; }
inline Vec3& operator/=(Vec3& a, float s) { a = a / s; return a; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline Vec3 cross(Vec3 a, Vec3 b) { return Vec3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length_squared(Vec3 a) { return dot(a, a); }
inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
inline Vec3 normalize(Vec3 a) { return a * (1.0f / length(a)); }
inline Vec3 lerp(Vec3 a, Vec3 b, float s) { return a * (1.0f - s) + b * s; }
inline float max_element(Vec3 a) { return f_max(a.x, f_max(a.y, a.z)); }
inline bool is_finite(Vec3 a) { return is_finite(a.x) && is_finite(a.y) && is_finite(a.z); }
inline Vec3 vclamp(Vec3 a, Vec3 lo, Vec3 hi) {
    return Vec3{f_min(f_max(a.x, lo.x), hi.x), f_min(f_max(a.y, lo.y), hi.y), f_min(f_max(a.z, lo.z), hi.z)};
}
struct Vec2 { float x, y; };

// Mat4 column-major; transform_point3 / transform_vector3 / mul_vec4 (glam)
struct Vec4 { float x, y, z, w; };
inline Vec3 transform_point3(const float* m, Vec3 p) {
    const float rx = ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12];
    const float ry = ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13];
    const float rz = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14];
    return Vec3{rx, ry, rz};
}
inline Vec3 transform_vector3(const float* m, Vec3 p) {
    const float rx = (m[0] * p.x + m[4] * p.y) + m[8] * p.z;
    const float ry = (m[1] * p.x + m[5] * p.y) + m[9] * p.z;
    const float rz = (m[2] * p.x + m[6] * p.y) + m[10] * p.z;
    return Vec3{rx, ry, rz};
}
inline Vec4 mul_vec4(const float* m, Vec4 v) {
    Vec4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}

// ------------------------------------------------------------- counters -------
struct Counters : PtxCounters { Counters() { std::memset(this, 0, sizeof(*this)); } };
thread_local Counters tl_cnt;
thread_local bool tl_in_any = false;   // inside Scene::hit_any: also feed the any_* share
Counters g_cnt;
std::mutex g_cnt_mutex;
void flush_counters() {
    std::lock_guard<std::mutex> lk(g_cnt_mutex);
    uint64_t* g = reinterpret_cast<uint64_t*>(static_cast<PtxCounters*>(&g_cnt));
    uint64_t* t = reinterpret_cast<uint64_t*>(static_cast<PtxCounters*>(&tl_cnt));
    for (size_t i = 0; i < sizeof(PtxCounters) / 8; ++i) { g[i] += t[i]; t[i] = 0; }
}

// ------------------------------------------------------------------ ray --------
// ray.rs:1-37
const float RAY_EPSILON = 0.001f;
struct Ray {
    Vec3 origin, direction, inv_direction;
    float t_min, t_max;
    static Ray make(Vec3 o, Vec3 d) {
        return Ray{o, d, Vec3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z}, RAY_EPSILON, INF};
    }
    Ray with_t_max(float t) const { Ray r = *this; r.t_max = t; return r; }
};
struct HitData {
    float t;
    uint32_t material_id;
    Vec3 normal, geo_normal, tangent, bitangent;
    Vec2 uv;
    // not in the reference's HitData: what its closures know and drop
    uint32_t inst_idx, tri_idx;
    float bary_u, bary_v;
};

// -------------------------------------------------------------- Bvh8 ----------
// accel/bvh.rs:171-223 — the AVX2 8-wide slab test, intrinsic for intrinsic.
inline void node_intersect(const PtxBvh8Node& n, const Ray& ray, float t_max, uint8_t* hit_mask, float distances[8]) {
    const __m256 origin_x = _mm256_set1_ps(ray.origin.x), origin_y = _mm256_set1_ps(ray.origin.y), origin_z = _mm256_set1_ps(ray.origin.z);
    const __m256 inv_dir_x = _mm256_set1_ps(ray.inv_direction.x), inv_dir_y = _mm256_set1_ps(ray.inv_direction.y), inv_dir_z = _mm256_set1_ps(ray.inv_direction.z);
    const __m256 t_min_v = _mm256_set1_ps(ray.t_min), t_max_v = _mm256_set1_ps(t_max);
    const __m256 min_x = _mm256_loadu_ps(n.min_x), max_x = _mm256_loadu_ps(n.max_x);
    const __m256 min_y = _mm256_loadu_ps(n.min_y), max_y = _mm256_loadu_ps(n.max_y);
    const __m256 min_z = _mm256_loadu_ps(n.min_z), max_z = _mm256_loadu_ps(n.max_z);
    const __m256 t1_x = _mm256_mul_ps(_mm256_sub_ps(min_x, origin_x), inv_dir_x);
    const __m256 t2_x = _mm256_mul_ps(_mm256_sub_ps(max_x, origin_x), inv_dir_x);
    const __m256 t1_y = _mm256_mul_ps(_mm256_sub_ps(min_y, origin_y), inv_dir_y);
    const __m256 t2_y = _mm256_mul_ps(_mm256_sub_ps(max_y, origin_y), inv_dir_y);
    const __m256 t1_z = _mm256_mul_ps(_mm256_sub_ps(min_z, origin_z), inv_dir_z);
    const __m256 t2_z = _mm256_mul_ps(_mm256_sub_ps(max_z, origin_z), inv_dir_z);
    const __m256 t_near_x = _mm256_min_ps(t1_x, t2_x), t_far_x = _mm256_max_ps(t1_x, t2_x);
    const __m256 t_near_y = _mm256_min_ps(t1_y, t2_y), t_far_y = _mm256_max_ps(t1_y, t2_y);
    const __m256 t_near_z = _mm256_min_ps(t1_z, t2_z), t_far_z = _mm256_max_ps(t1_z, t2_z);
    const __m256 t_enter = _mm256_max_ps(t_min_v, _mm256_max_ps(t_near_x, _mm256_max_ps(t_near_y, t_near_z)));
    const __m256 t_exit = _mm256_min_ps(t_max_v, _mm256_min_ps(t_far_x, _mm256_min_ps(t_far_y, t_far_z)));
    const __m256 hit_cmp = _mm256_cmp_ps(t_enter, t_exit, _CMP_LE_OQ);
    *hit_mask = (uint8_t)_mm256_movemask_ps(hit_cmp);
    _mm256_storeu_ps(distances, t_enter);
}

const uint32_t LEAF_FLAG = 1u << 31, PRIM_START_MASK = 0x00FFFFFFu, PRIM_COUNT_SHIFT = 24, PRIM_COUNT_MASK = 0x7F;

int g_stack_overflow = 0;  // the reference would panic on its bounds check

// accel/bvh.rs:308-368
template <class T, class F>
std::optional<T> traverse_closest(const PtxBvh8& bvh, const Ray& ray, F&& test_primitive) {
    if (bvh.n_nodes == 0) return std::nullopt;
    std::optional<T> closest;
    float closest_t = ray.t_max;
    std::pair<uint32_t, float> stack[64];
    size_t stack_ptr = 1;
    stack[0] = {0, 0.0f};
    while (stack_ptr > 0) {
        stack_ptr -= 1;
        if (stack[stack_ptr].second >= closest_t) continue;
        const PtxBvh8Node& node = bvh.nodes[stack[stack_ptr].first];
        uint8_t hit_mask;
        float distances[8];
        node_intersect(node, ray, closest_t, &hit_mask, distances);
        tl_cnt.node_visits += 1;
        tl_cnt.node_children += (uint64_t)__builtin_popcount(node.valid_mask);
        const uint8_t active = hit_mask & node.valid_mask;
        if (active == 0) continue;
        uint8_t mask = active;
        while (mask != 0) {
            const size_t slot = (size_t)__builtin_ctz(mask);
            mask &= (uint8_t)(mask - 1);
            const float dist = distances[slot];
            if (dist >= closest_t) continue;
            const uint32_t child = node.children[slot];
            if (child & LEAF_FLAG) {
                const size_t prim_start = child & PRIM_START_MASK;
                const size_t prim_count = ((child >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1;
                for (size_t i = prim_start; i < prim_start + prim_count; ++i) {
                    const uint32_t prim_idx = bvh.primitive_indices[i];
                    tl_cnt.prim_index_reads += 1;
                    std::optional<std::pair<float, T>> r = test_primitive(prim_idx, ray);
                    if (r && r->first < closest_t) { closest_t = r->first; closest = r->second; }
                }
            } else {
                if (stack_ptr >= 64) { g_stack_overflow = 1; return closest; }
                stack[stack_ptr] = {child, dist};
                stack_ptr += 1;
            }
        }
    }
    return closest;
}

// accel/bvh.rs:370-415
template <class F>
bool traverse_any(const PtxBvh8& bvh, const Ray& ray, F&& test_primitive) {
    if (bvh.n_nodes == 0) return false;
    uint32_t stack[64];
    size_t stack_ptr = 1;
    stack[0] = 0;
    while (stack_ptr > 0) {
        stack_ptr -= 1;
        const PtxBvh8Node& node = bvh.nodes[stack[stack_ptr]];
        uint8_t hit_mask;
        float distances[8];
        node_intersect(node, ray, ray.t_max, &hit_mask, distances);
        tl_cnt.node_visits += 1;
        tl_cnt.node_children += (uint64_t)__builtin_popcount(node.valid_mask);
        tl_cnt.any_node_visits += 1;
        tl_cnt.any_node_children += (uint64_t)__builtin_popcount(node.valid_mask);
        const uint8_t active = hit_mask & node.valid_mask;
        if (active == 0) continue;
        uint8_t mask = active;
        while (mask != 0) {
            const size_t slot = (size_t)__builtin_ctz(mask);
            mask &= (uint8_t)(mask - 1);
            const uint32_t child = node.children[slot];
            if (child & LEAF_FLAG) {
                const size_t prim_start = child & PRIM_START_MASK;
                const size_t prim_count = ((child >> PRIM_COUNT_SHIFT) & PRIM_COUNT_MASK) + 1;
                for (size_t i = prim_start; i < prim_start + prim_count; ++i) {
                    tl_cnt.prim_index_reads += 1;
                    tl_cnt.any_prim_index_reads += 1;
                    if (test_primitive(bvh.primitive_indices[i], ray)) return true;
                }
            } else {
                if (stack_ptr >= 64) { g_stack_overflow = 1; return false; }
                stack[stack_ptr] = child;
                stack_ptr += 1;
            }
        }
    }
    return false;
}

// -------------------------------------------------------- TriangleMesh --------
struct TriHit { uint32_t tri_idx; float t; Vec2 bary; };

inline Vec3 load3(const float* p, uint32_t i) { return Vec3{p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]}; }

// shape/triangle_mesh.rs:63-100
inline std::optional<std::pair<float, Vec2>> intersect_triangle(const PtxMesh& m, size_t tri_idx, const Ray& ray) {
    tl_cnt.triangle_tests += 1;
    if (tl_in_any) tl_cnt.any_triangle_tests += 1;
    const uint32_t* tri = &m.indices[3 * tri_idx];
    const Vec3 v0 = load3(m.positions, tri[0]), v1 = load3(m.positions, tri[1]), v2 = load3(m.positions, tri[2]);
    const Vec3 edge1 = v1 - v0, edge2 = v2 - v0;
    const Vec3 h = cross(ray.direction, edge2);
    const float a = dot(edge1, h);
    if (std::fabs(a) < 1e-8f) return std::nullopt;
    const float f = 1.0f / a;
    const Vec3 s = ray.origin - v0;
    const float u = f * dot(s, h);
    if (!(u >= 0.0f && u <= 1.0f)) return std::nullopt;
    const Vec3 q = cross(s, edge1);
    const float v = f * dot(ray.direction, q);
    if (v < 0.0f || u + v > 1.0f) return std::nullopt;
    const float t = f * dot(edge2, q);
    if (t >= ray.t_min && t <= ray.t_max) return std::make_pair(t, Vec2{u, v});
    return std::nullopt;
}

// shape/triangle_mesh.rs:150-159
inline void compute_tangent_space(Vec3 normal, Vec3* tangent, Vec3* bitangent) {
    const Vec3 up = (std::fabs(normal.y) < 0.999f) ? AXIS_Y : AXIS_X;
    *tangent = normalize(cross(up, normal));
    *bitangent = cross(normal, *tangent);
}

// shape/triangle_mesh.rs:102-148
inline HitData compute_hit_data(const PtxMesh& m, size_t tri_idx, float t, Vec2 bary) {
    tl_cnt.hitdata_built += 1;
    const uint32_t* tri = &m.indices[3 * tri_idx];
    const float u = bary.x, v = bary.y;
    const float w = 1.0f - u - v;
    const Vec3 v0 = load3(m.positions, tri[0]), v1 = load3(m.positions, tri[1]), v2 = load3(m.positions, tri[2]);
    const Vec3 face_normal = normalize(cross(v1 - v0, v2 - v0));
    Vec3 normal;
    if (m.n_normals != 0) {
        const Vec3 n0 = load3(m.normals, tri[0]), n1 = load3(m.normals, tri[1]), n2 = load3(m.normals, tri[2]);
        normal = normalize(n0 * w + n1 * u + n2 * v);
    } else {
        normal = face_normal;
    }
    const Vec3 geo_normal = (dot(face_normal, normal) >= 0.0f) ? face_normal : -face_normal;
    Vec2 uv;
    if (m.n_uvs != 0) {
        const float* a = &m.uvs[2 * (size_t)tri[0]]; const float* b = &m.uvs[2 * (size_t)tri[1]]; const float* c = &m.uvs[2 * (size_t)tri[2]];
        uv = Vec2{(a[0] * w + b[0] * u) + c[0] * v, (a[1] * w + b[1] * u) + c[1] * v};
    } else {
        uv = Vec2{u, v};
    }
    HitData h;
    h.t = t; h.material_id = m.material_id; h.normal = normal; h.geo_normal = geo_normal;
    compute_tangent_space(normal, &h.tangent, &h.bitangent);
    h.uv = uv;
    h.inst_idx = PTX_MISS; h.tri_idx = (uint32_t)tri_idx; h.bary_u = u; h.bary_v = v;
    return h;
}

// shape/triangle_mesh.rs:192-207
inline std::optional<HitData> mesh_hit(const PtxMesh& m, const Ray& ray) {
    std::optional<TriHit> r = traverse_closest<TriHit>(m.bvh, ray, [&](uint32_t tri_idx, const Ray& rr) -> std::optional<std::pair<float, TriHit>> {
        auto h = intersect_triangle(m, tri_idx, rr);
        if (!h) return std::nullopt;
        return std::make_pair(h->first, TriHit{tri_idx, h->first, h->second});
    });
    if (!r) return std::nullopt;
    return compute_hit_data(m, r->tri_idx, r->t, r->bary);
}
inline bool mesh_hit_any(const PtxMesh& m, const Ray& ray) {
    return traverse_any(m.bvh, ray, [&](uint32_t tri_idx, const Ray& rr) { return intersect_triangle(m, tri_idx, rr).has_value(); });
}

// ------------------------------------------------------------ Blas ------------
// accel/mod.rs:27-72.  The loader only ever builds Blas::build(vec![mesh]):
// Bvh8::build over one item with MIN_PRIMS_PER_LEAF 4 => one node whose slot 0
// is leaf(start 0, count 1) boxed by the mesh's unpadded bounds.
struct Blas {
    PtxBvh8Node node;
    uint32_t prim_index;
    PtxBvh8 bvh;
    const PtxMesh* mesh;
};
inline std::optional<HitData> blas_hit(const Blas& b, const Ray& ray) {
    return traverse_closest<HitData>(b.bvh, ray, [&](uint32_t, const Ray& rr) -> std::optional<std::pair<float, HitData>> {
        auto h = mesh_hit(*b.mesh, rr);
        if (!h) return std::nullopt;
        return std::make_pair(h->t, *h);
    });
}
inline bool blas_hit_any(const Blas& b, const Ray& ray) {
    return traverse_any(b.bvh, ray, [&](uint32_t, const Ray& rr) { return mesh_hit_any(*b.mesh, rr); });
}

// ---------------------------------------------------------- Instance ----------
// accel/mod.rs:122-146
inline std::optional<HitData> instance_hit(const PtxInstance& inst, const Ray& ray, const Blas& blas) {
    tl_cnt.instance_entries += 1;
    const Vec3 local_origin = transform_point3(inst.inverse_transform, ray.origin);
    const Vec3 local_direction = transform_vector3(inst.inverse_transform, ray.direction);
    const Ray local_ray = Ray::make(local_origin, local_direction);
    std::optional<HitData> hit = blas_hit(blas, local_ray);
    if (!hit) return std::nullopt;
    tl_cnt.instance_hits += 1;
    const Vec3 local_hit_point = local_origin + hit->t * local_direction;
    const Vec3 world_hit_point = transform_point3(inst.transform, local_hit_point);
    hit->t = dot(world_hit_point - ray.origin, ray.direction) / length_squared(ray.direction);
    hit->normal = normalize(transform_vector3(inst.normal_matrix, hit->normal));
    hit->geo_normal = normalize(transform_vector3(inst.normal_matrix, hit->geo_normal));
    return hit;
}
inline bool instance_hit_any(const PtxInstance& inst, const Ray& ray, const Blas& blas) {
    tl_cnt.instance_entries += 1;
    tl_cnt.any_instance_entries += 1;
    const Vec3 local_origin = transform_point3(inst.inverse_transform, ray.origin);
    const Vec3 local_direction = transform_vector3(inst.inverse_transform, ray.direction);
    return blas_hit_any(blas, Ray::make(local_origin, local_direction));
}

// ------------------------------------------------------------- Scene ----------
struct Scene {
    PtxSceneDesc d;          // borrowed arrays: the descriptor must outlive the Scene
    std::vector<Blas> blases;

    // scene.rs:44-50 -> accel/mod.rs:194-210
    std::optional<HitData> hit(const Ray& ray) const {
        tl_cnt.rays_closest += 1;
        return traverse_closest<HitData>(d.tlas, ray, [&](uint32_t inst_idx, const Ray& rr) -> std::optional<std::pair<float, HitData>> {
            const PtxInstance& inst = d.instances[inst_idx];
            auto h = instance_hit(inst, rr, blases[inst.blas_index]);
            if (!h) return std::nullopt;
            h->inst_idx = inst_idx;
            return std::make_pair(h->t, *h);
        });
    }
    bool hit_any(const Ray& ray) const {
        tl_cnt.rays_any += 1;
        tl_in_any = true;
        const bool r = traverse_any(d.tlas, ray, [&](uint32_t inst_idx, const Ray& rr) {
            const PtxInstance& inst = d.instances[inst_idx];
            return instance_hit_any(inst, rr, blases[inst.blas_index]);
        });
        tl_in_any = false;
        return r;
    }
    const PtxMaterial& get_material(uint32_t id) const { tl_cnt.material_fetches += 1; return d.materials[id]; }
    Vec3 sample_texture(uint32_t id, Vec2 uv) const { return texture_sample(d.textures[id], uv, &tl_cnt.texture_lookups); }

    // texture.rs:64-101
    static Vec3 texture_sample(const PtxTexture& t, Vec2 uv, uint64_t* counter) {
        if (!t.texels) return Vec3{t.solid[0], t.solid[1], t.solid[2]};
        *counter += 1;
        const float u = rem_euclid1(uv.x), v = rem_euclid1(uv.y);
        const float px = u * (float)t.width - 0.5f, py = v * (float)t.height - 0.5f;
        const int x0 = (int)std::floor(px), y0 = (int)std::floor(py);
        const int w = (int)t.width, h = (int)t.height;
        const size_t x0w = (size_t)rem_euclid_i(x0, w), y0h = (size_t)rem_euclid_i(y0, h);
        const size_t x1w = (size_t)rem_euclid_i(x0 + 1, w), y1h = (size_t)rem_euclid_i(y0 + 1, h);
        const float fx = px - std::floor(px), fy = py - std::floor(py);
        auto at = [&](size_t yy, size_t xx) { const float* p = &t.texels[(yy * t.width + xx) * 3]; return Vec3{p[0], p[1], p[2]}; };
        const Vec3 c0 = lerp(at(y0h, x0w), at(y0h, x1w), fx);
        const Vec3 c1 = lerp(at(y1h, x0w), at(y1h, x1w), fx);
        return lerp(c0, c1, fy);
    }

    // ---- TextureSky runtime (sky/texture.rs:41-65, 114-211) ----
    Vec3 rotate_direction(Vec3 dir) const {
        if (d.sky.rotation == 0.0f) return dir;
        const float c = m_cos(d.sky.rotation), s = m_sin(d.sky.rotation);
        return Vec3{dir.x * c - dir.z * s, dir.y, dir.x * s + dir.z * c};
    }
    Vec3 unrotate_direction(Vec3 dir) const {
        if (d.sky.rotation == 0.0f) return dir;
        const float c = m_cos(d.sky.rotation), s = m_sin(d.sky.rotation);
        return Vec3{dir.x * c + dir.z * s, dir.y, -dir.x * s + dir.z * c};
    }
    static Vec2 direction_to_uv(Vec3 dir) {
        const float u = 0.5f + m_atan2(dir.z, dir.x) / (2.0f * PI);
        const float v = 0.5f - m_asin(dir.y) / PI;
        return Vec2{u, v};
    }
    static Vec3 uv_to_direction(Vec2 uv) {
        const float phi = (uv.x - 0.5f) * 2.0f * PI;
        const float theta = (0.5f - uv.y) * PI;
        const float cos_theta = m_cos(theta);
        return Vec3{cos_theta * m_sin(phi), m_sin(theta), cos_theta * m_cos(phi)};
    }
    static size_t binary_search(const float* cdf, size_t len, float u) {
        size_t lo = 0, hi = len - 1;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            tl_cnt.cdf_probes += 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        return lo == 0 ? 0 : lo - 1;
    }
    Vec3 sample_sky(Vec3 direction) const {
        const Vec2 uv = direction_to_uv(rotate_direction(direction));
        return texture_sample(d.sky.texture, uv, &tl_cnt.sky_texel_lookups);
    }
    float sky_pdf(Vec3 direction) const {
        if (d.sky.total_weight <= 0.0f) return 1.0f / (4.0f * PI);
        tl_cnt.sky_pdf_evals += 1;
        const Vec2 uv = direction_to_uv(rotate_direction(direction));
        const uint32_t W = d.sky.width, H = d.sky.height;
        // `as u32` saturates (negative / NaN -> 0)
        auto to_u32 = [](float f) -> uint32_t { if (!(f > 0.0f)) return 0u; if (f >= 4294967296.0f) return 0xFFFFFFFFu; return (uint32_t)f; };
        const uint32_t x = std::min(to_u32(uv.x * (float)W), W - 1);
        const uint32_t y = std::min(to_u32(uv.y * (float)H), H - 1);
        const float marginal_pdf = d.sky.marginal_cdf[y + 1] - d.sky.marginal_cdf[y];
        const size_t row_offset = (size_t)y * (W + 1);
        const float conditional_pdf = d.sky.conditional_cdfs[row_offset + x + 1] - d.sky.conditional_cdfs[row_offset + x];
        const float pdf_uv = marginal_pdf * conditional_pdf * (float)(W * H);
        const float sin_theta = f_max(m_sin(PI * uv.y), 1e-10f);
        return pdf_uv / (2.0f * PI * PI * sin_theta);
    }
    struct SkySample { Vec3 direction, radiance; float pdf; };
    SkySample sample_sky_direction(float u1, float u2) const {
        if (d.sky.total_weight <= 0.0f) {
            const float phi = u1 * 2.0f * PI;
            const float cos_theta = 1.0f - 2.0f * u2;
            const float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
            const Vec3 direction{sin_theta * m_cos(phi), cos_theta, sin_theta * m_sin(phi)};
            return SkySample{direction, sample_sky(direction), 1.0f / (4.0f * PI)};
        }
        const uint32_t W = d.sky.width, H = d.sky.height;
        const size_t y = binary_search(d.sky.marginal_cdf, (size_t)H + 1, u1);
        const float v_cdf_lo = d.sky.marginal_cdf[y], v_cdf_hi = d.sky.marginal_cdf[y + 1];
        const float dv = (u1 - v_cdf_lo) / (v_cdf_hi - v_cdf_lo + 1e-10f);
        const float v = ((float)y + dv) / (float)H;
        const float* row_cdf = d.sky.conditional_cdfs + y * (size_t)(W + 1);
        const size_t x = binary_search(row_cdf, (size_t)W + 1, u2);
        const float u_cdf_lo = row_cdf[x], u_cdf_hi = row_cdf[x + 1];
        const float du = (u2 - u_cdf_lo) / (u_cdf_hi - u_cdf_lo + 1e-10f);
        const float u = ((float)x + du) / (float)W;
        const Vec2 uv{u, v};
        const Vec3 direction = unrotate_direction(uv_to_direction(uv));
        const Vec3 radiance = texture_sample(d.sky.texture, uv, &tl_cnt.sky_texel_lookups);
        return SkySample{direction, radiance, sky_pdf(direction)};
    }
};

// ------------------------------------------------------------ brdf.rs ----------
float ggx_d(float n_dot_h, float alpha) {
    n_dot_h = f_max(n_dot_h, 0.0f);
    const float a2 = alpha * alpha;
    const float denom = n_dot_h * n_dot_h * (a2 - 1.0f) + 1.0f;
    return a2 / (PI * denom * denom);
}
float ggx_g1(float n_dot_v, float alpha) {
    const float a2 = alpha * alpha;
    const float denom = n_dot_v + std::sqrt(a2 + (1.0f - a2) * n_dot_v * n_dot_v);
    return (2.0f * n_dot_v) / denom;
}
float ggx_g(float n_dot_l, float n_dot_v, float alpha) { return ggx_g1(n_dot_l, alpha) * ggx_g1(n_dot_v, alpha); }
float pdf_ggx(float n_dot_h, float v_dot_h, float alpha) {
    if (n_dot_h <= 0.0f || v_dot_h <= 0.0f) return 0.0f;
    const float d = ggx_d(n_dot_h, alpha);
    return d * n_dot_h / (4.0f * v_dot_h);
}
float pdf_cosine(float n_dot_l) { return n_dot_l / PI; }
float pdf_brdf(float n_dot_l, float n_dot_h, float v_dot_h, float alpha, float spec_prob) {
    const float pdf_spec = pdf_ggx(n_dot_h, v_dot_h, alpha);
    const float pdf_diff = pdf_cosine(n_dot_l);
    return spec_prob * pdf_spec + (1.0f - spec_prob) * pdf_diff;
}
Vec3 fresnel_schlick(float cos_theta, Vec3 f0) { return f0 + (ONE - f0) * powi5(f_clamp(1.0f - cos_theta, 0.0f, 1.0f)); }
float fresnel_dielectric(float cos_theta, float ior) {
    const float r0 = powi2((1.0f - ior) / (1.0f + ior));
    return r0 + (1.0f - r0) * powi5(f_clamp(1.0f - cos_theta, 0.0f, 1.0f));
}
std::optional<Vec3> refract(Vec3 incident, Vec3 normal, float eta) {
    const float cos_i = -dot(incident, normal);
    const float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
    if (sin2_t > 1.0f) return std::nullopt;
    const float cos_t = std::sqrt(1.0f - sin2_t);
    return eta * incident + (eta * cos_i - cos_t) * normal;
}
Vec3 reflect(Vec3 incident, Vec3 normal) { return normalize(incident - 2.0f * dot(incident, normal) * normal); }
float ggx_directional_albedo(float n_dot_v, float alpha) {
    return 1.0f - powi5(1.0f - n_dot_v) * powi2(1.0f - alpha) * 0.65f;
}
Vec3 multiscatter_compensation(float n_dot_v, float n_dot_l, float alpha, Vec3 f0) {
    const float e_v = ggx_directional_albedo(n_dot_v, alpha);
    const float e_l = ggx_directional_albedo(n_dot_l, alpha);
    const Vec3 f_avg = f0 + (ONE - f0) / 21.0f;
    const Vec3 f_ms = f_avg * f_avg * (1.0f - e_v) * (1.0f - e_l) / (PI * (ONE - f_avg * (1.0f - e_v)));
    return ONE + f_ms / (e_v * e_l + 0.001f);
}
Vec3 evaluate_brdf(Vec3 v, Vec3 l, Vec3 n, float alpha, Vec3 f0, Vec3 diffuse_color) {
    const float n_dot_l = dot(n, l), n_dot_v = dot(n, v);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return ZERO;
    const Vec3 h = normalize(v + l);
    const float n_dot_h = f_max(dot(n, h), 0.001f);
    const float v_dot_h = f_max(dot(v, h), 0.001f);
    const Vec3 f = fresnel_schlick(v_dot_h, f0);
    const float d = ggx_d(n_dot_h, alpha);
    const float g = ggx_g(n_dot_l, n_dot_v, alpha);
    const Vec3 ms = multiscatter_compensation(n_dot_v, n_dot_l, alpha, f0);
    const Vec3 specular = f * d * g / (4.0f * n_dot_l * n_dot_v) * ms;
    const Vec3 diffuse = diffuse_color * (ONE - f) / PI;
    return (specular + diffuse) * n_dot_l;
}

// --------------------------------------------------------- sampling.rs ---------
Vec3 sample_ggx(float u1, float u2, float alpha) {
    const float phi = 2.0f * PI * u1;
    const float cos_theta = std::sqrt((1.0f - u2) / (u2 * (alpha * alpha - 1.0f) + 1.0f));
    const float sin_theta = std::sqrt(f_max(1.0f - cos_theta * cos_theta, 0.0f));
    return Vec3{sin_theta * m_cos(phi), sin_theta * m_sin(phi), cos_theta};
}
Vec3 sample_cosine_hemisphere(float u1, float u2) {
    const float r = std::sqrt(u1);
    const float phi = 2.0f * PI * u2;
    return Vec3{r * m_cos(phi), r * m_sin(phi), std::sqrt(f_max(1.0f - u1, 0.0f))};
}
void build_basis(Vec3 n, Vec3* t, Vec3* b) {
    const Vec3 up = (std::fabs(n.y) < 0.999f) ? AXIS_Y : AXIS_X;
    *t = normalize(cross(up, n));
    *b = cross(n, *t);
}
Vec3 to_world(Vec3 local, Vec3 t, Vec3 b, Vec3 n) { return t * local.x + b * local.y + n * local.z; }
void half_vector_dots(Vec3 v, Vec3 l, Vec3 n, float* n_dot_h, float* v_dot_h) {
    const Vec3 h = normalize(v + l);
    *n_dot_h = f_max(dot(n, h), 0.001f);
    *v_dot_h = f_max(dot(v, h), 0.001f);
}
float mis_weight(float pdf_a, float pdf_b) {
    const float a2 = pdf_a * pdf_a, b2 = pdf_b * pdf_b;
    return a2 / (a2 + b2 + 1e-10f);
}
const float MAX_INV_PDF = 100.0f, MAX_SAMPLE_VALUE = 10.0f;
float safe_inv_pdf(float pdf) { return f_min(1.0f / pdf, MAX_INV_PDF); }
Vec3 clamp_radiance(Vec3 radiance) {
    if (!is_finite(radiance)) return ZERO;
    const float max_component = max_element(radiance);
    if (max_component > MAX_SAMPLE_VALUE) return radiance * (MAX_SAMPLE_VALUE / max_component);
    return radiance;
}
Vec3 fix_shading_normal(Vec3 shading_n, Vec3 geo_n, Vec3 v) {
    if (dot(shading_n, geo_n) < 0.0f) shading_n = -shading_n;
    const float n_dot_v = dot(shading_n, v);
    if (n_dot_v > 0.0f) return shading_n;
    const Vec3 projected = shading_n - v * n_dot_v;
    const float len = length(projected);
    if (len > 1e-6f) return normalize(projected / len * 0.5f + geo_n * 0.5f);
    return geo_n;
}

// -------------------------------------------------------- subsurface.rs --------
// PCG-XSH-RR 64/32 (subsurface.rs:15-35)
struct Rng {
    uint64_t state;
    explicit Rng(uint32_t seed) : state((uint64_t)seed) { next_u32(); }
    uint32_t next_u32() {
        const uint64_t old = state;
        state = old * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
        const uint32_t rot = (uint32_t)(old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    float next() { return (float)next_u32() / 4294967296.0f; }  // u32::MAX as f32 == 2^32
};

const uint32_t SSS_MAX_STEPS = 64;
float sample_scatter_distance(Rng& rng, float sigma_t) { return -m_ln(f_max(rng.next(), 1e-10f)) / sigma_t; }
Vec3 sample_hg_phase(Rng& rng, Vec3 wo, float g) {
    float cos_theta;
    if (std::fabs(g) < 0.001f) {
        cos_theta = 1.0f - 2.0f * rng.next();
    } else {
        const float g2 = g * g;
        const float sqr = (1.0f - g2) / (1.0f - g + 2.0f * g * rng.next());
        cos_theta = f_clamp((1.0f + g2 - sqr * sqr) / (2.0f * g), -1.0f, 1.0f);
    }
    const float sin_theta = std::sqrt(f_max(1.0f - cos_theta * cos_theta, 0.0f));
    const float phi = 2.0f * PI * rng.next();
    Vec3 t, b;
    const Vec3 n = -wo;
    build_basis(n, &t, &b);
    const Vec3 local{sin_theta * m_cos(phi), sin_theta * m_sin(phi), cos_theta};
    return to_world(local, t, b, n);
}
struct SubsurfaceResult { Vec3 exit_point, exit_normal, throughput; };
std::optional<SubsurfaceResult> random_walk_subsurface(Rng& rng, const Scene& scene, Vec3 entry_point, Vec3 entry_dir,
                                                       Vec3 radius, Vec3 albedo, float anisotropy) {
    const Vec3 sigma_t{1.0f / f_max(radius.x, 0.0001f), 1.0f / f_max(radius.y, 0.0001f), 1.0f / f_max(radius.z, 0.0001f)};
    const Vec3 sigma_s = sigma_t * albedo;
    Vec3 pos = entry_point, dir = entry_dir, throughput = ONE;
    for (uint32_t step = 0; step < SSS_MAX_STEPS; ++step) {
        tl_cnt.sss_steps += 1;
        const Vec3 weights = throughput * sigma_s;
        const float sum_weights = weights.x + weights.y + weights.z;
        if (sum_weights < 1e-10f) return std::nullopt;
        const float r = rng.next() * sum_weights;
        const size_t channel = (r < weights.x) ? 0 : ((r < weights.x + weights.y) ? 1 : 2);
        const float sigma_t_c = sigma_t[channel];
        const float dist = sample_scatter_distance(rng, sigma_t_c);
        const Ray internal_ray = Ray::make(pos, dir).with_t_max(dist);
        if (auto hit = scene.hit(internal_ray)) {
            const Vec3 hit_point = pos + dir * hit->t;
            const bool exiting = dot(dir, hit->normal) > 0.0f;
            if (exiting) {
                const float t = hit->t;
                const Vec3 tr{m_exp(-sigma_t.x * t), m_exp(-sigma_t.y * t), m_exp(-sigma_t.z * t)};
                const float pdf_sum = (m_exp(-sigma_t.x * t) * (weights.x / sum_weights) + m_exp(-sigma_t.y * t) * (weights.y / sum_weights)) +
                                      m_exp(-sigma_t.z * t) * (weights.z / sum_weights);
                throughput *= tr / f_max(pdf_sum, 1e-10f);
                return SubsurfaceResult{hit_point, hit->normal, throughput};
            }
            pos = hit_point + dir * RAY_EPSILON;
            continue;
        }
        pos += dir * dist;
        const Vec3 tr{m_exp(-sigma_t.x * dist), m_exp(-sigma_t.y * dist), m_exp(-sigma_t.z * dist)};
        const float pdf_sum = (sigma_t.x * m_exp(-sigma_t.x * dist) * (weights.x / sum_weights) +
                               sigma_t.y * m_exp(-sigma_t.y * dist) * (weights.y / sum_weights)) +
                              sigma_t.z * m_exp(-sigma_t.z * dist) * (weights.z / sum_weights);
        throughput *= tr * sigma_s / f_max(pdf_sum, 1e-10f);
        const float max_tp = max_element(throughput);
        if (max_tp < 0.1f) {
            const float survive = f_min(max_tp * 10.0f, 1.0f);
            if (rng.next() > survive) return std::nullopt;
            throughput /= survive;
        }
        if (!is_finite(throughput) || max_tp > 1000.0f) return std::nullopt;
        dir = sample_hg_phase(rng, dir, anisotropy);
    }
    return std::nullopt;
}

// -------------------------------------------------------- pathtracer.rs --------
struct AovData { Vec3 albedo{0, 0, 0}, normal{0, 0, 0}; };
struct BrdfContext { float alpha; Vec3 f0, diffuse_color; float spec_prob; };
struct SampledMaterial { Vec3 base_color; float roughness, metallic; Vec3 emissive, normal; float transmission, ior; };

// pathtracer.rs:35-87
SampledMaterial sample_material(const PtxMaterial& material, const HitData& hit, const Scene& scene) {
    const Vec3 bcf{material.base_color_factor[0], material.base_color_factor[1], material.base_color_factor[2]};
    const Vec3 base_color = material.base_color_texture >= 0 ? scene.sample_texture((uint32_t)material.base_color_texture, hit.uv) * bcf : bcf;
    float roughness, metallic;
    if (material.metallic_roughness_texture >= 0) {
        const Vec3 s = scene.sample_texture((uint32_t)material.metallic_roughness_texture, hit.uv);
        roughness = s.y * material.roughness_factor;
        metallic = s.z * material.metallic_factor;
    } else {
        roughness = material.roughness_factor;
        metallic = material.metallic_factor;
    }
    const Vec3 ef{material.emissive_factor[0], material.emissive_factor[1], material.emissive_factor[2]};
    const Vec3 emissive = material.emissive_texture >= 0 ? scene.sample_texture((uint32_t)material.emissive_texture, hit.uv) * ef : ef;
    Vec3 normal;
    if (material.normal_texture >= 0) {
        const Vec3 s = scene.sample_texture((uint32_t)material.normal_texture, hit.uv);
        Vec3 tn = normalize(s * 2.0f - ONE);
        tn = normalize(Vec3{tn.x * material.normal_scale, tn.y * material.normal_scale, tn.z});
        normal = normalize(hit.tangent * tn.x + hit.bitangent * tn.y + hit.normal * tn.z);
    } else {
        normal = hit.normal;
    }
    roughness = f_clamp(roughness, 0.04f, 1.0f);
    return SampledMaterial{base_color, roughness, metallic, emissive, normal, material.transmission, material.ior};
}

// pathtracer.rs:89-94
uint32_t ray_seed(const Ray& ray) {
    return f32_bits(ray.origin.x) + f32_bits(ray.origin.y) * 73856093u + f32_bits(ray.origin.z) * 19349663u +
           f32_bits(ray.direction.x) * 83492791u;
}

// pathtracer.rs:350-360
bool russian_roulette(Rng& rng, Vec3& throughput, uint32_t bounce) {
    if (bounce > 2) {
        const float p = f_clamp(max_element(throughput), 0.05f, 0.95f);
        if (rng.next() > p) return false;
        throughput /= p;
    }
    return true;
}
// pathtracer.rs:328-348
void sample_transmission(Rng& rng, Vec3 incident, Vec3 n, bool front_face, float ior, Vec3* new_dir, Vec3* offset_normal) {
    const float eta = front_face ? 1.0f / ior : ior;
    const float cos_theta = f_min(dot(-incident, n), 1.0f);
    const float fresnel = fresnel_dielectric(cos_theta, ior);
    if (rng.next() < fresnel) { *new_dir = reflect(incident, n); *offset_normal = n; return; }
    if (auto refracted = refract(incident, n, eta)) { *new_dir = normalize(*refracted); *offset_normal = -n; }
    else { *new_dir = reflect(incident, n); *offset_normal = n; }
}
// pathtracer.rs:362-370
float compute_spec_prob(float n_dot_v, Vec3 f0, Vec3 diffuse_color) {
    const Vec3 spec_albedo = fresnel_schlick(n_dot_v, f0);
    const Vec3 diff_albedo = diffuse_color * (ONE - spec_albedo);
    const Vec3 lum{0.2126f, 0.7152f, 0.0722f};
    const float spec_lum = dot(spec_albedo, lum), diff_lum = dot(diff_albedo, lum);
    return f_clamp(spec_lum / (spec_lum + diff_lum + 0.001f), 0.1f, 0.9f);
}
// pathtracer.rs:372-400
Vec3 sample_direct_light(const Scene& scene, Vec3 hit_point, Vec3 n, Vec3 geo_n, Vec3 v, const BrdfContext& c, Vec3 throughput,
                         Vec3 light_dir, Vec3 light_color, float light_pdf) {
    const float n_dot_l = dot(n, light_dir);
    if (n_dot_l <= 0.0f) return ZERO;
    const Ray shadow_ray = Ray::make(hit_point + geo_n * RAY_EPSILON, light_dir);
    if (scene.hit_any(shadow_ray)) return ZERO;
    const Vec3 brdf = evaluate_brdf(v, light_dir, n, c.alpha, c.f0, c.diffuse_color);
    float n_dot_h, v_dot_h;
    half_vector_dots(v, light_dir, n, &n_dot_h, &v_dot_h);
    const float brdf_pdf = pdf_brdf(n_dot_l, n_dot_h, v_dot_h, c.alpha, c.spec_prob);
    const float mis = mis_weight(light_pdf, brdf_pdf);
    return throughput * brdf * light_color * mis;
}
// pathtracer.rs:402-435
Vec3 sample_sky_light(Rng& rng, const Scene& scene, Vec3 hit_point, Vec3 n, Vec3 geo_n, Vec3 v, const BrdfContext& c, Vec3 throughput) {
    const float u1 = rng.next();
    const float u2 = rng.next();
    const Scene::SkySample sky_sample = scene.sample_sky_direction(u1, u2);
    const Vec3 l = sky_sample.direction;
    const float n_dot_l = dot(n, l);
    if (n_dot_l <= 0.0f || sky_sample.pdf <= 0.0f) return ZERO;
    const Ray shadow_ray = Ray::make(hit_point + geo_n * RAY_EPSILON, l);
    if (scene.hit_any(shadow_ray)) return ZERO;
    const Vec3 brdf = evaluate_brdf(v, l, n, c.alpha, c.f0, c.diffuse_color);
    float n_dot_h, v_dot_h;
    half_vector_dots(v, l, n, &n_dot_h, &v_dot_h);
    const float brdf_pdf = pdf_brdf(n_dot_l, n_dot_h, v_dot_h, c.alpha, c.spec_prob);
    const float mis = mis_weight(sky_sample.pdf, brdf_pdf);
    const Vec3 sky_contribution = clamp_radiance(sky_sample.radiance * safe_inv_pdf(sky_sample.pdf));
    return throughput * brdf * sky_contribution * mis;
}
// pathtracer.rs:437-479
struct BrdfSample { Vec3 dir, weight; float pdf; };
std::optional<BrdfSample> sample_brdf(Rng& rng, Vec3 v, Vec3 n, Vec3 t, Vec3 b, float n_dot_v, const BrdfContext& c) {
    if (rng.next() < c.spec_prob) {
        const float u1 = rng.next();
        const float u2 = rng.next();
        const Vec3 h = to_world(sample_ggx(u1, u2, c.alpha), t, b, n);
        const Vec3 l = normalize(2.0f * dot(v, h) * h - v);
        const float n_dot_l = dot(n, l);
        if (n_dot_l <= 0.0f) return std::nullopt;
        const float n_dot_h = f_max(dot(n, h), 0.001f);
        const float v_dot_h = f_max(dot(v, h), 0.001f);
        const Vec3 f = fresnel_schlick(v_dot_h, c.f0);
        const float g = ggx_g(n_dot_l, n_dot_v, c.alpha);
        const Vec3 weight = f * g * v_dot_h / (n_dot_v * n_dot_h * c.spec_prob);
        const float pdf = pdf_brdf(n_dot_l, n_dot_h, v_dot_h, c.alpha, c.spec_prob);
        return BrdfSample{l, weight, pdf};
    }
    const float u1 = rng.next();
    const float u2 = rng.next();
    const Vec3 l = to_world(sample_cosine_hemisphere(u1, u2), t, b, n);
    const float n_dot_l = f_max(dot(n, l), 0.001f);
    float n_dot_h, v_dot_h;
    half_vector_dots(v, l, n, &n_dot_h, &v_dot_h);
    const Vec3 f = fresnel_schlick(v_dot_h, c.f0);
    const Vec3 weight = c.diffuse_color * (ONE - f) / (1.0f - c.spec_prob);
    const float pdf = pdf_brdf(n_dot_l, n_dot_h, v_dot_h, c.alpha, c.spec_prob);
    return BrdfSample{l, weight, pdf};
}

// pathtracer.rs:101-326
Vec3 trace_inner(const Ray& initial_ray, const Scene& scene, uint32_t seed, uint32_t max_bounces, AovData* aov_out) {
    Rng rng(seed);
    Vec3 radiance = ZERO, throughput = ONE;
    Ray ray = initial_ray;
    bool first_hit = true;
    float last_brdf_pdf = 0.0f;
    bool last_was_specular = false;
    const bool has_sun = scene.d.sun.enabled != 0;
    const Vec3 sun_dir{scene.d.sun.direction[0], scene.d.sun.direction[1], scene.d.sun.direction[2]};
    const Vec3 sun_color{scene.d.sun.color[0], scene.d.sun.color[1], scene.d.sun.color[2]};

    for (uint32_t bounce = 0; bounce < max_bounces; ++bounce) {
        std::optional<HitData> hit_opt = scene.hit(ray);
        if (!hit_opt) {
            const Vec3 sky_radiance = scene.sample_sky(ray.direction);
            if (bounce == 0) {
                radiance += throughput * sky_radiance;
                if (aov_out && first_hit) { aov_out->albedo = vclamp(sky_radiance, ZERO, ONE); aov_out->normal = -ray.direction; }
            } else if (last_was_specular) {
                radiance += throughput * sky_radiance;
            } else {
                const float sky_pdf = scene.sky_pdf(ray.direction);
                const float mis = mis_weight(last_brdf_pdf, sky_pdf);
                radiance += throughput * sky_radiance * mis;
            }
            break;
        }
        const HitData& hit = *hit_opt;
        const PtxMaterial& material = scene.get_material(hit.material_id);
        const SampledMaterial mat = sample_material(material, hit, scene);

        if (aov_out && first_hit) {
            aov_out->albedo = mat.base_color;
            aov_out->normal = mat.normal * 0.5f + splat(0.5f);
            first_hit = false;
        }

        radiance += throughput * mat.emissive;

        const bool front_face = dot(ray.direction, hit.geo_normal) < 0.0f;
        const Vec3 geo_n = front_face ? hit.geo_normal : -hit.geo_normal;
        const Vec3 v = -normalize(ray.direction);
        const Vec3 shading_n = (dot(mat.normal, geo_n) > 0.0f) ? mat.normal : -mat.normal;
        const Vec3 n = fix_shading_normal(shading_n, geo_n, v);
        const Vec3 hit_point = ray.origin + ray.direction * hit.t;

        if (mat.transmission > 0.0f && rng.next() < mat.transmission) {
            Vec3 new_dir, offset_normal;
            sample_transmission(rng, ray.direction, n, front_face, mat.ior, &new_dir, &offset_normal);
            throughput *= mat.base_color;
            if (!russian_roulette(rng, throughput, bounce)) break;
            ray = Ray::make(hit_point + offset_normal * RAY_EPSILON, new_dir);
            last_was_specular = true;
            continue;
        }

        const float sss_prob = (material.subsurface > 0.0f && front_face) ? material.subsurface : 0.0f;
        if (sss_prob > 0.0f && rng.next() < sss_prob) {
            throughput /= sss_prob;
            const Vec3 radius{material.subsurface_radius[0], material.subsurface_radius[1], material.subsurface_radius[2]};
            const Vec3 albedo{material.subsurface_albedo[0], material.subsurface_albedo[1], material.subsurface_albedo[2]};
            const float cos_theta = f_min(dot(v, n), 1.0f);
            const float fresnel = fresnel_dielectric(cos_theta, material.ior);
            if (rng.next() < fresnel) {
                ray = Ray::make(hit_point + n * RAY_EPSILON, reflect(-v, n));
                last_was_specular = true;
                continue;
            }
            const float eta = 1.0f / material.ior;
            std::optional<Vec3> refracted = refract(-v, n, eta);
            if (!refracted) {
                ray = Ray::make(hit_point + n * RAY_EPSILON, reflect(-v, n));
                last_was_specular = true;
                continue;
            }
            const Vec3 entry_dir = normalize(*refracted);
            throughput *= Vec3{material.subsurface_entry_tint[0], material.subsurface_entry_tint[1], material.subsurface_entry_tint[2]};
            const Vec3 entry_point = hit_point - n * RAY_EPSILON;
            const Ray thickness_ray = Ray::make(entry_point, entry_dir);
            float thickness = 1.0f;
            if (auto back_hit = scene.hit(thickness_ray)) {
                if (dot(entry_dir, back_hit->normal) > 0.0f) thickness = back_hit->t;
            }
            const float thickness_scale = f_clamp(thickness / 1.0f, 0.1f, 2.0f);
            const Vec3 scaled_radius = radius * thickness_scale;
            auto sss = random_walk_subsurface(rng, scene, entry_point, entry_dir, scaled_radius, albedo, material.subsurface_anisotropy);
            if (!sss) break;
            throughput *= mat.base_color * sss->throughput *
                          Vec3{material.subsurface_exit_tint[0], material.subsurface_exit_tint[1], material.subsurface_exit_tint[2]};
            if (!russian_roulette(rng, throughput, bounce)) break;
            const Vec3 exit_n = sss->exit_normal, exit_point = sss->exit_point;
            const Vec3 diffuse_brdf = mat.base_color / PI;
            if (has_sun) {
                const float n_dot_l = dot(exit_n, sun_dir);
                if (n_dot_l > 0.0f) {
                    const Ray shadow_ray = Ray::make(exit_point + exit_n * RAY_EPSILON, sun_dir);
                    if (!scene.hit_any(shadow_ray)) radiance += throughput * diffuse_brdf * sun_color * n_dot_l;
                }
            }
            {
                const float u1 = rng.next();
                const float u2 = rng.next();
                const Scene::SkySample sky_sample = scene.sample_sky_direction(u1, u2);
                const float n_dot_l = dot(exit_n, sky_sample.direction);
                if (n_dot_l > 0.0f && sky_sample.pdf > 0.0f) {
                    const Ray shadow_ray = Ray::make(exit_point + exit_n * RAY_EPSILON, sky_sample.direction);
                    if (!scene.hit_any(shadow_ray)) {
                        const Vec3 sky_contribution = clamp_radiance(sky_sample.radiance * safe_inv_pdf(sky_sample.pdf));
                        radiance += throughput * diffuse_brdf * sky_contribution * n_dot_l;
                    }
                }
            }
            Vec3 exit_t, exit_b;
            build_basis(exit_n, &exit_t, &exit_b);
            const float u1 = rng.next();
            const float u2 = rng.next();
            const Vec3 exit_dir = to_world(sample_cosine_hemisphere(u1, u2), exit_t, exit_b, exit_n);
            ray = Ray::make(exit_point + exit_n * RAY_EPSILON, exit_dir);
            last_was_specular = false;
            last_brdf_pdf = pdf_cosine(f_max(dot(exit_n, exit_dir), 0.001f));
            continue;
        }
        if (sss_prob > 0.0f) throughput /= 1.0f - sss_prob;

        last_was_specular = false;
        const float alpha = mat.roughness * mat.roughness;
        const Vec3 f0 = lerp(splat(0.04f), mat.base_color, mat.metallic);
        const Vec3 diffuse_color = mat.base_color * (1.0f - mat.metallic);
        Vec3 t, b;
        build_basis(n, &t, &b);
        const float n_dot_v = f_max(dot(v, n), 0.001f);
        const float spec_prob = compute_spec_prob(n_dot_v, f0, diffuse_color);
        const BrdfContext ctx{alpha, f0, diffuse_color, spec_prob};

        if (has_sun) radiance += sample_direct_light(scene, hit_point, n, geo_n, v, ctx, throughput, sun_dir, sun_color, scene.d.sun.pdf);
        radiance += sample_sky_light(rng, scene, hit_point, n, geo_n, v, ctx, throughput);

        std::optional<BrdfSample> bs = sample_brdf(rng, v, n, t, b, n_dot_v, ctx);
        if (!bs) break;
        throughput *= bs->weight;
        if (!russian_roulette(rng, throughput, bounce)) break;
        ray = Ray::make(hit_point + geo_n * RAY_EPSILON, bs->dir);
        last_brdf_pdf = bs->pdf;
        if (has_sun && dot(bs->dir, sun_dir) >= scene.d.sun.cos_angle) {
            const float mis = mis_weight(bs->pdf, scene.d.sun.pdf);
            radiance += throughput * sun_color * mis;
        }
    }
    return clamp_radiance(radiance);
}

// pathtracer.rs:482-494
Vec3 pt_render(const Ray& ray, const Scene& scene, uint32_t sample_index, uint32_t max_bounces, AovData* aov) {
    const uint32_t seed = ray_seed(ray) + sample_index * 2654435761u;
    return trace_inner(ray, scene, seed, max_bounces, aov);
}
// renderer/unshaded.rs:7-23
Vec3 unshaded_render(const Ray& ray, const Scene& scene) {
    if (auto hit = scene.hit(ray)) {
        const PtxMaterial& material = scene.get_material(hit->material_id);
        const Vec3 bcf{material.base_color_factor[0], material.base_color_factor[1], material.base_color_factor[2]};
        const Vec3 base_color = material.base_color_texture >= 0 ? scene.sample_texture((uint32_t)material.base_color_texture, hit->uv) * bcf : bcf;
        const float n_dot_l = f_max(dot(hit->normal, -ray.direction), 0.0f);
        return base_color * (0.2f + 0.8f * n_dot_l);
    }
    return scene.sample_sky(ray.direction) * 0.5f;
}

// -------------------------------------------------- ray-gen (R1-R3) ------------
// bluenoise.rs:156-172
Vec2 bluenoise_sample(const PtxBlueNoise& bn, size_t x, size_t y, uint32_t sample_index) {
    const size_t tx = x % bn.size, ty = y % bn.size;
    const size_t idx = ty * bn.size + tx;
    const float bx = bn.samples[2 * idx], by = bn.samples[2 * idx + 1];
    const float R2_A1 = 0.7548777f, R2_A2 = 0.5698403f;
    const float n = (float)sample_index;
    const float offset_x = fract(n * R2_A1), offset_y = fract(n * R2_A2);
    return Vec2{fract(bx + offset_x), fract(by + offset_y)};
}
// camera.rs:66-95
Ray generate_ray_dof(const PtxCamera& cam, float u, float v, float lens_u, float lens_v) {
    const float ndc_x = 2.0f * u - 1.0f, ndc_y = 1.0f - 2.0f * v;
    const Vec4 np = mul_vec4(cam.inv_vp_matrix, Vec4{ndc_x, ndc_y, -1.0f, 1.0f});
    const Vec4 fp = mul_vec4(cam.inv_vp_matrix, Vec4{ndc_x, ndc_y, 1.0f, 1.0f});
    const Vec3 near_point = Vec3{np.x, np.y, np.z} / np.w;
    const Vec3 far_point = Vec3{fp.x, fp.y, fp.z} / fp.w;
    const Vec3 direction = normalize(far_point - near_point);
    if (cam.aperture <= 0.0f) return Ray::make(near_point, direction);
    const Vec3 focus_point = near_point + direction * cam.focus_distance;
    const float r = std::sqrt(lens_u);
    const float theta = lens_v * TAU;
    const float lens_offset_x = r * m_cos(theta) * cam.aperture;
    const float lens_offset_y = r * m_sin(theta) * cam.aperture;
    const Vec3 right{cam.right[0], cam.right[1], cam.right[2]}, up{cam.up[0], cam.up[1], cam.up[2]};
    const Vec3 new_origin = near_point + right * lens_offset_x + up * lens_offset_y;
    return Ray::make(new_origin, normalize(focus_point - new_origin));
}
// renderloop/parallel.rs:31-56
Ray primary_ray(const Scene& scene, const PtxCamera& cam, size_t x, size_t y, size_t width, size_t height, uint32_t sample_index) {
    tl_cnt.primary_rays += 1;
    const Vec2 pixel_sample = bluenoise_sample(scene.d.blue_noise_pixel, x, y, sample_index);
    const Vec2 lens_sample = bluenoise_sample(scene.d.blue_noise_lens, x, y, sample_index);
    const float u = ((float)x + pixel_sample.x) / (float)width;
    const float v = ((float)y + pixel_sample.y) / (float)height;
    return generate_ray_dof(cam, u, v, lens_sample.x, lens_sample.y);
}

// ---------------------------------------------------- rayon stand-in -----------
int g_threads = 0;
template <class F>
void parallel_for(size_t n, F&& body) {
    int nt = g_threads > 0 ? g_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    const size_t chunk = 256;
    if (nt == 1 || n <= chunk) {
        for (size_t i = 0; i < n; ++i) body(i);
        flush_counters();
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([&] {
            for (;;) {
                const size_t b = next.fetch_add(chunk);
                if (b >= n) break;
                const size_t e = std::min(n, b + chunk);
                for (size_t i = b; i < e; ++i) body(i);
            }
            flush_counters();
        });
    for (auto& th : pool) th.join();
    flush_counters();
}

// ------------------------------------------------------- framebuffer.rs --------
struct AdaptiveFramebuffer {
    size_t width, height;
    bool adaptive;
    std::vector<Vec3> sum, sum_sq;
    std::vector<uint32_t> counts;
    uint32_t global_count = 0;  // Framebuffer::sample_count
};

}  // namespace

// =================================================================== C ABI ======
struct OrcScene { Scene scene; };
struct OrcFb { AdaptiveFramebuffer fb; };

static void render_pass_impl(const Scene& scene, const PtxCamera& cam, size_t w, size_t h, uint32_t sample_index,
                             uint32_t max_bounces, float* out_rgb, float* out_albedo, float* out_normal, int mode) {
    parallel_for(w * h, [&](size_t idx) {
        const size_t x = idx % w, y = idx / w;
        const Ray ray = primary_ray(scene, cam, x, y, w, h, sample_index);
        Vec3 c;
        if (mode == 2) c = unshaded_render(ray, scene);
        else if (mode == 1) {
            AovData aov;
            c = pt_render(ray, scene, sample_index, max_bounces, &aov);
            out_albedo[3 * idx] = aov.albedo.x; out_albedo[3 * idx + 1] = aov.albedo.y; out_albedo[3 * idx + 2] = aov.albedo.z;
            out_normal[3 * idx] = aov.normal.x; out_normal[3 * idx + 1] = aov.normal.y; out_normal[3 * idx + 2] = aov.normal.z;
        } else c = pt_render(ray, scene, sample_index, max_bounces, nullptr);
        tl_cnt.pixel_samples_written += 1;
        out_rgb[3 * idx] = c.x; out_rgb[3 * idx + 1] = c.y; out_rgb[3 * idx + 2] = c.z;
    });
}

namespace {
// ------------------------------------------------------------ tone mapping -----
// exporter/tonemapping.rs:12-115.  mode: 0 None, 1 Aces, 2 Reinhard, 3 Agx.
inline Vec3 tm_aces(Vec3 c) {                                   // :27-44
    const Vec3 a0{(c.x * 0.59719f + c.y * 0.35458f) + c.z * 0.04823f,
                  (c.x * 0.07600f + c.y * 0.90834f) + c.z * 0.01566f,
                  (c.x * 0.02840f + c.y * 0.13383f) + c.z * 0.83777f};
    auto rrt = [](float v) {
        const float a = v * (v + 0.0245786f) - 0.000090537f;
        const float b = v * (v * 0.983729f + 0.432951f) + 0.238081f;
        return a / b;
    };
    const Vec3 r{rrt(a0.x), rrt(a0.y), rrt(a0.z)};
    const Vec3 o{(r.x * 1.60475f + r.y * -0.53108f) + r.z * -0.07367f,
                 (r.x * -0.10208f + r.y * 1.10813f) + r.z * -0.00605f,
                 (r.x * -0.00327f + r.y * -0.07276f) + r.z * 1.07602f};
    // glam Vec3::clamp = self.max(min).min(max)
    return Vec3{f_min(f_max(o.x, 0.0f), 1.0f), f_min(f_max(o.y, 0.0f), 1.0f), f_min(f_max(o.z, 0.0f), 1.0f)};
}
inline float tm_agx_contrast(float x) {                         // :50-61, left to right
    const float x2 = x * x;
    const float x4 = x2 * x2;
    return (((((x4 * x2 * 15.5f - x4 * x * 40.14f) + x4 * 31.96f) - x2 * x * 6.868f) + x2 * 0.4298f) + x * 0.1191f) - 0.00232f;
}
inline Vec3 tm_agx(Vec3 v) {                                    // :63-99; glam Mat3 * Vec3 = (x_axis*v.x + y_axis*v.y) + z_axis*v.z
    const float MIN_EV = -12.47393f, MAX_EV = 4.026069f;
    const Vec3 m{(0.842479062253094f * v.x + 0.0423282422610123f * v.y) + 0.0423756549057051f * v.z,
                 (0.0784335999999992f * v.x + 0.878468636469772f * v.y) + 0.0784336f * v.z,
                 (0.0792237451477643f * v.x + 0.0791661274605434f * v.y) + 0.879142973793104f * v.z};
    auto enc = [&](float c) { return (f_clamp(m_log2(f_max(c, 1e-10f)), MIN_EV, MAX_EV) - MIN_EV) / (MAX_EV - MIN_EV); };
    const Vec3 e{tm_agx_contrast(enc(m.x)), tm_agx_contrast(enc(m.y)), tm_agx_contrast(enc(m.z))};
    const Vec3 o{(1.19687900512017f * e.x + -0.0528968517574562f * e.y) + -0.0529716355144438f * e.z,
                 (-0.0980208811401368f * e.x + 1.15190312990417f * e.y) + -0.0980434501171241f * e.z,
                 (-0.0990297440797205f * e.x + -0.0989611768448433f * e.y) + 1.15107367264116f * e.z};
    return Vec3{m_pow(o.x, 2.2f), m_pow(o.y, 2.2f), m_pow(o.z, 2.2f)};
}
inline Vec3 tm_apply(int mode, Vec3 c, float exposure) {        // apply_with_exposure :22-24
    c = Vec3{c.x * exposure, c.y * exposure, c.z * exposure};
    switch (mode) {
        case 1: return tm_aces(c);
        case 2: return Vec3{c.x / (c.x + 1.0f), c.y / (c.y + 1.0f), c.z / (c.z + 1.0f)};
        case 3: return tm_agx(c);
        default: return c;
    }
}
inline uint8_t tm_srgb_u8(float c) {                            // linear_to_srgb_u8 :101-115; `as u8` saturates, NaN -> 0
    c = f_clamp(c, 0.0f, 1.0f);
    const float s = (c <= 0.0031308f) ? 12.92f * c : 1.055f * m_pow(c, 1.0f / 2.4f) - 0.055f;
    const float v = s * 255.0f;
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
}  // namespace

extern "C" {

OrcScene* orc_scene_new(const PtxSceneDesc* desc) {
    if (!desc || desc->abi_version != PTX_ABI_VERSION) return nullptr;
    auto* s = new OrcScene();
    s->scene.d = *desc;
    s->scene.blases.resize(desc->n_meshes);
    for (uint32_t i = 0; i < desc->n_meshes; ++i) {
        Blas& b = s->scene.blases[i];
        std::memset(&b.node, 0, sizeof(b.node));
        for (int k = 0; k < 8; ++k) {
            b.node.min_x[k] = b.node.min_y[k] = b.node.min_z[k] = INF;
            b.node.max_x[k] = b.node.max_y[k] = b.node.max_z[k] = -INF;
        }
        const PtxMesh& m = desc->meshes[i];
        b.node.min_x[0] = m.bounds_min[0]; b.node.min_y[0] = m.bounds_min[1]; b.node.min_z[0] = m.bounds_min[2];
        b.node.max_x[0] = m.bounds_max[0]; b.node.max_y[0] = m.bounds_max[1]; b.node.max_z[0] = m.bounds_max[2];
        b.node.children[0] = LEAF_FLAG | 0u;  // leaf(start 0, count 1)
        b.node.valid_mask = 1;
        b.prim_index = 0;
        b.mesh = &desc->meshes[i];
    }
    for (auto& b : s->scene.blases) { b.bvh.nodes = &b.node; b.bvh.n_nodes = 1; b.bvh.primitive_indices = &b.prim_index; b.bvh.n_primitive_indices = 1; }
    return s;
}
void orc_scene_free(OrcScene* s) { delete s; }
void orc_set_threads(int n) { g_threads = n; }
int orc_is_det_math(void) {
#ifdef PT_ORACLE_DET_MATH
    return 1;
#else
    return 0;
#endif
}
int orc_stack_overflowed(void) { return g_stack_overflow; }

void orc_reset_counters(void) { std::lock_guard<std::mutex> lk(g_cnt_mutex); g_cnt = Counters(); tl_cnt = Counters(); }
void orc_get_counters(PtxCounters* out) { flush_counters(); std::lock_guard<std::mutex> lk(g_cnt_mutex); *out = g_cnt; }

void orc_trace_batch(const OrcScene* s, const PtxRay* rays, uint64_t n, int32_t mode, PtxHitRecord* out_hits, uint8_t* out_any) {
    const Scene& scene = s->scene;
    parallel_for((size_t)n, [&](size_t i) {
        const PtxRay& r = rays[i];
        const Ray ray = Ray::make(Vec3{r.origin[0], r.origin[1], r.origin[2]}, Vec3{r.direction[0], r.direction[1], r.direction[2]}).with_t_max(r.t_max);
        if (mode == PTX_TRACE_ANY) { out_any[i] = scene.hit_any(ray) ? 1 : 0; return; }
        PtxHitRecord rec{PTX_MISS, PTX_MISS, INF, 0.0f, 0.0f};
        if (auto h = scene.hit(ray)) { rec.inst_idx = h->inst_idx; rec.tri_idx = h->tri_idx; rec.t = h->t; rec.u = h->bary_u; rec.v = h->bary_v; }
        out_hits[i] = rec;
    });
}

// full HitData of the closest hit, 16 floats per ray: t, material_id(bits), normal, geo_normal, tangent, bitangent, uv
void orc_trace_hitdata(const OrcScene* s, const PtxRay* rays, uint64_t n, float* out16) {
    const Scene& scene = s->scene;
    parallel_for((size_t)n, [&](size_t i) {
        const PtxRay& r = rays[i];
        const Ray ray = Ray::make(Vec3{r.origin[0], r.origin[1], r.origin[2]}, Vec3{r.direction[0], r.direction[1], r.direction[2]}).with_t_max(r.t_max);
        float* o = out16 + 16 * i;
        for (int k = 0; k < 16; ++k) o[k] = 0.0f;
        o[0] = INF;
        if (auto h = scene.hit(ray)) {
            o[0] = h->t; std::memcpy(&o[1], &h->material_id, 4);
            o[2] = h->normal.x; o[3] = h->normal.y; o[4] = h->normal.z;
            o[5] = h->geo_normal.x; o[6] = h->geo_normal.y; o[7] = h->geo_normal.z;
            o[8] = h->tangent.x; o[9] = h->tangent.y; o[10] = h->tangent.z;
            o[11] = h->bitangent.x; o[12] = h->bitangent.y; o[13] = h->bitangent.z;
            o[14] = h->uv.x; o[15] = h->uv.y;
        }
    });
}

void orc_generate_primary_rays(const OrcScene* s, const PtxCamera* cam, uint32_t w, uint32_t h, uint32_t sample_index, PtxRay* out) {
    parallel_for((size_t)w * h, [&](size_t idx) {
        const Ray r = primary_ray(s->scene, *cam, idx % w, idx / w, w, h, sample_index);
        out[idx] = PtxRay{{r.origin.x, r.origin.y, r.origin.z}, {r.direction.x, r.direction.y, r.direction.z}, r.t_max};
    });
}

void orc_render_pass(const OrcScene* s, const PtxCamera* cam, uint32_t w, uint32_t h, uint32_t sample_index, uint32_t max_bounces, float* out_rgb) {
    render_pass_impl(s->scene, *cam, w, h, sample_index, max_bounces, out_rgb, nullptr, nullptr, 0);
}
void orc_render_first_pass_aov(const OrcScene* s, const PtxCamera* cam, uint32_t w, uint32_t h, uint32_t max_bounces, float* out_rgb, float* out_albedo, float* out_normal) {
    render_pass_impl(s->scene, *cam, w, h, 0, max_bounces, out_rgb, out_albedo, out_normal, 1);
}
void orc_render_pass_unshaded(const OrcScene* s, const PtxCamera* cam, uint32_t w, uint32_t h, uint32_t sample_index, float* out_rgb) {
    render_pass_impl(s->scene, *cam, w, h, sample_index, 0, out_rgb, nullptr, nullptr, 2);
}
// renderloop/parallel.rs:58-78
void orc_render_pixels(const OrcScene* s, const PtxCamera* cam, uint32_t w, uint32_t h, const uint32_t* xys, uint64_t n, uint32_t max_bounces, float* out_rgb) {
    parallel_for((size_t)n, [&](size_t i) {
        const size_t x = xys[3 * i], y = xys[3 * i + 1];
        const uint32_t sample_index = xys[3 * i + 2];
        const Ray ray = primary_ray(s->scene, *cam, x, y, w, h, sample_index);
        const Vec3 c = pt_render(ray, s->scene, sample_index, max_bounces, nullptr);
        tl_cnt.pixel_samples_written += 1;
        out_rgb[3 * i] = c.x; out_rgb[3 * i + 1] = c.y; out_rgb[3 * i + 2] = c.z;
    });
}

// ---- framebuffers (framebuffer.rs) ----
OrcFb* orc_fb_new(uint32_t w, uint32_t h, int32_t adaptive) {
    auto* f = new OrcFb();
    f->fb.width = w; f->fb.height = h; f->fb.adaptive = adaptive != 0;
    f->fb.sum.assign((size_t)w * h, ZERO);
    if (adaptive) f->fb.sum_sq.assign((size_t)w * h, ZERO);
    f->fb.counts.assign((size_t)w * h, 0);
    return f;
}
void orc_fb_free(OrcFb* f) { delete f; }
// Framebuffer::accumulate (framebuffer.rs:44-50) / AdaptiveFramebuffer::accumulate (:118-125)
void orc_fb_accumulate(OrcFb* f, const float* pixels) {
    AdaptiveFramebuffer& fb = f->fb;
    for (size_t i = 0; i < fb.sum.size(); ++i) {
        const Vec3 c{pixels[3 * i], pixels[3 * i + 1], pixels[3 * i + 2]};
        fb.sum[i] += c;
        if (fb.adaptive) fb.sum_sq[i] += c * c;
        fb.counts[i] += 1;
    }
    fb.global_count += 1;
}
// main.rs:124-133
void orc_fb_render_passes(OrcFb* f, const OrcScene* s, const PtxCamera* cam, uint32_t first_sample, uint32_t count, uint32_t max_bounces) {
    std::vector<float> px(f->fb.sum.size() * 3);
    for (uint32_t k = 0; k < count; ++k) {
        orc_render_pass(s, cam, (uint32_t)f->fb.width, (uint32_t)f->fb.height, first_sample + k, max_bounces, px.data());
        orc_fb_accumulate(f, px.data());
    }
}
// framebuffer.rs:181-214 + renderloop/parallel.rs:80-123
uint64_t orc_fb_adaptive_pass(OrcFb* f, const OrcScene* s, const PtxCamera* cam, const PtxAdaptiveConfig* cfg, uint32_t max_bounces) {
    AdaptiveFramebuffer& fb = f->fb;
    struct Noisy { size_t x, y; float err; };
    std::vector<Noisy> noisy;
    for (size_t idx = 0; idx < fb.sum.size(); ++idx) {
        const size_t x = idx % fb.width, y = idx / fb.width;
        const uint32_t count = fb.counts[idx];
        if (count < cfg->min_samples) { noisy.push_back(Noisy{x, y, INF}); continue; }
        const float n = (float)count;
        const Vec3 mean = fb.sum[idx] / n, mean_sq = fb.sum_sq[idx] / n;
        const Vec3 variance = mean_sq - mean * mean;
        const float var_scalar = (f_max(variance.x, 0.0f) + f_max(variance.y, 0.0f) + f_max(variance.z, 0.0f)) / 3.0f;
        const float err = std::sqrt(var_scalar / n);
        if (err > cfg->variance_threshold) noisy.push_back(Noisy{x, y, err});
    }
    if (noisy.empty()) return 0;
    std::stable_sort(noisy.begin(), noisy.end(), [](const Noisy& a, const Noisy& b) { return b.err < a.err; });
    if (noisy.size() > cfg->max_samples_per_pass) noisy.resize((size_t)cfg->max_samples_per_pass);
    std::vector<uint32_t> work(noisy.size() * 3);
    for (size_t i = 0; i < noisy.size(); ++i) {
        work[3 * i] = (uint32_t)noisy[i].x; work[3 * i + 1] = (uint32_t)noisy[i].y;
        work[3 * i + 2] = fb.counts[noisy[i].y * fb.width + noisy[i].x];
    }
    std::vector<float> out(noisy.size() * 3);
    orc_render_pixels(s, cam, (uint32_t)fb.width, (uint32_t)fb.height, work.data(), noisy.size(), max_bounces, out.data());
    for (size_t i = 0; i < noisy.size(); ++i) {  // add_sample (framebuffer.rs:110-116)
        const size_t idx = noisy[i].y * fb.width + noisy[i].x;
        const Vec3 c{out[3 * i], out[3 * i + 1], out[3 * i + 2]};
        fb.sum[idx] += c;
        fb.sum_sq[idx] += c * c;
        fb.counts[idx] += 1;
    }
    return noisy.size();
}
void orc_fb_read(const OrcFb* f, float* out_sum, float* out_sum_sq, uint32_t* out_counts) {
    const AdaptiveFramebuffer& fb = f->fb;
    if (out_sum) std::memcpy(out_sum, fb.sum.data(), fb.sum.size() * sizeof(Vec3));
    if (out_sum_sq && fb.adaptive) std::memcpy(out_sum_sq, fb.sum_sq.data(), fb.sum_sq.size() * sizeof(Vec3));
    if (out_counts) std::memcpy(out_counts, fb.counts.data(), fb.counts.size() * 4);
}
// get_averaged / to_color_buffer (framebuffer.rs:52-59, 216-228)
void orc_fb_read_mean(const OrcFb* f, float* out_mean) {
    const AdaptiveFramebuffer& fb = f->fb;
    for (size_t i = 0; i < fb.sum.size(); ++i) {
        const uint32_t c = fb.counts[i];
        const Vec3 m = c == 0 ? ZERO : fb.sum[i] / (float)c;
        out_mean[3 * i] = m.x; out_mean[3 * i + 1] = m.y; out_mean[3 * i + 2] = m.z;
    }
}

// ---- unit hooks for the known-answer tests ----
void orc_rng_stream(uint32_t seed, uint32_t n, uint32_t* out_u32, float* out_f32) {
    Rng a(seed), b(seed);
    for (uint32_t i = 0; i < n; ++i) { out_u32[i] = a.next_u32(); out_f32[i] = b.next(); }
}
void orc_bluenoise_sample(const OrcScene* s, int which, uint32_t x, uint32_t y, uint32_t sample_index, float out[2]) {
    const Vec2 v = bluenoise_sample(which ? s->scene.d.blue_noise_lens : s->scene.d.blue_noise_pixel, x, y, sample_index);
    out[0] = v.x; out[1] = v.y;
}
float orc_mis_weight(float a, float b) { return mis_weight(a, b); }
float orc_safe_inv_pdf(float p) { return safe_inv_pdf(p); }
void orc_clamp_radiance(const float in[3], float out[3]) { const Vec3 r = clamp_radiance(Vec3{in[0], in[1], in[2]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_evaluate_brdf(const float v[3], const float l[3], const float n[3], float alpha, const float f0[3], const float diffuse[3], float out[3]) {
    const Vec3 r = evaluate_brdf(Vec3{v[0], v[1], v[2]}, Vec3{l[0], l[1], l[2]}, Vec3{n[0], n[1], n[2]}, alpha, Vec3{f0[0], f0[1], f0[2]}, Vec3{diffuse[0], diffuse[1], diffuse[2]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_fresnel_dielectric(float cos_theta, float ior) { return fresnel_dielectric(cos_theta, ior); }
void orc_sky_sample_direction(const OrcScene* s, float u1, float u2, float out[7]) {
    const Scene::SkySample ss = s->scene.sample_sky_direction(u1, u2);
    out[0] = ss.direction.x; out[1] = ss.direction.y; out[2] = ss.direction.z;
    out[3] = ss.radiance.x; out[4] = ss.radiance.y; out[5] = ss.radiance.z; out[6] = ss.pdf;
}
float orc_sky_pdf(const OrcScene* s, const float d[3]) { return s->scene.sky_pdf(Vec3{d[0], d[1], d[2]}); }
void orc_sky_sample(const OrcScene* s, const float d[3], float out[3]) { const Vec3 r = s->scene.sample_sky(Vec3{d[0], d[1], d[2]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_texture_sample(const OrcScene* s, uint32_t tex, float u, float v, float out[3]) {
    const Vec3 r = s->scene.sample_texture(tex, Vec2{u, v}); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// fn: 0 sin, 1 cos, 2 exp, 3 ln, 4 atan2(x,y), 5 asin — through this build's math backend
void orc_math(int fn, const float* x, const float* y, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        switch (fn) {
            case 0: out[i] = m_sin(x[i]); break;
            case 1: out[i] = m_cos(x[i]); break;
            case 2: out[i] = m_exp(x[i]); break;
            case 3: out[i] = m_ln(x[i]); break;
            case 4: out[i] = m_atan2(x[i], y[i]); break;
            case 6: out[i] = m_pow(x[i], y[i]); break;
            case 7: out[i] = m_log2(x[i]); break;
            default: out[i] = m_asin(x[i]); break;
        }
    }
}
// ToneMap::apply_with_exposure + linear_to_srgb_u8 over n pixels (exporter/tonemapping.rs:22-24, 101-115)
void orc_tonemap_srgb8(int mode, float exposure, const float* rgb, uint64_t n, uint8_t* out) {
    for (uint64_t i = 0; i < n; ++i) {
        const Vec3 c = tm_apply(mode, Vec3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]}, exposure);
        out[3 * i] = tm_srgb_u8(c.x); out[3 * i + 1] = tm_srgb_u8(c.y); out[3 * i + 2] = tm_srgb_u8(c.z);
    }
}
void orc_tonemap(int mode, float exposure, const float* rgb, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        const Vec3 c = tm_apply(mode, Vec3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]}, exposure);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}
// one path from an explicit ray + seed (trace_inner), for targeted tests
void orc_trace_path(const OrcScene* s, const PtxRay* ray, uint32_t seed, uint32_t max_bounces, float out[3]) {
    const Ray r = Ray::make(Vec3{ray->origin[0], ray->origin[1], ray->origin[2]}, Vec3{ray->direction[0], ray->direction[1], ray->direction[2]});
    const Vec3 c = trace_inner(r, s->scene, seed, max_bounces, nullptr);
    flush_counters();
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

}  // extern "C"
