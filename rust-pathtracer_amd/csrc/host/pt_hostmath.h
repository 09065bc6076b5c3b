// pt_hostmath.h — scalar f32 vector/matrix arithmetic with the operation order
// of glam 0.30 (the reference's math crate, Cargo.toml:12; its sources are not
// in /root/reference, so this restates glam's published formulas: SURVEY.md
// §8(c)).  Host-side only.  Build with -ffp-contract=off: rustc never fuses.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace pth {

static const float INF = std::numeric_limits<float>::infinity();
static const float PI_F = 3.14159265358979323846264338327950288f;  // std::f32::consts::PI

// Rust f32::min / f32::max (IEEE minNum/maxNum as rustc lowers them on x86-64).
static inline float rs_min(float a, float b) { return (a != a) ? b : ((b < a) ? b : a); }
static inline float rs_max(float a, float b) { return (a != a) ? b : ((b > a) ? b : a); }
static inline float rs_clamp(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 splat(float s) { return V3{s, s, s}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
static inline float length(V3 a) { return std::sqrt(dot(a, a)); }
static inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }
static inline V3 vmin(V3 a, V3 b) { return V3{rs_min(a.x, b.x), rs_min(a.y, b.y), rs_min(a.z, b.z)}; }
static inline V3 vmax(V3 a, V3 b) { return V3{rs_max(a.x, b.x), rs_max(a.y, b.y), rs_max(a.z, b.z)}; }
static inline V3 lerp(V3 a, V3 b, float s) { return a * (1.0f - s) + b * s; }

struct V4 { float x, y, z, w; };
static inline V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline V4 operator-(V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
static inline V4 operator*(V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }

// Column-major 4x4, c[0..3] = x_axis..w_axis.
struct M4 {
    V4 c[4];
    static M4 identity() {
        return M4{{V4{1, 0, 0, 0}, V4{0, 1, 0, 0}, V4{0, 0, 1, 0}, V4{0, 0, 0, 1}}};
    }
    static M4 from_array(const float* a) {
        M4 m;
        std::memcpy(&m, a, sizeof(float) * 16);
        return m;
    }
    void to_array(float* a) const { std::memcpy(a, this, sizeof(float) * 16); }
};

// Mat4::mul_vec4: ((x_axis*x + y_axis*y) + z_axis*z) + w_axis*w
static inline V4 mul_vec4(const M4& m, V4 v) {
    V4 r = m.c[0] * v.x;
    r = r + m.c[1] * v.y;
    r = r + m.c[2] * v.z;
    r = r + m.c[3] * v.w;
    return r;
}
static inline M4 mul(const M4& a, const M4& b) {
    return M4{{mul_vec4(a, b.c[0]), mul_vec4(a, b.c[1]), mul_vec4(a, b.c[2]), mul_vec4(a, b.c[3])}};
}
// Mat4::transform_point3: x_axis*x + y_axis*y + z_axis*z + w_axis (xyz)
static inline V3 transform_point3(const M4& m, V3 p) {
    V4 r = m.c[0] * p.x;
    r = m.c[1] * p.y + r;
    r = m.c[2] * p.z + r;
    r = m.c[3] + r;
    return V3{r.x, r.y, r.z};
}
static inline V3 transform_vector3(const M4& m, V3 p) {
    V4 r = m.c[0] * p.x;
    r = m.c[1] * p.y + r;
    r = m.c[2] * p.z + r;
    return V3{r.x, r.y, r.z};
}
static inline M4 transpose(const M4& m) {
    return M4{{V4{m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x}, V4{m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y},
               V4{m.c[0].z, m.c[1].z, m.c[2].z, m.c[3].z}, V4{m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w}}};
}

// Mat4::inverse — cofactor expansion in the order glam's scalar path (and GLM,
// which it derives from) evaluates it.
static inline M4 inverse(const M4& m) {
    const float m00 = m.c[0].x, m01 = m.c[0].y, m02 = m.c[0].z, m03 = m.c[0].w;
    const float m10 = m.c[1].x, m11 = m.c[1].y, m12 = m.c[1].z, m13 = m.c[1].w;
    const float m20 = m.c[2].x, m21 = m.c[2].y, m22 = m.c[2].z, m23 = m.c[2].w;
    const float m30 = m.c[3].x, m31 = m.c[3].y, m32 = m.c[3].z, m33 = m.c[3].w;

    const float coef00 = m22 * m33 - m32 * m23;
    const float coef02 = m12 * m33 - m32 * m13;
    const float coef03 = m12 * m23 - m22 * m13;
    const float coef04 = m21 * m33 - m31 * m23;
    const float coef06 = m11 * m33 - m31 * m13;
    const float coef07 = m11 * m23 - m21 * m13;
    const float coef08 = m21 * m32 - m31 * m22;
    const float coef10 = m11 * m32 - m31 * m12;
    const float coef11 = m11 * m22 - m21 * m12;
    const float coef12 = m20 * m33 - m30 * m23;
    const float coef14 = m10 * m33 - m30 * m13;
    const float coef15 = m10 * m23 - m20 * m13;
    const float coef16 = m20 * m32 - m30 * m22;
    const float coef18 = m10 * m32 - m30 * m12;
    const float coef19 = m10 * m22 - m20 * m12;
    const float coef20 = m20 * m31 - m30 * m21;
    const float coef22 = m10 * m31 - m30 * m11;
    const float coef23 = m10 * m21 - m20 * m11;

    const V4 fac0{coef00, coef00, coef02, coef03};
    const V4 fac1{coef04, coef04, coef06, coef07};
    const V4 fac2{coef08, coef08, coef10, coef11};
    const V4 fac3{coef12, coef12, coef14, coef15};
    const V4 fac4{coef16, coef16, coef18, coef19};
    const V4 fac5{coef20, coef20, coef22, coef23};

    const V4 vec0{m10, m00, m00, m00};
    const V4 vec1{m11, m01, m01, m01};
    const V4 vec2{m12, m02, m02, m02};
    const V4 vec3{m13, m03, m03, m03};

    const V4 inv0 = (vec1 * fac0 - vec2 * fac1) + vec3 * fac2;
    const V4 inv1 = (vec0 * fac0 - vec2 * fac3) + vec3 * fac4;
    const V4 inv2 = (vec0 * fac1 - vec1 * fac3) + vec3 * fac5;
    const V4 inv3 = (vec0 * fac2 - vec1 * fac4) + vec2 * fac5;

    const V4 sign_a{1.0f, -1.0f, 1.0f, -1.0f};
    const V4 sign_b{-1.0f, 1.0f, -1.0f, 1.0f};
    M4 inv{{inv0 * sign_a, inv1 * sign_b, inv2 * sign_a, inv3 * sign_b}};

    const V4 col0{inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x};
    const V4 dot0 = m.c[0] * col0;
    const float dot1 = ((dot0.x + dot0.y) + dot0.z) + dot0.w;
    const float rcp_det = 1.0f / dot1;
    for (int i = 0; i < 4; ++i) inv.c[i] = inv.c[i] * rcp_det;
    return inv;
}

// Mat4::perspective_rh (depth 0..1) and Mat4::look_at_rh.
static inline M4 perspective_rh(float fov_y, float aspect, float z_near, float z_far) {
    const float s = std::sin(0.5f * fov_y), c = std::cos(0.5f * fov_y);
    const float h = c / s;
    const float w = h / aspect;
    const float r = z_far / (z_near - z_far);
    return M4{{V4{w, 0, 0, 0}, V4{0, h, 0, 0}, V4{0, 0, r, -1.0f}, V4{0, 0, r * z_near, 0}}};
}
static inline M4 look_at_rh(V3 eye, V3 center, V3 up) {
    const V3 f = normalize(center - eye);
    const V3 s = normalize(cross(f, up));
    const V3 u = cross(s, f);
    return M4{{V4{s.x, u.x, -f.x, 0}, V4{s.y, u.y, -f.y, 0}, V4{s.z, u.z, -f.z, 0},
               V4{-dot(eye, s), -dot(eye, u), dot(eye, f), 1.0f}}};
}

// Affine helpers for the procedural scenes (Mat4::from_scale_rotation_translation-like).
static inline M4 translation(V3 t) {
    M4 m = M4::identity();
    m.c[3] = V4{t.x, t.y, t.z, 1.0f};
    return m;
}
static inline M4 scaling(V3 s) {
    M4 m = M4::identity();
    m.c[0].x = s.x; m.c[1].y = s.y; m.c[2].z = s.z;
    return m;
}
static inline M4 rotation_axis(V3 axis, float angle) {
    const V3 a = normalize(axis);
    const float s = std::sin(angle), c = std::cos(angle), t = 1.0f - c;
    M4 m = M4::identity();
    m.c[0] = V4{t * a.x * a.x + c, t * a.x * a.y + s * a.z, t * a.x * a.z - s * a.y, 0};
    m.c[1] = V4{t * a.x * a.y - s * a.z, t * a.y * a.y + c, t * a.y * a.z + s * a.x, 0};
    m.c[2] = V4{t * a.x * a.z + s * a.y, t * a.y * a.z - s * a.x, t * a.z * a.z + c, 0};
    return m;
}

}  // namespace pth
