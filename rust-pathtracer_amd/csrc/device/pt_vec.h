// pt_vec.h — f32 vector arithmetic for the gfx950 kernels.
//
// Every operation is spelled out in the order glam 0.30 evaluates it (the
// reference's math crate: dot = x*x + y*y + z*z left to right, cross, normalize
// = v * (1/sqrt(dot)), lerp = a*(1-s) + b*s), and the translation unit is built
// with -ffp-contract=off, so that no v_fma is formed where rustc emits mul+add.
// min/max follow Rust's f32::min/max as lowered on x86-64 (what the CPU path
// executes), not v_min_f32/v_max_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

namespace ptd {

constexpr float PT_INF = __builtin_inff();
constexpr float PT_PI = 3.14159265358979323846264338327950288f;
constexpr float PT_TAU = 6.28318530717958647692528676655900577f;
constexpr float RAY_EPSILON = 0.001f;   // ray.rs:1

PT_DEV float f_min(float a, float b) { return (a != a) ? b : ((b < a) ? b : a); }
PT_DEV float f_max(float a, float b) { return (a != a) ? b : ((b > a) ? b : a); }
PT_DEV float f_clamp(float x, float lo, float hi) { return (x < lo) ? lo : ((x > hi) ? hi : x); }
// _mm256_min_ps / _mm256_max_ps: second operand when either is NaN (accel/bvh.rs:205-213)
PT_DEV float x86_min(float a, float b) { return (a < b) ? a : b; }
PT_DEV float x86_max(float a, float b) { return (a > b) ? a : b; }
PT_DEV float powi5(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x * x4; }
PT_DEV float powi2(float x) { return x * x; }
PT_DEV float fractf(float x) { return x - truncf(x); }
PT_DEV bool finitef(float x) { return fabsf(x) < PT_INF; }   // false for NaN and +-inf
// f32::rem_euclid(1.0) (texture.rs:73-74).  fmod(x, 1) is x - trunc(x) exactly (the fractional part of a float is a float)
// with the sign of x also on a zero result; OCML's general fmodf — an exponent loop, ~45 instructions — computed the same
// bits.  Non-finite x gives NaN either way.
PT_DEV float rem_euclid1(float x) { const float r = __builtin_copysignf(x - truncf(x), x); return (r < 0.0f) ? r + 1.0f : r; }
// i32::rem_euclid (texture.rs:82-85).  A bilinear tap index lies in [-b, 2b) for every finite uv, where one conditional
// add or subtract is the residue; the integer division (~25 instructions a piece) is kept for whatever else arrives.
PT_DEV int rem_euclid_i(int a, int b) {
    int r = a < 0 ? a + b : (a >= b ? a - b : a);
    if (b <= 0 || (unsigned)r >= (unsigned)b) { r = a % b; r = r < 0 ? r + b : r; }
    return r;
}

struct V3 { float x, y, z; };
PT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV V3 splat(float s) { return v3(s, s, s); }
PT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PT_DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
PT_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
PT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PT_DEV float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
PT_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
PT_DEV float length(V3 a) { return sqrtf(dot(a, a)); }
PT_DEV V3 normalize(V3 a) { return a * (1.0f / length(a)); }
PT_DEV V3 lerp(V3 a, V3 b, float s) { return a * (1.0f - s) + b * s; }
PT_DEV float max_element(V3 a) { return f_max(a.x, f_max(a.y, a.z)); }
PT_DEV bool is_finite(V3 a) { return finitef(a.x) && finitef(a.y) && finitef(a.z); }
PT_DEV V3 clamp01(V3 a) { return v3(f_min(f_max(a.x, 0.0f), 1.0f), f_min(f_max(a.y, 0.0f), 1.0f), f_min(f_max(a.z, 0.0f), 1.0f)); }
PT_DEV float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

}  // namespace ptd
