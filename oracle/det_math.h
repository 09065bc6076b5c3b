// det_math.h — TEST INFRASTRUCTURE (oracle side).
//
// Deterministic f32 transcendentals: every function evaluates a fixed sequence
// of IEEE-754 double-precision +,-,*,/,sqrt (no FMA: build with
// -ffp-contract=off) and rounds once to f32.  The HIP kernels carry their own
// copy of the same sequences (rust-pathtracer_amd/csrc/device/pt_detmath.h),
// so the `det` build of the oracle and the GPU agree bit-for-bit on sin, cos,
// exp, ln, atan2 and asin — which lets tests demand bit-exact images — while
// the `libm` build of the oracle calls glibc, which is what the reference's
// Rust std f32 methods resolve to.  Results are within
// 1 ulp(f32) of glibc's (tests/test_oracle_math.py measures it).
//
// Polynomial coefficients: sin/cos are the classic fdlibm kernel minimax
// coefficients; exp/log/atan are plain Taylor/Maclaurin series, long enough
// that the truncation error is below 2^-60 on the reduced range.
#pragma once
#include <cstdint>
#include <cstring>

namespace detm {

static inline double bits_to_f64(uint64_t b) { double d; std::memcpy(&d, &b, 8); return d; }
static inline uint64_t f64_to_bits(double d) { uint64_t b; std::memcpy(&b, &d, 8); return b; }
static inline double sqrt_f64(double x) { return __builtin_sqrt(x); }

// ---- sin / cos ---------------------------------------------------------------
static inline double k_sin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = r * r;
    const double p = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    return r + (r * z) * p;
}
static inline double k_cos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = r * r;
    const double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return (1.0 - 0.5 * z) + z * p;
}
// x = k*(pi/2) + r, |r| <= pi/4 (+eps); returns k mod 4
static inline unsigned reduce_pio2(double x, double* r) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double MAGIC = 6755399441055744.0;            // 1.5 * 2^52
    const double t = x * TWO_OVER_PI + MAGIC;
    const double kd = t - MAGIC;
    *r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    return (unsigned)(f64_to_bits(t) & 3u);
}
static inline float sinf_det(float xf) {
    double r;
    const unsigned q = reduce_pio2((double)xf, &r);
    double v;
    if (q == 0) v = k_sin(r);
    else if (q == 1) v = k_cos(r);
    else if (q == 2) v = -k_sin(r);
    else v = -k_cos(r);
    return (float)v;
}
static inline float cosf_det(float xf) {
    double r;
    const unsigned q = reduce_pio2((double)xf, &r);
    double v;
    if (q == 0) v = k_cos(r);
    else if (q == 1) v = -k_sin(r);
    else if (q == 2) v = -k_cos(r);
    else v = k_sin(r);
    return (float)v;
}

// ---- exp -----------------------------------------------------------------------
static inline float expf_det(float xf) {
    if (xf != xf) return xf;
    if (xf > 89.0f) return __builtin_inff();
    if (xf < -104.0f) return 0.0f;
    const double x = (double)xf;
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double MAGIC = 6755399441055744.0;
    const double t = x * INV_LN2 + MAGIC;
    const double kd = t - MAGIC;
    const int64_t k = (int64_t)(int32_t)(uint32_t)(f64_to_bits(t) & 0xFFFFFFFFu);
    const double r = (x - kd * LN2_HI) - kd * LN2_LO;
    // Taylor, degree 13
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const double scale = bits_to_f64((uint64_t)(k + 1023) << 52);
    return (float)(p * scale);
}

// ---- ln ------------------------------------------------------------------------
static inline float logf_det(float xf) {
    if (xf != xf) return xf;
    if (xf < 0.0f) return __builtin_nanf("");
    if (xf == 0.0f) return -__builtin_inff();
    if (xf == __builtin_inff()) return xf;
    const double x = (double)xf;  // exact; normal in f64 even for f32 subnormals
    uint64_t b = f64_to_bits(x);
    int64_t e = (int64_t)((b >> 52) & 0x7FF) - 1023;
    b = (b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = bits_to_f64(b);                    // [1, 2)
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double ed = (double)e;
    return (float)(ed * LN2_HI + ((2.0 * s) * p + ed * LN2_LO));
}

// ---- powf / log2f (tone mapping only: exporter/tonemapping.rs) ------------------------
// The same ln / exp cores as above, kept in f64 between the two so that x^y is rounded to f32 once.
static inline double log_f64_core(double x) {          // x finite, > 0
    uint64_t b = f64_to_bits(x);
    int64_t e = (int64_t)((b >> 52) & 0x7FF) - 1023;
    b = (b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = bits_to_f64(b);
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double ed = (double)e;
    return ed * LN2_HI + ((2.0 * s) * p + ed * LN2_LO);
}
static inline double exp_f64_core(double x) {          // |x| < 700
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double MAGIC = 6755399441055744.0;
    const double t = x * INV_LN2 + MAGIC;
    const double kd = t - MAGIC;
    const int64_t k = (int64_t)(int32_t)(uint32_t)(f64_to_bits(t) & 0xFFFFFFFFu);
    const double r = (x - kd * LN2_HI) - kd * LN2_LO;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return p * bits_to_f64((uint64_t)(k + 1023) << 52);
}
// x^y for the tone mapper's arguments: y is a positive constant (2.2, 1/2.4); x < 0 (AgX's inverse matrix can
// produce it) is NaN like libm's non-integer power; 0 -> 0; +inf -> +inf.
static inline float powf_det(float xf, float yf) {
    if (xf != xf || yf != yf) return __builtin_nanf("");
    if (xf < 0.0f) return __builtin_nanf("");
    if (xf == 0.0f) return 0.0f;
    if (xf == __builtin_inff()) return xf;
    const double e = (double)yf * log_f64_core((double)xf);
    if (e > 89.0) return __builtin_inff();
    if (e < -104.0) return 0.0f;
    return (float)exp_f64_core(e);
}
static inline float log2f_det(float xf) {
    if (xf != xf) return xf;
    if (xf < 0.0f) return __builtin_nanf("");
    if (xf == 0.0f) return -__builtin_inff();
    if (xf == __builtin_inff()) return xf;
    return (float)(log_f64_core((double)xf) * 1.44269504088896338700e+00);
}

// ---- atan / atan2 / asin ---------------------------------------------------------
static inline double atan_pos(double z) {  // z >= 0 (may be +inf)
    const double PIO2 = 1.57079632679489655800e+00, PIO6 = 5.23598775598298815659e-01;
    const double SQRT3 = 1.73205080756887719318e+00, TAN_PIO12 = 2.67949192431122695840e-01;
    bool inv = false;
    if (z > 1.0) { z = 1.0 / z; inv = true; }
    bool shift = false;
    if (z > TAN_PIO12) { z = (z * SQRT3 - 1.0) / (SQRT3 + z); shift = true; }
    const double w = z * z;
    double p = 1.0 / 29.0;     // alternating series, Horner from the top
    p = 1.0 / 27.0 - w * p;
    p = 1.0 / 25.0 - w * p;
    p = 1.0 / 23.0 - w * p;
    p = 1.0 / 21.0 - w * p;
    p = 1.0 / 19.0 - w * p;
    p = 1.0 / 17.0 - w * p;
    p = 1.0 / 15.0 - w * p;
    p = 1.0 / 13.0 - w * p;
    p = 1.0 / 11.0 - w * p;
    p = 1.0 / 9.0 - w * p;
    p = 1.0 / 7.0 - w * p;
    p = 1.0 / 5.0 - w * p;
    p = 1.0 / 3.0 - w * p;
    p = 1.0 - w * p;
    double a = z * p;
    if (shift) a = a + PIO6;
    if (inv) a = PIO2 - a;
    return a;
}
static inline double atan2_f64(double y, double x) {
    const double PI = 3.14159265358979311600e+00, PIO4 = 7.85398163397448278999e-01;
    const double INF = __builtin_inf();
    if (x != x || y != y) return x + y;
    const bool xneg = (f64_to_bits(x) >> 63) != 0, yneg = (f64_to_bits(y) >> 63) != 0;
    const double ax = xneg ? -x : x, ay = yneg ? -y : y;
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ax == INF && ay == INF) a = PIO4;
    else a = atan_pos(ay / ax);
    if (xneg) a = PI - a;
    if (yneg) a = -a;
    return a;
}
static inline float atan2f_det(float y, float x) { return (float)atan2_f64((double)y, (double)x); }
static inline float asinf_det(float xf) {
    if (xf != xf) return xf;
    const double x = (double)xf;
    if (x > 1.0 || x < -1.0) return __builtin_nanf("");
    return (float)atan2_f64(x, sqrt_f64((1.0 - x) * (1.0 + x)));
}

}  // namespace detm
