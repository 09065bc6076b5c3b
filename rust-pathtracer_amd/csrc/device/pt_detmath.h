// pt_detmath.h — f32 transcendentals for the kernels, evaluated in f64.
//
// The reference calls Rust std's f32 sin/cos/exp/ln/atan2/asin, i.e. glibc
// libm on the CPU.  ROCm's OCML versions differ from glibc in the last ulp
// for some arguments, which would send a fraction of paths down different
// branches.  These versions instead evaluate a FIXED sequence of IEEE f64
// +,-,*,/,sqrt (gfx950 runs f64 vector math at half the f32 rate, and these
// calls are a tiny share of a path's work) and round once to f32: the result
// is the correctly rounded value except in near-tie cases, so it agrees with
// glibc on all but a vanishing fraction of inputs, and it is reproducible
// anywhere the same sequence is evaluated without contraction (the test
// oracle's `det` build does exactly that).
#pragma once
#include "pt_vec.h"

namespace ptd {

PT_DEV double bits_to_f64(uint64_t b) { return __longlong_as_double((long long)b); }
PT_DEV uint64_t f64_to_bits(double d) { return (uint64_t)__double_as_longlong(d); }

PT_DEV double k_sin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = r * r;
    const double p = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    return r + (r * z) * p;
}
PT_DEV double k_cos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = r * r;
    const double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return (1.0 - 0.5 * z) + z * p;
}
PT_DEV unsigned reduce_pio2(double x, double* r) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double MAGIC = 6755399441055744.0;
    const double t = x * TWO_OVER_PI + MAGIC;
    const double kd = t - MAGIC;
    *r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    return (unsigned)(f64_to_bits(t) & 3u);
}
PT_DEV float d_sinf(float xf) {
    double r;
    const unsigned q = reduce_pio2((double)xf, &r);
    double v;
    if (q == 0) v = k_sin(r);
    else if (q == 1) v = k_cos(r);
    else if (q == 2) v = -k_sin(r);
    else v = -k_cos(r);
    return (float)v;
}
PT_DEV float d_cosf(float xf) {
    double r;
    const unsigned q = reduce_pio2((double)xf, &r);
    double v;
    if (q == 0) v = k_cos(r);
    else if (q == 1) v = -k_sin(r);
    else if (q == 2) v = -k_cos(r);
    else v = k_sin(r);
    return (float)v;
}
// both at once (one range reduction): same values as d_sinf / d_cosf
PT_DEV void d_sincosf(float xf, float* s, float* c) {
    double r;
    const unsigned q = reduce_pio2((double)xf, &r);
    const double ks = k_sin(r), kc = k_cos(r);
    double vs, vc;
    if (q == 0) { vs = ks; vc = kc; }
    else if (q == 1) { vs = kc; vc = -ks; }
    else if (q == 2) { vs = -ks; vc = -kc; }
    else { vs = -kc; vc = ks; }
    *s = (float)vs;
    *c = (float)vc;
}

PT_DEV float d_expf(float xf) {
    if (xf != xf) return xf;
    if (xf > 89.0f) return PT_INF;
    if (xf < -104.0f) return 0.0f;
    const double x = (double)xf;
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
    const double scale = bits_to_f64((uint64_t)(k + 1023) << 52);
    return (float)(p * scale);
}

PT_DEV float d_logf(float xf) {
    if (xf != xf) return xf;
    if (xf < 0.0f) return __builtin_nanf("");
    if (xf == 0.0f) return -PT_INF;
    if (xf == PT_INF) return xf;
    const double x = (double)xf;
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
    return (float)(ed * LN2_HI + ((2.0 * s) * p + ed * LN2_LO));
}

// ---- powf / log2f (tone mapping only: exporter/tonemapping.rs) ------------------------
// The same ln / exp cores as above, kept in f64 between the two so that x^y is rounded to f32 once.
PT_DEV double log_f64_core(double x) {          // x finite, > 0
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
PT_DEV double exp_f64_core(double x) {          // |x| < 700
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
PT_DEV float powf_det(float xf, float yf) {
    if (xf != xf || yf != yf) return __builtin_nanf("");
    if (xf < 0.0f) return __builtin_nanf("");
    if (xf == 0.0f) return 0.0f;
    if (xf == PT_INF) return xf;
    const double e = (double)yf * log_f64_core((double)xf);
    if (e > 89.0) return PT_INF;
    if (e < -104.0) return 0.0f;
    return (float)exp_f64_core(e);
}
PT_DEV float log2f_det(float xf) {
    if (xf != xf) return xf;
    if (xf < 0.0f) return __builtin_nanf("");
    if (xf == 0.0f) return -PT_INF;
    if (xf == PT_INF) return xf;
    return (float)(log_f64_core((double)xf) * 1.44269504088896338700e+00);
}

PT_DEV double atan_pos(double z) {
    const double PIO2 = 1.57079632679489655800e+00, PIO6 = 5.23598775598298815659e-01;
    const double SQRT3 = 1.73205080756887719318e+00, TAN_PIO12 = 2.67949192431122695840e-01;
    bool inv = false;
    if (z > 1.0) { z = 1.0 / z; inv = true; }
    bool shift = false;
    if (z > TAN_PIO12) { z = (z * SQRT3 - 1.0) / (SQRT3 + z); shift = true; }
    const double w = z * z;
    double p = 1.0 / 29.0;
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
PT_DEV double atan2_f64(double y, double x) {
    const double PI = 3.14159265358979311600e+00, PIO4 = 7.85398163397448278999e-01;
    const double DINF = __builtin_inf();
    if (x != x || y != y) return x + y;
    const bool xneg = (f64_to_bits(x) >> 63) != 0, yneg = (f64_to_bits(y) >> 63) != 0;
    const double ax = xneg ? -x : x, ay = yneg ? -y : y;
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ax == DINF && ay == DINF) a = PIO4;
    else a = atan_pos(ay / ax);
    if (xneg) a = PI - a;
    if (yneg) a = -a;
    return a;
}
PT_DEV float d_atan2f(float y, float x) { return (float)atan2_f64((double)y, (double)x); }
PT_DEV float d_asinf(float xf) {
    if (xf != xf) return xf;
    const double x = (double)xf;
    if (x > 1.0 || x < -1.0) return __builtin_nanf("");
    return (float)atan2_f64(x, sqrt((1.0 - x) * (1.0 + x)));
}

}  // namespace ptd
