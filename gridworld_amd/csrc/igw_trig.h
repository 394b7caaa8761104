// igw_trig.h -- the build's own double-precision sincos / atan2 (SURVEY.md "A-fly").
//
// Used where the 5-degree walking LUT does not apply (flying mode, arbitrary poses).  ONE source for
// host and device: every operation is an IEEE binary64 add / mul / div / fma, so the gfx950 kernel and
// a host compile of this header produce identical bits (tests/test_trig.py, tests/test_gpu_flying.py).
//
// Method: double-double evaluation, then round.
//   sincos: Cody-Waite reduction by pi/2 (triple-double, exact products via fma), table of
//           sin/cos(j/64) in double-double, degree-9/8 Taylor tails on |d| <= 1/128 with the
//           leading correction terms in double-double.
//   atan2 : ratio in double-double, table of atan(j/64), atan(u) series on |u| <= 1/128.
// The double-double result carries ~100 bits, so the returned double is the correctly rounded
// value except for astronomically rare ties-at-100-bits; on 2*10^5 random arguments each function
// matches mpmath's correctly rounded result everywhere (tests/test_trig.py).  glibc 2.35 (what the
// Python reference calls through `math`) is itself NOT correctly rounded on ~0.1-0.2 % of arguments
// (SURVEY.md F12), which bounds how often the two can differ, always by one ulp.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define IGW_HD __host__ __device__ inline
#else
#define IGW_HD inline
#endif

#include "igw_trig_tables.h"

namespace igw {

struct dd_t {
    double hi, lo;
};

IGW_HD dd_t dd_two_sum(double a, double b) {
    const double s = a + b;
    const double bb = s - a;
    const double e = (a - (s - bb)) + (b - bb);
    return dd_t{s, e};
}
IGW_HD dd_t dd_fast_two_sum(double a, double b) {  // |a| >= |b| or a == 0
    const double s = a + b;
    const double e = b - (s - a);
    return dd_t{s, e};
}
IGW_HD dd_t dd_two_prod(double a, double b) {
    const double p = a * b;
    const double e = __builtin_fma(a, b, -p);
    return dd_t{p, e};
}
IGW_HD dd_t dd_add(dd_t a, dd_t b) {
    dd_t s = dd_two_sum(a.hi, b.hi);
    const dd_t t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_fast_two_sum(s.hi, s.lo);
}
IGW_HD dd_t dd_add_d(dd_t a, double b) {
    dd_t s = dd_two_sum(a.hi, b);
    s.lo += a.lo;
    return dd_fast_two_sum(s.hi, s.lo);
}
IGW_HD dd_t dd_neg(dd_t a) { return dd_t{-a.hi, -a.lo}; }
IGW_HD dd_t dd_mul(dd_t a, dd_t b) {
    dd_t p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_fast_two_sum(p.hi, p.lo);
}
IGW_HD dd_t dd_mul_d(dd_t a, double b) {
    dd_t p = dd_two_prod(a.hi, b);
    p.lo = __builtin_fma(a.lo, b, p.lo);
    return dd_fast_two_sum(p.hi, p.lo);
}
IGW_HD dd_t dd_div(dd_t a, dd_t b) {
    const double q1 = a.hi / b.hi;
    dd_t r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
    const double q2 = r.hi / b.hi;
    r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
    const double q3 = r.hi / b.hi;
    dd_t q = dd_fast_two_sum(q1, q2);
    return dd_add_d(q, q3);
}

// sin and cos of x (radians), |x| < 2^20
IGW_HD void igw_sincos(double x, double* s_out, double* c_out) {
    using namespace trigtab;
    if (x == 0.0) {
        *s_out = x;  // keeps the sign of zero
        *c_out = 1.0;
        return;
    }
    // reduction: r = x - k * pi/2, |r| <= pi/4 (+ a rounding sliver)
    const double kd = __builtin_rint(x * TWO_OVER_PI);
    dd_t r;
    if (kd == 0.0) {
        r = dd_t{x, 0.0};
    } else {
        const dd_t p1 = dd_two_prod(kd, PIO2_1);
        const double r0 = x - p1.hi;  // exact (Sterbenz)
        r = dd_two_sum(r0, -p1.lo);
        const dd_t p2 = dd_two_prod(kd, PIO2_2);
        r = dd_add(r, dd_neg(p2));
        r = dd_add_d(r, -(kd * PIO2_3));
    }
    const bool neg = r.hi < 0.0;
    if (neg) r = dd_neg(r);
    // table point j/64 and remainder d, |d| <= 1/128
    const double jd = __builtin_rint(r.hi * 64.0);
    const int j = (int)jd;
    const dd_t d = dd_two_sum(r.hi - jd * 0.015625, r.lo);
    const dd_t S = dd_t{SINCOS[j][0], SINCOS[j][1]};
    const dd_t C = dd_t{SINCOS[j][2], SINCOS[j][3]};
    const dd_t d2 = dd_mul(d, d);
    const double q = d2.hi;
    // sin d = d - d^3/6 + d^5 (1/120 - q/5040 + q^2/362880)
    const dd_t d3 = dd_mul(d2, d);
    const double d5 = (q * q) * d.hi;
    const double ps = d5 * (0x1.1111111111111p-7 + q * (-0x1.a01a01a01a01ap-13 + q * 0x1.71de3a556c734p-19));
    dd_t sind = dd_add_d(dd_neg(dd_mul(d3, dd_t{SIXTH_HI, SIXTH_LO})), ps);
    sind = dd_add(d, sind);
    // cos d - 1 = -d^2/2 + d^4 (1/24 - q/720 + q^2/40320 - q^3/3628800)
    const double pc = (q * q) * (0x1.5555555555555p-5 + q * (-0x1.6c16c16c16c17p-10 + q * (0x1.a01a01a01a01ap-16 + q * -0x1.27e4fb7789f5cp-22)));
    const dd_t cm1 = dd_add_d(dd_t{-0.5 * d2.hi, -0.5 * d2.lo}, pc);
    // sin(xj + d) = S + (C sin d + S (cos d - 1));  cos(xj + d) = C + (C (cos d - 1) - S sin d)
    dd_t sr = dd_add(S, dd_add(dd_mul(C, sind), dd_mul(S, cm1)));
    const dd_t cr = dd_add(C, dd_add(dd_mul(C, cm1), dd_neg(dd_mul(S, sind))));
    if (neg) sr = dd_neg(sr);
    const long long k = (long long)kd;
    const int n = (int)(((k % 4) + 4) % 4);
    double sn, cs;
    if (n == 0) { sn = sr.hi; cs = cr.hi; }
    else if (n == 1) { sn = cr.hi; cs = -sr.hi; }
    else if (n == 2) { sn = -sr.hi; cs = -cr.hi; }
    else { sn = -cr.hi; cs = sr.hi; }
    *s_out = sn;
    *c_out = cs;
}

// atan2(y, x) for finite arguments
IGW_HD double igw_atan2(double y, double x) {
    using namespace trigtab;
    const bool xneg = __builtin_signbit(x);
    if (y == 0.0) return xneg ? __builtin_copysign(PI_HI, y) : __builtin_copysign(0.0, y);
    if (x == 0.0) return __builtin_copysign(PIO2_HI, y);
    const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    const bool swap = ay > ax;
    const double num = swap ? ax : ay, den = swap ? ay : ax;
    // t = num / den in double-double
    const double t1 = num / den;
    const double rem = __builtin_fma(-t1, den, num);  // exact remainder
    const dd_t t = dd_fast_two_sum(t1, rem / den);
    const double jd = __builtin_rint(t.hi * 64.0);
    const int j = (int)jd;
    const double tj = jd * 0.015625;
    // u = (t - tj) / (1 + t tj), |u| <= 1/128
    const dd_t un = dd_two_sum(t.hi - tj, t.lo);
    const dd_t ud = dd_add_d(dd_mul_d(t, tj), 1.0);
    const dd_t u = (j == 0) ? t : dd_div(un, ud);
    // atan u = u - u^3/3 + u^5 (1/5 - w/7 + w^2/9 - w^3/11), w = u^2
    const dd_t u2 = dd_mul(u, u);
    const double w = u2.hi;
    const dd_t u3 = dd_mul(u2, u);
    const double u5 = (w * w) * u.hi;
    const double pa = u5 * (0x1.999999999999ap-3 + w * (-0x1.2492492492492p-3 + w * (0x1.c71c71c71c71cp-4 + w * -0x1.745d1745d1746p-4)));
    dd_t a = dd_add_d(dd_neg(dd_mul(u3, dd_t{THIRD_HI, THIRD_LO})), pa);
    a = dd_add(u, a);
    a = dd_add(dd_t{ATAN[j][0], ATAN[j][1]}, a);
    if (swap) a = dd_add(dd_t{PIO2_HI, PIO2_LO}, dd_neg(a));
    if (xneg) a = dd_add(dd_t{PI_HI, PI_LO}, dd_neg(a));
    return __builtin_copysign(a.hi, y);
}

}  // namespace igw
