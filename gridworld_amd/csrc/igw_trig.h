// igw_trig.h -- the build's own double-precision sincos / atan2 (SURVEY.md "A-fly").
//
// Used where the 5-degree walking LUT does not apply (flying mode, arbitrary poses).  ONE source for
// host and device: every operation is an IEEE binary64 add / mul / div / fma, so the gfx950 kernel and
// a host compile of this header produce identical bits (tests/test_trig.py, tests/test_gpu_flying.py).
//
// Method: double-double evaluation, then round.
//   sincos: Cody-Waite reduction by pi/2 (33-bit pieces in the quick evaluation, a triple-double with exact
//           products via fma in the accurate one), table of
//           sin/cos(j/256) in double-double, degree-9/8 Taylor tails on |d| <= 1/512 with the
//           leading correction terms in double-double; a quick evaluation with an error bound first
//           (Ziv's strategy), the full double-double one only when the bound straddles a rounding boundary.
//   atan2 : ratio in double-double, table of atan(j/256), atan(u) series on |u| <= 1/512; quick evaluation first.
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

// 1/b for the quick atan2's two quotients, which only need it to within 2^-51 (they are refined with an exact
// remainder).  On the GPU: v_rcp_f64 (at least 20 good bits) + two Newton steps = 5 instructions, against 12 for the
// IEEE division sequence; on the host: the division.  The two differ in the last bit or two of an INTERMEDIATE; the
// returned atan2 is the correctly rounded one on both as long as each passes its rounding test, which is what
// tests/test_trig.py (host) and tests/test_gpu_flying.py::test_device_trig_self_check (GPU) check.
// b is finite, normal and far from the overflow / underflow thresholds (|float32 value| or 1 + t tj).
IGW_HD double igw_recip(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    return 1.0 / b;
#endif
}

// Argument reduction shared by the two evaluations below: x = kd * pi/2 + r, |r| <= pi/4 (+ a rounding sliver),
// r as a double-double.
IGW_HD dd_t sincos_reduce(double x, double* kd_out) {
    using namespace trigtab;
    const double kd = __builtin_rint(x * TWO_OVER_PI);
    *kd_out = kd;
    if (kd == 0.0) return dd_t{x, 0.0};
    const dd_t p1 = dd_two_prod(kd, PIO2_1);
    const double r0 = x - p1.hi;  // exact (Sterbenz)
    dd_t r = dd_two_sum(r0, -p1.lo);
    const dd_t p2 = dd_two_prod(kd, PIO2_2);
    r = dd_add(r, dd_neg(p2));
    return dd_add_d(r, -(kd * PIO2_3));
}
IGW_HD void sincos_quadrant(double kd, double sr, double cr, double* s_out, double* c_out) {
    const int n = (int)kd & 3;  // |kd| < 2^20; two's complement & 3 == mod 4
    double sn, cs;
    if (n == 0) { sn = sr; cs = cr; }
    else if (n == 1) { sn = cr; cs = -sr; }
    else if (n == 2) { sn = -sr; cs = -cr; }
    else { sn = -cr; cs = sr; }
    *s_out = sn;
    *c_out = cs;
}

// The accurate evaluation: everything in double-double (~100 bits), then round.
IGW_HD void igw_sincos_accurate(double x, double* s_out, double* c_out) {
    using namespace trigtab;
    double kd;
    dd_t r = sincos_reduce(x, &kd);
    const bool neg = r.hi < 0.0;
    if (neg) r = dd_neg(r);
    // table point j/256 and remainder d, |d| <= 1/512
    const double jd = __builtin_rint(r.hi * 256.0);
    const int j = (int)jd;
    const dd_t d = dd_two_sum(r.hi - jd * 0.00390625, r.lo);
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
    sincos_quadrant(kd, sr.hi, cr.hi, s_out, c_out);
}

// The quick evaluation (Ziv's strategy): the same table, but only the leading products exact (two_prod) and
// everything small in plain double -- about a third of the flops -- as a value hi + lo with an error bound E; the
// result is final when hi + (lo - E) and hi + (lo + E) round to the same double, which they do except on about one
// argument in 2^19.  Returns false otherwise (the caller then runs the accurate evaluation).
// Reduction: pi/2 in 33-bit pieces A + B + C + CT.  For |kd| < 2^20 the products kd A, kd B, kd C are exact and so
// is x - kd A (Cody-Waite), hence r = ((x - kd A) - kd B) - kd (C + CT) with ONE exact two_sum and the small terms in
// plain double: |error| <= |kd| 2^-116.7.  The closing fast_two_sum needs |hi| >= |lo| (|lo| < 2^-45): arguments
// within 2^-40 of a non-zero multiple of pi/2 are left to the accurate evaluation.  For kd == 0, r = x exactly.
// d = r - j/256 is kept as dh + dl with dh = |r.hi| - j/256 (exact; zero or a multiple of ulp(r.hi) >= 2 |r.lo|) and
// dl = +-r.lo, |dl| <= 2^-54: every formula below is an expansion in dl around dh that drops dl^2 (< 2^-108).
// Error budget, absolute, |d| <= 2^-9.  Table points j >= 1 (results >= 2^-9 in magnitude): the largest plain-
// double tail is C (sin d - d), <= 2^-29.6; it is computed with a relative error <= 5 * 2^-53 (coefficient, q,
// three products, one sum): 2^-80.3; the five sums that contain it add <= 2^-82.6 each; the dropped series terms
// (d^9/9!, d^8/8!) are < 2^-87; the reduction < 2^-96.  Total < 2^-79; E = 2^-76.  j == 0 (sin x ~ x, possibly
// tiny): every tail scales with d^3, the error is < 2^-71 |d| + the reduction's; E = 2^-68 |d| + |kd| 2^-113.
// tests/test_trig.py: no wrong acceptance, and none with E / 16 either.
IGW_HD bool igw_sincos_quick(double x, double* s_out, double* c_out) {
    using namespace trigtab;
    const double kd = __builtin_rint(x * TWO_OVER_PI);
    const double r0 = x - kd * PIO2_A;
    const dd_t rs = dd_two_sum(r0, -(kd * PIO2_B));
    const double rlo = rs.lo - (kd * PIO2_C + kd * PIO2_CT);
    const bool reduced = __builtin_fabs(rs.hi) >= 0x1p-40 || kd == 0.0;
    const dd_t r = dd_fast_two_sum(rs.hi, rlo);
    const bool neg = r.hi < 0.0;
    const double rh = __builtin_fabs(r.hi), rl = neg ? -r.lo : r.lo;
    const double jd = __builtin_rint(rh * 256.0);
    const int j = (int)jd;
    const double dh = rh - jd * 0.00390625, dl = rl;
    const double Sh = SINCOS[j][0], Sl = SINCOS[j][1], Ch = SINCOS[j][2], Cl = SINCOS[j][3];
    const dd_t d2 = dd_two_prod(dh, dh);  // dh^2 exactly
    const double q = d2.hi;
    // sin d = dh + ts:  ts = dl (1 - q/2) + dh^3 (-1/6 + q/120 - q^2/5040)
    const double ps = (q * dh) * (-0x1.5555555555555p-3 + q * (0x1.1111111111111p-7 + q * -0x1.a01a01a01a01ap-13));
    const double ts = (dl - 0.5 * q * dl) + ps;
    // cos d - 1 = ch + tc:  ch = -dh^2/2 (hi part, exact), tc = -d2.lo/2 - dh dl + d^4 (1/24 - q/720)
    const double pc = (q * q) * (0x1.5555555555555p-5 + q * -0x1.6c16c16c16c17p-10);
    const double ch = -0.5 * d2.hi;
    const double tc = (pc - dh * dl) - 0.5 * d2.lo;
    // sin(xj + d) = S + C sin d + S (cos d - 1)
    const dd_t p1 = dd_two_prod(Ch, dh), p2 = dd_two_prod(Sh, ch);
    dd_t a = dd_fast_two_sum(Sh, p1.hi);   // |Sh| >= |Ch dh| for j >= 1; Sh == 0 for j == 0
    dd_t b = dd_fast_two_sum(a.hi, p2.hi);
    const double s_hi = b.hi;
    const double s_lo = (((Ch * ts + Cl * dh) + (Sh * tc + Sl * ch)) + ((p1.lo + p2.lo) + Sl)) + (a.lo + b.lo);
    // cos(xj + d) = C + C (cos d - 1) - S sin d
    const dd_t p3 = dd_two_prod(Ch, ch), p4 = dd_two_prod(Sh, dh);
    a = dd_fast_two_sum(Ch, -p4.hi);       // Ch >= 0.7 > |Sh dh|
    b = dd_fast_two_sum(a.hi, p3.hi);
    const double c_hi = b.hi;
    const double c_lo = (((Ch * tc + Cl * ch) - (Sh * ts + Sl * dh)) + ((p3.lo - p4.lo) + Cl)) + (a.lo + b.lo);
#ifndef IGW_QUICK_E_SCALE   // tests only: shrink E to find where the first wrong acceptance appears (the margin)
#define IGW_QUICK_E_SCALE 1.0
#endif
    const double es = (j == 0 ? 0x1p-68 * dh + __builtin_fabs(kd) * 0x1p-113 : 0x1p-76) * IGW_QUICK_E_SCALE, ec = 0x1p-76 * IGW_QUICK_E_SCALE;
    // the rounding test: both perturbed sums round to the same double => so does the exact value
    const double su = s_hi + (s_lo - es), sv = s_hi + (s_lo + es);
    const double cu = c_hi + (c_lo - ec), cv = c_hi + (c_lo + ec);
    // quadrant n = kd mod 4:  (sin, cos) = (sr, cr), (cr, -sr), (-sr, -cr), (-cr, sr)
    const int n = (int)kd & 3;
    const double sr = neg ? -sv : sv;
    double sn = (n & 1) ? cv : sr, cs = (n & 1) ? sr : cv;
    if (n & 2) sn = -sn;
    if ((n + 1) & 2) cs = -cs;
    *s_out = sn;
    *c_out = cs;
    return reduced && su == sv && cu == cv;
}

// sin and cos of x (radians), |x| < 2^20
IGW_HD void igw_sincos(double x, double* s_out, double* c_out) {
    if (x == 0.0) {
        *s_out = x;  // keeps the sign of zero
        *c_out = 1.0;
        return;
    }
    if (!igw_sincos_quick(x, s_out, c_out)) igw_sincos_accurate(x, s_out, c_out);
}

// atan2(y, x) for finite arguments: the accurate evaluation (double-double throughout)
IGW_HD double igw_atan2_accurate(double y, double x) {
    using namespace trigtab;
    const bool xneg = __builtin_signbit(x);
    const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    const bool swap = ay > ax;
    const double num = swap ? ax : ay, den = swap ? ay : ax;
    // t = num / den in double-double
    const double t1 = num / den;
    const double rem = __builtin_fma(-t1, den, num);  // exact remainder
    const dd_t t = dd_fast_two_sum(t1, rem / den);
    const double jd = __builtin_rint(t.hi * 256.0);
    const int j = (int)jd;
    const double tj = jd * 0.00390625;
    // u = (t - tj) / (1 + t tj), |u| <= 1/512
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

// The quick evaluation (see igw_sincos_quick): two reciprocals instead of five divisions, the leading sums exact, the
// rest in plain double, as hi + lo with an error bound E.  Error budget, absolute, |u| <= 2^-9: the quotients t and u
// are formed as q1 + q2 with q1 = a * rb, q2 = (remainder a - q1 b, one fma) * rb, rb = (1/b)(1 + e), |e| <= 2^-51
// (igw_recip): q1 + q2 - a/b = (a/b - q1)(b rb - 1), <= 2^-101 relative; the largest plain-double tail is
// u^3/3 <= 2^-28.6, computed to <= 5 * 2^-53 relative: 2^-79.3; three sums containing it: <= 2^-81.6 each; dropped
// series term u^9/9 < 2^-84.  Total < 2^-78; E = 2^-75.  For j == 0 without a reflection the result is ~t and may be
// tiny: every tail scales with t^3, the error is < 2^-71 t; E = 2^-68 t.
IGW_HD bool igw_atan2_quick(double y, double x, double* out) {
    using namespace trigtab;
    const bool xneg = __builtin_signbit(x);
    const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    const bool swap = ay > ax;
    const double num = swap ? ax : ay, den = swap ? ay : ax;
    const double rd = igw_recip(den);
    const double t1 = num * rd;
    const double t2 = __builtin_fma(-t1, den, num) * rd;  // remainder of a quotient that is within 2^-50 of num/den
    const dd_t t = dd_fast_two_sum(t1, t2);
    const double jd = __builtin_rint(t.hi * 256.0);
    const int j = (int)jd;
    const double tj = jd * 0.00390625;
    double uh = t.hi, ul = t.lo;
    if (j != 0) {
        // u = un / ud:  un = t - tj as unh + t.lo (unh exact; zero or a multiple of ulp(t.hi) >= 2 |t.lo|),
        // ud = 1 + t tj as a double-double
        const double unh = t.hi - tj;
        const dd_t m = dd_two_prod(t.hi, tj);
        dd_t ud = dd_fast_two_sum(1.0, m.hi);
        ud.lo += m.lo + t.lo * tj;
        const double ru = igw_recip(ud.hi);
        const double u1 = unh * ru;
        const double u2 = ((__builtin_fma(-u1, ud.hi, unh) + t.lo) - u1 * ud.lo) * ru;
        const dd_t u = dd_fast_two_sum(u1, u2);  // u1 == 0 or |u1| > |u2|
        uh = u.hi; ul = u.lo;
    }
    // atan u = uh + ta:  ta = ul + uh^3 (-1/3 + w/5 - w^2/7), w = uh^2   (ul (1 - w) ~ ul: w ul < 2^-80)
    const double w = uh * uh;
    const double ta = ul + (w * uh) * (-0x1.5555555555555p-2 + w * (0x1.999999999999ap-3 + w * -0x1.2492492492492p-3));
    const double Ah = ATAN[j][0], Al = ATAN[j][1];
    dd_t a = dd_fast_two_sum(Ah, uh);  // Ah >= atan(1/256) > |u| for j >= 1; Ah == 0 for j == 0
    a.lo += Al + ta;
    // the octant: a | pi/2 - a (swap) | pi - a (x < 0) | pi/2 + a (both) = B +- a in one sum (B == 0 or B > a)
    const double Bh = swap ? PIO2_HI : (xneg ? PI_HI : 0.0), Bl = swap ? PIO2_LO : (xneg ? PI_LO : 0.0);
    const bool minus = swap != xneg;
    const dd_t s = dd_fast_two_sum(Bh, minus ? -a.hi : a.hi);
    const double lo = s.lo + (Bl + (minus ? -a.lo : a.lo));
#ifndef IGW_QUICK_E_SCALE
#define IGW_QUICK_E_SCALE 1.0
#endif
    const double e = ((j == 0 && !swap && !xneg) ? 0x1p-68 * uh : 0x1p-75) * IGW_QUICK_E_SCALE;
    const double u = s.hi + (lo - e), v = s.hi + (lo + e);
    *out = __builtin_copysign(v, y);
    return u == v;
}

// atan2(y, x) for finite arguments
IGW_HD double igw_atan2(double y, double x) {
    using namespace trigtab;
    const bool xneg = __builtin_signbit(x);
    if (y == 0.0) return xneg ? __builtin_copysign(PI_HI, y) : __builtin_copysign(0.0, y);
    if (x == 0.0) return __builtin_copysign(PIO2_HI, y);
    double a;
    if (igw_atan2_quick(y, x, &a)) return a;
    return igw_atan2_accurate(y, x);
}

}  // namespace igw
