// libm_sincosf.hpp -- sinf / cosf as the reference's host libm returns them, for the device (and, for its test, the host).
//
// The reference's CPU path takes std::cos / std::sin of a FLOAT (src/engine/glyph_kernels.cu:126-128 rotation, :236-237 Line
// direction): glibc's cosf / sinf.  A Line's end points are ROUNDED to cells (:245-250), so a result that differs in its last
// bit moves an end point across a rounding boundary now and then -- once in ~10^8 segments, found by the round-5 soak (mixed
// pipeline fuzz, seed 48227: one segment of 6 457 drawn a row higher, four cells off by its value).  Rounds 1-4 used
// (float)cos((double)a), believing glibc's float routines to be correctly rounded.  They are not: they evaluate a short
// polynomial in double and round once, with up to 0.56 ulp of error -- 2.7 % of random directions come out one ulp away from the
// correctly rounded value.
//
// So this is glibc's own algorithm, restated: the third-party code the reference's arithmetic lives in is glibc >= 2.28
// (this image: Ubuntu GLIBC 2.35), sysdeps/ieee754/flt-32/{s_sinf.c, s_cosf.c, sincosf.h} -- Szabolcs Nagy's sincosf from
// ARM's optimized-routines (MIT).  Published algorithm: |y| < pi/4: polynomials in double on y; |y| < 120: n = round(y *
// 2/pi) by a scaled float-to-int conversion, x = y - n * pi/2 in double, sign and polynomial chosen by n mod 4; larger: a
// 192-bit 4/pi table and 64-bit integer arithmetic give x and n; the result is the double rounded to float.  The constants
// are the ones in this image's libm.so.6 (__sincosf_table, __inv_pio4; read out of its .rodata and compared), the x86-64 build's
// fused multiply-adds included (its ifunc picks the FMA variant on every CPU that has one; without the fusion in the
// reduction 0.8 % of the arguments next to a multiple of pi/2 differ in cosf).  tests/test_libm_sincosf.py compiles this
// header for the host and compares it with the system's sinf / cosf BIT FOR BIT over ~3e7 arguments of every range.
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define PCR_LIBM_HD __host__ __device__ __forceinline__
#else
#define PCR_LIBM_HD inline
#endif

namespace pcrhip {
namespace libm {

PCR_LIBM_HD uint32_t as_u32(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

// sinf_poly of sincosf.h: n even -> the sine polynomial, n odd -> the cosine polynomial; `neg`: the second table (cosine
// coefficients negated: quadrants 2 and 3)
PCR_LIBM_HD double poly(double x, double x2, int n, bool neg) {
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double k = neg ? -1.0 : 1.0;
    const double c0 = k, c1 = k * -0x1.ffffffd0c621cp-2, c2 = k * 0x1.55553e1068f19p-5, c3 = k * -0x1.6c087e89a359dp-10,
                 c4 = k * 0x1.99343027bf8c3p-16;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t = __builtin_fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double s = __builtin_fma(x3, s1, x);
        return __builtin_fma(x7, t, s);
    }
    const double x4 = x2 * x2;
    const double t2 = __builtin_fma(x2, c4, c3);
    const double t1 = __builtin_fma(x2, c1, c0);
    const double x6 = x4 * x2;
    const double c = __builtin_fma(x4, c2, t1);
    return __builtin_fma(x6, t2, c);
}

// reduce_large: 4/pi to 192 bits, the argument's 24 mantissa bits times the three words that matter
PCR_LIBM_HD double reduce_large(uint32_t xi, int& np) {
    const uint32_t inv_pio4[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                   0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                   0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    uint64_t res0 = (uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4];
    const uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    np = (int)n;
    return (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
}

// s = sinf(y), c = cosf(y), each as glibc's separate routine returns it
PCR_LIBM_HD void sincosf(float y, float& s, float& c) {
    const uint32_t xi = as_u32(y), top = (xi >> 20) & 0x7ff;
    double x = (double)y;
    if (top < 0x3f4u) {                                   // |y| < pi/4
        if (top < 0x398u) { s = y; c = 1.0f; return; }    // |y| < 2^-12
        const double x2 = x * x;
        s = (float)poly(x, x2, 0, false);
        c = (float)poly(x, x2, 1, false);
        return;
    }
    int n, q;
    if (top < 0x42fu) {                                   // |y| < 120: one multiply-subtract
        const double r = x * 0x1.45F306DC9C883p+23;       // 2/pi * 2^24: the quadrant ends up in bits 24..31
        n = ((int32_t)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
        q = n;
    } else if (top < 0x7f8u) {
        x = reduce_large(xi, n);
        q = n + (int)(xi >> 31);
    } else {                                              // inf, NaN
        s = c = y - y;
        return;
    }
    const double sg = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;     // sign[] = {1, -1, -1, 1}
    const bool neg = (q & 2) != 0;
    const double xs = x * sg, x2 = x * x;
    s = (float)poly(xs, x2, n, neg);
    c = (float)poly(xs, x2, n ^ 1, neg);
}

}  // namespace libm
}  // namespace pcrhip
