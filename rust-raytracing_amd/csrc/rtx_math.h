// rtx_math.h -- f64 Vector3 arithmetic in the reference's operation order, usable from host and
// device code.  Every function is the inlined form of one src/math item of the reference
// (cited per function).  The translation units that include this header are compiled with
// -ffp-contract=off: Rust never fuses a*b+c, so neither may these (the results of the exact
// path must have the reference's roundings).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define RTX_HD __host__ __device__ __forceinline__

namespace rtx {

struct V3 { double x, y, z; };                                       // math/vector.rs:12-20

RTX_HD V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RTX_HD V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }     // vector/add.rs:16-24
RTX_HD V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }     // vector/sub.rs:16-24
RTX_HD V3 vmuls(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }      // vector/mul.rs:11-20
RTX_HD V3 vmulv(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }    // vector/mul.rs:22-30 (element-wise)
RTX_HD V3 vdivs(V3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }      // vector/div.rs:11-20 (true divisions)
RTX_HD V3 vneg(V3 a) { return mk(-a.x, -a.y, -a.z); }                          // vector.rs:115-121

RTX_HD double dot(V3 a, V3 b)                                        // vector.rs:85-87: (x*x' + y*y') + z*z'
{
    double xx = a.x * b.x;
    double yy = a.y * b.y;
    double zz = a.z * b.z;
    double s = xx + yy;
    return s + zz;
}

RTX_HD V3 cross(V3 a, V3 b)                                          // vector.rs:89-95
{
    return mk(a.y * b.z - a.z * b.y,
              a.z * b.x - a.x * b.z,
              a.x * b.y - a.y * b.x);
}

RTX_HD double vlen(V3 a)                                             // vector.rs:101-103
{
    double xx = a.x * a.x;
    double yy = a.y * a.y;
    double zz = a.z * a.z;
    double s = xx + yy;
    s = s + zz;
    return sqrt(s);
}

RTX_HD V3 vnorm(V3 a) { return vdivs(a, vlen(a)); }                  // vector.rs:105-107

// Counter-based RNG that stands in for fastrand's thread-local generator
// (math/vector.rs:31-33,37-38).  Integer arithmetic only, so the test-suite's CPU checker can
// reproduce every draw bit for bit (DESIGN.md "RNG").
RTX_HD uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

RTX_HD uint64_t rng_key(uint64_t seed, uint64_t pixel_index, uint64_t sample_index)
{
    uint64_t a = mix64(seed + 0x9E3779B97F4A7C15ULL * (pixel_index + 1));
    return mix64(a ^ (0xD1B54A32D192ED03ULL * (sample_index + 1)));
}

RTX_HD double rng_u01(uint64_t key, uint32_t draw_index)              // fastrand::f64(): [1,2) - 1.0
{
    uint64_t z = mix64(key + 0x9E3779B97F4A7C15ULL * ((uint64_t)draw_index + 1));
    uint64_t bits = 0x3FF0000000000000ULL | (z >> 12);
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)bits) - 1.0;
#else
    double d;
    __builtin_memcpy(&d, &bits, sizeof d);
    return d - 1.0;
#endif
}

// sin and cos of an angle in [0, 2 pi] -- Vector3::random_direction's theta = u * 2 * pi (vector.rs:38-42), the path's only
// transcendental call.  The reference calls the platform libm; no two libms agree on the last bit of sin / cos everywhere, so this is
// the one place the device may differ from a CPU run (tests: <= 1 ulp from glibc; images within 1e-9 unless a path flips).  The device
// library's sincos() is a general routine (any argument: ~175 instructions and enough registers to push the sphere kernels' ray state
// into scratch -- 33-59 spilled registers, 4-6 without it) that differs from glibc's on ~3 % of arguments.  For this range:
//     n = rint(x * 2/pi) in 0..4;  y = x - n * pi/2 with pi/2 in three pieces (33 + 33 + 53 bits): the first product and difference
//     are exact, the second difference's rounding error is recovered (Fast2Sum) and joins the third piece in the tail: y0 + y1;
//     fdlibm's polynomials for sin and cos on [-pi/4, pi/4] (S1..S6, C1..C6), evaluated so that the result is ONE rounding of
//     leading term + everything small: the squares' and cube's rounding errors (fma tails: zl, vl) are carried in the small sums,
//     which brings the worst error found from fdlibm's ~0.8 ulp to 0.58 (against 200-bit values);  the quadrant swaps / negates.
// Differs from glibc's sincos() on 1.4 % of arguments (never by more than one ulp).  ~80 instructions, no memory, few registers.
// Explicit fma() / mul / add in a fixed order, nothing left to contraction: the same operations on a CPU give the same bits -- the
// test suite's CPU checker has a mode that restates this routine, and then the kernels must equal it bit for bit
// (tests/test_gpu_parity.py).  (A 33-row table of sin / cos(n pi/16) with Taylor remainders was 0.23 % and fewer instructions, but
// its eight table values stayed live beside the ray state: the spills came back and with them the time -- LAB_NOTEBOOK R4.9.)
RTX_HD void sincos_2pi(double x, double *sn, double *cs)
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double fn = rint(x * invpio2);
    const double r1 = x - fn * pio2_1, w2 = fn * pio2_2;              // both exact (33-bit pieces, n <= 4; Sterbenz)
    const double r = r1 - w2;
    const double w = fn * pio2_2t - ((r1 - r) - w2);                  // the third piece minus what the difference above dropped
    const double y0 = r - w, y1 = (r - y0) - w;                       // the reduced angle and its tail, |y| <= pi/4
    const double z = y0 * y0, zl = fma(y0, y0, -z);                   // y0^2 = z + zl
    const double v = z * y0, vl = fma(z, y0, -v) + zl * y0;           // y0^3 = v + vl
    // sin y = y0 + [ y1 + (v + vl) S1 + v z (S2 + ...) - z y1 / 2 ]
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double sm = z * (v * rs - 0.5 * y1) + y1;
    sm = fma(vl, S1, sm);
    sm = fma(v, S1, sm);
    const double s = y0 + sm;
    // cos y = (1 - z/2) + [ rounding of that difference - zl/2 + z (z (C1 + ...)) - y0 y1 ]
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double h = 0.5 * z, w1 = 1.0 - h;
    double cm = fma(z, rc, -(y0 * y1));
    cm = fma(-0.5, zl, cm);
    cm = ((1.0 - w1) - h) + cm;
    const double c = w1 + cm;
    const int n = (int)fn & 3;
    *sn = n == 0 ? s : (n == 1 ? c : (n == 2 ? -s : -c));
    *cs = n == 0 ? c : (n == 1 ? -s : (n == 2 ? -c : s));
}

// f64::is_normal() && is_sign_positive()  (raytracing/scene.rs:249)
RTX_HD bool is_normal_positive(double d)
{
    return d >= 2.2250738585072014e-308 && d <= 1.7976931348623157e308;   // NaN fails both
}

}  // namespace rtx
