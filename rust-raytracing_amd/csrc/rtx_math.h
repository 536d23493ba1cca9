// rtx_math.h -- f64 Vector3 arithmetic in the reference's operation order, usable from host and
// device code.  Every function is the inlined form of one src/math item of the reference
// (cited per function).  The translation units that include this header are compiled with
// -ffp-contract=off: Rust never fuses a*b+c, so neither may these (the results of the exact
// path must have the reference's roundings).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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

// f64::is_normal() && is_sign_positive()  (raytracing/scene.rs:249)
RTX_HD bool is_normal_positive(double d)
{
    return d >= 2.2250738585072014e-308 && d <= 1.7976931348623157e308;   // NaN fails both
}

}  // namespace rtx
