// glibc's single-precision power function, restated so that the DEVICE computes the very
// number `np.float32(x) ** alpha` yields on the host (NumPy's scalar power calls libm's
// powf): pfrl/replay_buffers/prioritized.py:47-55 is `(clip(error) + eps) ** alpha` on
// np.float32 scalars, and one leaf that differs by one ulp changes every later prefix sum of
// the priority tree.
//
// Third-party algorithm (not vendored by the reference): GNU libc 2.35 (Ubuntu 22.04, the
// libm this image and the GPU boxes run) sysdeps/ieee754/flt-32/e_powf.c with the tables of
// powf_log2_data.c and exp2f_data.c -- Szabolcs Nagy's powf from ARM's optimized-routines
// (log2 of x through a 16-entry table and a degree-5 polynomial in double precision, times y,
// exp2 through a 32-entry table and a cubic; TOINT_INTRINSICS = 0, POWF_SCALE = 1 on x86-64).
// The constants below are the published table values (checked against the bytes of the
// installed libm.so.6, and the whole function against libm's powf on every float32 in
// [2^-7, 2) for several exponents: tests/test_powf_glibc.py).
//
// powf is not correctly rounded and x86-64 glibc ships TWO builds of this code behind an
// ifunc: `__powf_sse2` (separate multiply and add) and `__powf_fma` (-mfma: every a*b+c is one
// fused operation).  They differ in the last bit for a small fraction of inputs, so both are
// restated (template parameter FMA) and the host picks the one its libm uses
// (pfrl_powf_host_variant, hostplan.hip).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define PFRL_POWF_HD __host__ __device__ __forceinline__
#else
#define PFRL_POWF_HD __host__ __device__ inline
#endif

namespace pfrl_powf {

struct Log2Entry {
    double invc, logc;
};

// __powf_log2_data.tab / .poly (powf_log2_data.c)
__device__ __constant__ static const Log2Entry kLog2TabDev[16] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
static const Log2Entry kLog2TabHost[16] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};

// __exp2f_data.tab (exp2f_data.c): tab[i] = bits(2^(i/32)) - (i << 47)
#define PFRL_EXP2F_TAB                                                                             \
    {0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,   \
     0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,   \
     0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,   \
     0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,   \
     0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,   \
     0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,   \
     0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,   \
     0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull}
__device__ __constant__ static const uint64_t kExp2TabDev[32] = PFRL_EXP2F_TAB;
static const uint64_t kExp2TabHost[32] = PFRL_EXP2F_TAB;

PFRL_POWF_HD uint32_t as_u32(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
PFRL_POWF_HD float as_f32(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
PFRL_POWF_HD uint64_t as_u64(double f) {
    uint64_t u;
    memcpy(&u, &f, 8);
    return u;
}
PFRL_POWF_HD double as_f64(uint64_t u) {
    double f;
    memcpy(&f, &u, 8);
    return f;
}

// one rounding per call in both forms; the translation units are built with
// -ffp-contract=off, so `a * b + c` below stays two operations
template <bool FMA>
PFRL_POWF_HD double madd(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return FMA ? __fma_rn(a, b, c) : __dadd_rn(__dmul_rn(a, b), c);
#else
    if (FMA) return __builtin_fma(a, b, c);
    volatile double p = a * b;   // (volatile: no contraction whatever the flags)
    return p + c;
#endif
}

PFRL_POWF_HD const Log2Entry &log2_entry(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return kLog2TabDev[i];
#else
    return kLog2TabHost[i];
#endif
}
PFRL_POWF_HD uint64_t exp2_entry(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return kExp2TabDev[i];
#else
    return kExp2TabHost[i];
#endif
}

// e_powf.c log2_inline: x = 2^k z, z in [OFF, 2 OFF); log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k
template <bool FMA>
PFRL_POWF_HD double log2_inline(uint32_t ix) {
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp0;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;   // arithmetic shift
    const double invc = log2_entry(i).invc, logc = log2_entry(i).logc;
    const double z = (double)as_f32(iz);
    const double r = madd<FMA>(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = madd<FMA>(A0, r, A1);
    const double p = madd<FMA>(A2, r, A3);
    const double r4 = r2 * r2;
    double q = madd<FMA>(A4, r, y0);
    q = madd<FMA>(p, r2, q);
    y = madd<FMA>(y, r4, q);
    return y;
}

// e_powf.c exp2_inline: x = k/N + r, 2^x = 2^(k/N) * (C0 r^3 + C1 r^2 + C2 r + 1)
template <bool FMA>
PFRL_POWF_HD double exp2_inline(double xd, uint32_t sign_bias) {
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    const double SHIFT = 0x1.8p+52 / 32;
    double kd = xd + SHIFT;
    const uint64_t ki = as_u64(kd);
    kd -= SHIFT;
    const double r = xd - kd;
    uint64_t t = exp2_entry((int)(ki % 32));
    const uint64_t ski = ki + sign_bias;
    t += ski << (52 - 5);
    const double s = as_f64(t);
    const double z = madd<FMA>(C0, r, C1);
    const double r2 = r * r;
    double y = madd<FMA>(C2, r, 1.0);
    y = madd<FMA>(z, r2, y);
    return y * s;
}

PFRL_POWF_HD int zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }

// 0: not an integer, 1: odd integer, 2: even integer
PFRL_POWF_HD int checkint(uint32_t iy) {
    const int e = iy >> 23 & 0xff;
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}

// e_powf.c __powf (round-to-nearest; errno / exception flags are not modelled)
template <bool FMA>
PFRL_POWF_HD float powf_glibc(float x, float y) {
    uint32_t sign_bias = 0;
    uint32_t ix = as_u32(x);
    const uint32_t iy = as_u32(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || zeroinfnan(iy)) {
        if (zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
            if (2 * ix == 2 * 0x3f800000u) return 1.0f;
            if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1 / x2 : x2;
        }
        if (ix & 0x80000000u) {
            const int yint = checkint(iy);
            if (yint == 0) return as_f32(0x7fc00000u);
            if (yint == 1) sign_bias = 1u << (5 + 11);
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {
            ix = as_u32(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    const double logx = log2_inline<FMA>(ix);
    const double ylogx = (double)y * logx;
    if ((as_u64(ylogx) >> 47 & 0xffff) >= (as_u64(126.0) >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return as_f32(sign_bias ? 0xff800000u : 0x7f800000u);
        if (ylogx <= -150.0) return as_f32(sign_bias ? 0x80000000u : 0u);
    }
    return (float)exp2_inline<FMA>(ylogx, sign_bias);
}

}  // namespace pfrl_powf
