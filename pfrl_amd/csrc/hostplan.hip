// Host-side native helpers of the data path (no kernels in this file): the probe that picks
// the device restatement of libm's powf, and the step planner that walks the reference's
// NumPy draws for a whole batched env step.
#include <math.h>
#include <stdint.h>

#include "common.h"
#include "powf_glibc.h"

extern "C" int pfrl_powf_host(int pow_mode, const float *host_x, float alpha, float *host_out,
                              int64_t n) {
    PFRL_CHECK_ARG(host_x && host_out && n >= 0, "pfrl_powf_host: null argument");
    PFRL_CHECK_ARG(pow_mode == PFRL_POW_GLIBC || pow_mode == PFRL_POW_GLIBC_FMA,
                   "pfrl_powf_host: pow_mode must be PFRL_POW_GLIBC or PFRL_POW_GLIBC_FMA");
    if (pow_mode == PFRL_POW_GLIBC_FMA)
        for (int64_t i = 0; i < n; ++i) host_out[i] = pfrl_powf::powf_glibc<true>(host_x[i], alpha);
    else
        for (int64_t i = 0; i < n; ++i) host_out[i] = pfrl_powf::powf_glibc<false>(host_x[i], alpha);
    return 0;
}

extern "C" int pfrl_powf_host_variant(float alpha, int64_t n_probe) {
    // inputs: a 64-bit LCG over the float32 values of (2^-20, 2) -- the priority transform's
    // domain is (eps, error_max + eps] -- with the caller's alpha and a few common ones
    const float alphas[] = {alpha, 0.5f, 0.6f, 0.7f};
    bool ok_plain = true, ok_fma = true;
    uint64_t state = 0x9E3779B97F4A7C15ull;
    for (int64_t i = 0; i < n_probe && (ok_plain || ok_fma); ++i) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t bits = 0x35800000u + (uint32_t)((state >> 33) % (0x40000000u - 0x35800000u));
        const float x = pfrl_powf::as_f32(bits);
        const float a = alphas[i & 3];
        volatile float xv = x, av = a;          // keep the call to libm (no constant folding)
        const uint32_t want = pfrl_powf::as_u32(powf(xv, av));
        if (ok_plain && pfrl_powf::as_u32(pfrl_powf::powf_glibc<false>(x, a)) != want) ok_plain = false;
        if (ok_fma && pfrl_powf::as_u32(pfrl_powf::powf_glibc<true>(x, a)) != want) ok_fma = false;
    }
    if (ok_fma) return PFRL_POW_GLIBC_FMA;
    if (ok_plain) return PFRL_POW_GLIBC;
    return -1;
}
