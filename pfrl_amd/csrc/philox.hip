// torch.randn on the device generator, restated: every factorised-noise draw of a NoisyNet pass
// (pfrl/nn/noisy_linear.py:52-60: one torch.normal(0, 1, size = in + out) per layer and forward
// pass) in ONE launch for all layers and passes of an update, bit for bit what the separate
// torch.randn calls would have written.
//
// What torch does for a float normal_ of `numel` elements (ATen/native/cuda/DistributionTemplates.h,
// distribution_elementwise_grid_stride_kernel<float, 4>; on ROCm curand is rocRAND):
//   grid = min(CUs * (max threads per CU / 256), ceil(numel / 256)) blocks of 256 threads, T threads
//   thread idx: Philox4x32-10 state (seed, subsequence = idx, offset = the generator's offset)
//   per round: n = rocrand_normal4(state) (Box-Muller on the four 32-bit outputs of one Philox
//   block); element idx + T * ii takes n[ii], ii = 0..3; rounds advance idx by 4 T
//   the generator's offset then grows by 4 * ceil(numel / (4 T)) (rounded up to a multiple of 4)
// The integer part is rocRAND's own engine (its header); the Box-Muller arithmetic is restated in
// the two forms a compiler may give it (with / without contraction of a * b + c into an fma):
// `variant` selects, tests/test_philox.py pins which one is torch's on this stack.
#include <rocrand/rocrand_philox4x32_10.h>

#include "common.h"

namespace {

constexpr int kMaxCalls = 16;

struct PhiloxCalls {
    unsigned long long offset[kMaxCalls];   // the generator's offset at each call
    long long numel[kMaxCalls];
    long long out_off[kMaxCalls];           // element offset of the call's output in `out`
    int grid[kMaxCalls];                    // blocks of 256 threads torch would launch
    int block_end[kMaxCalls];               // running total of `grid` (exclusive prefix ends)
    int n;
};

template <bool FMA>
__device__ __forceinline__ void box_muller2(unsigned int x, unsigned int y, float &a, float &b) {
    const float inv = 2.3283064e-10f, inv2pi = 1.46291807e-09f;   // ROCRAND_2POW32_INV[_2PI]
    float u, v;
    if (FMA) {
        u = __fmaf_rn((float)x, inv, inv);
        v = __fmaf_rn((float)y, inv2pi, inv2pi);
    } else {
        u = __fadd_rn(inv, __fmul_rn((float)x, inv));
        v = __fadd_rn(inv2pi, __fmul_rn((float)y, inv2pi));
    }
    const float s = sqrtf(__fmul_rn(-2.0f, logf(u)));
    float sn, cs;
    __sincosf(v, &sn, &cs);
    a = __fmul_rn(sn, s);
    b = __fmul_rn(cs, s);
}

template <bool FMA>
__global__ __launch_bounds__(256) void k_philox_normal(unsigned long long seed, PhiloxCalls c,
                                                       float *__restrict__ out) {
    int call = 0;
    while (call < c.n - 1 && (int)blockIdx.x >= c.block_end[call]) ++call;
    const int first = call == 0 ? 0 : c.block_end[call - 1];
    const long long T = (long long)c.grid[call] * 256;
    const long long idx = (long long)((int)blockIdx.x - first) * 256 + threadIdx.x;
    const long long numel = c.numel[call];
    float *dst = out + c.out_off[call];
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, (unsigned long long)idx, c.offset[call], &st);
    const long long rounded = ((numel - 1) / (T * 4) + 1) * T * 4;
    for (long long li = idx; li < rounded; li += T * 4) {
        const uint4 r = rocrand4(&st);
        float n[4];
        box_muller2<FMA>(r.x, r.y, n[0], n[1]);
        box_muller2<FMA>(r.z, r.w, n[2], n[3]);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const long long e = li + T * ii;
            // normal_(0, 1): rand * std + mean in float
            if (e < numel) dst[e] = __fadd_rn(__fmul_rn(n[ii], 1.0f), 0.0f);
        }
    }
}

}  // namespace

extern "C" int pfrl_philox_normal(uint64_t seed, uint64_t base_offset, int32_t n_calls,
                                  const uint64_t *offsets, const int64_t *numel,
                                  const int64_t *out_offsets, const int32_t *grids, float *out,
                                  int32_t variant, void *stream) {
    PFRL_CHECK_ARG(n_calls >= 1 && n_calls <= kMaxCalls && offsets && numel && out_offsets && grids && out,
                   "pfrl_philox_normal: 1..16 calls");
    PhiloxCalls c;
    int blocks = 0;
    for (int i = 0; i < kMaxCalls; ++i) {
        const int j = i < n_calls ? i : 0;
        PFRL_CHECK_ARG(numel[j] >= 1 && grids[j] >= 1, "pfrl_philox_normal: empty call");
        c.offset[i] = base_offset + offsets[j];
        c.numel[i] = numel[j];
        c.out_off[i] = out_offsets[j];
        c.grid[i] = grids[j];
        if (i < n_calls) blocks += grids[j];
        c.block_end[i] = blocks;
    }
    c.n = n_calls;
    if (variant)
        hipLaunchKernelGGL(k_philox_normal<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed, c, out);
    else
        hipLaunchKernelGGL(k_philox_normal<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed, c, out);
    PFRL_LAUNCH_CHECK();
}
