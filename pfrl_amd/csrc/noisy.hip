// Factorised NoisyNet weights (pfrl/nn/noisy_linear.py:52-70), forward and backward.
//
// The reference builds the perturbed weights of every noisy layer, every
// forward pass, out of eight pointwise launches (abs, sqrt, abs, sign, mul for
// f(r) = sign(r) sqrt|r|; ger; two addcmul).  Rainbow runs nine noisy forwards per
// update at minibatch 32, so those ~70 launches of 4-5 us each cost more than
// the GEMMs they feed.  Here:
//   forward   W[o][i] = mu_W + sigma_W * (f(r[in+o]) * f(r[i])),  b[o] = mu_b + sigma_b * f(r[in+o])
//   backward  g_sigma_W = g_W * (f_o * f_i),  g_sigma_b = g_b * f_o   (g_mu_* = g_* pass through)
// one launch each, HBM-bound: 12 B (fwd) / 12 B (bwd) per weight.
#include "common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float shaped(float r) {
    // |sqrt(|r|)| * sign(r), sign(0) = 0
    const float s = sqrtf(fabsf(r));
    return r > 0.0f ? s : (r < 0.0f ? -s : r * 0.0f);
}

// VEC = 4: in_features % 4 == 0 and 16-byte aligned rows; VEC = 1 otherwise.
template <int VEC>
__global__ __launch_bounds__(kThreads) void k_noisy_fwd(
    const float *__restrict__ mu_w, const float *__restrict__ sigma_w,
    const float *__restrict__ mu_b, const float *__restrict__ sigma_b,
    const float *__restrict__ r, float *__restrict__ w, float *__restrict__ b, int64_t out_f,
    int64_t in_f) {
    const int64_t per_row = in_f / VEC;
    const int64_t n_w = out_f * per_row;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t it = (int64_t)blockIdx.x * kThreads + threadIdx.x; it < n_w + out_f;
         it += stride) {
        if (it >= n_w) {
            if (b != nullptr) {
                const int64_t o = it - n_w;
                b[o] = mu_b[o] + sigma_b[o] * shaped(r[in_f + o]);
            }
            continue;
        }
        const int64_t o = it / per_row;
        const int64_t i = (it - o * per_row) * VEC;
        const float fo = shaped(r[in_f + o]);
        if constexpr (VEC == 4) {
            const float4 m = *reinterpret_cast<const float4 *>(mu_w + o * in_f + i);
            const float4 s = *reinterpret_cast<const float4 *>(sigma_w + o * in_f + i);
            const float4 ri = *reinterpret_cast<const float4 *>(r + i);
            float4 v;
            v.x = m.x + s.x * (fo * shaped(ri.x));
            v.y = m.y + s.y * (fo * shaped(ri.y));
            v.z = m.z + s.z * (fo * shaped(ri.z));
            v.w = m.w + s.w * (fo * shaped(ri.w));
            *reinterpret_cast<float4 *>(w + o * in_f + i) = v;
        } else {
            w[o * in_f + i] = mu_w[o * in_f + i] + sigma_w[o * in_f + i] * (fo * shaped(r[i]));
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void k_noisy_bwd(
    const float *__restrict__ g_w, const float *__restrict__ g_b, const float *__restrict__ r,
    float *__restrict__ g_sigma_w, float *__restrict__ g_sigma_b, int64_t out_f, int64_t in_f) {
    const int64_t per_row = in_f / VEC;
    const int64_t n_w = out_f * per_row;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t it = (int64_t)blockIdx.x * kThreads + threadIdx.x; it < n_w + out_f;
         it += stride) {
        if (it >= n_w) {
            if (g_sigma_b != nullptr) {
                const int64_t o = it - n_w;
                g_sigma_b[o] = g_b[o] * shaped(r[in_f + o]);
            }
            continue;
        }
        const int64_t o = it / per_row;
        const int64_t i = (it - o * per_row) * VEC;
        const float fo = shaped(r[in_f + o]);
        if constexpr (VEC == 4) {
            const float4 g = *reinterpret_cast<const float4 *>(g_w + o * in_f + i);
            const float4 ri = *reinterpret_cast<const float4 *>(r + i);
            float4 v;
            v.x = g.x * (fo * shaped(ri.x));
            v.y = g.y * (fo * shaped(ri.y));
            v.z = g.z * (fo * shaped(ri.z));
            v.w = g.w * (fo * shaped(ri.w));
            *reinterpret_cast<float4 *>(g_sigma_w + o * in_f + i) = v;
        } else {
            g_sigma_w[o * in_f + i] = g_w[o * in_f + i] * (fo * shaped(r[i]));
        }
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline unsigned grid_for(int64_t items) {
    int64_t blocks = (items + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" int pfrl_noisy_weights_fwd(const float *mu_w, const float *sigma_w, const float *mu_b,
                                      const float *sigma_b, const float *r, float *w_out,
                                      float *b_out, int64_t out_features, int64_t in_features,
                                      void *stream) {
    PFRL_CHECK_ARG(out_features > 0 && in_features > 0, "pfrl_noisy_weights_fwd: empty layer");
    PFRL_CHECK_ARG((b_out == nullptr) || (mu_b != nullptr && sigma_b != nullptr),
                   "pfrl_noisy_weights_fwd: bias output without bias parameters");
    const bool v4 = (in_features & 3) == 0 && aligned16(mu_w) && aligned16(sigma_w) &&
                    aligned16(r) && aligned16(w_out);
    const int64_t items = out_features * (in_features / (v4 ? 4 : 1)) + out_features;
    if (v4)
        hipLaunchKernelGGL(k_noisy_fwd<4>, dim3(grid_for(items)), dim3(kThreads), 0,
                           (hipStream_t)stream, mu_w, sigma_w, mu_b, sigma_b, r, w_out, b_out,
                           out_features, in_features);
    else
        hipLaunchKernelGGL(k_noisy_fwd<1>, dim3(grid_for(items)), dim3(kThreads), 0,
                           (hipStream_t)stream, mu_w, sigma_w, mu_b, sigma_b, r, w_out, b_out,
                           out_features, in_features);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_noisy_weights_bwd(const float *g_w, const float *g_b, const float *r,
                                      float *g_sigma_w, float *g_sigma_b, int64_t out_features,
                                      int64_t in_features, void *stream) {
    PFRL_CHECK_ARG(out_features > 0 && in_features > 0, "pfrl_noisy_weights_bwd: empty layer");
    PFRL_CHECK_ARG((g_sigma_b == nullptr) || (g_b != nullptr),
                   "pfrl_noisy_weights_bwd: bias gradient output without its input");
    const bool v4 = (in_features & 3) == 0 && aligned16(g_w) && aligned16(r) &&
                    aligned16(g_sigma_w);
    const int64_t items = out_features * (in_features / (v4 ? 4 : 1)) + out_features;
    if (v4)
        hipLaunchKernelGGL(k_noisy_bwd<4>, dim3(grid_for(items)), dim3(kThreads), 0,
                           (hipStream_t)stream, g_w, g_b, r, g_sigma_w, g_sigma_b, out_features,
                           in_features);
    else
        hipLaunchKernelGGL(k_noisy_bwd<1>, dim3(grid_for(items)), dim3(kThreads), 0,
                           (hipStream_t)stream, g_w, g_b, r, g_sigma_w, g_sigma_b, out_features,
                           in_features);
    PFRL_LAUNCH_CHECK();
}
