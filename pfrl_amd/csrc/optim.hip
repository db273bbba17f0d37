// Fused multi-tensor RMSprop step (centered or not, no momentum): ONE launch
// updates every parameter of the network instead of the ~6 foreach kernels x
// parameter-list chunks PyTorch issues (73 us -> a few us per DQN update at
// 1.69 M parameters).  HBM-bound elementwise: reads p, g, square_avg,
// (grad_avg) and writes p, square_avg, (grad_avg): 28 B per element centered.
// Same arithmetic, in the same order, as torch.optim.RMSprop's foreach path
// (torch/optim/rmsprop.py _multi_tensor_rmsprop): every step below is one
// rounded f32 operation.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPerThread = 4;
constexpr int kChunk = kThreads * kPerThread;

struct RmsArgs {
    float *p[PFRL_OPT_MAX_TENSORS];
    const float *g[PFRL_OPT_MAX_TENSORS];
    float *sq[PFRL_OPT_MAX_TENSORS];
    float *ga[PFRL_OPT_MAX_TENSORS];
    int64_t numel[PFRL_OPT_MAX_TENSORS];
    int32_t chunk_end[PFRL_OPT_MAX_TENSORS];  // exclusive prefix of chunks
    int32_t n;
};

template <bool CENTERED>
__global__ __launch_bounds__(kThreads) void k_rmsprop(RmsArgs a, float lr, float alpha, float eps,
                                                      float weight_decay) {
    int t = 0;
    const int b = blockIdx.x;
    while (t < a.n - 1 && b >= a.chunk_end[t]) ++t;
    const int first = t == 0 ? 0 : a.chunk_end[t - 1];
    const int64_t base = (int64_t)(b - first) * kChunk;
    float *__restrict__ p = a.p[t];
    const float *__restrict__ g = a.g[t];
    float *__restrict__ sq = a.sq[t];
    float *__restrict__ ga = a.ga[t];
    const int64_t n = a.numel[t];
    const float oma = __fsub_rn(1.0f, alpha);
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
        const int64_t i = base + u * kThreads + threadIdx.x;
        if (i >= n) break;
        float gi = g[i];
        if (weight_decay != 0.0f) gi = __fadd_rn(gi, __fmul_rn(weight_decay, p[i]));
        // square_avg.mul_(alpha).addcmul_(grad, grad, value=1 - alpha)
        float s = __fmul_rn(sq[i], alpha);
        s = __fadd_rn(s, __fmul_rn(oma, __fmul_rn(gi, gi)));
        sq[i] = s;
        float avg;
        if (CENTERED) {
            // grad_avg.lerp_(grad, 1 - alpha)
            float m = ga[i];
            m = __fadd_rn(m, __fmul_rn(oma, __fsub_rn(gi, m)));
            ga[i] = m;
            // addcmul(square_avg, grad_avg, grad_avg, value=-1).sqrt_().add_(eps)
            avg = __fadd_rn(__fsqrt_rn(__fsub_rn(s, __fmul_rn(m, m))), eps);
        } else {
            avg = __fadd_rn(__fsqrt_rn(s), eps);
        }
        // param.addcdiv_(grad, avg, value=-lr)
        p[i] = __fadd_rn(p[i], __fmul_rn(-lr, __fdiv_rn(gi, avg)));
    }
}

}  // namespace

extern "C" int pfrl_rmsprop_step(int32_t n_tensors, float *const *params, const float *const *grads,
                                 float *const *square_avg, float *const *grad_avg,
                                 const int64_t *numel, float lr, float alpha, float eps,
                                 float weight_decay, int centered, void *stream) {
    PFRL_CHECK_ARG(n_tensors >= 0, "pfrl_rmsprop_step: bad tensor count");
    for (int lo = 0; lo < n_tensors; lo += PFRL_OPT_MAX_TENSORS) {
        const int n = (n_tensors - lo) < PFRL_OPT_MAX_TENSORS ? (n_tensors - lo)
                                                              : PFRL_OPT_MAX_TENSORS;
        RmsArgs a;
        int chunks = 0;
        for (int t = 0; t < n; ++t) {
            a.p[t] = params[lo + t];
            a.g[t] = grads[lo + t];
            a.sq[t] = square_avg[lo + t];
            a.ga[t] = centered ? grad_avg[lo + t] : nullptr;
            a.numel[t] = numel[lo + t];
            chunks += (int)((numel[lo + t] + kChunk - 1) / kChunk);
            a.chunk_end[t] = chunks;
        }
        a.n = n;
        if (chunks == 0) continue;
        if (centered)
            hipLaunchKernelGGL(k_rmsprop<true>, dim3(chunks), dim3(kThreads), 0, (hipStream_t)stream,
                               a, lr, alpha, eps, weight_decay);
        else
            hipLaunchKernelGGL(k_rmsprop<false>, dim3(chunks), dim3(kThreads), 0,
                               (hipStream_t)stream, a, lr, alpha, eps, weight_decay);
    }
    PFRL_LAUNCH_CHECK();
}
