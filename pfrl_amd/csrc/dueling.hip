// Distributional dueling head (pfrl/q_functions/dueling_dqn.py:116-127): mean over
// actions, advantage centring, state-value broadcast add and the softmax over
// atoms -- forward in one launch instead of five, backward in one instead of ~8.
//
//   logits[b][a][z] = (ya[b][a][z] - (sum_a' ya[b][a'][z]) / A) + ys[b][z]
//   q[b][a][.]      = softmax_z(logits[b][a][.])
//
// One 64-lane wave per sample, lane = atom (n_atoms <= 64), four samples per
// workgroup; no cross-workgroup traffic.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWavesPerBlock = kThreads / 64;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(kThreads) void k_dueling_softmax_fwd(
    const float *__restrict__ ya, const float *__restrict__ ys, float *__restrict__ q, int64_t B,
    int A, int Z) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (b >= B) return;
    const bool on = lane < Z;
    const int zl = min(lane, Z - 1);
    const float *row = ya + b * A * Z;
    const float v = ys[b * Z + zl];
    float colsum = 0.0f;
    for (int a0 = 0; a0 < A; a0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = row[min(a0 + u, A - 1) * Z + zl];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (a0 + u < A) colsum += t[u];
    }
    const float mean = colsum / (float)A;
    float *out = q + b * A * Z;
    for (int a0 = 0; a0 < A; a0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = row[min(a0 + u, A - 1) * Z + zl];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (a0 + u < A) {
                const float x = (t[u] - mean) + v;
                const float m = wmax(on ? x : -INFINITY);
                const float e = on ? expf(x - m) : 0.0f;
                const float s = wsum(e);
                if (on) out[(a0 + u) * Z + lane] = e / s;
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_dueling_softmax_bwd(
    const float *__restrict__ gq, const float *__restrict__ q, float *__restrict__ g_ya,
    float *__restrict__ g_ys, int64_t B, int A, int Z) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (b >= B) return;
    const bool on = lane < Z;
    const int zl = min(lane, Z - 1);
    const float *gr = gq + b * A * Z;
    const float *qr = q + b * A * Z;
    float *out = g_ya + b * A * Z;
    // pass 1: softmax backward per action, column sums over actions; the per-action
    // logit gradients are parked in g_ya and centred in pass 2
    float colsum = 0.0f;
    for (int a0 = 0; a0 < A; a0 += 8) {
        float g[8], p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int aa = min(a0 + u, A - 1);
            g[u] = gr[aa * Z + zl];
            p[u] = qr[aa * Z + zl];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (a0 + u < A) {
                const float dot = wsum(on ? g[u] * p[u] : 0.0f);
                const float gl = on ? p[u] * (g[u] - dot) : 0.0f;
                colsum += gl;
                if (on) out[(a0 + u) * Z + lane] = gl;
            }
        }
    }
    if (on) g_ys[b * Z + lane] = colsum;
    const float shift = colsum / (float)A;
    for (int a = 0; a < A; ++a)
        if (on) out[a * Z + lane] -= shift;   // same lane wrote it: no hazard
}

}  // namespace

extern "C" int pfrl_dueling_softmax_fwd(const float *ya, const float *ys, float *q, int64_t B,
                                        int32_t A, int32_t Z, void *stream) {
    PFRL_CHECK_ARG(A > 0 && Z >= 1 && Z <= 64, "pfrl_dueling_softmax_fwd: n_atoms must be <= 64");
    if (B <= 0) return 0;
    const unsigned blocks = (unsigned)((B + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(k_dueling_softmax_fwd, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream,
                       ya, ys, q, B, (int)A, (int)Z);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_dueling_softmax_bwd(const float *gq, const float *q, float *g_ya, float *g_ys,
                                        int64_t B, int32_t A, int32_t Z, void *stream) {
    PFRL_CHECK_ARG(A > 0 && Z >= 1 && Z <= 64, "pfrl_dueling_softmax_bwd: n_atoms must be <= 64");
    if (B <= 0) return 0;
    const unsigned blocks = (unsigned)((B + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(k_dueling_softmax_bwd, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream,
                       gq, q, g_ya, g_ys, B, (int)A, (int)Z);
    PFRL_LAUNCH_CHECK();
}
