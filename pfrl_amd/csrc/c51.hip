// Categorical (C51) DQN loss, fused: greedy next action, Bellman shift of the
// support, categorical projection (Algorithm 1 of arXiv:1707.06887), cross
// entropy against the online distribution, its gradient, the per-sample KL
// priorities and Q(s, a) -- one launch instead of ~45 pointwise / scatter
// launches (pfrl/agents/categorical_dqn.py:7-57,60-104,150-204,
// categorical_double_dqn.py:10-52).
//
// One 64-lane wave per sample, lane = atom (n_atoms <= 64).  The projection is
// a deterministic gather: target bin k sums the contributions of all source
// atoms j in increasing j (the reference scatter-adds with atomics in arbitrary
// order; sums agree to fp32 rounding).
#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxBatch = 4096;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(kThreads) void k_c51_loss(
    const float *__restrict__ q_dist, const int64_t *__restrict__ action,
    const float *__restrict__ next_dist, const float *__restrict__ next_select,
    const float *__restrict__ z_values, const float *__restrict__ reward,
    const float *__restrict__ discount, const float *__restrict__ terminal,
    const float *__restrict__ weights, int B, int A, int Z, int mean, float *__restrict__ out_loss,
    float *__restrict__ out_grad, float *__restrict__ out_q, float *__restrict__ out_delta) {
    __shared__ float s_wl[kWaves][64];
    __shared__ float s_wu[kWaves][64];
    __shared__ int s_lo[kWaves][64];
    __shared__ int s_up[kWaves][64];
    __shared__ float s_part[kMaxBatch];
    const int lane = threadIdx.x & 63;
    // (in an SGPR: the loop over this wave's samples and their scalar operands become uniform)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool on = lane < Z;
    const float z = on ? z_values[lane] : 0.0f;
    const float v_min = z_values[0];
    const float v_max = z_values[Z - 1];
    const float delta_z = __fsub_rn(z_values[1], z_values[0]);
    const float inv_b = 1.0f / (float)B;
    for (int b = wave; b < B; b += kWaves) {
        // Everything this sample needs from HBM is requested up front (the loads are
        // independent; issued one after another they would each cost a full memory
        // round trip): the taken action's online distribution, and per action the
        // selection distribution plus the target distribution, eight actions at a time.
        const int64_t act = action[b];
        const int zl = min(lane, Z - 1);
        const float y_ = q_dist[((int64_t)b * A + act) * Z + zl];
        const float y = on ? y_ : 1.0f;
        const float rew = reward[b], term = terminal[b], disc = discount[b];
        const float *sel = next_select + (int64_t)b * A * Z;
        const float *nd = next_dist + (int64_t)b * A * Z;
        // greedy next action under next_select (first maximum, like torch.argmax);
        // the target distribution of the running best is carried along
        float best = 0.0f, p = 0.0f;
        for (int a0 = 0; a0 < A; a0 += 8) {
            float sv[8], pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // unconditional clamped loads keep all 16 in one basic block (in flight
                // together); out-of-range lanes / actions are zeroed afterwards
                const int aa = min(a0 + u, A - 1);
                const float s_ = sel[aa * Z + zl];
                const float p_ = nd[aa * Z + zl];
                sv[u] = on ? s_ : 0.0f;
                pv[u] = on ? p_ : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (a0 + u < A) {
                    const float qv = wave_sum(sv[u] * z);
                    if ((a0 + u) == 0 || qv > best) {
                        best = qv;
                        p = pv[u];
                    }
                }
            }
        }
        // Bellman shift + projection weights of source atom j = lane
        const float scale = __fmul_rn(__fsub_rn(1.0f, term), disc);
        float tz = __fadd_rn(rew, __fmul_rn(scale, z));
        tz = fminf(fmaxf(tz, v_min), v_max);
        float bj = __fdiv_rn(__fsub_rn(tz, v_min), delta_z);
        bj = fminf(fmaxf(bj, 0.0f), (float)(Z - 1));
        const float lo = floorf(bj), up = ceilf(bj);
        const float frac = __fsub_rn(bj, lo);
        s_lo[wave][lane] = on ? (int)lo : -1;
        s_up[wave][lane] = on ? (int)up : -1;
        s_wl[wave][lane] = __fmul_rn(p, __fsub_rn(1.0f, frac));
        s_wu[wave][lane] = __fmul_rn(p, frac);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float t = 0.0f;
        for (int j = 0; j < Z; ++j) {
            if (s_lo[wave][j] == lane) t += s_wl[wave][j];
            if (s_up[wave][j] == lane) t += s_wu[wave][j];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // cross entropy against the online distribution of the taken action
        const float yc = fminf(fmaxf(y, 1e-10f), 1.0f);
        const float el = on ? -t * logf(yc) : 0.0f;
        const float d = wave_sum(el);
        const float qsa = wave_sum(on ? y * z : 0.0f);
        float coef = weights != nullptr ? weights[b] : 1.0f;
        if (mean) coef *= inv_b;
        // d/dy of -t*log(clamp(y)): clamp passes the gradient on [1e-10, 1]
        const float gy = (y >= 1e-10f && y <= 1.0f) ? -t / yc * coef : 0.0f;
        if (on) {
            float *grow = out_grad + (int64_t)b * A * Z;
            for (int a = 0; a < A; ++a) grow[a * Z + lane] = (a == act) ? gy : 0.0f;
        }
        if (lane == 0) {
            out_delta[b] = d;
            out_q[b] = qsa;
            s_part[b] = d * coef;
        }
    }
    __syncthreads();
    if (wave == 0) {
        float acc = 0.0f;
        for (int b = lane; b < B; b += 64) acc += s_part[b];
        acc = wave_sum(acc);
        if (lane == 0) out_loss[0] = acc;
    }
}

}  // namespace

extern "C" int pfrl_c51_loss(const float *q_dist, const int64_t *action, const float *next_dist,
                             const float *next_select, const float *z_values,
                             const float *reward, const float *discount, const float *terminal,
                             const float *weights, int32_t B, int32_t A, int32_t Z, int32_t mean,
                             float *out_loss, float *out_grad_q, float *out_q, float *out_delta,
                             void *stream) {
    PFRL_CHECK_ARG(B > 0 && B <= kMaxBatch, "pfrl_c51_loss: batch size must be in [1, 4096]");
    PFRL_CHECK_ARG(A > 0, "pfrl_c51_loss: no actions");
    PFRL_CHECK_ARG(Z >= 2 && Z <= 64, "pfrl_c51_loss: n_atoms must be in [2, 64]");
    if (next_select == nullptr) next_select = next_dist;
    hipLaunchKernelGGL(k_c51_loss, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, q_dist, action,
                       next_dist, next_select, z_values, reward, discount, terminal, weights,
                       (int)B, (int)A, (int)Z, (int)mean, out_loss, out_grad_q, out_q, out_delta);
    PFRL_LAUNCH_CHECK();
}
