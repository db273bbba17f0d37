// Actor-critic update helpers for gfx950: the stretches of the SAC / TD3 / DDPG update
// (pfrl/agents/soft_actor_critic.py:213-330) that PyTorch issues as dozens of 256-element
// elementwise launches, each ~4 us inside a captured graph whatever its size.
//
//   squashed Gaussian   a = tanh(loc + eps * scale) and log pi(a) of the example policy head
//                       (TransformedDistribution(Independent(Normal), [TanhTransform]),
//                       examples/mujoco/reproduction/soft_actor_critic/
//                       train_soft_actor_critic.py:128-141): 23 launches -> 1, backward 45 -> 1
//   soft target update  theta' <- (1 - tau) theta' + tau theta (pfrl/utils/copy_param.py:10-28)
//                       for every tensor of several networks in one launch
//   Adam                torch.optim.Adam's step for all parameters of an optimizer in one
//                       launch, the step counters advanced by the last workgroup to finish
//
// All HBM / latency bound elementwise work; every arithmetic step is one rounded f32
// operation in the order of the PyTorch code it replaces (built with -ffp-contract=off).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------
// tanh-squashed diagonal Gaussian: sample and log-probability, one wave per row
// ---------------------------------------------------------------------------------
// torch.distributions arithmetic (Normal.log_prob, TanhTransform.log_abs_det_jacobian,
// TransformedDistribution.log_prob with the transform's cached pre-image x):
//   x = loc + eps * scale;  a = tanh(x)
//   log N(x) = -((x - loc)^2) / (2 scale^2) - log(scale) - log(sqrt(2 pi))
//   ladj(x)  = 2 (log 2 - x - softplus(-2 x))
//   log pi   = (0 - sum_a ladj) + sum_a log N
__global__ __launch_bounds__(kThreads) void k_squashed_gaussian_fwd(
    const float *__restrict__ loc, int64_t ld_loc, const float *__restrict__ scale, int64_t ld_scale,
    const float *__restrict__ eps, float *__restrict__ action, float *__restrict__ logp,
    float *__restrict__ neg_logp, int B, int A) {
    const int row = blockIdx.x * (kThreads / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float s_ladj = 0.f, s_nlp = 0.f;
    for (int a = lane; a < A; a += 64) {
        const float l = loc[(int64_t)row * ld_loc + a], s = scale[(int64_t)row * ld_scale + a];
        const float e = eps[(int64_t)row * A + a];
        const float x = l + e * s;
        action[(int64_t)row * A + a] = tanhf(x);
        const float d = x - l;
        const float nlp = -(d * d) / (2.f * (s * s)) - logf(s) - 0.9189385332046727f;
        const float z = -2.f * x;
        const float sp = z > 20.f ? z : log1pf(expf(z));
        s_ladj += 2.f * (0.6931471805599453f - x - sp);
        s_nlp += nlp;
    }
    s_ladj = wave_sum(s_ladj);
    s_nlp = wave_sum(s_nlp);
    if (lane == 0) {
        const float lp = (0.f - s_ladj) + s_nlp;
        logp[row] = lp;
        if (neg_logp != nullptr) neg_logp[row] = -lp;   // the entropy estimate the agents record
    }
}

// Gradients w.r.t. loc and scale given dL/da (g_action, may be null) and dL/dlog pi
// (g_logp, may be null).  With y = tanh(x):  da/dloc = 1 - y^2, da/dscale = (1 - y^2) eps,
// dlogpi/dloc = 2 y (the Normal term cancels between x and loc),
// dlogpi/dscale = 2 y eps - 1 / scale.
__global__ __launch_bounds__(kThreads) void k_squashed_gaussian_bwd(
    const float *__restrict__ g_action, const float *__restrict__ g_logp,
    const float *__restrict__ action, const float *__restrict__ eps, const float *__restrict__ scale,
    int64_t ld_scale, float *__restrict__ g_loc, float *__restrict__ g_scale, int B, int A) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (int64_t)B * A) return;
    const int row = (int)(i / A), a = (int)(i - (int64_t)row * A);
    const float y = action[i], e = eps[i], s = scale[(int64_t)row * ld_scale + a];
    const float ga = g_action != nullptr ? g_action[i] : 0.f;
    const float gl = g_logp != nullptr ? g_logp[row] : 0.f;
    const float t = ga * (1.f - y * y);
    const float y2 = 2.f * y;
    g_loc[i] = t + gl * y2;
    g_scale[i] = t * e + gl * (y2 * e - 1.f / s);
}

// The example policy head folded into the same two launches: x [B, 2A] is the last Linear's output,
// (mean | log_scale) side by side; scale = sqrt(exp(2 clamp(log_scale, lo, hi))) (MODE 0, the
// arithmetic of train_soft_actor_critic.py:128-141, one rounded f32 operation per torch op) or
// exp(clamp(log_scale, lo, hi)) (MODE 1).  Replaces chunk / clamp / mul / exp / sqrt forward and
// their ten backward launches (sqrt, exp, mul, two compares, and, where, fill x2, cat).
template <int MODE>
__device__ __forceinline__ float head_scale(float ls, float lo, float hi, float &v) {
    float c = ls < lo ? lo : ls;          // torch.clamp: min(max(x, lo), hi), NaN passes through
    c = c > hi ? hi : c;
    if (MODE == 0) {
        v = expf(c * 2.f);
        return sqrtf(v);
    }
    v = expf(c);
    return v;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_squashed_head_fwd(
    const float *__restrict__ x, int64_t ldx, float lo, float hi, const float *__restrict__ eps,
    float *__restrict__ action, float *__restrict__ logp, float *__restrict__ neg_logp, int B, int A) {
    const int row = blockIdx.x * (kThreads / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float s_ladj = 0.f, s_nlp = 0.f;
    for (int a = lane; a < A; a += 64) {
        float v;
        const float l = x[(int64_t)row * ldx + a];
        const float s = head_scale<MODE>(x[(int64_t)row * ldx + A + a], lo, hi, v);
        const float e = eps[(int64_t)row * A + a];
        const float u = l + e * s;
        action[(int64_t)row * A + a] = tanhf(u);
        const float d = u - l;
        const float nlp = -(d * d) / (2.f * (s * s)) - logf(s) - 0.9189385332046727f;
        const float z = -2.f * u;
        const float sp = z > 20.f ? z : log1pf(expf(z));
        s_ladj += 2.f * (0.6931471805599453f - u - sp);
        s_nlp += nlp;
    }
    s_ladj = wave_sum(s_ladj);
    s_nlp = wave_sum(s_nlp);
    if (lane == 0) {
        const float lp = (0.f - s_ladj) + s_nlp;
        logp[row] = lp;
        if (neg_logp != nullptr) neg_logp[row] = -lp;
    }
}

// d/dx of the above: the mean half takes g_loc as is; the log-scale half takes g_scale through
// sqrt (g / (2 s)), exp (g v), the doubling and the clamp mask (lo <= log_scale <= hi), each the
// rounded operation of the autograd formula it replaces.
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_squashed_head_bwd(
    const float *__restrict__ g_action, const float *__restrict__ g_logp,
    const float *__restrict__ action, const float *__restrict__ eps, const float *__restrict__ x,
    int64_t ldx, float lo, float hi, float *__restrict__ g_x, int B, int A) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (int64_t)B * A) return;
    const int row = (int)(i / A), a = (int)(i - (int64_t)row * A);
    const float ls = x[(int64_t)row * ldx + A + a];
    float v;
    const float s = head_scale<MODE>(ls, lo, hi, v);
    const float y = action[i], e = eps[i];
    const float ga = g_action != nullptr ? g_action[i] : 0.f;
    const float gl = g_logp != nullptr ? g_logp[row] : 0.f;
    const float t = ga * (1.f - y * y);
    const float y2 = 2.f * y;
    const float g_scale = t * e + gl * (y2 * e - 1.f / s);
    float g_c;
    if (MODE == 0) g_c = ((g_scale / (2.f * s)) * v) * 2.f;
    else g_c = g_scale * v;
    const bool inside = ls >= lo && ls <= hi;
    g_x[(int64_t)row * 2 * A + a] = t + gl * y2;
    g_x[(int64_t)row * 2 * A + A + a] = inside ? g_c : 0.f;
}

// ---------------------------------------------------------------------------------
// multi-tensor elementwise launches
// ---------------------------------------------------------------------------------
struct SoftArgs {
    float *dst[PFRL_OPT_MAX_TENSORS];
    const float *src[PFRL_OPT_MAX_TENSORS];
    int64_t numel[PFRL_OPT_MAX_TENSORS];
    int32_t chunk_end[PFRL_OPT_MAX_TENSORS];
    int32_t n;
};

__global__ __launch_bounds__(kThreads) void k_soft_update(SoftArgs a, float one_minus_tau, float tau) {
    int t = 0;
    const int b = blockIdx.x;
    while (t < a.n - 1 && b >= a.chunk_end[t]) ++t;
    const int first = t == 0 ? 0 : a.chunk_end[t - 1];
    const int64_t base = (int64_t)(b - first) * kChunk;
    float *__restrict__ dst = a.dst[t];
    const float *__restrict__ src = a.src[t];
    const int64_t n = a.numel[t];
    float d[4], s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * kThreads + threadIdx.x;
        const int64_t ii = i < n ? i : n - 1;
        d[u] = dst[ii];
        s[u] = src[ii];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * kThreads + threadIdx.x;
        // dst.mul_(1 - tau); dst.add_(tau * src)
        if (i < n) dst[i] = __fadd_rn(__fmul_rn(d[u], one_minus_tau), __fmul_rn(tau, s[u]));
    }
}

struct AdamArgs {
    float *p[PFRL_OPT_MAX_TENSORS];
    const float *g[PFRL_OPT_MAX_TENSORS];
    float *m[PFRL_OPT_MAX_TENSORS];
    float *v[PFRL_OPT_MAX_TENSORS];
    float *step[PFRL_OPT_MAX_TENSORS];
    float *soft[PFRL_OPT_MAX_TENSORS];            // target-network tensor soft-updated from p (or NULL)
    int64_t slab_stride[PFRL_OPT_MAX_TENSORS];    // n_slabs > 1: g = sum_s g[s * slab_stride + i]
    int32_t n_slabs[PFRL_OPT_MAX_TENSORS];
    int64_t numel[PFRL_OPT_MAX_TENSORS];
    int32_t chunk_end[PFRL_OPT_MAX_TENSORS];
    int32_t n;
};

// torch.optim.Adam (no amsgrad), the arithmetic of _single_tensor_adam:
//   g += wd * p;  m.lerp_(g, 1 - b1);  v.mul_(b2).addcmul_(g, g, value = 1 - b2)
//   denom = sqrt(v) / sqrt(1 - b2^t) + eps;  p.addcdiv_(m, denom, value = -lr / (1 - b1^t))
// with t = step + 1 read from the tensor's device-side step counter.  The counters are
// advanced by whichever workgroup finishes last (ticket), after every workgroup has read them.
// Two things may ride along (pfrl_adam_step_ex): a gradient still in split-K slabs is summed here,
// slab 0 first, exactly as pfrl_splitk_reduce sums it (that launch then never runs), and a target
// network's tensor takes its soft update from the parameter value just written (pfrl_soft_update's
// arithmetic: dst * (1 - tau) + tau * src).
__global__ __launch_bounds__(kThreads) void k_adam(AdamArgs a, double lr, double b1d, double b2d,
                                                   float eps, float weight_decay,
                                                   unsigned int *ticket, int reps, float one_minus_tau,
                                                   float tau) {
    __shared__ float s_step_size, s_bc2_sqrt;
    __shared__ bool s_last;
    int t = 0;
    const int b = blockIdx.x;
    while (t < a.n - 1 && b >= a.chunk_end[t]) ++t;
    const int first = t == 0 ? 0 : a.chunk_end[t - 1];
    // a workgroup walks `reps` consecutive chunks of its tensor: the launch stays at a couple
    // of thousand workgroups however large the network is (the ticket below is one atomic per
    // workgroup on ONE address: 6 700 of them were 60 of the 86 us of a 6.9 M-parameter step)
    const int64_t base0 = (int64_t)(b - first) * kChunk * reps;
    if (threadIdx.x == 0) {
        const double step = (double)(*a.step[t]) + 1.0;
        const double bc1 = 1.0 - pow(b1d, step), bc2 = 1.0 - pow(b2d, step);
        s_step_size = (float)(-(lr / bc1));
        s_bc2_sqrt = (float)sqrt(bc2);
    }
    float *__restrict__ p = a.p[t];
    const float *__restrict__ g = a.g[t];
    float *__restrict__ m = a.m[t];
    float *__restrict__ v = a.v[t];
    float *__restrict__ soft = a.soft[t];
    const int64_t n = a.numel[t];
    const int S = a.n_slabs[t];
    const int64_t sstride = a.slab_stride[t];
    __syncthreads();
    const float step_size = s_step_size, bc2_sqrt = s_bc2_sqrt;
    // the scalars as PyTorch hands them to f32 kernels: computed in double, then rounded
    const float w1 = (float)(1.0 - b1d), b2 = (float)b2d, omb2 = (float)(1.0 - b2d);
    for (int r = 0; r < reps; ++r) {
        const int64_t base = base0 + (int64_t)r * kChunk;
        if (base >= n) break;   // (uniform)
        float pv[4], gv[4], mv[4], vv[4], sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = base + u * kThreads + threadIdx.x;
            const int64_t ii = i < n ? i : n - 1;
            pv[u] = p[ii]; gv[u] = g[ii]; mv[u] = m[ii]; vv[u] = v[ii];
            sv[u] = soft != nullptr ? soft[ii] : 0.f;
        }
        if (S > 1) {   // (uniform) the slabs beyond the first, eight loads per element in flight
#pragma unroll
            for (int u = 0; u < 4; ++u) gv[u] = __fadd_rn(0.f, gv[u]);
            for (int k = 1; k < S; k += 8) {
                float w8[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = base + u * kThreads + threadIdx.x;
                    const int64_t ii = i < n ? i : n - 1;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        w8[u][q] = g[(int64_t)(k + q < S ? k + q : S - 1) * sstride + ii];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k + q < S) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) gv[u] = __fadd_rn(gv[u], w8[u][q]);
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = base + u * kThreads + threadIdx.x;
            if (i >= n) continue;
            float gi = gv[u];
            if (weight_decay != 0.0f) gi = __fadd_rn(gi, __fmul_rn(weight_decay, pv[u]));
            // lerp_(g, w1): the two forms of at::native::lerp
            const float dm = __fsub_rn(gi, mv[u]);
            const float mi = w1 < 0.5f ? __fadd_rn(mv[u], __fmul_rn(w1, dm))
                                       : __fsub_rn(gi, __fmul_rn(dm, __fsub_rn(1.0f, w1)));
            const float vi = __fadd_rn(__fmul_rn(vv[u], b2), __fmul_rn(__fmul_rn(omb2, gi), gi));
            const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), bc2_sqrt), eps);
            m[i] = mi;
            v[i] = vi;
            const float pn = __fadd_rn(pv[u], __fdiv_rn(__fmul_rn(step_size, mi), denom));
            p[i] = pn;
            if (soft != nullptr) soft[i] = __fadd_rn(__fmul_rn(sv[u], one_minus_tau), __fmul_rn(tau, pn));
        }
    }
    // Every workgroup has read (and used) its step counter before it takes a ticket; nothing
    // it wrote has to be visible to the others, so no fence here: a release fence is a whole
    // L2 write-back on this part, ~15 us per launch when every workgroup issues one.
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                 gridDim.x - 1;
    __syncthreads();
    if (s_last) {
        if (threadIdx.x < a.n) *a.step[threadIdx.x] = *a.step[threadIdx.x] + 1.0f;
        if (threadIdx.x == 0) *ticket = 0u;
    }
}

// ---------------------------------------------------------------------------------
// SAC losses (pfrl/agents/soft_actor_critic.py:214-308), each a single workgroup: B is the
// minibatch (256), the cost is the launch, not the arithmetic
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float *sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__device__ __forceinline__ float torch_min(float a, float b) {   // NaN-propagating, as torch.min
    return (a < b || a != a) ? a : b;
}

__device__ __forceinline__ float temperature_of(const float *log_t, float t_val) {
    return log_t != nullptr ? expf(*log_t) : t_val;   // TemperatureHolder: exp(log_temperature)
}

// target_q = reward + discount * (1 - terminal) * (min(next_q1, next_q2) - T * next_log_prob)
__global__ __launch_bounds__(kThreads) void k_sac_target_q(
    const float *__restrict__ reward, const float *__restrict__ discount, const float *__restrict__ terminal,
    const float *__restrict__ nq1, const float *__restrict__ nq2, const float *__restrict__ nlogp,
    const float *__restrict__ log_t, float t_val, float *__restrict__ out, int B) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= B) return;
    const float T = temperature_of(log_t, t_val);
    const float soft = torch_min(nq1[i], nq2[i]) - T * nlogp[i];
    out[i] = reward[i] + (discount[i] * (1.0f - terminal[i])) * soft;
}

// loss = 0.5 * mean((target - pred)^2); blockIdx.y picks one of up to two predictions of the
// same target (the twin Q-networks)
struct HalfMseArgs {
    const float *pred[2], *g_loss[2];
    float *loss[2], *g_pred[2];
};

__global__ __launch_bounds__(kThreads) void k_half_mse_fwd(const float *__restrict__ target, HalfMseArgs a,
                                                           int B) {
    __shared__ float sh[4];
    const float *__restrict__ pred = a.pred[blockIdx.y];
    float s = 0.f;
    // (g_pred given: also the gradient for dL/dloss = 1, the arithmetic of k_half_mse_bwd)
    float *__restrict__ unit = a.g_pred[blockIdx.y];
    const float gB = 0.5f / (float)B;
    for (int i = threadIdx.x; i < B; i += kThreads) {
        const float d = target[i] - pred[i];
        s += d * d;
        if (unit != nullptr) unit[i] = -((2.0f * d) * gB);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) a.loss[blockIdx.y][0] = 0.5f * (s / (float)B);
}

__global__ __launch_bounds__(kThreads) void k_half_mse_bwd(const float *__restrict__ target, HalfMseArgs a,
                                                           int B) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= B) return;
    const float *g_loss = a.g_loss[blockIdx.y];
    const float g = 0.5f * (g_loss != nullptr ? g_loss[0] : 0.f);
    a.g_pred[blockIdx.y][i] = -((2.0f * (target[i] - a.pred[blockIdx.y][i])) * (g / (float)B));
}

// loss = mean(T * log_prob - min(q1, q2))
__global__ __launch_bounds__(kThreads) void k_sac_policy_loss_fwd(
    const float *__restrict__ logp, const float *__restrict__ q1, const float *__restrict__ q2,
    const float *__restrict__ log_t, float t_val, float *__restrict__ loss, float *__restrict__ u_logp,
    float *__restrict__ u_q1, float *__restrict__ u_q2, int B) {
    __shared__ float sh[4];
    const float T = temperature_of(log_t, t_val);
    const float g = 1.0f / (float)B;      // (u_*: the gradients for dL/dloss = 1, as k_sac_policy_loss_bwd)
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += kThreads) {
        const float a = q1[i], b = q2[i];
        s += T * logp[i] - torch_min(a, b);
        if (u_logp != nullptr) {
            u_logp[i] = g * T;
            u_q1[i] = -(g * (a < b ? 1.f : (a == b ? 0.5f : 0.f)));
            u_q2[i] = -(g * (b < a ? 1.f : (a == b ? 0.5f : 0.f)));
        }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) loss[0] = s / (float)B;
}

// d loss / d log_prob = T / B;  d loss / d q = -1 / B to the smaller one (halves on a tie,
// as torch.min's backward splits it)
__global__ __launch_bounds__(kThreads) void k_sac_policy_loss_bwd(
    const float *__restrict__ g_loss, const float *__restrict__ q1, const float *__restrict__ q2,
    const float *__restrict__ log_t, float t_val, float *__restrict__ g_logp, float *__restrict__ g_q1,
    float *__restrict__ g_q2, int B) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= B) return;
    const float T = temperature_of(log_t, t_val);
    const float g = g_loss[0] / (float)B;
    const float a = q1[i], b = q2[i];
    g_logp[i] = g * T;
    const float w1 = a < b ? 1.f : (a == b ? 0.5f : 0.f);
    const float w2 = b < a ? 1.f : (a == b ? 0.5f : 0.f);
    g_q1[i] = -(g * w1);
    g_q2[i] = -(g * w2);
}

// loss = -mean(T * (log_prob + entropy_target)), T = exp(log_t)   (soft_actor_critic.py:264-271)
__global__ __launch_bounds__(kThreads) void k_sac_temperature_loss(const float *__restrict__ log_t,
                                                                   const float *__restrict__ logp,
                                                                   float entropy_target,
                                                                   float *__restrict__ loss, int B) {
    __shared__ float sh[4];
    const float T = expf(*log_t);
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += kThreads) s += T * (logp[i] + entropy_target);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) loss[0] = -(s / (float)B);
}

// ... and torch.optim.Adam's step on log_t with that loss as the gradient (k_adam's arithmetic for
// one element, the step counter read as t - 1 and advanced), in the same single workgroup: the
// temperature update of soft_actor_critic.py:264-271 as ONE launch instead of loss + optimizer.
__global__ __launch_bounds__(kThreads) void k_sac_temperature_step(
    float *__restrict__ log_t, const float *__restrict__ logp, float entropy_target,
    float *__restrict__ loss, float *__restrict__ m, float *__restrict__ v, float *__restrict__ step,
    double lr, double b1d, double b2d, float eps, float weight_decay, int B) {
    __shared__ float sh[4];
    const float p0 = *log_t;
    const float T = expf(p0);
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += kThreads) s += T * (logp[i] + entropy_target);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        const float l = -(s / (float)B);
        loss[0] = l;
        const double t = (double)(*step) + 1.0;
        const double bc1 = 1.0 - pow(b1d, t), bc2 = 1.0 - pow(b2d, t);
        const float step_size = (float)(-(lr / bc1)), bc2_sqrt = (float)sqrt(bc2);
        const float w1 = (float)(1.0 - b1d), b2 = (float)b2d, omb2 = (float)(1.0 - b2d);
        float gi = l;
        if (weight_decay != 0.0f) gi = __fadd_rn(gi, __fmul_rn(weight_decay, p0));
        const float m0 = *m, v0 = *v;
        const float dm = __fsub_rn(gi, m0);
        const float mi = w1 < 0.5f ? __fadd_rn(m0, __fmul_rn(w1, dm))
                                   : __fsub_rn(gi, __fmul_rn(dm, __fsub_rn(1.0f, w1)));
        const float vi = __fadd_rn(__fmul_rn(v0, b2), __fmul_rn(__fmul_rn(omb2, gi), gi));
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), bc2_sqrt), eps);
        *m = mi;
        *v = vi;
        *log_t = __fadd_rn(p0, __fdiv_rn(__fmul_rn(step_size, mi), denom));
        *step = *step + 1.0f;
    }
}

template <typename Args>
int fill_chunks(Args &a, int n, const int64_t *numel, int lo, int64_t per_block = kChunk) {
    int chunks = 0;
    for (int t = 0; t < n; ++t) {
        a.numel[t] = numel[lo + t];
        chunks += (int)((numel[lo + t] + per_block - 1) / per_block);
        a.chunk_end[t] = chunks;
    }
    a.n = n;
    return chunks;
}

}  // namespace

extern "C" int pfrl_squashed_gaussian_fwd(const float *loc, int64_t ld_loc, const float *scale,
                                          int64_t ld_scale, const float *eps, float *action, float *logp,
                                          float *neg_logp, int32_t B, int32_t A, void *stream) {
    PFRL_CHECK_ARG(B >= 0 && A >= 1 && ld_loc >= A && ld_scale >= A, "pfrl_squashed_gaussian_fwd: bad shape");
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_squashed_gaussian_fwd, dim3((B + 3) / 4), dim3(kThreads), 0, (hipStream_t)stream,
                       loc, ld_loc, scale, ld_scale, eps, action, logp, neg_logp, B, A);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_squashed_gaussian_bwd(const float *g_action, const float *g_logp, const float *action,
                                          const float *eps, const float *scale, int64_t ld_scale,
                                          float *g_loc, float *g_scale, int32_t B, int32_t A,
                                          void *stream) {
    PFRL_CHECK_ARG(B >= 0 && A >= 1 && ld_scale >= A, "pfrl_squashed_gaussian_bwd: bad shape");
    if (B == 0) return 0;
    const int64_t n = (int64_t)B * A;
    hipLaunchKernelGGL(k_squashed_gaussian_bwd, dim3((unsigned)((n + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, (hipStream_t)stream, g_action, g_logp, action, eps, scale,
                       ld_scale, g_loc, g_scale, B, A);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_squashed_head_fwd(const float *x, int64_t ldx, float clamp_lo, float clamp_hi,
                                      int32_t mode, const float *eps, float *action, float *logp,
                                      float *neg_logp, int32_t B, int32_t A, void *stream) {
    PFRL_CHECK_ARG(B >= 0 && A >= 1 && ldx >= 2 * (int64_t)A && (mode == 0 || mode == 1) && clamp_lo <= clamp_hi,
                   "pfrl_squashed_head_fwd: bad shape");
    if (B == 0) return 0;
    if (mode == 0)
        hipLaunchKernelGGL(k_squashed_head_fwd<0>, dim3((B + 3) / 4), dim3(kThreads), 0, (hipStream_t)stream,
                           x, ldx, clamp_lo, clamp_hi, eps, action, logp, neg_logp, B, A);
    else
        hipLaunchKernelGGL(k_squashed_head_fwd<1>, dim3((B + 3) / 4), dim3(kThreads), 0, (hipStream_t)stream,
                           x, ldx, clamp_lo, clamp_hi, eps, action, logp, neg_logp, B, A);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_squashed_head_bwd(const float *g_action, const float *g_logp, const float *action,
                                      const float *eps, const float *x, int64_t ldx, float clamp_lo,
                                      float clamp_hi, int32_t mode, float *g_x, int32_t B, int32_t A,
                                      void *stream) {
    PFRL_CHECK_ARG(B >= 0 && A >= 1 && ldx >= 2 * (int64_t)A && (mode == 0 || mode == 1),
                   "pfrl_squashed_head_bwd: bad shape");
    if (B == 0) return 0;
    const int64_t n = (int64_t)B * A;
    const dim3 grid((unsigned)((n + kThreads - 1) / kThreads));
    if (mode == 0)
        hipLaunchKernelGGL(k_squashed_head_bwd<0>, grid, dim3(kThreads), 0, (hipStream_t)stream, g_action,
                           g_logp, action, eps, x, ldx, clamp_lo, clamp_hi, g_x, B, A);
    else
        hipLaunchKernelGGL(k_squashed_head_bwd<1>, grid, dim3(kThreads), 0, (hipStream_t)stream, g_action,
                           g_logp, action, eps, x, ldx, clamp_lo, clamp_hi, g_x, B, A);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_soft_update(int32_t n_tensors, float *const *dst, const float *const *src,
                                const int64_t *numel, double tau, void *stream) {
    PFRL_CHECK_ARG(n_tensors >= 0, "pfrl_soft_update: bad tensor count");
    for (int lo = 0; lo < n_tensors; lo += PFRL_OPT_MAX_TENSORS) {
        const int n = (n_tensors - lo) < PFRL_OPT_MAX_TENSORS ? (n_tensors - lo) : PFRL_OPT_MAX_TENSORS;
        SoftArgs a;
        for (int t = 0; t < n; ++t) {
            a.dst[t] = dst[lo + t];
            a.src[t] = src[lo + t];
        }
        const int chunks = fill_chunks(a, n, numel, lo);
        if (chunks == 0) continue;
        // the scalars as PyTorch hands them to an f32 kernel: 1 - tau in double, then rounded
        hipLaunchKernelGGL(k_soft_update, dim3(chunks), dim3(kThreads), 0, (hipStream_t)stream, a,
                           (float)(1.0 - tau), (float)tau);
    }
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_adam_step_ex(int32_t n_tensors, float *const *params, const float *const *grads,
                                 const int64_t *slab_stride, const int32_t *n_slabs,
                                 float *const *exp_avg, float *const *exp_avg_sq, float *const *steps,
                                 float *const *soft_dst, double tau, const int64_t *numel, double lr,
                                 double beta1, double beta2, double eps, double weight_decay,
                                 void *ticket, void *stream) {
    PFRL_CHECK_ARG(n_tensors >= 0 && ticket != nullptr, "pfrl_adam_step: bad arguments");
    if (n_slabs != nullptr)
        for (int t = 0; t < n_tensors; ++t)
            PFRL_CHECK_ARG(n_slabs[t] >= 1 && (n_slabs[t] == 1 || (slab_stride != nullptr && slab_stride[t] >= numel[t])),
                           "pfrl_adam_step_ex: slabs need a stride >= numel");
    for (int lo = 0; lo < n_tensors; lo += PFRL_OPT_MAX_TENSORS) {
        const int n = (n_tensors - lo) < PFRL_OPT_MAX_TENSORS ? (n_tensors - lo) : PFRL_OPT_MAX_TENSORS;
        AdamArgs a;
        for (int t = 0; t < n; ++t) {
            a.p[t] = params[lo + t];
            a.g[t] = grads[lo + t];
            a.m[t] = exp_avg[lo + t];
            a.v[t] = exp_avg_sq[lo + t];
            a.step[t] = steps[lo + t];
            a.soft[t] = soft_dst != nullptr ? soft_dst[lo + t] : nullptr;
            a.n_slabs[t] = n_slabs != nullptr ? n_slabs[lo + t] : 1;
            a.slab_stride[t] = slab_stride != nullptr ? slab_stride[lo + t] : 0;
        }
        int64_t total = 0;
        for (int t = 0; t < n; ++t) total += numel[lo + t];
        // chunks of 1 024 elements per workgroup, `reps` of them once that would exceed ~2 048
        // workgroups
        const int reps = (int)((total / kChunk + 2047) / 2048) > 1 ? (int)((total / kChunk + 2047) / 2048) : 1;
        const int chunks = fill_chunks(a, n, numel, lo, (int64_t)kChunk * reps);
        PFRL_CHECK_ARG(chunks > 0, "pfrl_adam_step: empty parameters");
        hipLaunchKernelGGL(k_adam, dim3(chunks), dim3(kThreads), 0, (hipStream_t)stream, a, lr, beta1,
                           beta2, (float)eps, (float)weight_decay, (unsigned int *)ticket, reps,
                           (float)(1.0 - tau), (float)tau);
    }
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_adam_step(int32_t n_tensors, float *const *params, const float *const *grads,
                              float *const *exp_avg, float *const *exp_avg_sq, float *const *steps,
                              const int64_t *numel, double lr, double beta1, double beta2, double eps,
                              double weight_decay, void *ticket, void *stream) {
    return pfrl_adam_step_ex(n_tensors, params, grads, nullptr, nullptr, exp_avg, exp_avg_sq, steps, nullptr,
                             0.0, numel, lr, beta1, beta2, eps, weight_decay, ticket, stream);
}

extern "C" int pfrl_sac_target_q(const float *reward, const float *discount, const float *terminal,
                                 const float *next_q1, const float *next_q2, const float *next_log_prob,
                                 const float *log_temperature, float temperature, float *target_q,
                                 int32_t B, void *stream) {
    PFRL_CHECK_ARG(B >= 0, "pfrl_sac_target_q: bad batch");
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_sac_target_q, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, reward, discount, terminal, next_q1, next_q2, next_log_prob,
                       log_temperature, temperature, target_q, B);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_half_mse_fwd(const float *target, const float *pred, float *loss, int32_t B,
                                 void *stream) {
    PFRL_CHECK_ARG(B >= 1, "pfrl_half_mse_fwd: empty batch");
    HalfMseArgs a{{pred, nullptr}, {nullptr, nullptr}, {loss, nullptr}, {nullptr, nullptr}};
    hipLaunchKernelGGL(k_half_mse_fwd, dim3(1, 1), dim3(kThreads), 0, (hipStream_t)stream, target, a, B);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_half_mse_bwd(const float *g_loss, const float *target, const float *pred,
                                 float *g_pred, int32_t B, void *stream) {
    PFRL_CHECK_ARG(B >= 1, "pfrl_half_mse_bwd: empty batch");
    HalfMseArgs a{{pred, nullptr}, {g_loss, nullptr}, {nullptr, nullptr}, {g_pred, nullptr}};
    hipLaunchKernelGGL(k_half_mse_bwd, dim3((B + kThreads - 1) / kThreads, 1), dim3(kThreads), 0,
                       (hipStream_t)stream, target, a, B);
    PFRL_LAUNCH_CHECK();
}

// The two losses of twin predictions of one target, and both gradients, in one launch each
// (host arrays of two device pointers; a NULL g_loss entry is a zero upstream gradient).
extern "C" int pfrl_half_mse_twin_fwd(const float *target, const float *const *pred, float *const *loss,
                                      float *const *unit_g_pred, int32_t B, void *stream) {
    PFRL_CHECK_ARG(B >= 1, "pfrl_half_mse_twin_fwd: empty batch");
    HalfMseArgs a{{pred[0], pred[1]}, {nullptr, nullptr}, {loss[0], loss[1]}, {nullptr, nullptr}};
    if (unit_g_pred != nullptr) { a.g_pred[0] = unit_g_pred[0]; a.g_pred[1] = unit_g_pred[1]; }
    hipLaunchKernelGGL(k_half_mse_fwd, dim3(1, 2), dim3(kThreads), 0, (hipStream_t)stream, target, a, B);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_half_mse_twin_bwd(const float *const *g_loss, const float *target,
                                      const float *const *pred, float *const *g_pred, int32_t B,
                                      void *stream) {
    PFRL_CHECK_ARG(B >= 1, "pfrl_half_mse_twin_bwd: empty batch");
    HalfMseArgs a{{pred[0], pred[1]}, {g_loss[0], g_loss[1]}, {nullptr, nullptr}, {g_pred[0], g_pred[1]}};
    hipLaunchKernelGGL(k_half_mse_bwd, dim3((B + kThreads - 1) / kThreads, 2), dim3(kThreads), 0,
                       (hipStream_t)stream, target, a, B);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_sac_policy_loss_fwd(const float *log_prob, const float *q1, const float *q2,
                                        const float *log_temperature, float temperature, float *loss,
                                        float *unit_g_log_prob, float *unit_g_q1, float *unit_g_q2,
                                        int32_t B, void *stream) {
    PFRL_CHECK_ARG(B >= 1, "pfrl_sac_policy_loss_fwd: empty batch");
    PFRL_CHECK_ARG((unit_g_log_prob == nullptr) == (unit_g_q1 == nullptr) &&
                   (unit_g_q1 == nullptr) == (unit_g_q2 == nullptr),
                   "pfrl_sac_policy_loss_fwd: the three unit gradients or none");
    hipLaunchKernelGGL(k_sac_policy_loss_fwd, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, log_prob,
                       q1, q2, log_temperature, temperature, loss, unit_g_log_prob, unit_g_q1, unit_g_q2, B);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_sac_policy_loss_bwd(const float *g_loss, const float *q1, const float *q2,
                                        const float *log_temperature, float temperature, float *g_log_prob,
                                        float *g_q1, float *g_q2, int32_t B, void *stream) {
    PFRL_CHECK_ARG(B >= 1, "pfrl_sac_policy_loss_bwd: empty batch");
    hipLaunchKernelGGL(k_sac_policy_loss_bwd, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       (hipStream_t)stream, g_loss, q1, q2, log_temperature, temperature, g_log_prob, g_q1,
                       g_q2, B);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_sac_temperature_loss(const float *log_temperature, const float *log_prob,
                                         float entropy_target, float *loss, int32_t B, void *stream) {
    PFRL_CHECK_ARG(B >= 1 && log_temperature != nullptr, "pfrl_sac_temperature_loss: bad arguments");
    hipLaunchKernelGGL(k_sac_temperature_loss, dim3(1), dim3(kThreads), 0, (hipStream_t)stream,
                       log_temperature, log_prob, entropy_target, loss, B);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_sac_temperature_step(float *log_temperature, const float *log_prob, float entropy_target,
                                         float *loss, float *exp_avg, float *exp_avg_sq, float *step,
                                         double lr, double beta1, double beta2, double eps,
                                         double weight_decay, int32_t B, void *stream) {
    PFRL_CHECK_ARG(B >= 1 && log_temperature != nullptr && loss != nullptr && exp_avg != nullptr &&
                       exp_avg_sq != nullptr && step != nullptr,
                   "pfrl_sac_temperature_step: bad arguments");
    hipLaunchKernelGGL(k_sac_temperature_step, dim3(1), dim3(kThreads), 0, (hipStream_t)stream,
                       log_temperature, log_prob, entropy_target, loss, exp_avg, exp_avg_sq, step, lr, beta1,
                       beta2, (float)eps, (float)weight_decay, B);
    PFRL_LAUNCH_CHECK();
}
