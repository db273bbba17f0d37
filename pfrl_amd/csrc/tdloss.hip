// Fused DQN TD loss: replaces the ~20 tiny elementwise / gather / reduce /
// scatter kernels PyTorch launches for
//     y = Q(s)[a];  t = r + disc * (1 - term) * max_a' Q_target(s')[a']
//     loss = sum_b w_b * Huber(y_b - t_b)      (pfrl/agents/dqn.py:388-470,
//                                                compute_value_loss :44-104)
// and for its backward pass (the gradient w.r.t. Q(s) is analytic) with ONE
// launch.  At minibatch 32 those kernels were 27 % of the update's device time
// (each ~4-5 us of pure launch latency for 32 scalars of work).
#include "common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float wave_sum_f(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(kThreads) void k_dqn_td_loss(
    const float *__restrict__ q, const int64_t *__restrict__ action,
    const float *__restrict__ target_q, const float *__restrict__ next_q_online,
    const float *__restrict__ reward, const float *__restrict__ discount,
    const float *__restrict__ terminal, const float *__restrict__ weights, int64_t B, int A,
    int clip_delta, int mean, float *__restrict__ out_loss, float *__restrict__ out_grad_q,
    float *__restrict__ out_y, float *__restrict__ out_abs_delta) {
    __shared__ float s_part[kThreads / 64];
    float local = 0.0f;
    const float scale = mean ? 1.0f / (float)B : 1.0f;
    for (int64_t b = threadIdx.x; b < B; b += kThreads) {
        const float *qt = target_q + b * A;
        // greedy next action: first maximum (torch.argmax tie rule)
        const float *sel = next_q_online ? next_q_online + b * A : qt;
        int best = 0;
        float bestv = sel[0];
        for (int a = 1; a < A; ++a) {
            const float v = sel[a];
            if (v > bestv) {
                bestv = v;
                best = a;
            }
        }
        const float next = qt[best];
        const int act = (int)action[b];
        const float y = q[b * A + act];
        // t = r + (disc * (1 - term)) * next   -- same association as the reference
        const float coef = __fmul_rn(discount[b], __fsub_rn(1.0f, terminal[b]));
        const float t = __fadd_rn(reward[b], __fmul_rn(coef, next));
        const float d = __fsub_rn(y, t);
        const float ad = fabsf(d);
        float l, g;
        if (clip_delta) {
            l = ad < 1.0f ? __fmul_rn(__fmul_rn(0.5f, ad), ad) : __fsub_rn(ad, 0.5f);
            g = ad < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f);
        } else {
            l = __fmul_rn(0.5f, __fmul_rn(d, d));
            g = d;
        }
        const float w = weights ? weights[b] : 1.0f;
        local += l * w;
        const float gq = g * w * scale;
        for (int a = 0; a < A; ++a) out_grad_q[b * A + a] = (a == act) ? gq : 0.0f;
        out_y[b] = y;
        out_abs_delta[b] = ad;
    }
    local = wave_sum_f(local);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.0f;
        for (int k = 0; k < kThreads / 64; ++k) tot += s_part[k];
        out_loss[0] = tot * scale;
    }
}

}  // namespace

extern "C" int pfrl_dqn_td_loss(const float *q, const int64_t *action, const float *target_q,
                                const float *next_q_online, const float *reward,
                                const float *discount, const float *terminal,
                                const float *weights, int64_t B, int32_t A, int clip_delta,
                                int mean, float *out_loss, float *out_grad_q, float *out_y,
                                float *out_abs_delta, void *stream) {
    PFRL_CHECK_ARG(B > 0 && A > 0, "pfrl_dqn_td_loss: empty batch");
    hipLaunchKernelGGL(k_dqn_td_loss, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, q, action,
                       target_q, next_q_online, reward, discount, terminal, weights, B, (int)A,
                       clip_delta, mean, out_loss, out_grad_q, out_y, out_abs_delta);
    PFRL_LAUNCH_CHECK();
}

// ===================================================================================
// Head + TD loss + head backward in ONE launch.
//
//   q = h W^T + b  (the `Linear(512, n_actions)` head of the example Q-functions,
//                   examples/atari/train_dqn_batch_ale.py:35-41)
//   loss, y, |delta|, dL/dq as k_dqn_td_loss above
//   dL/dh = dL/dq W,  dL/dW = dL/dq^T h,  dL/db = sum_m dL/dq
//
// At B = 32 these were three launches (narrow-head forward 4.6 us, TD loss 4.6 us, narrow-head
// backward 5.8 us) around 32 x 6 numbers.  Here ONE WAVE OWNS ONE ROW: it loads the row of h
// and all of W (lanes across k: k = lane + 64 j), reduces the A dot products, evaluates the
// row's loss term in every lane, and writes dL/dh[m] = g_m W[a_m] (dL/dq has one nonzero per
// row, the taken action).  What couples the rows -- dL/dW[a] = sum_{m: a_m = a} g_m h[m],
// dL/db, the loss sum -- leaves as one partial slab per workgroup (4 rows, folded in wave
// order through LDS): [ceil(B/4)][A*K + 32], the layout pfrl_splitk_reduce folds; the caller
// puts them into the fold launch that ends the trunk's backward anyway.  Every load is
// unconditional and issued before the first use (a single workgroup walking 32 rows through
// LDS with barriers measured 19-30 us).
// ===================================================================================
namespace {

template <int A, int KJ>
__device__ __forceinline__ void head_td_row(
    const float *__restrict__ h, const float *__restrict__ W, const float *__restrict__ bias,
    const int64_t *__restrict__ action, const float *__restrict__ target_q,
    const float *__restrict__ next_q_online, const float *__restrict__ reward,
    const float *__restrict__ discount, const float *__restrict__ terminal,
    const float *__restrict__ weights, int B, int clip_delta, int mean, float *__restrict__ out_y,
    float *__restrict__ out_abs_delta, float *__restrict__ dh, const int m, float (*s_c)[64 * KJ],
    float *s_g, float *s_l, int *s_act, const float *__restrict__ h_part, int h_splits,
    int64_t h_stride, const float *__restrict__ h_bias, float *__restrict__ h_out,
    float *__restrict__ dh_masked, float dh_scale) {
    constexpr int K = 64 * KJ;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (optional inputs are read through a pointer that is always valid: no branch at a load)
    const float *__restrict__ sel_q = next_q_online ? next_q_online : target_q;
    const float *__restrict__ wt_src = weights ? weights : reward;
    float hv[KJ], wv[A][KJ], tqv[A], nqv[A], bv[A];
    if (h_part == nullptr) {
#pragma unroll
        for (int j = 0; j < KJ; ++j) hv[j] = h[(size_t)m * K + lane + 64 * j];
    } else {
        // the hidden layer's forward left split-K slabs [h_splits][B][K]: this row's part of the
        // fold launch -- h = relu(0 + slab 0 + slab 1 + ... + bias), the order of
        // pfrl_splitk_reduce -- happens here, eight slabs in flight, and h is written out for
        // the backward pass (ReLU mask, the head's own weight gradient)
#pragma unroll
        for (int j = 0; j < KJ; ++j) hv[j] = 0.f;
        const float *row = h_part + (size_t)m * K + lane;
        for (int s0 = 0; s0 < h_splits; s0 += 8) {
            float v[8][KJ];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sc = min(s0 + u, h_splits - 1);
#pragma unroll
                for (int j = 0; j < KJ; ++j) v[u][j] = row[(size_t)sc * h_stride + 64 * j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < h_splits) {
#pragma unroll
                    for (int j = 0; j < KJ; ++j) hv[j] = __fadd_rn(hv[j], v[u][j]);
                }
        }
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            hv[j] = fmaxf(__fadd_rn(hv[j], h_bias[lane + 64 * j]), 0.f);
            h_out[(size_t)m * K + lane + 64 * j] = hv[j];
        }
    }
#pragma unroll
    for (int a = 0; a < A; ++a) {
#pragma unroll
        for (int j = 0; j < KJ; ++j) wv[a][j] = W[a * K + lane + 64 * j];
        tqv[a] = target_q[(size_t)m * A + a];
        nqv[a] = sel_q[(size_t)m * A + a];
        bv[a] = bias[a];
    }
    const int act = (int)action[m];
    const float rew = reward[m], disc = discount[m], term = terminal[m];
    const float wt_raw = wt_src[m];
    const float wt = weights ? wt_raw : 1.0f;
    const float scale = mean ? 1.0f / (float)B : 1.0f;
    // y = q[act] = h[m] . W[act] + b[act]
    float y = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < KJ; ++j) p = fmaf(hv[j], wv[a][j], p);
        p = wave_sum_f(p) + bv[a];
        y = (a == act) ? p : y;
    }
    int best = 0;
    float bestv = nqv[0];
#pragma unroll
    for (int a = 1; a < A; ++a) {
        if (nqv[a] > bestv) {
            bestv = nqv[a];
            best = a;
        }
    }
    float next = tqv[0];
#pragma unroll
    for (int a = 1; a < A; ++a) next = (a == best) ? tqv[a] : next;
    const float coef = __fmul_rn(disc, __fsub_rn(1.0f, term));
    const float t = __fadd_rn(rew, __fmul_rn(coef, next));
    const float d = __fsub_rn(y, t);
    const float ad = fabsf(d);
    float l, g;
    if (clip_delta) {
        l = ad < 1.0f ? __fmul_rn(__fmul_rn(0.5f, ad), ad) : __fsub_rn(ad, 0.5f);
        g = ad < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f);
    } else {
        l = __fmul_rn(0.5f, __fmul_rn(d, d));
        g = d;
    }
    const float gq = g * wt * scale;
    float wsel[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        wsel[j] = wv[0][j];
#pragma unroll
        for (int a = 1; a < A; ++a) wsel[j] = (a == act) ? wv[a][j] : wsel[j];
        const float dv = gq * wsel[j];
        dh[(size_t)m * K + lane + 64 * j] = dv;
        // (data parallel: the hidden layer's batch matrix dy as its low-rank exchange wants it --
        // ReLU mask of h applied, times 1 / world size -- written here instead of by two
        // elementwise launches in front of the all-gather, distributed.lowrank_ready)
        if (dh_masked != nullptr)
            dh_masked[(size_t)m * K + lane + 64 * j] =
                hv[j] > 0.f ? (dh_scale == 1.0f ? dv : __fmul_rn(dv, dh_scale)) : 0.f;
    }
    if (lane == 0) {
        out_y[m] = y;
        out_abs_delta[m] = ad;
    }
    // the workgroup's four rows fold into ONE slab, in wave order (deterministic): each wave
    // parks g_m h[m] and (a_m, g_m, loss term); then every thread sums, per action, the rows
    // that took it
#pragma unroll
    for (int j = 0; j < KJ; ++j) s_c[wave][lane + 64 * j] = gq * hv[j];
    if (lane == 0) {
        s_act[wave] = act;
        s_g[wave] = gq;
        s_l[wave] = l * wt * scale;
    }
}

template <int A, int KJ>
__global__ __launch_bounds__(kThreads) void k_dqn_head_td_rows(
    const float *__restrict__ h, const float *__restrict__ W, const float *__restrict__ bias,
    const int64_t *__restrict__ action, const float *__restrict__ target_q,
    const float *__restrict__ next_q_online, const float *__restrict__ reward,
    const float *__restrict__ discount, const float *__restrict__ terminal,
    const float *__restrict__ weights, int B, int clip_delta, int mean, float *__restrict__ out_y,
    float *__restrict__ out_abs_delta, float *__restrict__ dh, float *__restrict__ part,
    const float *__restrict__ h_part, int h_splits, int64_t h_stride,
    const float *__restrict__ h_bias, float *__restrict__ h_out, float *__restrict__ dh_masked,
    float dh_scale) {
    constexpr int K = 64 * KJ;
    constexpr int STRIDE = A * K + 32;
    __shared__ float s_c[4][K];
    __shared__ float s_g[4], s_l[4];
    __shared__ int s_act[4];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 4) {
        s_act[tid] = -1;          // rows past the batch take no action
        s_g[tid] = 0.f;
        s_l[tid] = 0.f;
    }
    __syncthreads();
    const int m = blockIdx.x * 4 + wave;
    if (m < B)
        head_td_row<A, KJ>(h, W, bias, action, target_q, next_q_online, reward, discount, terminal,
                           weights, B, clip_delta, mean, out_y, out_abs_delta, dh, m, s_c, s_g, s_l,
                           s_act, h_part, h_splits, h_stride, h_bias, h_out, dh_masked, dh_scale);
    __syncthreads();
    float *pr = part + (size_t)blockIdx.x * STRIDE;
    for (int e = tid; e < A * K; e += kThreads) {
        const int a = e / K, k = e - a * K;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += (s_act[w] == a) ? s_c[w][k] : 0.f;
        pr[e] = v;
    }
    if (tid < 32) {
        // [A*K, A*K+16): dL/db of the slab; [A*K+16]: its loss terms; the rest zero
        float v = 0.f;
        if (tid < A) {
#pragma unroll
            for (int w = 0; w < 4; ++w) v += (s_act[w] == tid) ? s_g[w] : 0.f;
        }
        if (tid == 16) v = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
        pr[A * K + tid] = v;
    }
}

}  // namespace

// partials: [ceil(B/4)][A*K + 32] floats (see above); fold with pfrl_splitk_reduce: dw = sum
// over the slabs of [0, A*K), db of [A*K, A*K + A), loss of [A*K + 16].
extern "C" int pfrl_dqn_head_td_loss(const float *h, const float *w, const float *bias,
                                     const int64_t *action, const float *target_q,
                                     const float *next_q_online, const float *reward,
                                     const float *discount, const float *terminal, const float *weights,
                                     int32_t B, int32_t K, int32_t A, int clip_delta, int mean,
                                     float *out_y, float *out_abs_delta, float *dh, float *partials,
                                     const float *h_part, int32_t h_splits, int64_t h_stride,
                                     const float *h_bias, float *h_out, float *dh_masked,
                                     float dh_scale, void *stream) {
    PFRL_CHECK_ARG(B >= 1 && A >= 1 && A <= 16 && (K == 512 || K == 256),
                   "pfrl_dqn_head_td_loss: A <= 16, K = 256 or 512");
    PFRL_CHECK_ARG(h_part == nullptr || (h_splits >= 1 && h_bias && h_out && h_stride >= (int64_t)B * K),
                   "pfrl_dqn_head_td_loss: bad hidden-layer fold arguments");
    const dim3 grid((B + 3) / 4);
#define CALL_HT(AA)                                                                                \
    do {                                                                                           \
        if (K == 512)                                                                              \
            hipLaunchKernelGGL((k_dqn_head_td_rows<AA, 8>), grid, dim3(kThreads), 0,               \
                               (hipStream_t)stream, h, w, bias, action, target_q, next_q_online,   \
                               reward, discount, terminal, weights, B, clip_delta, mean, out_y,    \
                               out_abs_delta, dh, partials, h_part, h_splits, h_stride, h_bias,    \
                               h_out, dh_masked, dh_scale);                                        \
        else                                                                                       \
            hipLaunchKernelGGL((k_dqn_head_td_rows<AA, 4>), grid, dim3(kThreads), 0,               \
                               (hipStream_t)stream, h, w, bias, action, target_q, next_q_online,   \
                               reward, discount, terminal, weights, B, clip_delta, mean, out_y,    \
                               out_abs_delta, dh, partials, h_part, h_splits, h_stride, h_bias,    \
                               h_out, dh_masked, dh_scale);                                        \
    } while (0)
    switch (A) {
        case 1: CALL_HT(1); break;   case 2: CALL_HT(2); break;   case 3: CALL_HT(3); break;
        case 4: CALL_HT(4); break;   case 5: CALL_HT(5); break;   case 6: CALL_HT(6); break;
        case 7: CALL_HT(7); break;   case 8: CALL_HT(8); break;   case 9: CALL_HT(9); break;
        case 10: CALL_HT(10); break; case 11: CALL_HT(11); break; case 12: CALL_HT(12); break;
        case 13: CALL_HT(13); break; case 14: CALL_HT(14); break; case 15: CALL_HT(15); break;
        default: CALL_HT(16); break;
    }
#undef CALL_HT
    PFRL_LAUNCH_CHECK();
}
