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
// backward 5.8 us) around 32 x 6 numbers: one workgroup does all of it.  dL/dq has one nonzero
// per row (the taken action), so dL/dh[m] = g_m W[a_m] and dL/dW[a] = sum_{m: a_m = a} g_m h[m].
// A wave owns 8 rows of a 32-row pass with lanes across k (k = lane + 64 j); W sits in LDS.
// ===================================================================================
namespace {

constexpr int HT_ROWS = 32;  // rows per pass

// 64 per-lane partial sums v[0..63] -> lane i returns sum over all lanes of v[i].  Halving
// butterfly: at offset o a lane keeps the half of its values selected by its bit o and adds
// the partner's copy of that half: 63 shuffles in all, against 6 per value (384) for one
// xor-reduction each -- those, serialised behind one another, were 14 us of this kernel.
template <int N>
__device__ __forceinline__ void halve_step(float (&v)[64], int lane, int o) {
    const bool up = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const float send = up ? v[i] : v[i + N / 2];
        const float keep = up ? v[i + N / 2] : v[i];
        v[i] = keep + __shfl_xor(send, o, 64);
    }
}

__device__ __forceinline__ float wave_transpose_reduce64(float (&v)[64], int lane) {
    halve_step<64>(v, lane, 32);
    halve_step<32>(v, lane, 16);
    halve_step<16>(v, lane, 8);
    halve_step<8>(v, lane, 4);
    halve_step<4>(v, lane, 2);
    halve_step<2>(v, lane, 1);
    return v[0];
}

// HT_KJ = K / 64 is a template parameter: with a run-time bound every `if (j < KJ)` around a
// load is a branch, and behind a branch the load is waited for on the spot (64 row loads one
// after the other were 20 us of this kernel).
template <int A, int HT_KJ>
__global__ __launch_bounds__(kThreads) void k_dqn_head_td_loss(
    const float *__restrict__ h, const float *__restrict__ W, const float *__restrict__ bias,
    const int64_t *__restrict__ action, const float *__restrict__ target_q,
    const float *__restrict__ next_q_online, const float *__restrict__ reward,
    const float *__restrict__ discount, const float *__restrict__ terminal,
    const float *__restrict__ weights, int B, int K, int clip_delta, int mean,
    float *__restrict__ out_loss, float *__restrict__ out_y, float *__restrict__ out_abs_delta,
    float *__restrict__ dh, float *__restrict__ dW, float *__restrict__ db) {
    extern __shared__ float sm[];
    float *Ws = sm;                      // [A][K]
    float *red = Ws + A * K;             // [4][K]
    float *qs = red + 4 * K;             // [HT_ROWS][A]
    float *gs = qs + HT_ROWS * A;        // [B]  dL/dq of the taken action, all rows
    int *acts = reinterpret_cast<int *>(gs + B);   // [B]
    float *lsum = reinterpret_cast<float *>(acts + B);   // [HT_ROWS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float hv[8][HT_KJ];
    // this wave's 8 rows of a 32-row pass; every load is issued before anything waits
    auto load_rows = [&](int m0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int m = m0 + wave * 8 + r;
            const int mm = m < B ? m : B - 1;
#pragma unroll
            for (int j = 0; j < HT_KJ; ++j)
                hv[r][j] = h[(size_t)mm * K + lane + 64 * j];
        }
    };
    load_rows(0);
    // (inside the lane-0 branches below a global load would be waited for on the spot, 48
    // times over: the bias lives in registers)
    float bv[A];
#pragma unroll
    for (int a = 0; a < A; ++a) bv[a] = bias[a];
    // W into LDS, eight loads in flight per thread
    for (int e0 = tid; e0 < A * K; e0 += kThreads * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = W[min(e0 + kThreads * u, A * K - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + kThreads * u < A * K) Ws[e0 + kThreads * u] = v[u];
    }
    if (tid < HT_ROWS) lsum[tid] = 0.f;
    __syncthreads();
    const float scale = mean ? 1.0f / (float)B : 1.0f;
    float dWacc[A][HT_KJ];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int j = 0; j < HT_KJ; ++j) dWacc[a][j] = 0.f;

    for (int m0 = 0; m0 < B; m0 += HT_ROWS) {
        if (m0 > 0) load_rows(m0);
        // the row's loss inputs, in flight while the head is computed
        const bool rowt = tid < HT_ROWS && m0 + tid < B;
        const int brow = rowt ? m0 + tid : 0;
        float tqv[A], nqv[A];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            tqv[a] = target_q[(size_t)brow * A + a];
            nqv[a] = next_q_online ? next_q_online[(size_t)brow * A + a] : tqv[a];
        }
        const int act_in = (int)action[brow];
        const float rew_in = reward[brow], disc_in = discount[brow], term_in = terminal[brow];
        const float wt_in = weights ? weights[brow] : 1.0f;
        // q = h W^T (+ b where it is read): every lane's partial dot products of the wave's
        // 8 rows x A actions, then one transpose-reduction; lane i ends with element i
        static_assert(8 * A <= 128, "two reductions of 64 cover 8 rows x 16 actions");
#pragma unroll
        for (int part = 0; part < (8 * A + 63) / 64; ++part) {
            float pv[64];
#pragma unroll
            for (int e = 0; e < 64; ++e) {
                const int idx = part * 64 + e;           // = r * A + a
                const int r = idx / A, a = idx - r * A;
                float p = 0.f;
                if (idx < 8 * A) {
#pragma unroll
                    for (int j = 0; j < HT_KJ; ++j) p = fmaf(hv[r][j], Ws[a * K + lane + 64 * j], p);
                }
                pv[e] = p;
            }
            const float tot = wave_transpose_reduce64(pv, lane);
            const int idx = part * 64 + lane;
            if (idx < 8 * A) qs[wave * 8 * A + idx] = tot;
        }
        __syncthreads();
        // TD loss of the pass's rows (one thread per row), as k_dqn_td_loss
        if (rowt) {
            const int b = brow;
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
            const int act = act_in;
            float bact = bv[0];
#pragma unroll
            for (int a = 1; a < A; ++a) bact = (a == act) ? bv[a] : bact;
            const float y = qs[tid * A + act] + bact;
            const float coef = __fmul_rn(disc_in, __fsub_rn(1.0f, term_in));
            const float t = __fadd_rn(rew_in, __fmul_rn(coef, next));
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
            const float w = wt_in;
            lsum[tid] += l * w;
            gs[b] = g * w * scale;
            acts[b] = act;
            out_y[b] = y;
            out_abs_delta[b] = ad;
        }
        __syncthreads();
        // dL/dh rows and this wave's share of dL/dW
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int m = m0 + wave * 8 + r;
            if (m >= B) continue;   // (uniform per wave)
            const float g = gs[m];
            const int am = acts[m];
#pragma unroll
            for (int j = 0; j < HT_KJ; ++j)
                dh[(size_t)m * K + lane + 64 * j] = g * Ws[am * K + lane + 64 * j];
#pragma unroll
            for (int a = 0; a < A; ++a) {
                const float ga = (a == am) ? g : 0.f;
#pragma unroll
                for (int j = 0; j < HT_KJ; ++j) dWacc[a][j] = fmaf(ga, hv[r][j], dWacc[a][j]);
            }
        }
        __syncthreads();
    }
    // loss: the per-row-slot sums folded by wave 0
    if (wave == 0) {
        float v = lane < HT_ROWS ? lsum[lane] : 0.f;
        v = wave_sum_f(v);
        if (lane == 0) out_loss[0] = v * scale;
    }
    // dL/db[a] = sum over the rows that took action a, in row order
    if (tid < A) {
        float s = 0.f;
        for (int m = 0; m < B; ++m) s += (acts[m] == tid) ? gs[m] : 0.f;
        db[tid] = s;
    }
    // dL/dW: fold the four waves' shares, one action at a time
#pragma unroll
    for (int a = 0; a < A; ++a) {
#pragma unroll
        for (int j = 0; j < HT_KJ; ++j) red[wave * K + lane + 64 * j] = dWacc[a][j];
        __syncthreads();
        for (int k = tid; k < K; k += kThreads)
            dW[(size_t)a * K + k] = (red[k] + red[K + k]) + (red[2 * K + k] + red[3 * K + k]);
        __syncthreads();
    }
}

}  // namespace

extern "C" int pfrl_dqn_head_td_loss(const float *h, const float *w, const float *bias,
                                     const int64_t *action, const float *target_q,
                                     const float *next_q_online, const float *reward,
                                     const float *discount, const float *terminal, const float *weights,
                                     int32_t B, int32_t K, int32_t A, int clip_delta, int mean,
                                     float *out_loss, float *out_y, float *out_abs_delta, float *dh,
                                     float *dw, float *db, void *stream) {
    PFRL_CHECK_ARG(B >= 1 && B <= 1024 && A >= 1 && A <= 16 && (K == 512 || K == 256),
                   "pfrl_dqn_head_td_loss: B <= 1024, A <= 16, K = 256 or 512");
    const size_t lds = ((size_t)A * K + 4 * K + HT_ROWS * A + 2 * (size_t)B + HT_ROWS) * sizeof(float);
    PFRL_CHECK_ARG(lds <= 64 * 1024, "pfrl_dqn_head_td_loss: LDS budget");
#define CALL_HT(AA)                                                                                \
    do {                                                                                           \
        if (K == 512)                                                                              \
            hipLaunchKernelGGL((k_dqn_head_td_loss<AA, 8>), dim3(1), dim3(kThreads), lds,          \
                               (hipStream_t)stream, h, w, bias, action, target_q, next_q_online,   \
                               reward, discount, terminal, weights, B, K, clip_delta, mean,        \
                               out_loss, out_y, out_abs_delta, dh, dw, db);                        \
        else                                                                                       \
            hipLaunchKernelGGL((k_dqn_head_td_loss<AA, 4>), dim3(1), dim3(kThreads), lds,          \
                               (hipStream_t)stream, h, w, bias, action, target_q, next_q_online,   \
                               reward, discount, terminal, weights, B, K, clip_delta, mean,        \
                               out_loss, out_y, out_abs_delta, dh, dw, db);                        \
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
