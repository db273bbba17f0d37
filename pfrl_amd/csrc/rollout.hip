// On-policy rollout kernels: GAE / return reverse scans, advantage statistics
// with wavefront (64-lane) shuffle reductions, PPO minibatch assembly.
// Layout [T][N] with the env index minor, so the 64 lanes of a wave read 64
// consecutive envs of one time step (coalesced) and each lane owns one env's
// sequential scan (pfrl/agents/ppo.py:36-47 is inherently sequential in t).
#include <hip/hip_ext.h>

#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int kThreads = 256;

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_gae_scan(int64_t T, int64_t N,
                                                       const double *__restrict__ reward,
                                                       const float *__restrict__ v_pred,
                                                       const float *__restrict__ next_v_pred,
                                                       const uint8_t *__restrict__ nonterminal,
                                                       const uint8_t *__restrict__ cut, double gamma,
                                                       double lambd, float *__restrict__ adv_out,
                                                       float *__restrict__ vt_out) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= N) return;
    const double gl = __dmul_rn(gamma, lambd);
    if (MODE == 0) {
        const float glf = (float)gl;
        float adv = 0.0f;
        for (int64_t t = T - 1; t >= 0; --t) {
            const int64_t i = t * N + e;
            if (cut[i]) adv = 0.0f;
            const double gn = nonterminal[i] ? gamma : __dmul_rn(gamma, 0.0);
            const float prod = __fmul_rn((float)gn, next_v_pred[i]);
            const float s1 = __fadd_rn((float)reward[i], prod);
            const float td = __fsub_rn(s1, v_pred[i]);
            const float ga = __fmul_rn(glf, adv);
            adv = __fadd_rn(td, ga);
            adv_out[i] = adv;
            vt_out[i] = __fadd_rn(adv, v_pred[i]);
        }
    } else {
        double adv = 0.0;
        for (int64_t t = T - 1; t >= 0; --t) {
            const int64_t i = t * N + e;
            if (cut[i]) adv = 0.0;
            const double gn = nonterminal[i] ? gamma : __dmul_rn(gamma, 0.0);
            // MODE 1: np.float32 values under NEP 50 (the product rounds to f32);
            // MODE 2: Python floats throughout (the recurrent dataset, ppo.py:98-107): f64
            const double prod = MODE == 2 ? __dmul_rn(gn, (double)next_v_pred[i])
                                          : (double)__fmul_rn((float)gn, next_v_pred[i]);
            const double s1 = __dadd_rn(reward[i], prod);
            const double td = __dsub_rn(s1, (double)v_pred[i]);
            const double ga = __dmul_rn(gl, adv);
            adv = __dadd_rn(td, ga);
            adv_out[i] = (float)adv;
            vt_out[i] = (float)__dadd_rn(adv, (double)v_pred[i]);
        }
    }
}

// The same scan with the rollout staged through LDS (the form pfrl_gae_scan launches).  The scan
// of pfrl/agents/ppo.py:36-47 is sequential in t and its rounding is part of the parity contract
// (adv = fl(td + fl(gl * adv))), so it cannot become an associative prefix scan; what CAN leave
// the dependent chain is everything else.  A workgroup owns E envs (8, or 16 from 4 096 envs on):
//   1. all 256 threads load the five input columns of the E envs for every t (independent,
//      coalesced in 64..128-byte runs) and compute td[t][e] -- the part with the memory latency --
//      into LDS, together with v_pred and the cut flag;
//   2. one lane per env runs the recurrence out of LDS, 16 steps at a time through registers: the
//      chain is one multiply and one add per step, no load in it;
//   3. all threads write adv and v_teacher back, coalesced.
// One lane per env in global memory (k_gae_scan above) was 8 waves on the chip at N = 512 walking
// 128 dependent iterations of 5 global loads each; here N / 16 workgroups issue all loads at once.
constexpr int kGaeChunk = 16;

template <int MODE, int E>
__global__ __launch_bounds__(kThreads) void k_gae_scan_lds(
    int T, int64_t N, const double *__restrict__ reward, const float *__restrict__ v_pred,
    const float *__restrict__ next_v_pred, const uint8_t *__restrict__ nonterminal,
    const uint8_t *__restrict__ cut, double gamma, double lambd, float *__restrict__ adv_out,
    float *__restrict__ vt_out) {
    using acc_t = typename std::conditional<MODE == 0, float, double>::type;
    extern __shared__ unsigned char lds_raw[];
    acc_t *s_td = reinterpret_cast<acc_t *>(lds_raw);                 // [T][E], becomes adv
    float *s_v = reinterpret_cast<float *>(s_td + (size_t)T * E);     // [T][E]
    uint8_t *s_cut = reinterpret_cast<uint8_t *>(s_v + (size_t)T * E);
    const int64_t e0 = (int64_t)blockIdx.x * E;
    const int total = T * E;
    const double gl = __dmul_rn(gamma, lambd);
    const double g0 = __dmul_rn(gamma, 0.0);
    // (no branch around the loads: an env past N reads env N - 1 and is never looked at again --
    // behind a branch every load is waited for on the spot, five round trips per element)
    for (int q0 = 0; q0 < total; q0 += kThreads) {
        const int q = min(q0 + (int)threadIdx.x, total - 1);
        const int t = q / E, j = q % E;
        const int64_t e = min(e0 + j, N - 1);
        const int64_t i = (int64_t)t * N + e;
        const double r = reward[i];
        const float v = v_pred[i], nv = next_v_pred[i];
        const uint8_t nt = nonterminal[i], c = cut[i];
        const double gn = nt ? gamma : g0;
        const float prod = __fmul_rn((float)gn, nv);
        if (MODE == 0) {
            const float s1 = __fadd_rn((float)r, prod);
            s_td[q] = (acc_t)__fsub_rn(s1, v);
        } else {
            // (MODE 2: every operand a Python float -- the product stays in f64)
            const double pd = MODE == 2 ? __dmul_rn(gn, (double)nv) : (double)prod;
            const double s1 = __dadd_rn(r, pd);
            s_td[q] = (acc_t)__dsub_rn(s1, (double)v);
        }
        s_v[q] = v;
        s_cut[q] = c;
    }
    __syncthreads();
    if (threadIdx.x < E && e0 + threadIdx.x < N) {
        const int j = threadIdx.x;
        const acc_t glx = (acc_t)gl;
        acc_t adv = (acc_t)0;
        for (int t1 = T; t1 > 0; t1 -= kGaeChunk) {
            acc_t td[kGaeChunk];
            uint8_t c[kGaeChunk];
#pragma unroll
            for (int u = 0; u < kGaeChunk; ++u) {
                const int t = t1 - 1 - u;
                const int q = (t >= 0 ? t : 0) * E + j;
                td[u] = s_td[q];
                c[u] = s_cut[q];
            }
            // (no early exit for a ragged last chunk: steps past t = 0 come LAST in the chain, run on
            // clamped operands and are never stored -- a `break` here made the compiler index td[] /
            // c[] at run time: 16-way select chains, 65 instructions per step, 28 us per launch)
#pragma unroll
            for (int u = 0; u < kGaeChunk; ++u) {
                if (c[u]) adv = (acc_t)0;
                if (MODE == 0) adv = (acc_t)__fadd_rn((float)td[u], __fmul_rn((float)glx, (float)adv));
                else adv = (acc_t)__dadd_rn((double)td[u], __dmul_rn((double)glx, (double)adv));
                td[u] = adv;
            }
#pragma unroll
            for (int u = 0; u < kGaeChunk; ++u) {
                const int t = t1 - 1 - u;
                if (t >= 0) s_td[t * E + j] = td[u];
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < total; q += kThreads) {
        const int t = q / E, j = q % E;
        const int64_t e = e0 + j;
        if (e >= N) continue;
        const int64_t i = (int64_t)t * N + e;
        if (MODE == 0) {
            const float a = (float)s_td[q];
            adv_out[i] = a;
            vt_out[i] = __fadd_rn(a, s_v[q]);
        } else {
            const double a = (double)s_td[q];
            adv_out[i] = (float)a;
            vt_out[i] = (float)__dadd_rn(a, (double)s_v[q]);
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_a2c_returns(int64_t T, int64_t N,
                                                          const float *__restrict__ rewards,
                                                          const float *__restrict__ masks,
                                                          const float *__restrict__ value_preds,
                                                          float *__restrict__ returns, double gamma_d,
                                                          double tau_d, int use_gae) {
    const int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (e >= N) return;
    // Python floats meet f32 tensors: each scalar is rounded to f32 once.
    const float gamma = (float)gamma_d;
    if (use_gae) {
        // a2c.py:151-161 ; gae starts as Python int 0; gamma*tau is an f64 product
        float gae = 0.0f;
        const float gt = (float)__dmul_rn(gamma_d, tau_d);
        for (int64_t t = T - 1; t >= 0; --t) {
            const int64_t i = t * N + e;
            const float a = __fmul_rn(gamma, value_preds[i + N]);
            const float b = __fmul_rn(a, masks[i]);
            const float c = __fadd_rn(rewards[i], b);
            const float delta = __fsub_rn(c, value_preds[i]);
            const float g2 = __fmul_rn(gt, masks[i]);
            const float g3 = __fmul_rn(g2, gae);
            gae = __fadd_rn(delta, g3);
            returns[i] = __fadd_rn(gae, value_preds[i]);
        }
    } else {
        // a2c.py:162-167
        float nxt = returns[T * N + e];
        for (int64_t t = T - 1; t >= 0; --t) {
            const int64_t i = t * N + e;
            const float a = __fmul_rn(gamma, nxt);
            const float b = __fmul_rn(a, masks[i]);
            nxt = __fadd_rn(rewards[i], b);
            returns[i] = nxt;
        }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Stage 1: per-block partial (sum, sum of squares) in f64.
__global__ __launch_bounds__(kThreads) void k_adv_partial(const float *__restrict__ adv, int64_t n,
                                                          double *__restrict__ partial) {
    __shared__ double s1[kThreads / 64], s2[kThreads / 64];
    double a = 0.0, b = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const double x = (double)adv[i];
        a += x;
        b += x * x;
    }
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) {
        s1[w] = a;
        s2[w] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double x = 0.0, y = 0.0;
        for (int k = 0; k < kThreads / 64; ++k) {
            x += s1[k];
            y += s2[k];
        }
        partial[2 * blockIdx.x] = x;
        partial[2 * blockIdx.x + 1] = y;
    }
}

// Stage 2: one wave folds the partials; std with unbiased=False.
__global__ __launch_bounds__(64) void k_adv_final(const double *__restrict__ partial, int nblocks,
                                                  int64_t n, float *__restrict__ out) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) {
        a += partial[2 * i];
        b += partial[2 * i + 1];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0) {
        const double mean = a / (double)n;
        double var = b / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        out[0] = (float)mean;
        out[1] = (float)sqrt(var);
    }
}

__global__ __launch_bounds__(kThreads) void k_ppo_minibatch(
    int64_t M, const int64_t *__restrict__ idx, const float *__restrict__ adv,
    const float *__restrict__ mean_std, int standardize, const float *__restrict__ log_prob,
    const float *__restrict__ v_pred, const float *__restrict__ v_teacher,
    const int64_t *__restrict__ action, const int32_t *__restrict__ state_refs, int32_t k,
    float *__restrict__ out_adv, float *__restrict__ out_logp, float *__restrict__ out_v,
    float *__restrict__ out_vt, int64_t *__restrict__ out_action, int32_t *__restrict__ out_refs) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= M) return;
    const int64_t p = idx[i];
    float a = adv[p];
    if (standardize) {
        // ppo.py:494-495  (advs - mean_advs) / (std_advs + 1e-8)
        const float den = __fadd_rn(mean_std[1], 1e-8f);
        a = __fdiv_rn(__fsub_rn(a, mean_std[0]), den);
    }
    out_adv[i] = a;
    out_logp[i] = log_prob[p];
    out_v[i] = v_pred[p];
    out_vt[i] = v_teacher[p];
    out_action[i] = action[p];
    for (int j = 0; j < k; ++j) out_refs[i * k + j] = state_refs[p * k + j];
}

// ---------------------------------------------------------------------------------------
// The two narrow heads of the PPO example network on the acting path, in ONE launch:
//   Branched(Sequential(Linear(K, A), SoftmaxCategoricalHead()), Linear(K, 1))
// (examples/atari/train_ppo_ale.py:257-263) followed by what PPO.batch_act does with the result
// (pfrl/agents/ppo.py:759-778): action ~ Categorical(logits), its entropy (entropy_record) and the
// state value (value_record).  Through torch.distributions that is two library GEMMs and ~35
// elementwise / reduction launches on [N, A] numbers (softmax, logsumexp, multinomial's exponential
// draw + argmax, entropy's masked products, two host-assert kernels): 180 us per 512-env step
// against 135 us for the convolution trunk in front of them.
// One wave per row: lanes split K, A + 1 dot products by wave reduction, then every lane holds the
// logits; softmax / entropy in registers; the action by inverse CDF on one uniform per row (drawn
// by the caller: torch's Philox stream, one rand launch).  A <= 31.
// ---------------------------------------------------------------------------------------
constexpr int ACT_MAX_OUT = 32;

// (A is a template parameter: with a run-time bound every weight load sits behind a branch and is
// waited for on the spot -- 20 us for 512 rows; with the loops unrolled at compile time all loads of
// a 64-column slice are in flight together)
template <int A>
__global__ __launch_bounds__(256) void k_ppo_act_head(
    const float *__restrict__ h, const float *__restrict__ wp, const float *__restrict__ bp,
    const float *__restrict__ wv, const float *__restrict__ bv, const float *__restrict__ u,
    const int64_t *__restrict__ given, int64_t *__restrict__ action, float *__restrict__ entropy,
    float *__restrict__ value, float *__restrict__ log_prob, int N, int K,
    const int32_t *__restrict__ rows) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x * 4 + wave;
    if (row >= N) return;
    // (rows != NULL: a captured rollout step writes straight into the rollout's columns -- the
    // action into row rows[0] of an [T][N] column, entropy / value into slot rows[1] of a ring of
    // [2][N] blocks -- the indices arrive with the step's staging transfer, so the graph's
    // addresses stay fixed and no copy launch follows it, agents/ppo.py::_ActGraph)
    if (rows != nullptr) {
        const int ra = rows[0], rs = rows[1];
        if (action != nullptr) action += (size_t)ra * N;
        if (entropy != nullptr) entropy += (size_t)rs * 2 * N;
        value += (size_t)rs * 2 * N;
    }
    float acc[A + 1];
#pragma unroll
    for (int j = 0; j <= A; ++j) acc[j] = 0.f;
    const float *hr = h + (size_t)row * K;
    for (int k0 = 0; k0 < K; k0 += 128) {
        // two 64-column slices per round: 2 (A + 2) independent loads in flight
        float x[2], w[2][A + 1];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = k0 + 64 * s + lane;
            const int kk = k < K ? k : 0;
            x[s] = hr[kk];
#pragma unroll
            for (int j = 0; j < A; ++j) w[s][j] = wp[(size_t)j * K + kk];
            w[s][A] = wv[kk];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float xs = (k0 + 64 * s + lane) < K ? x[s] : 0.f;
#pragma unroll
            for (int j = 0; j <= A; ++j) acc[j] = fmaf(xs, w[s][j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j <= A; ++j) {
        float v = acc[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[j] = v + (j < A ? bp[j] : bv[0]);
    }
    if (lane != 0) return;
    float m = acc[0];
#pragma unroll
    for (int j = 1; j < A; ++j) m = fmaxf(m, acc[j]);
    float sum = 0.f;
    float e[A];
#pragma unroll
    for (int j = 0; j < A; ++j) {
        e[j] = expf(acc[j] - m);
        sum += e[j];
    }
    const float lse = m + logf(sum);
    // Categorical.entropy(): -sum p log p with log p = logits - logsumexp (probabilities of exactly
    // zero contribute nothing); the action: first j with u * sum < e_0 + .. + e_j
    // (given != NULL: the value pass of an update -- log pi(a | s) of the RECORDED action, no draw)
    float ent = 0.f, cum = 0.f, lp_a = 0.f;
    const float ur = given == nullptr ? u[row] * sum : 0.f;
    const int ga = given != nullptr ? (int)given[row] : -1;
    int a = -1;
#pragma unroll
    for (int j = 0; j < A; ++j) {
        const float lp = acc[j] - lse;
        const float p = e[j] / sum;
        ent -= p > 0.f ? p * lp : 0.f;
        cum += e[j];
        const bool take = given != nullptr ? j == ga : (a < 0 && (ur < cum || j == A - 1));
        if (take) {
            a = j;
            lp_a = lp;
        }
    }
    if (action != nullptr) action[row] = a;
    if (entropy != nullptr) entropy[row] = ent;
    value[row] = acc[A];
    if (log_prob != nullptr) log_prob[row] = lp_a;
}

// ---------------------------------------------------------------------------------------
// PPO._lossfun (pfrl/agents/ppo.py:634-671) on the logits and values of a minibatch, forward AND
// the gradient with respect to both, in one launch + a 1-workgroup finish:
//   prob_ratio = exp(log pi(a|s) - log_prob_old)
//   loss_policy = -mean(min(prob_ratio adv, clamp(prob_ratio, 1 - eps, 1 + eps) adv))
//   loss_value  = mean((v - v_teacher)^2)                                   (clip_eps_vf < 0)
//               = mean(max((v - vt)^2, (clip(v, v_old -+ eps_vf) - vt)^2))  (otherwise)
//   loss_entropy = -mean(H(Categorical(logits)))
//   loss = loss_policy + value_func_coef loss_value + entropy_coef loss_entropy
// Through torch.distributions + autograd that is ~45 elementwise / reduction launches forward and
// as many backward on [M, A] numbers: 0.5 ms of a 2 ms update at the 8-GPU rank's minibatch of
// 2 048 (profiles/r06_ppo_rank_shape.txt), 4 % at 16 384.  One thread per row; torch's tie rules
// are kept (min / max split the gradient of equal operands, clamp passes it at the bounds).
// The sums over the batch are f64 per workgroup, folded in index order by the finish kernel.
// ---------------------------------------------------------------------------------------
// one row of the loss: logits z, value v -> the gradient with respect to both (g, gv: 1 / M and the
// coefficients included) and the row's three terms (-surrogate, value loss, entropy)
template <int A>
__device__ __forceinline__ void ppo_loss_row(const float (&z)[A], float v, int a, float ad, float lpo,
                                             float vo, float vt, float inv_m, float clip_eps,
                                             float clip_eps_vf, float vf_coef, float ent_coef,
                                             float (&g)[A], float &gv_out, double &pol, double &val,
                                             double &ent) {
    float mx = z[0];
#pragma unroll
    for (int j = 1; j < A; ++j) mx = fmaxf(mx, z[j]);
    float e[A], sum = 0.f;
#pragma unroll
    for (int j = 0; j < A; ++j) {
        e[j] = expf(z[j] - mx);
        sum += e[j];
    }
    const float lse = mx + logf(sum);
    float H = 0.f, lpa = 0.f, p[A], lp[A];
#pragma unroll
    for (int j = 0; j < A; ++j) {
        lp[j] = z[j] - lse;
        p[j] = e[j] / sum;
        H -= p[j] > 0.f ? p[j] * lp[j] : 0.f;
        if (j == a) lpa = lp[j];
    }
    const float ratio = expf(lpa - lpo);
    const float lo = 1.0f - clip_eps, hi = 1.0f + clip_eps;
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float s1 = ratio * ad, s2 = rc * ad;
    const float surr = fminf(s1, s2);
    // d surr / d ratio: both operands of min carry it inside the clip range (a tie: half
    // each, summing to adv); outside only the unclipped product does, if it is the minimum
    const bool inside = ratio >= lo && ratio <= hi;
    float ds = 0.f;
    if (inside) ds = ad;
    else if (s1 < s2) ds = ad;
    else if (s1 == s2) ds = 0.5f * ad;
    const float g_lpa = -inv_m * ds * ratio;
    const float ge = ent_coef * inv_m;
#pragma unroll
    for (int j = 0; j < A; ++j) {
        const float onehot = j == a ? 1.f : 0.f;
        float gj = g_lpa * (onehot - p[j]);
        gj += p[j] > 0.f ? ge * p[j] * (lp[j] + H) : 0.f;
        g[j] = gj;
    }
    const float d1 = v - vt;
    float lv = d1 * d1, gv = 2.f * d1;
    if (clip_eps_vf >= 0.f) {
        const float vlo = vo - clip_eps_vf, vhi = vo + clip_eps_vf;
        const float vc = fminf(fmaxf(v, vlo), vhi);
        const float d2 = vc - vt;
        const float l2 = d2 * d2;
        // d vc / d v: torch.min(torch.max(v, lo), hi) -- 1 strictly inside, 1/2 at a bound
        // (max / min split ties), 0 outside
        float dvc = (v > vlo && v < vhi) ? 1.f : ((v == vlo || v == vhi) ? 0.5f : 0.f);
        if (l2 > lv) {
            lv = l2;
            gv = 2.f * d2 * dvc;
        } else if (l2 == lv) {
            gv = 0.5f * (2.f * d1) + 0.5f * (2.f * d2 * dvc);
        }
    }
    gv_out = vf_coef * inv_m * gv;
    pol = -(double)surr;
    val = (double)lv;
    ent = (double)H;
}

template <int A>
__global__ __launch_bounds__(kThreads) void k_ppo_loss(
    const float *__restrict__ logits, const float *__restrict__ value,
    const int64_t *__restrict__ action, const float *__restrict__ adv,
    const float *__restrict__ logp_old, const float *__restrict__ v_old,
    const float *__restrict__ v_teacher, int M, float clip_eps, float clip_eps_vf, float vf_coef,
    float ent_coef, float *__restrict__ dlogits, float *__restrict__ dvalue,
    double *__restrict__ partial) {
    __shared__ double s_red[3][kThreads / 64];
    const int m = blockIdx.x * kThreads + threadIdx.x;
    const float inv_m = 1.0f / (float)M;
    double pol = 0.0, val = 0.0, ent = 0.0;
    if (m < M) {
        float z[A], g[A], gv;
#pragma unroll
        for (int j = 0; j < A; ++j) z[j] = logits[(size_t)m * A + j];
        ppo_loss_row<A>(z, value[m], (int)action[m], adv[m], logp_old[m],
                        clip_eps_vf >= 0.f ? v_old[m] : 0.f, v_teacher[m], inv_m, clip_eps, clip_eps_vf,
                        vf_coef, ent_coef, g, gv, pol, val, ent);
#pragma unroll
        for (int j = 0; j < A; ++j) dlogits[(size_t)m * A + j] = g[j];
        dvalue[m] = gv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        pol += __shfl_xor(pol, o, 64);
        val += __shfl_xor(val, o, 64);
        ent += __shfl_xor(ent, o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_red[0][wave] = pol;
        s_red[1][wave] = val;
        s_red[2][wave] = ent;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) t += s_red[threadIdx.x][w];
        partial[(size_t)blockIdx.x * 3 + threadIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------
// The two narrow heads of the example network, the loss and the heads' backward in ONE launch:
// logits = h Wp^T + bp, v = h Wv^T + bv, the row's loss gradient (ppo_loss_row), then
//   dh = g Wp + gv Wv           (written: where backward of the body starts)
//   dWp, dbp, dWv, dbv          (per-workgroup partial slabs [A + 1][K] + [A + 1], folded afterwards)
// h is read once and dh written once (2 x 33.5 MB at 16 384 x 512) where the library route runs
// five narrow GEMMs, two elementwise adds and the loss launch over the same rows (~180 us per
// minibatch, profiles/r06_ppo_kernel_stats.csv).  A wave owns a row: lane l holds columns
// 4 l .. 4 l + 3 of every 256-column block of h, of the A + 1 weight rows (registers, loaded once)
// and of the A + 1 gradient rows it accumulates; the A + 1 dot products are folded across the wave
// by butterflies.  K = 256 or 512, A <= 9.
// ---------------------------------------------------------------------------------------
template <int A, int KQ>
__global__ __launch_bounds__(256) void k_ppo_head_loss(
    const float *__restrict__ h, const float *__restrict__ wp, const float *__restrict__ bp,
    const float *__restrict__ wv, const float *__restrict__ bv, const int64_t *__restrict__ action,
    const float *__restrict__ adv, const float *__restrict__ logp_old, const float *__restrict__ v_old,
    const float *__restrict__ v_teacher, int M, int rows_per_block, float clip_eps, float clip_eps_vf,
    float vf_coef, float ent_coef, float *__restrict__ dh, float *__restrict__ dw_part,
    double *__restrict__ partial) {
    constexpr int K = 256 * KQ, NO = A + 1;
    __shared__ float s_w[NO * K + NO];
    __shared__ double s_red[3][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float inv_m = 1.0f / (float)M;
    float4 w[NO][KQ], dw[NO][KQ];
    float db[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const float *src = j < A ? wp + (size_t)j * K : wv;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            w[j][q] = *reinterpret_cast<const float4 *>(src + 256 * q + 4 * lane);
            dw[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        db[j] = 0.f;
    }
    float bias[NO];
#pragma unroll
    for (int j = 0; j < A; ++j) bias[j] = bp[j];
    bias[A] = bv[0];
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
    double pol = 0.0, val = 0.0, ent = 0.0;
    float4 hv[KQ], hn[KQ];
    int r = r0 + wave;
    if (r < r1) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) hn[q] = *reinterpret_cast<const float4 *>(h + (size_t)r * K + 256 * q + 4 * lane);
    }
    for (; r < r1; r += 4) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) hv[q] = hn[q];
        if (r + 4 < r1) {       // the next row of this wave is in flight while this one is worked on
#pragma unroll
            for (int q = 0; q < KQ; ++q)
                hn[q] = *reinterpret_cast<const float4 *>(h + (size_t)(r + 4) * K + 256 * q + 4 * lane);
        }
        float z[NO];
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                s = fmaf(hv[q].x, w[j][q].x, s);
                s = fmaf(hv[q].y, w[j][q].y, s);
                s = fmaf(hv[q].z, w[j][q].z, s);
                s = fmaf(hv[q].w, w[j][q].w, s);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            z[j] = s + bias[j];
        }
        float zl[A], g[A], gv;
#pragma unroll
        for (int j = 0; j < A; ++j) zl[j] = z[j];
        double rp, rv, re;
        ppo_loss_row<A>(zl, z[A], (int)action[r], adv[r], logp_old[r],
                        clip_eps_vf >= 0.f ? v_old[r] : 0.f, v_teacher[r], inv_m, clip_eps, clip_eps_vf,
                        vf_coef, ent_coef, g, gv, rp, rv, re);
        pol += rp;
        val += rv;
        ent += re;
        float gg[NO];
#pragma unroll
        for (int j = 0; j < A; ++j) gg[j] = g[j];
        gg[A] = gv;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                d.x = fmaf(gg[j], w[j][q].x, d.x);
                d.y = fmaf(gg[j], w[j][q].y, d.y);
                d.z = fmaf(gg[j], w[j][q].z, d.z);
                d.w = fmaf(gg[j], w[j][q].w, d.w);
                dw[j][q].x = fmaf(gg[j], hv[q].x, dw[j][q].x);
                dw[j][q].y = fmaf(gg[j], hv[q].y, dw[j][q].y);
                dw[j][q].z = fmaf(gg[j], hv[q].z, dw[j][q].z);
                dw[j][q].w = fmaf(gg[j], hv[q].w, dw[j][q].w);
            }
            *reinterpret_cast<float4 *>(dh + (size_t)r * K + 256 * q + 4 * lane) = d;
        }
#pragma unroll
        for (int j = 0; j < NO; ++j) db[j] += gg[j];
    }
    // the four waves' gradient rows folded in wave order through LDS, then one slab per workgroup
    for (int wv_i = 0; wv_i < 4; ++wv_i) {
        if (wave == wv_i) {
#pragma unroll
            for (int j = 0; j < NO; ++j) {
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    float4 *dst = reinterpret_cast<float4 *>(&s_w[j * K + 256 * q + 4 * lane]);
                    float4 v = dw[j][q];
                    if (wv_i > 0) {
                        const float4 o = *dst;
                        v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
                    }
                    *dst = v;
                }
                if (lane == 0) s_w[NO * K + j] = (wv_i > 0 ? s_w[NO * K + j] : 0.f) + db[j];
            }
            if (lane == 0) {
                s_red[0][wave] = pol;
                s_red[1][wave] = val;
                s_red[2][wave] = ent;
            }
        }
        __syncthreads();
    }
    // (slab stride: the bias block padded to a multiple of 4 floats, pfrl_splitk_reduce reads float4)
    float *slab = dw_part + (size_t)blockIdx.x * (NO * K + (NO + 3) / 4 * 4);
    for (int e = tid; e < NO * K + NO; e += 256) slab[e] = s_w[e];
    if (tid < 3) partial[(size_t)blockIdx.x * 3 + tid] = ((s_red[tid][0] + s_red[tid][1]) + s_red[tid][2]) + s_red[tid][3];
}

// out[0] = loss, out[1] = loss_policy, out[2] = loss_value, out[3] = mean entropy
__global__ __launch_bounds__(64) void k_ppo_loss_finish(const double *__restrict__ partial, int nblk,
                                                        int M, float vf_coef, float ent_coef,
                                                        float *__restrict__ out) {
    const int lane = threadIdx.x;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = lane; b < nblk; b += 64) {
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += partial[(size_t)b * 3 + k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
    }
    if (lane == 0) {
        const float pol = (float)(s[0] / M), val = (float)(s[1] / M), ent = (float)(s[2] / M);
        out[1] = pol;
        out[2] = val;
        out[3] = ent;
        out[0] = (pol + vf_coef * val) + ent_coef * (-ent);
    }
}

}  // namespace

extern "C" int pfrl_gae_scan(int64_t T, int64_t N, const double *reward, const float *v_pred,
                             const float *next_v_pred, const uint8_t *nonterminal,
                             const uint8_t *cut, double gamma, double lambd, int mode, float *adv,
                             float *v_teacher, void *stream) {
    PFRL_CHECK_ARG(T >= 0 && N >= 0 && mode >= 0 && mode <= 2, "pfrl_gae_scan: bad shape / mode");
    if (T == 0 || N == 0) return 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    pfrl_profile_events(PFRL_PROFILE_GAE_SCAN, T * N, &e0, &e1);
    // rollouts that fit a workgroup's LDS (T * E envs * 9 or 13 bytes <= 64 KB)
    // take the LDS-staged form; PFRL_GAE_LDS=0 keeps the lane-per-env loop for A/B runs
    static const bool use_lds = [] {
        const char *e = getenv("PFRL_GAE_LDS");
        return !(e != nullptr && e[0] == '0');
    }();
    const int E = N >= 4096 ? 16 : 8;
    const size_t lds = (size_t)T * E * ((mode == 0 ? 4 : 8) + 4 + 1);
    if (use_lds && lds <= 64 * 1024 && T < (1 << 20)) {
        const unsigned blocks = (unsigned)((N + E - 1) / E);
#define PFRL_GAE_LAUNCH(MODE_, E_)                                                                   \
    hipExtLaunchKernelGGL((k_gae_scan_lds<MODE_, E_>), dim3(blocks), dim3(kThreads), lds,            \
                          (hipStream_t)stream, e0, e1, 0, (int)T, N, reward, v_pred, next_v_pred,    \
                          nonterminal, cut, gamma, lambd, adv, v_teacher)
        if (mode == 0 && E == 8) PFRL_GAE_LAUNCH(0, 8);
        else if (mode == 0) PFRL_GAE_LAUNCH(0, 16);
        else if (mode == 1 && E == 8) PFRL_GAE_LAUNCH(1, 8);
        else if (mode == 1) PFRL_GAE_LAUNCH(1, 16);
        else if (E == 8) PFRL_GAE_LAUNCH(2, 8);
        else PFRL_GAE_LAUNCH(2, 16);
#undef PFRL_GAE_LAUNCH
        PFRL_LAUNCH_CHECK();
    }
    const unsigned blocks = (unsigned)((N + kThreads - 1) / kThreads);
    if (mode == 0)
        hipExtLaunchKernelGGL(k_gae_scan<0>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, e0, e1,
                              0, T, N, reward, v_pred, next_v_pred, nonterminal, cut, gamma, lambd, adv,
                              v_teacher);
    else if (mode == 1)
        hipExtLaunchKernelGGL(k_gae_scan<1>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, e0, e1,
                              0, T, N, reward, v_pred, next_v_pred, nonterminal, cut, gamma, lambd, adv,
                              v_teacher);
    else
        hipExtLaunchKernelGGL(k_gae_scan<2>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, e0, e1,
                              0, T, N, reward, v_pred, next_v_pred, nonterminal, cut, gamma, lambd, adv,
                              v_teacher);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_a2c_returns(int64_t T, int64_t N, const float *rewards, const float *masks,
                                const float *value_preds, float *returns, double gamma, double tau,
                                int use_gae, void *stream) {
    PFRL_CHECK_ARG(T >= 0 && N >= 0, "pfrl_a2c_returns: bad shape");
    if (T == 0 || N == 0) return 0;
    const unsigned blocks = (unsigned)((N + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_a2c_returns, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, T, N,
                       rewards, masks, value_preds, returns, gamma, tau, use_gae);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_adv_stats(const float *adv, int64_t n, float *out_mean_std, void *partial_ws,
                              void *stream) {
    PFRL_CHECK_ARG(n > 0 && partial_ws, "pfrl_adv_stats: need n > 0 and a workspace of 2*1024 f64");
    int blocks = (int)((n + kThreads * 8 - 1) / (kThreads * 8));
    if (blocks > 1024) blocks = 1024;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    pfrl_profile_events(PFRL_PROFILE_ADV_STATS, n, &e0, &e1);
    hipExtLaunchKernelGGL(k_adv_partial, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, e0, e1, 0,
                          adv, n, (double *)partial_ws);
    hipLaunchKernelGGL(k_adv_final, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const double *)partial_ws, blocks, n, out_mean_std);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_ppo_minibatch(int64_t M, const int64_t *idx, const float *adv,
                                  const float *mean_std, int standardize, const float *log_prob,
                                  const float *v_pred, const float *v_teacher,
                                  const int64_t *action, const int32_t *state_refs, int32_t k,
                                  float *out_adv, float *out_logp, float *out_v, float *out_vt,
                                  int64_t *out_action, int32_t *out_refs, void *stream) {
    if (M <= 0) return 0;
    const unsigned blocks = (unsigned)((M + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_ppo_minibatch, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, M,
                       idx, adv, mean_std, standardize, log_prob, v_pred, v_teacher, action,
                       state_refs, k, out_adv, out_logp, out_v, out_vt, out_action, out_refs);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_ppo_head_loss(const float *h, const float *w_policy, const float *b_policy,
                                  const float *w_value, const float *b_value, const int64_t *action,
                                  const float *adv, const float *log_prob_old, const float *v_pred_old,
                                  const float *v_teacher, int32_t M, int32_t K, int32_t A,
                                  float clip_eps, float clip_eps_vf, float value_func_coef,
                                  float entropy_coef, float *dh, float *dw_part, int32_t blocks,
                                  double *partial_ws, float *out4, void *stream) {
    PFRL_CHECK_ARG(M >= 1 && blocks >= 1 && A >= 1 && A <= 9 && (K == 256 || K == 512),
                   "pfrl_ppo_head_loss: 1 <= A <= 9, K = 256 or 512");
    PFRL_CHECK_ARG(h && w_policy && b_policy && w_value && b_value && action && adv && log_prob_old &&
                       v_teacher && dh && dw_part && partial_ws && out4 && (clip_eps_vf < 0.f || v_pred_old),
                   "pfrl_ppo_head_loss: null pointer");
    PFRL_CHECK_ARG((((uintptr_t)h | (uintptr_t)w_policy | (uintptr_t)w_value | (uintptr_t)dh) & 15) == 0,
                   "pfrl_ppo_head_loss: 16-byte aligned rows");
    const int rpb = (M + blocks - 1) / blocks;
#define HL_CALL(AA, KQ)                                                                             \
    hipLaunchKernelGGL((k_ppo_head_loss<AA, KQ>), dim3((unsigned)blocks), dim3(256), 0,             \
                       (hipStream_t)stream, h, w_policy, b_policy, w_value, b_value, action, adv,   \
                       log_prob_old, v_pred_old, v_teacher, M, rpb, clip_eps, clip_eps_vf,          \
                       value_func_coef, entropy_coef, dh, dw_part, partial_ws)
#define HL_CASE(AA)                                                                                 \
    case AA:                                                                                        \
        if (K == 256) HL_CALL(AA, 1);                                                               \
        else HL_CALL(AA, 2);                                                                        \
        break;
    switch (A) {
        HL_CASE(1) HL_CASE(2) HL_CASE(3) HL_CASE(4) HL_CASE(5) HL_CASE(6) HL_CASE(7) HL_CASE(8) HL_CASE(9)
    }
#undef HL_CASE
#undef HL_CALL
    hipLaunchKernelGGL(k_ppo_loss_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, partial_ws, blocks,
                       M, value_func_coef, entropy_coef, out4);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_ppo_act_head(const float *h, const float *w_policy, const float *b_policy,
                                 const float *w_value, const float *b_value, const float *u01,
                                 const int64_t *given_action, int64_t *out_action, float *out_entropy,
                                 float *out_value, float *out_log_prob, int32_t N, int32_t K, int32_t A,
                                 const int32_t *rows, void *stream) {
    PFRL_CHECK_ARG(N >= 0 && K >= 1 && A >= 1 && A < ACT_MAX_OUT, "pfrl_ppo_act_head: 1 <= A <= 31");
    PFRL_CHECK_ARG(h && w_policy && b_policy && w_value && b_value && out_value,
                   "pfrl_ppo_act_head: null pointer");
    PFRL_CHECK_ARG(given_action != nullptr || (u01 != nullptr && out_action != nullptr),
                   "pfrl_ppo_act_head: sampling needs u01 and out_action");
    if (N == 0) return 0;
#define ACT_CALL(AA)                                                                              \
    case AA:                                                                                      \
        hipLaunchKernelGGL(k_ppo_act_head<AA>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0,       \
                           (hipStream_t)stream, h, w_policy, b_policy, w_value, b_value, u01,     \
                           given_action, out_action, out_entropy, out_value, out_log_prob, N, K,  \
                           rows);                                                                 \
        break;
    switch (A) {
        ACT_CALL(1) ACT_CALL(2) ACT_CALL(3) ACT_CALL(4) ACT_CALL(5) ACT_CALL(6) ACT_CALL(7) ACT_CALL(8)
        ACT_CALL(9) ACT_CALL(10) ACT_CALL(11) ACT_CALL(12) ACT_CALL(13) ACT_CALL(14) ACT_CALL(15)
        ACT_CALL(16) ACT_CALL(17) ACT_CALL(18) ACT_CALL(19) ACT_CALL(20) ACT_CALL(21) ACT_CALL(22)
        ACT_CALL(23) ACT_CALL(24) ACT_CALL(25) ACT_CALL(26) ACT_CALL(27) ACT_CALL(28) ACT_CALL(29)
        ACT_CALL(30) ACT_CALL(31)
    }
#undef ACT_CALL
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_ppo_loss(const float *logits, const float *value, const int64_t *action,
                             const float *adv, const float *log_prob_old, const float *v_pred_old,
                             const float *v_teacher, int32_t M, int32_t A, float clip_eps,
                             float clip_eps_vf, float value_func_coef, float entropy_coef,
                             float *dlogits, float *dvalue, double *partial_ws, float *out4,
                             void *stream) {
    PFRL_CHECK_ARG(M >= 1 && A >= 1 && A < ACT_MAX_OUT, "pfrl_ppo_loss: 1 <= A <= 31, M >= 1");
    PFRL_CHECK_ARG(logits && value && action && adv && log_prob_old && v_teacher && dlogits && dvalue &&
                       partial_ws && out4 && (clip_eps_vf < 0.f || v_pred_old),
                   "pfrl_ppo_loss: null pointer");
    const unsigned blocks = (unsigned)((M + kThreads - 1) / kThreads);
#define LOSS_CALL(AA)                                                                              \
    case AA:                                                                                       \
        hipLaunchKernelGGL(k_ppo_loss<AA>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream,   \
                           logits, value, action, adv, log_prob_old, v_pred_old, v_teacher, M,     \
                           clip_eps, clip_eps_vf, value_func_coef, entropy_coef, dlogits, dvalue,  \
                           partial_ws);                                                            \
        break;
    switch (A) {
        LOSS_CALL(1) LOSS_CALL(2) LOSS_CALL(3) LOSS_CALL(4) LOSS_CALL(5) LOSS_CALL(6) LOSS_CALL(7)
        LOSS_CALL(8) LOSS_CALL(9) LOSS_CALL(10) LOSS_CALL(11) LOSS_CALL(12) LOSS_CALL(13) LOSS_CALL(14)
        LOSS_CALL(15) LOSS_CALL(16) LOSS_CALL(17) LOSS_CALL(18) LOSS_CALL(19) LOSS_CALL(20)
        LOSS_CALL(21) LOSS_CALL(22) LOSS_CALL(23) LOSS_CALL(24) LOSS_CALL(25) LOSS_CALL(26)
        LOSS_CALL(27) LOSS_CALL(28) LOSS_CALL(29) LOSS_CALL(30) LOSS_CALL(31)
    }
#undef LOSS_CALL
    hipLaunchKernelGGL(k_ppo_loss_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, partial_ws,
                       (int)blocks, M, value_func_coef, entropy_coef, out4);
    PFRL_LAUNCH_CHECK();
}
