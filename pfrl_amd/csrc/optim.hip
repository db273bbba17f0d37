// Fused multi-tensor RMSprop step (centered or not, no momentum): ONE launch
// updates every parameter of the network instead of the ~6 foreach kernels x
// parameter-list chunks PyTorch issues (73 us -> a few us per DQN update at
// 1.69 M parameters).  HBM-bound elementwise: reads p, g, square_avg,
// (grad_avg) and writes p, square_avg, (grad_avg): 28 B per element centered.
// Same arithmetic, in the same order, as torch.optim.RMSprop's foreach path
// (torch/optim/rmsprop.py _multi_tensor_rmsprop): every step below is one
// rounded f32 operation.
#include "common.h"
#include "rms_update.h"

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
        float pi = p[i], si = sq[i], mi = CENTERED ? ga[i] : 0.0f;
        rms_update<CENTERED>(pi, g[i], si, mi, lr, alpha, oma, eps, weight_decay);
        p[i] = pi;
        sq[i] = si;
        if (CENTERED) ga[i] = mi;
    }
}

// ---------------------------------------------------------------------------------------
// The optimizer step that FINISHES the gradients (DQN.update at minibatch size,
// pfrl/agents/dqn.py:360-365).  At B = 32 the backward pass of the example Q-network leaves
//   * split-K partial slabs of the convolution / head gradients (a fold launch away from being
//     gradients), and
//   * for the hidden Linear(3136, 512) layer -- 95 % of the parameters -- a weight gradient that
//     is a rank-B product dW = dy^T x of two small L2-resident matrices,
// and the optimizer then reads every gradient exactly once.  This launch takes the gradients
// in the form the backward pass left them: it sums the slabs while loading (task SLABS), forms
// the hidden layer's gradient tile by tile on the matrix cores (exact-f32 16x16x4 MFMA, task
// LOWRANK) and applies RMSprop straight from the accumulators, so the 6.4 MB weight gradient
// is never written or read, and the fold launch and the weight-gradient half of the hidden
// layer's backward launch disappear.  Arithmetic per element is rms_update() above.
// ---------------------------------------------------------------------------------------
constexpr int kMaxTasks = PFRL_OPT_MAX_TASKS;
constexpr int kTileK = 256, kTileF = 16, kMaxM = 32;   // one block: 16 rows of W x 1 KiB of each

struct FusedArgs {
    pfrl_opt_task_t t[kMaxTasks];
    int32_t block_end[kMaxTasks];
    int32_t n;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float slab_sum(const float *__restrict__ src, int64_t i, int32_t n_slabs,
                                          int64_t stride) {
    // the order of k_splitk_reduce (csrc/qnet.hip): 0 + slab 0 + slab 1 + ...; eight slabs in
    // flight per thread (the slabs come from memory: one round trip each if read one by one)
    float s = 0.0f;
    int k = 0;
    for (; k + 8 <= n_slabs; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + u) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = __fadd_rn(s, v[u]);
    }
    if (k < n_slabs) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)min(k + u, n_slabs - 1) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k + u < n_slabs) s = __fadd_rn(s, v[u]);
    }
    return s;
}

template <bool CENTERED>
__global__ __launch_bounds__(kThreads) void k_rmsprop_fused(FusedArgs a, float lr, float alpha,
                                                            float eps, float weight_decay) {
    __shared__ float s_x[kMaxM][kTileK + 4];    // x tile [m][kk]  (+4: bank spread of the rows)
    __shared__ float s_dy[kMaxM][kTileF + 4];   // masked dy tile [m][co]
    int ti = 0;
    const int b = blockIdx.x;
    while (ti < a.n - 1 && b >= a.block_end[ti]) ++ti;
    const pfrl_opt_task_t &t = a.t[ti];
    const int lb = b - (ti == 0 ? 0 : a.block_end[ti - 1]);
    const float oma = __fsub_rn(1.0f, alpha);
    const int tid = threadIdx.x;
    if (t.mode == PFRL_OPT_LOWRANK) {
        // tile (kt, ft) of W [F][K]: 16 rows (co) x 256 columns (kk) -- 1 KiB contiguous per row
        // and array, consecutive blocks walking along the rows (DRAM pages; 64-float tiles ran
        // at 1.6 TB/s).  Rows of the MFMA result = input index kk, columns = co, so that a
        // lane's four accumulator registers are four consecutive kk of one row of W.  The last
        // column tile may be partial (K % 64 == 0: whole waves drop out).
        const int tiles_k = (t.K + kTileK - 1) / kTileK;
        const int kt = lb % tiles_k, ft = lb / tiles_k;
        const int kk0 = kt * kTileK, co0 = ft * kTileF;
        const int kw = min(kTileK, t.K - kk0);             // valid columns of this tile
        const int M = t.M;
        // both operand tiles: every load of the thread is issued before the first LDS store
        // (a load behind a per-iteration branch is waited for on the spot: eight round trips)
        {
            float4 v[kMaxM / 4];
            const int c4 = (tid & 63) * 4, mrow = tid >> 6;      // 64 float4 per row, 4 rows / pass
            const int cc = min(c4, kw - 4);                       // (kw % 64 == 0, kw >= 64)
#pragma unroll
            for (int u = 0; u < kMaxM / 4; ++u) {
                const int m = min(u * 4 + mrow, M - 1);
                v[u] = *reinterpret_cast<const float4 *>(t.x + (int64_t)m * t.K + kk0 + cc);
            }
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f), r = make_float4(1.f, 1.f, 1.f, 1.f);
            const int dm = tid >> 2, dc = (tid & 3) * 4;          // dy tile: M rows x 4 float4
            if (dm < M) {
                d = *reinterpret_cast<const float4 *>(t.src + (int64_t)dm * t.F + co0 + dc);
                if (t.mask) r = *reinterpret_cast<const float4 *>(t.mask + (int64_t)dm * t.F + co0 + dc);
            }
#pragma unroll
            for (int u = 0; u < kMaxM / 4; ++u) {
                const int m = u * 4 + mrow;
                if (m < M && c4 < kw) {
                    s_x[m][c4] = v[u].x; s_x[m][c4 + 1] = v[u].y;
                    s_x[m][c4 + 2] = v[u].z; s_x[m][c4 + 3] = v[u].w;
                }
            }
            if (dm < M) {
                s_dy[dm][dc] = r.x > 0.f ? d.x : 0.f; s_dy[dm][dc + 1] = r.y > 0.f ? d.y : 0.f;
                s_dy[dm][dc + 2] = r.z > 0.f ? d.z : 0.f; s_dy[dm][dc + 3] = r.w > 0.f ? d.w : 0.f;
            }
        }
        __syncthreads();
        const int lane = tid & 63, wave = tid >> 6;
        const int l15 = lane & 15, l4 = lane >> 4;
        if (wave * 64 >= kw) return;                        // (partial last tile)
        const int co = co0 + l15;                           // this lane's column of the result
        const int kb = kk0 + wave * 64;                     // this wave's 64 columns of W
        // parameter and state of this lane's 4 x 4 outputs: requested before the products so
        // that their memory round trip runs under the LDS reads and the MFMA chain
        float4 pv[4], sv[4], mv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t off = (int64_t)co * t.K + kb + q * 16 + l4 * 4;
            pv[q] = *reinterpret_cast<const float4 *>(t.p + off);
            sv[q] = *reinterpret_cast<const float4 *>(t.sq + off);
            mv[q] = CENTERED ? *reinterpret_cast<const float4 *>(t.ga + off)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int m0 = 0; m0 < M; m0 += 4) {
            // B[k = m][col = co]; A[row = kk][k = m] = x[m][kk]
            const float bv = s_dy[m0 + l4][l15];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float av = s_x[m0 + l4][wave * 64 + q * 16 + l15];
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[q], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // rows 4 * l4 + r (r = 0..3) of tile q: W[co][kb + 16 q + 4 l4 .. + 3]
            const int64_t off = (int64_t)co * t.K + kb + q * 16 + l4 * 4;
            rms_update<CENTERED>(pv[q].x, acc[q][0], sv[q].x, mv[q].x, lr, alpha, oma, eps, weight_decay);
            rms_update<CENTERED>(pv[q].y, acc[q][1], sv[q].y, mv[q].y, lr, alpha, oma, eps, weight_decay);
            rms_update<CENTERED>(pv[q].z, acc[q][2], sv[q].z, mv[q].z, lr, alpha, oma, eps, weight_decay);
            rms_update<CENTERED>(pv[q].w, acc[q][3], sv[q].w, mv[q].w, lr, alpha, oma, eps, weight_decay);
            *reinterpret_cast<float4 *>(t.p + off) = pv[q];
            *reinterpret_cast<float4 *>(t.sq + off) = sv[q];
            if (CENTERED) *reinterpret_cast<float4 *>(t.ga + off) = mv[q];
        }
        return;
    }
    // elementwise tasks: a thread owns four ADJACENT elements (one float4 per array), so that
    // the slab sums keep eight 16-byte loads in flight instead of walking slab after slab
    const int64_t i0 = (int64_t)lb * kChunk + (int64_t)tid * 4;
    if (i0 >= t.numel) return;
    const int cnt = (int)min((int64_t)4, t.numel - i0);
    const bool vec = cnt == 4 && ((reinterpret_cast<uintptr_t>(t.src) | (uintptr_t)(t.slab_stride * 4)) & 15) == 0;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (t.mode == PFRL_OPT_PLAIN) {
        if (vec) {
            const float4 v = *reinterpret_cast<const float4 *>(t.src + i0);
            g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
        } else {
            for (int c = 0; c < cnt; ++c) g[c] = t.src[i0 + c];
        }
    } else if (t.mode == PFRL_OPT_LOWRANK_BIAS) {
        // db[co] = sum over the batch of the masked dy column
        for (int m = 0; m < t.M; ++m)
            for (int c = 0; c < cnt; ++c) {
                const float v = t.src[(int64_t)m * t.F + i0 + c];
                g[c] = __fadd_rn(g[c], (t.mask == nullptr || t.mask[(int64_t)m * t.F + i0 + c] > 0.f) ? v : 0.f);
            }
    } else if (vec) {
        // the order of k_splitk_reduce (csrc/qnet.hip): 0 + slab 0 + slab 1 + ...
        const float *base = t.src + i0;
        const int S = t.n_slabs;
        int k = 0;
        for (; k + 8 <= S; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = *reinterpret_cast<const float4 *>(base + (int64_t)(k + u) * t.slab_stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                g[0] = __fadd_rn(g[0], v[u].x); g[1] = __fadd_rn(g[1], v[u].y);
                g[2] = __fadd_rn(g[2], v[u].z); g[3] = __fadd_rn(g[3], v[u].w);
            }
        }
        if (k < S) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = *reinterpret_cast<const float4 *>(base + (int64_t)min(k + u, S - 1) * t.slab_stride);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k + u < S) {
                    g[0] = __fadd_rn(g[0], v[u].x); g[1] = __fadd_rn(g[1], v[u].y);
                    g[2] = __fadd_rn(g[2], v[u].z); g[3] = __fadd_rn(g[3], v[u].w);
                }
        }
    } else {
        for (int c = 0; c < cnt; ++c) g[c] = slab_sum(t.src, i0 + c, t.n_slabs, t.slab_stride);
    }
    if (t.mode == PFRL_OPT_FOLD) {
        for (int c = 0; c < cnt; ++c) t.out[i0 + c] = g[c];
        return;
    }
    const bool pvec = cnt == 4 && ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.sq) |
                                   reinterpret_cast<uintptr_t>(t.ga)) & 15) == 0;
    if (pvec) {
        float4 pv = *reinterpret_cast<float4 *>(t.p + i0);
        float4 sv = *reinterpret_cast<float4 *>(t.sq + i0);
        float4 mv = CENTERED ? *reinterpret_cast<float4 *>(t.ga + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
        rms_update<CENTERED>(pv.x, g[0], sv.x, mv.x, lr, alpha, oma, eps, weight_decay);
        rms_update<CENTERED>(pv.y, g[1], sv.y, mv.y, lr, alpha, oma, eps, weight_decay);
        rms_update<CENTERED>(pv.z, g[2], sv.z, mv.z, lr, alpha, oma, eps, weight_decay);
        rms_update<CENTERED>(pv.w, g[3], sv.w, mv.w, lr, alpha, oma, eps, weight_decay);
        *reinterpret_cast<float4 *>(t.p + i0) = pv;
        *reinterpret_cast<float4 *>(t.sq + i0) = sv;
        if (CENTERED) *reinterpret_cast<float4 *>(t.ga + i0) = mv;
        return;
    }
    for (int c = 0; c < cnt; ++c) {
        float pi = t.p[i0 + c], si = t.sq[i0 + c], mi = CENTERED ? t.ga[i0 + c] : 0.0f;
        rms_update<CENTERED>(pi, g[c], si, mi, lr, alpha, oma, eps, weight_decay);
        t.p[i0 + c] = pi;
        t.sq[i0 + c] = si;
        if (CENTERED) t.ga[i0 + c] = mi;
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

extern "C" int pfrl_rmsprop_fused_step(int32_t n_tasks, const pfrl_opt_task_t *host_tasks, float lr,
                                       float alpha, float eps, float weight_decay, int centered,
                                       void *stream) {
    PFRL_CHECK_ARG(n_tasks >= 0 && (n_tasks == 0 || host_tasks), "pfrl_rmsprop_fused_step: bad tasks");
    for (int lo = 0; lo < n_tasks; lo += kMaxTasks) {
        const int n = (n_tasks - lo) < kMaxTasks ? (n_tasks - lo) : kMaxTasks;
        FusedArgs a;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            const pfrl_opt_task_t &t = host_tasks[lo + i];
            a.t[i] = t;
            if (t.mode == PFRL_OPT_LOWRANK) {
                PFRL_CHECK_ARG(t.K % 64 == 0 && t.F % kTileF == 0 && t.M % 4 == 0 && t.M >= 4 &&
                                   t.M <= kMaxM && t.numel == (int64_t)t.F * t.K && t.x && t.src,
                               "pfrl_rmsprop_fused_step: LOWRANK needs K % 64 == 0, F % 16 == 0, "
                               "M % 4 == 0, M <= 32");
                blocks += ((t.K + kTileK - 1) / kTileK) * (t.F / kTileF);
            } else {
                PFRL_CHECK_ARG(t.mode >= PFRL_OPT_PLAIN && t.mode <= PFRL_OPT_FOLD && t.src,
                               "pfrl_rmsprop_fused_step: bad task mode");
                PFRL_CHECK_ARG(t.mode != PFRL_OPT_LOWRANK_BIAS || t.numel == t.F,
                               "pfrl_rmsprop_fused_step: LOWRANK_BIAS covers F elements");
                blocks += (int)((t.numel + kChunk - 1) / kChunk);
            }
            PFRL_CHECK_ARG(t.mode == PFRL_OPT_FOLD ? t.out != nullptr
                                                   : (t.p && t.sq && (!centered || t.ga)),
                           "pfrl_rmsprop_fused_step: missing parameter / state pointer");
            a.block_end[i] = blocks;
        }
        a.n = n;
        if (blocks == 0) continue;
        if (centered)
            hipLaunchKernelGGL(k_rmsprop_fused<true>, dim3(blocks), dim3(kThreads), 0,
                               (hipStream_t)stream, a, lr, alpha, eps, weight_decay);
        else
            hipLaunchKernelGGL(k_rmsprop_fused<false>, dim3(blocks), dim3(kThreads), 0,
                               (hipStream_t)stream, a, lr, alpha, eps, weight_decay);
    }
    PFRL_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// torch.nn.utils.clip_grad_norm_(parameters, max_norm) (pfrl/agents/ppo.py:602-605: between
// loss.backward() and optimizer.step() of every minibatch) in three launches instead of twelve
// (_foreach_norm + lpnorm_cleanup, stack, vector_norm, add eps, reciprocal, mul, clamp,
// _foreach_mul x 2 ...): 60 us of a 1.7 ms update at the 8-GPU rank's minibatch.
//   1. per 4 096-element chunk of every gradient: sum of squares (f64) -> partial[chunk]
//   2. one wavefront: total = sum of partials in index order, norm = sqrtf((float)total),
//      coef = min(max_norm / (norm + 1e-6), 1)      (torch's clip_coef_clamped)
//   3. every gradient *= coef  (also when coef == 1, as torch does)
// ---------------------------------------------------------------------------------------
namespace {

constexpr int kClipChunk = 4096;

struct ClipArgs {
    float *g[PFRL_OPT_MAX_TENSORS];
    int64_t numel[PFRL_OPT_MAX_TENSORS];
    int32_t chunk_end[PFRL_OPT_MAX_TENSORS];
    int32_t n;
};

__global__ __launch_bounds__(kThreads) void k_grad_sumsq(ClipArgs a, double *__restrict__ partial) {
    __shared__ double s_w[kThreads / 64];
    int t = 0;
    const int b = blockIdx.x;
    while (t < a.n - 1 && b >= a.chunk_end[t]) ++t;
    const int64_t base = (int64_t)(b - (t == 0 ? 0 : a.chunk_end[t - 1])) * kClipChunk;
    const float *__restrict__ g = a.g[t];
    const int64_t n = a.numel[t];
    float v[kClipChunk / kThreads];
#pragma unroll
    for (int u = 0; u < kClipChunk / kThreads; ++u) {
        const int64_t i = base + u * kThreads + threadIdx.x;
        v[u] = i < n ? g[i] : 0.f;
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < kClipChunk / kThreads; ++u) acc += (double)v[u] * (double)v[u];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) s += s_w[w];
        partial[b] = s;
    }
}

// out[0] = total norm, out[1] = clamped clip coefficient
__global__ __launch_bounds__(64) void k_clip_coef(const double *__restrict__ partial, int nblk,
                                                  float max_norm, float *__restrict__ out) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        const float norm = sqrtf((float)s);
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = norm;
        out[1] = coef < 1.0f ? coef : 1.0f;
    }
}

__global__ __launch_bounds__(kThreads) void k_grad_scale(ClipArgs a, const float *__restrict__ coef_p) {
    int t = 0;
    const int b = blockIdx.x;
    while (t < a.n - 1 && b >= a.chunk_end[t]) ++t;
    const int64_t base = (int64_t)(b - (t == 0 ? 0 : a.chunk_end[t - 1])) * kClipChunk;
    float *__restrict__ g = a.g[t];
    const int64_t n = a.numel[t];
    const float coef = coef_p[1];
#pragma unroll
    for (int u = 0; u < kClipChunk / kThreads; ++u) {
        const int64_t i = base + u * kThreads + threadIdx.x;
        if (i < n) g[i] = __fmul_rn(g[i], coef);
    }
}

}  // namespace

extern "C" int pfrl_clip_grad_norm(int32_t n_tensors, float *const *grads, const int64_t *numel,
                                   float max_norm, double *partial_ws, float *out_norm_coef,
                                   void *stream) {
    PFRL_CHECK_ARG(n_tensors >= 1 && n_tensors <= PFRL_OPT_MAX_TENSORS && grads && numel && partial_ws &&
                       out_norm_coef && max_norm > 0.f,
                   "pfrl_clip_grad_norm: 1 <= tensors <= 24, max_norm > 0");
    ClipArgs a;
    int chunks = 0;
    for (int t = 0; t < n_tensors; ++t) {
        a.g[t] = grads[t];
        a.numel[t] = numel[t];
        chunks += (int)((numel[t] + kClipChunk - 1) / kClipChunk);
        a.chunk_end[t] = chunks;
    }
    a.n = n_tensors;
    if (chunks == 0) return 0;
    hipLaunchKernelGGL(k_grad_sumsq, dim3(chunks), dim3(kThreads), 0, (hipStream_t)stream, a, partial_ws);
    hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(64), 0, (hipStream_t)stream, partial_ws, chunks, max_norm,
                       out_norm_coef);
    hipLaunchKernelGGL(k_grad_scale, dim3(chunks), dim3(kThreads), 0, (hipStream_t)stream, a, out_norm_coef);
    PFRL_LAUNCH_CHECK();
}
