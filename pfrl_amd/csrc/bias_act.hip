// Fused bias + ReLU for the conv trunk, forward and backward, on row-major
// [rows][C] activations (channels_last conv outputs flattened over N*H*W).
//
// PyTorch-ROCm runs conv -> bias add -> ReLU as three kernels and, backward,
// ReLU-grad and the bias-gradient reduction as two more; at minibatch 32 each of
// them is 4-12 us of mostly launch latency.  Here:
//   forward   y = max(x + b[c], 0)                                   one launch
//   backward  gx = gy * (y > 0);  gb[c] = sum_rows gx[., c]          one launch
// The backward kernel reduces columns inside each workgroup (registers -> LDS),
// publishes one partial row per workgroup and lets the LAST arriving workgroup
// fold the partials (cdna_hip_programming.md guideline 16, data-tagged granule
// form): no second launch, no atomics on floats (the result is deterministic).
#include "common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void k_bias_relu_fwd(const float *__restrict__ x,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ y, int64_t n4,
                                                            int C) {
    // C % 4 == 0: one float4 never straddles a row
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    const int c4 = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const float4 b = reinterpret_cast<const float4 *>(bias)[i % c4];
        float4 o;
        o.x = fmaxf(__fadd_rn(v.x, b.x), 0.0f);
        o.y = fmaxf(__fadd_rn(v.y, b.y), 0.0f);
        o.z = fmaxf(__fadd_rn(v.z, b.z), 0.0f);
        o.w = fmaxf(__fadd_rn(v.w, b.w), 0.0f);
        reinterpret_cast<float4 *>(y)[i] = o;
    }
}

// Same, but y is written PLANAR ([N][C][HW], i.e. plain NCHW) from a channels-last x:
// used for the last convolution of a trunk, whose output is flattened for a linear
// layer -- the flatten is then a view instead of a layout-copy launch (and the same in
// backward, see k_bias_relu_bwd<true>).  Consecutive threads take consecutive pixels of
// one channel quad: coalesced stores, strided 16 B loads of a tensor the convolution
// has just left in L2.
__global__ __launch_bounds__(kThreads) void k_bias_relu_fwd_planar(
    const float *__restrict__ x, const float *__restrict__ bias, float *__restrict__ y,
    int64_t n4, int C, int HW) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    const int cq_n = C >> 2;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        const int p = (int)(i % HW);
        const int64_t t = i / HW;
        const int cq = (int)(t % cq_n);
        const int64_t n = t / cq_n;
        const float4 v = reinterpret_cast<const float4 *>(x)[(n * HW + p) * cq_n + cq];
        const float4 b = reinterpret_cast<const float4 *>(bias)[cq];
        float *o = y + (n * C + cq * 4) * HW + p;
        o[0] = fmaxf(__fadd_rn(v.x, b.x), 0.0f);
        o[HW] = fmaxf(__fadd_rn(v.y, b.y), 0.0f);
        o[2 * (int64_t)HW] = fmaxf(__fadd_rn(v.z, b.z), 0.0f);
        o[3 * (int64_t)HW] = fmaxf(__fadd_rn(v.w, b.w), 0.0f);
    }
}

// grid = nblk workgroups; workgroup w owns rows [w*rows_per_blk, ...).
// thread t: column quad t % (C/4), row lane t / (C/4)  (C % 4 == 0, C <= 256, 256 % C == 0).
//
// Cross-workgroup fold without fences: every partial is published as ONE
// naturally aligned 8-byte granule {value, epoch} with a write-through (sc1)
// agent-scope store; the last arriver (relaxed ticket) reads the granules with
// sc1 loads and accepts a granule only if its tag equals this launch's epoch,
// so a stale line in another XCD's L2 can never be mistaken for fresh data.
// The epoch comes from a monotonically increasing device counter, which keeps
// the protocol valid under HIP-graph replay (kernel arguments are frozen there).
// A release fence here would write back the whole L2 -- including the gx tile
// this workgroup just produced -- and cost 15-35 us per launch (measured).
// PLANAR: gy and y are [N][C][HW] (NCHW); gx is always channels-last rows.
template <bool PLANAR>
__global__ __launch_bounds__(kThreads) void k_bias_relu_bwd(
    const float *__restrict__ gy, const float *__restrict__ y, float *__restrict__ gx,
    float *__restrict__ gb, unsigned long long *__restrict__ granules,
    unsigned long long *__restrict__ ctr, int64_t rows, int C, int64_t rows_per_blk, int HW) {
    __shared__ float4 s_acc4[kThreads];
    __shared__ float s_acc[kThreads];
    __shared__ unsigned int s_epoch;
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        const unsigned long long e =
            __hip_atomic_fetch_add(&ctr[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_epoch = (unsigned int)(e / gridDim.x) + 1u;
    }
    // Main pass, four channels per thread (float4; C % 4 == 0): thread t owns column
    // quad cq = t % (C/4) and walks rows rl, rl + rstep, ...  A wave covers 1 KiB of
    // contiguous memory per load instruction.
    const int cq_n = C >> 2;
    const int cq = threadIdx.x % cq_n;
    const int rl = threadIdx.x / cq_n;
    const int rstep = kThreads / cq_n;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > rows) r1 = rows;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 *y4 = reinterpret_cast<const float4 *>(y);
    const float4 *gy4 = reinterpret_cast<const float4 *>(gy);
    float4 *gx4 = reinterpret_cast<float4 *>(gx);
    for (int64_t r = r0 + rl; r < r1; r += rstep) {
        const int64_t i = r * cq_n + cq;
        float4 yv, gv;
        if (PLANAR) {
            const int64_t n = r / HW;
            const int64_t base = (n * C + cq * 4) * HW + (r - n * HW);
            yv = make_float4(y[base], y[base + HW], y[base + 2 * (int64_t)HW],
                             y[base + 3 * (int64_t)HW]);
            gv = make_float4(gy[base], gy[base + HW], gy[base + 2 * (int64_t)HW],
                             gy[base + 3 * (int64_t)HW]);
        } else {
            yv = y4[i];
            gv = gy4[i];
        }
        float4 g;
        g.x = yv.x > 0.0f ? gv.x : 0.0f;
        g.y = yv.y > 0.0f ? gv.y : 0.0f;
        g.z = yv.z > 0.0f ? gv.z : 0.0f;
        g.w = yv.w > 0.0f ? gv.w : 0.0f;
        gx4[i] = g;
        acc.x += g.x;
        acc.y += g.y;
        acc.z += g.z;
        acc.w += g.w;
    }
    // per-workgroup column sums: s_acc4[row lane][column]
    float *s_flat = reinterpret_cast<float *>(s_acc4);
    s_acc4[threadIdx.x] = acc;          // thread t = (rl, cq): element t*4 + j = rl*C + cq*4 + j
    __syncthreads();
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x;
        float tot = 0.0f;
        for (int k = 0; k < rstep; ++k) tot += s_flat[k * C + c];
        const unsigned long long g =
            ((unsigned long long)s_epoch << 32) | (unsigned long long)__float_as_uint(tot);
        __hip_atomic_store(&granules[(int64_t)blockIdx.x * C + c], g, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t =
            __hip_atomic_fetch_add(&ctr[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t % gridDim.x) == (unsigned long long)(gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last) return;
    // last workgroup: fold the partial rows.  All 256 threads load granules in
    // parallel (8 independent write-through loads in flight per thread; a serial
    // walk would pay one uncached round trip per workgroup), then LDS-reduce.
    {
        const unsigned int epoch = s_epoch;
        const int64_t total = (int64_t)gridDim.x * C;   // granule (w, c) at w*C + c
        float part = 0.0f;
        for (int64_t base = threadIdx.x; base < total; base += (int64_t)kThreads * 8) {
            unsigned long long g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t idx = base + (int64_t)u * kThreads;
                g[u] = idx < total ? __hip_atomic_load(&granules[idx], __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT)
                                   : ((unsigned long long)epoch << 32);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t idx = base + (int64_t)u * kThreads;
                int spins = 0;
                while ((unsigned int)(g[u] >> 32) != epoch && ++spins < (1 << 22))
                    g[u] = __hip_atomic_load(&granules[idx], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
                part += __uint_as_float((unsigned int)g[u]);
            }
        }
        // kThreads % C == 0, so thread t only ever saw column t % C
        __syncthreads();
        s_acc[threadIdx.x] = part;
        __syncthreads();
        if ((int)threadIdx.x < C) {
            float tot = 0.0f;
            for (int k = 0; k < kThreads / C; ++k) tot += s_acc[k * C + threadIdx.x];
            gb[threadIdx.x] = tot;
        }
    }
}

// Few rows, any width (the hidden linear layer of a Q-network at minibatch 32: rows = 32,
// C = 512): ONE workgroup, thread t owns columns t, t + 256, ... and walks the rows;
// loads are coalesced across the threads and nothing crosses workgroups.
__global__ __launch_bounds__(kThreads) void k_bias_relu_bwd_small(
    const float *__restrict__ gy, const float *__restrict__ y, float *__restrict__ gx,
    float *__restrict__ gb, int rows, int C) {
    for (int c = threadIdx.x; c < C; c += kThreads) {
        float acc = 0.0f;
        for (int r = 0; r < rows; ++r) {
            const int64_t i = (int64_t)r * C + c;
            const float g = y[i] > 0.0f ? gy[i] : 0.0f;
            gx[i] = g;
            acc += g;
        }
        gb[c] = acc;
    }
}

}  // namespace

extern "C" int pfrl_bias_relu_fwd(const float *x, const float *bias, float *y, int64_t rows,
                                  int32_t C, int64_t planar_hw, void *stream) {
    PFRL_CHECK_ARG(C > 0 && (C & 3) == 0, "pfrl_bias_relu_fwd: C must be a multiple of 4");
    PFRL_CHECK_ARG(planar_hw >= 0 && (planar_hw == 0 || rows % planar_hw == 0),
                   "pfrl_bias_relu_fwd: rows must be a multiple of planar_hw");
    if (rows <= 0) return 0;
    const int64_t n4 = rows * C / 4;
    int64_t blocks = (n4 + kThreads - 1) / kThreads;
    if (blocks > 2048) blocks = 2048;
    if (planar_hw > 0)
        hipLaunchKernelGGL(k_bias_relu_fwd_planar, dim3((unsigned)blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, x, bias, y, n4, (int)C, (int)planar_hw);
    else
        hipLaunchKernelGGL(k_bias_relu_fwd, dim3((unsigned)blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, x, bias, y, n4, (int)C);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_bias_relu_bwd(const float *gy, const float *y, float *gx, float *gb,
                                  uint64_t *granule_ws, uint64_t *counters, int64_t rows,
                                  int32_t C, int32_t blocks, int64_t planar_hw, void *stream) {
    if (blocks == 0) {
        // single-workgroup form: no workspace, no cross-workgroup fold
        PFRL_CHECK_ARG(C > 0 && rows <= 1024 && planar_hw == 0,
                       "pfrl_bias_relu_bwd: the single-workgroup form takes <= 1024 plain rows");
        if (rows <= 0) return 0;
        hipLaunchKernelGGL(k_bias_relu_bwd_small, dim3(1), dim3(kThreads), 0, (hipStream_t)stream,
                           gy, y, gx, gb, (int)rows, (int)C);
        PFRL_LAUNCH_CHECK();
    }
    PFRL_CHECK_ARG(C > 0 && C <= kThreads && kThreads % C == 0 && (C & 3) == 0,
                   "pfrl_bias_relu_bwd: C must be a multiple of 4 that divides 256");
    PFRL_CHECK_ARG(blocks > 0, "pfrl_bias_relu_bwd: blocks must be positive");
    PFRL_CHECK_ARG(planar_hw >= 0 && (planar_hw == 0 || rows % planar_hw == 0),
                   "pfrl_bias_relu_bwd: rows must be a multiple of planar_hw");
    if (rows <= 0) return 0;
    const int64_t rows_per_blk = (rows + blocks - 1) / blocks;
    unsigned long long *gr = reinterpret_cast<unsigned long long *>(granule_ws);
    unsigned long long *ct = reinterpret_cast<unsigned long long *>(counters);
    if (planar_hw > 0)
        hipLaunchKernelGGL(k_bias_relu_bwd<true>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, gy, y, gx, gb, gr, ct, rows, (int)C, rows_per_blk,
                           (int)planar_hw);
    else
        hipLaunchKernelGGL(k_bias_relu_bwd<false>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, gy, y, gx, gb, gr, ct, rows, (int)C, rows_per_blk,
                           1);
    PFRL_LAUNCH_CHECK();
}
