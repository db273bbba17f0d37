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
// fold the partials -- agent-scope release on the producers, acquire on the
// reducer (cdna_hip_programming.md guideline 16): no second launch, no atomics
// on floats (the result is deterministic).
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

// grid = nblk workgroups; workgroup w owns rows [w*rows_per_blk, ...).
// thread t: column c = t % C, row lane rl = t / C (kThreads % C == 0).
__global__ __launch_bounds__(kThreads) void k_bias_relu_bwd(
    const float *__restrict__ gy, const float *__restrict__ y, float *__restrict__ gx,
    float *__restrict__ gb, float *__restrict__ partial, unsigned int *__restrict__ counter,
    int64_t rows, int C, int64_t rows_per_blk) {
    __shared__ float s_acc[kThreads];
    __shared__ int s_last;
    const int c = threadIdx.x % C;
    const int rl = threadIdx.x / C;
    const int rstep = kThreads / C;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    int64_t r1 = r0 + rows_per_blk;
    if (r1 > rows) r1 = rows;
    float acc = 0.0f;
    for (int64_t r = r0 + rl; r < r1; r += rstep) {
        const int64_t i = r * C + c;
        const float g = y[i] > 0.0f ? gy[i] : 0.0f;
        gx[i] = g;
        acc += g;
    }
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        float tot = 0.0f;
        for (int k = 0; k < rstep; ++k) tot += s_acc[k * C + c];
        partial[(int64_t)blockIdx.x * C + c] = tot;
    }
    // publish the partial row, take a ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int ticket =
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == gridDim.x - 1;
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;
    // last workgroup: fold the partial rows in a fixed order
    if (threadIdx.x < C) {
        float tot = 0.0f;
        for (unsigned int w = 0; w < gridDim.x; ++w) tot += partial[(int64_t)w * C + threadIdx.x];
        gb[threadIdx.x] = tot;
    }
    if (threadIdx.x == 0)
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

extern "C" int pfrl_bias_relu_fwd(const float *x, const float *bias, float *y, int64_t rows,
                                  int32_t C, void *stream) {
    PFRL_CHECK_ARG(C > 0 && (C & 3) == 0, "pfrl_bias_relu_fwd: C must be a multiple of 4");
    if (rows <= 0) return 0;
    const int64_t n4 = rows * C / 4;
    int64_t blocks = (n4 + kThreads - 1) / kThreads;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_bias_relu_fwd, dim3((unsigned)blocks), dim3(kThreads), 0,
                       (hipStream_t)stream, x, bias, y, n4, (int)C);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_bias_relu_bwd(const float *gy, const float *y, float *gx, float *gb,
                                  float *partial_ws, uint32_t *counter, int64_t rows, int32_t C,
                                  int32_t max_blocks, void *stream) {
    PFRL_CHECK_ARG(C > 0 && C <= kThreads && kThreads % C == 0,
                   "pfrl_bias_relu_bwd: C must divide 256");
    if (rows <= 0) return 0;
    // ~64 rows per row-lane keeps every workgroup busy and the partial table small
    const int rstep = kThreads / C;
    int64_t rows_per_blk = (int64_t)rstep * 16;
    int64_t blocks = (rows + rows_per_blk - 1) / rows_per_blk;
    if (blocks > max_blocks) {
        blocks = max_blocks;
        rows_per_blk = (rows + blocks - 1) / blocks;
        blocks = (rows + rows_per_blk - 1) / rows_per_blk;
    }
    hipLaunchKernelGGL(k_bias_relu_bwd, dim3((unsigned)blocks), dim3(kThreads), 0,
                       (hipStream_t)stream, gy, y, gx, gb, partial_ws, counter, rows, (int)C,
                       rows_per_blk);
    PFRL_LAUNCH_CHECK();
}
