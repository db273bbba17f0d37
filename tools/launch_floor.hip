// What does a dependent kernel boundary cost on this box?  Chains of small kernels,
// (a) eager on one stream, (b) replayed from a hipGraph captured from one stream,
// (c) a graph with two parallel branches per stage (fork / join), (d) kernels that each
// touch a few MB (L2 write-back at the boundary).
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o tools/launch_floor && tools/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_tiny(float *p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

// distinct kernels with a few KB of code each: does a chain of DIFFERENT kernels cost more
// per boundary than a chain of one kernel (instruction-cache misses at every launch)?
template <int ID>
__global__ void k_var(float *p, int n, int sel) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    // `sel` is 0 at run time; the branches keep ~ID-specific code in the kernel
#pragma unroll
    for (int u = 0; u < 96; ++u)
        if (sel == u + 1) v = v * (1.0f + 0.001f * (ID * 97 + u)) + __sinf(v + ID + u) * (u + ID);
    p[i] = v * 1.0001f + 1.0f;
}
// one kernel holding all the variants, selected at run time (same code object every launch)
__global__ void k_uber(float *p, int n, int sel, int id) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (id == k) {
#pragma unroll
            for (int u = 0; u < 96; ++u)
                if (sel == u + 1) v = v * (1.0f + 0.001f * (k * 97 + u)) + __sinf(v + k + u) * (u + k);
        }
    p[i] = v * 1.0001f + 1.0f;
}

template <int ID>
static void launch_var(int blocks, hipStream_t s, float *buf, int n) {
    hipLaunchKernelGGL(k_var<ID>, dim3(blocks), dim3(256), 0, s, buf, n, 0);
}
static void launch_id(int id, int blocks, hipStream_t s, float *buf, int n) {
    switch (id & 7) {
        case 0: launch_var<0>(blocks, s, buf, n); break;
        case 1: launch_var<1>(blocks, s, buf, n); break;
        case 2: launch_var<2>(blocks, s, buf, n); break;
        case 3: launch_var<3>(blocks, s, buf, n); break;
        case 4: launch_var<4>(blocks, s, buf, n); break;
        case 5: launch_var<5>(blocks, s, buf, n); break;
        case 6: launch_var<6>(blocks, s, buf, n); break;
        default: launch_var<7>(blocks, s, buf, n); break;
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const int CHAIN = 64, REPS = 200;
    float *buf;
    const int big = 1 << 22;   // 16 MB
    CK(hipMalloc(&buf, big * sizeof(float)));
    CK(hipMemset(buf, 0, big * sizeof(float)));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int pass = 0; pass < 3; ++pass) {
        const int n = pass == 0 ? 256 : (pass == 1 ? 65536 : big);   // 1 block, 256 blocks, 16 MB
        const int blocks = (n + 255) / 256;
        // (a) eager
        for (int i = 0; i < CHAIN; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, buf, n);
        CK(hipStreamSynchronize(s));
        double t0 = now_us();
        for (int r = 0; r < REPS; ++r)
            for (int i = 0; i < CHAIN; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, buf, n);
        CK(hipStreamSynchronize(s));
        double eager = (now_us() - t0) / (REPS * CHAIN);
        // (b) graph, one stream
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < CHAIN; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, buf, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = now_us();
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double graph = (now_us() - t0) / (REPS * CHAIN);
        // (c) graph, two parallel branches per stage
        hipGraph_t g2;
        hipGraphExec_t ge2;
        hipEvent_t ef, ej;
        CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < CHAIN / 2; ++i) {
            CK(hipEventRecord(ef, s));
            CK(hipStreamWaitEvent(s2, ef, 0));
            hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, buf, n / 2);
            hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s2, buf + n / 2, n / 2);
            CK(hipEventRecord(ej, s2));
            CK(hipStreamWaitEvent(s, ej, 0));
        }
        CK(hipStreamEndCapture(s, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge2, s));
        CK(hipStreamSynchronize(s));
        t0 = now_us();
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge2, s));
        CK(hipStreamSynchronize(s));
        double forked = (now_us() - t0) / (REPS * CHAIN / 2);
        // (d) chain of 8 distinct kernels vs the same kernel, vs one uber kernel, graph replay
        double var[3];
        for (int mode = 0; mode < 3; ++mode) {
            hipGraph_t g3;
            hipGraphExec_t ge3;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < CHAIN; ++i) {
                if (mode == 0) launch_id(0, blocks, s, buf, n);
                else if (mode == 1) launch_id(i, blocks, s, buf, n);
                else hipLaunchKernelGGL(k_uber, dim3(blocks), dim3(256), 0, s, buf, n, 0, i & 7);
            }
            CK(hipStreamEndCapture(s, &g3));
            CK(hipGraphInstantiate(&ge3, g3, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge3, s));
            CK(hipStreamSynchronize(s));
            t0 = now_us();
            for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge3, s));
            CK(hipStreamSynchronize(s));
            var[mode] = (now_us() - t0) / (REPS * CHAIN);
        }
        printf("   code-size variants: same kernel %.2f, 8 distinct kernels %.2f, one uber kernel %.2f us/kernel\n",
               var[0], var[1], var[2]);
        printf("n=%8d (%5d blocks): eager %.2f us/kernel, graph %.2f us/kernel, forked graph %.2f us/stage (2 kernels)\n",
               n, blocks, eager, graph, forked);
    }
    return 0;
}
