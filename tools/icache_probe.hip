// What does STRAIGHT-LINE code cost a short kernel on MI355X?  The B = 32 tile programs of
// csrc/qnet.hip are 1 500-1 800 instructions long, almost all executed once (stages of 8 chunks
// fully unrolled), and their in-kernel clocks (tools/qnet_phase.py) show ~1 us before the first
// load is issued for ~280 instructions of address arithmetic.  This probe separates instruction
// FETCH from instruction ISSUE: the same number of FMAs (4 independent chains) executed
//   (a) as one straight line of N unique instructions (every cache line of code touched once),
//   (b) as a loop over a 64-instruction body (code resident after the first trip),
// 256 workgroups x 256 threads, chains of 32 launches replayed from a hipGraph, in-kernel
// wall_clock64 (100 MHz) around the arithmetic of workgroup 0 and hipEvents around the graph.
//   hipcc --offload-arch=gfx950 -O3 tools/icache_probe.hip -o tools/icache_probe && tools/icache_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N>
__device__ __forceinline__ void line(float &a, float &b, float &c, float &d) {
    // N FMAs with distinct immediates: the compiler cannot roll them back into a loop
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        a = __builtin_fmaf(a, 1.0001f, (float)(4 * i + 1));
        b = __builtin_fmaf(b, 0.9999f, (float)(4 * i + 2));
        c = __builtin_fmaf(c, 1.0002f, (float)(4 * i + 3));
        d = __builtin_fmaf(d, 0.9998f, (float)(4 * i + 4));
    }
}

template <int N, bool LOOP>
__global__ __launch_bounds__(256) void k_code(const float *__restrict__ in, float *__restrict__ out,
                                              unsigned long long *__restrict__ clk, int trips) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a = in[i], b = a + 1.f, c = a + 2.f, d = a + 3.f;
    const unsigned long long t0 = wall_clock64();
    if (LOOP) {
        for (int t = 0; t < trips; ++t) {
            line<64>(a, b, c, d);
            asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    } else {
        line<N>(a, b, c, d);
    }
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    const unsigned long long t1 = wall_clock64();
    out[i] = a + b + c + d;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int N, bool LOOP>
static int run(const char *what, float *in, float *out, unsigned long long *clk, hipStream_t s) {
    const int CHAIN = 32, REPS = 100, WG = 256;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < CHAIN; ++i)
        hipLaunchKernelGGL((k_code<N, LOOP>), dim3(WG), dim3(256), 0, s, (i & 1) ? out : in, (i & 1) ? in : out,
                           clk, N / 64);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256];
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < WG; ++i) mean += (double)h[i];
    mean /= WG;
    printf("%-34s %6d instr  launch %6.2f us   arithmetic (in-kernel, mean of 256 wg) %6.2f us  = %5.1f clk/instr @2.4 GHz\n",
           what, N, 1e3 * ms / (REPS * CHAIN), mean / 100.0, mean / 100.0 * 2400.0 / N);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return 0;
}

int main() {
    float *in, *out;
    unsigned long long *clk;
    CK(hipMalloc(&in, 256 * 256 * 4));
    CK(hipMalloc(&out, 256 * 256 * 4));
    CK(hipMalloc(&clk, 256 * 8));
    CK(hipMemset(in, 0, 256 * 256 * 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (run<256, false>("straight line", in, out, clk, s)) return 1;
    if (run<256, true>("loop over a 64-instruction body", in, out, clk, s)) return 1;
    if (run<1024, false>("straight line", in, out, clk, s)) return 1;
    if (run<1024, true>("loop over a 64-instruction body", in, out, clk, s)) return 1;
    if (run<2048, false>("straight line", in, out, clk, s)) return 1;
    if (run<2048, true>("loop over a 64-instruction body", in, out, clk, s)) return 1;
    if (run<4096, false>("straight line", in, out, clk, s)) return 1;
    if (run<4096, true>("loop over a 64-instruction body", in, out, clk, s)) return 1;
    return 0;
}
