// The floor of one DEPENDENT PHASE on MI355X: a kernel that reads what the previous kernel wrote
// (from other CUs / XCDs: rotated index), does nothing with it, and writes its own output --
// chains of 64 replayed from a hipGraph.  This is what every launch of the B = 32 DQN update pays
// before its own work: boundary + first waves + one cold memory round trip + store.
//   hipcc --offload-arch=gfx950 -O3 tools/phase_floor.hip -o tools/phase_floor && tools/phase_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// out[i] = in[(i + rot) % n] + 1: every workgroup reads lines another workgroup (another XCD: block
// b runs on XCD b % 8, the rotation moves by 1/3 of the array) wrote in the previous launch
__global__ __launch_bounds__(256) void k_phase(const float4 *__restrict__ in, float4 *__restrict__ out,
                                               int n4, int rot4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    int j = i + rot4;
    if (j >= n4) j -= n4;
    float4 v = in[j];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    out[i] = v;
}

// the same with TWO dependent round trips (an index chain: what a gather-style loader does)
__global__ __launch_bounds__(256) void k_phase2(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                const int *__restrict__ idx, int n4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int j = idx[i];
    float4 v = in[j];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    out[i] = v;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const int CHAIN = 64, REPS = 200;
    const int maxn4 = 1 << 20;       // 16 MB
    float4 *a, *b;
    int *idx;
    CK(hipMalloc(&a, maxn4 * sizeof(float4)));
    CK(hipMalloc(&b, maxn4 * sizeof(float4)));
    CK(hipMalloc(&idx, maxn4 * sizeof(int)));
    CK(hipMemset(a, 0, maxn4 * sizeof(float4)));
    CK(hipMemset(b, 0, maxn4 * sizeof(float4)));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int sizes_kb[] = {4, 64, 1024, 4096, 16384};
    printf("%10s %8s %12s %12s\n", "bytes", "blocks", "1 trip us", "2 trips us");
    for (int kb : sizes_kb) {
        const int n4 = kb * 1024 / 16;
        const int blocks = (n4 + 255) / 256;
        const int rot4 = (n4 / 3) & ~63;
        int *h = (int *)malloc(n4 * sizeof(int));
        for (int i = 0; i < n4; ++i) h[i] = (i + rot4) % n4;
        CK(hipMemcpy(idx, h, n4 * sizeof(int), hipMemcpyHostToDevice));
        free(h);
        double res[2];
        for (int variant = 0; variant < 2; ++variant) {
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < CHAIN; ++i) {
                const float4 *src = (i & 1) ? b : a;
                float4 *dst = (i & 1) ? a : b;
                if (variant == 0)
                    hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(256), 0, s, src, dst, n4, rot4);
                else
                    hipLaunchKernelGGL(k_phase2, dim3(blocks), dim3(256), 0, s, src, dst, idx, n4);
            }
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const double t0 = now_us();
            for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            res[variant] = (now_us() - t0) / (REPS * CHAIN);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
        printf("%10d %8d %12.2f %12.2f\n", kb * 1024, blocks, res[0], res[1]);
    }
    return 0;
}
