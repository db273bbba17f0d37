// Which XCDs does a CU-masked stream (hipExtStreamCreateWithCUMask) run on?  For a few mask
// shapes: launch 4096 workgroups that each record HW_REG_XCC_ID and their CU, print the
// histogram of XCC ids and the number of distinct (xcc, se, cu) triples seen.
//   hipcc --offload-arch=gfx950 -O3 tools/cumask_probe.hip -o tools/cumask_probe && tools/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_where(unsigned *out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the workgroup alive a little so that the whole mask gets used
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc & 0xf;
        out[2 * blockIdx.x + 1] = hw;
    }
    if (v == 12345.f) out[0] = 0;
}

static int run(const char *name, const std::vector<uint32_t> &mask) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    const int nb = 4096;
    unsigned *d;
    CK(hipMalloc(&d, 2 * nb * sizeof(unsigned)));
    hipLaunchKernelGGL(k_where, dim3(nb), dim3(256), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * nb);
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    int hist[16] = {0};
    std::set<unsigned long long> cus;
    for (int i = 0; i < nb; ++i) {
        hist[h[2 * i] & 15]++;
        // HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13 (gfx9)
        const unsigned hw = h[2 * i + 1];
        cus.insert(((unsigned long long)h[2 * i] << 32) | (hw & 0xff00));
    }
    int bits = 0;
    for (auto w : mask) bits += __builtin_popcount(w);
    printf("%-28s mask bits %3d  distinct CUs seen %3zu  xcc histogram:", name, bits, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %4d", hist[x]);
    printf("\n");
    CK(hipFree(d));
    CK(hipStreamDestroy(s));
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int W = 8;        // 256 bits
    std::vector<uint32_t> all(W, 0xffffffffu);
    run("all", all);
    {   // the first 32 bits
        std::vector<uint32_t> m(W, 0);
        m[0] = 0xffffffffu;
        run("bits 0..31", m);
    }
    {   // every 8th bit
        std::vector<uint32_t> m(W, 0);
        for (int b = 0; b < 256; b += 8) m[b / 32] |= 1u << (b % 32);
        run("bits b % 8 == 0", m);
    }
    {   // every 8th bit, complement
        std::vector<uint32_t> m(W, 0);
        for (int b = 0; b < 256; ++b) if (b % 8 != 0) m[b / 32] |= 1u << (b % 32);
        run("bits b % 8 != 0", m);
    }
    {   // bits 32..255
        std::vector<uint32_t> m(W, 0xffffffffu);
        m[0] = 0;
        run("bits 32..255", m);
    }
    {   // 4 CUs of every XCD if interleaved: b % 8 == x for all x, b < 32
        std::vector<uint32_t> m(W, 0);
        m[0] = 0xffffffffu;
        m[1] = 0xffffffffu;
        run("bits 0..63", m);
    }
    return 0;
}
