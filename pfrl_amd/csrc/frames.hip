// Observation store kernels: frame ring writes and the gather-by-index
// u8 -> f32 minibatch assembly that replaces batch_states(phi(...)).
//
// Roofline: HBM.  Per gathered frame the kernel reads frame_bytes and writes
// 4*frame_bytes (u8 path); nothing is reused on chip except frames shared by
// consecutive stacks (served by L2/MALL).  Work decomposition: one 256-thread
// workgroup per output frame; each lane loads dwords strided by the workgroup
// (256 B per wave instruction) and stores one float4 per dword, so every store
// instruction of a wave covers 1 KiB of contiguous HBM.
#include <hip/hip_ext.h>

#include "common.h"
#include "nhwc.h"

static char g_err[512] = "";
extern "C" void pfrl_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
extern "C" const char *pfrl_amd_last_error(void) { return g_err; }
extern "C" int pfrl_amd_version(void) { return 100; }

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 8;

__device__ __forceinline__ float4 cvt_div(uint32_t w, float d) {
    float4 o;
    o.x = __fdiv_rn((float)(w & 0xffu), d);
    o.y = __fdiv_rn((float)((w >> 8) & 0xffu), d);
    o.z = __fdiv_rn((float)((w >> 16) & 0xffu), d);
    o.w = __fdiv_rn((float)(w >> 24), d);
    return o;
}

__device__ __forceinline__ float4 cvt_only(uint32_t w) {
    float4 o;
    o.x = (float)(w & 0xffu);
    o.y = (float)((w >> 8) & 0xffu);
    o.z = (float)((w >> 16) & 0xffu);
    o.w = (float)(w >> 24);
    return o;
}

// One workgroup converts one frame of `nd` dwords.  NT = non-temporal stores,
// selected for launches whose output is far larger than L2 + MALL.
template <bool DIV, bool NT>
__device__ __forceinline__ void convert_frame(const uint32_t *__restrict__ src,
                                              float4 *__restrict__ dst, int nd, float d) {
    const int tid = threadIdx.x;
    for (int base = 0; base < nd; base += kThreads * kUnroll) {
        uint32_t w[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            int i = base + u * kThreads + tid;
            w[u] = (i < nd) ? src[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            int i = base + u * kThreads + tid;
            if (i < nd) {
                const float4 o = DIV ? cvt_div(w[u], d) : cvt_only(w[u]);
                if (NT) {
                    __builtin_nontemporal_store(o.x, &dst[i].x);
                    __builtin_nontemporal_store(o.y, &dst[i].y);
                    __builtin_nontemporal_store(o.z, &dst[i].z);
                    __builtin_nontemporal_store(o.w, &dst[i].w);
                } else {
                    dst[i] = o;
                }
            }
        }
    }
}

template <bool DIV, bool NT>
__global__ __launch_bounds__(kThreads) void k_batch_states_u8(const uint8_t *__restrict__ frames,
                                                              int64_t frame_bytes,
                                                              const int32_t *__restrict__ refs,
                                                              float d, float *__restrict__ out) {
    const int64_t f = blockIdx.x;
    const int64_t slot = refs[f];
    const uint32_t *src = reinterpret_cast<const uint32_t *>(frames + slot * frame_bytes);
    float4 *dst = reinterpret_cast<float4 *>(out + f * frame_bytes);
    convert_frame<DIV, NT>(src, dst, (int)(frame_bytes >> 2), d);
}

// Channels-last variant for stacks of four frames: grid = n_obs * tiles workgroups.
template <bool DIV, bool NT>
__global__ __launch_bounds__(kThreads) void k_batch_states_u8_nhwc4(
    const uint8_t *__restrict__ frames, int64_t frame_bytes, const int32_t *__restrict__ refs,
    float d, float *__restrict__ out, int tiles) {
    const int64_t obs = blockIdx.x / tiles;
    const int tile = (int)(blockIdx.x - obs * tiles);
    const int32_t *r = refs + obs * 4;
    const uint8_t *f0 = frames + (int64_t)r[0] * frame_bytes;
    const uint8_t *f1 = frames + (int64_t)r[1] * frame_bytes;
    const uint8_t *f2 = frames + (int64_t)r[2] * frame_bytes;
    const uint8_t *f3 = frames + (int64_t)r[3] * frame_bytes;
    float4 *dst = reinterpret_cast<float4 *>(out + obs * 4 * frame_bytes);
    pfrl_nhwc::convert_tile<DIV, NT>(f0, f1, f2, f3, dst, tile, (int)frame_bytes, d);
}

// Stacks of four u8 frames as u8 NHWC4 pixels: out[obs][pixel] = one dword holding the pixel's
// byte in each of the four frames (frame 0 in the low byte) -- the observation in channels-last
// order WITHOUT the fp32 conversion, for consumers that evaluate phi in their own operand loader
// (pfrl_conv2d_u8nhwc4_fwd / _bwd_weight, csrc/qnet.hip).  A lane takes four consecutive pixels:
// one dword from each frame in, one 16-byte store out; grid = n_obs * tiles.
__global__ __launch_bounds__(kThreads) void k_batch_states_u8_raw_nhwc4(
    const uint8_t *__restrict__ frames, int64_t frame_bytes, const int32_t *__restrict__ refs,
    uint4 *__restrict__ out, int tiles, int ndw) {
    const int64_t obs = blockIdx.x / tiles;
    const int tile = (int)(blockIdx.x - obs * tiles);
    const int i = tile * kThreads + threadIdx.x;
    if (i >= ndw) return;
    const int32_t *r = refs + obs * 4;
    const uint32_t w0 = reinterpret_cast<const uint32_t *>(frames + (int64_t)r[0] * frame_bytes)[i];
    const uint32_t w1 = reinterpret_cast<const uint32_t *>(frames + (int64_t)r[1] * frame_bytes)[i];
    const uint32_t w2 = reinterpret_cast<const uint32_t *>(frames + (int64_t)r[2] * frame_bytes)[i];
    const uint32_t w3 = reinterpret_cast<const uint32_t *>(frames + (int64_t)r[3] * frame_bytes)[i];
    uint4 o;
    o.x = (w0 & 0xffu) | ((w1 & 0xffu) << 8) | ((w2 & 0xffu) << 16) | (w3 << 24);
    o.y = ((w0 >> 8) & 0xffu) | (w1 & 0xff00u) | ((w2 & 0xff00u) << 8) | ((w3 & 0xff00u) << 16);
    o.z = ((w0 >> 16) & 0xffu) | ((w1 >> 8) & 0xff00u) | (w2 & 0xff0000u) | ((w3 & 0xff0000u) << 8);
    o.w = (w0 >> 24) | ((w1 >> 16) & 0xff00u) | ((w2 >> 8) & 0xff0000u) | (w3 & 0xff000000u);
    out[obs * ndw + i] = o;
}

// f32 frames: plain gather, 16 B per lane when the frame size allows.
template <typename VecT>
__global__ __launch_bounds__(kThreads) void k_batch_states_f32(const uint8_t *__restrict__ frames,
                                                               int64_t frame_bytes,
                                                               const int32_t *__restrict__ refs,
                                                               int64_t n_refs,
                                                               uint8_t *__restrict__ out) {
    const uint32_t nv = (uint32_t)(frame_bytes / sizeof(VecT));
    const int64_t total = n_refs * (int64_t)nv;
    for (int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * kThreads) {
        const int64_t f = g / nv;
        const uint32_t i = (uint32_t)(g - f * nv);
        const VecT *src = reinterpret_cast<const VecT *>(frames + (int64_t)refs[f] * frame_bytes);
        reinterpret_cast<VecT *>(out + f * frame_bytes)[i] = src[i];
    }
}

__global__ __launch_bounds__(kThreads) void k_frames_scatter(uint8_t *__restrict__ frames,
                                                             int64_t frame_bytes,
                                                             const uint8_t *__restrict__ src,
                                                             const int32_t *__restrict__ slots) {
    const int64_t f = blockIdx.x;
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src + f * frame_bytes);
    uint32_t *d = reinterpret_cast<uint32_t *>(frames + (int64_t)slots[f] * frame_bytes);
    const int nd = (int)(frame_bytes >> 2);
    for (int i = threadIdx.x; i < nd; i += kThreads) d[i] = s[i];
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// slots == nullptr: frame f goes to ring position (seq0 + f) % n_slots (a whole env batch takes
// consecutive positions, so the host need not upload a slot list)
__global__ __launch_bounds__(kThreads) void k_frames_synth(uint8_t *__restrict__ frames,
                                                           int64_t frame_bytes,
                                                           const int32_t *__restrict__ slots,
                                                           int64_t seq0, int64_t n_slots,
                                                           uint64_t seed, int64_t env_id0,
                                                           int64_t step) {
    const int64_t f = blockIdx.x;
    const int64_t slot = slots ? (int64_t)slots[f] : (seq0 + f) % n_slots;
    uint2 *d = reinterpret_cast<uint2 *>(frames + slot * frame_bytes);
    const int nq = (int)(frame_bytes >> 3);
    const uint64_t key = mix64(seed ^ mix64((uint64_t)(env_id0 + f) * 0x9e3779b97f4a7c15ull +
                                            (uint64_t)step));
    for (int i = threadIdx.x; i < nq; i += kThreads) {
        uint64_t r = mix64(key + (uint64_t)i * 0xd1342543de82ef95ull);
        d[i] = make_uint2((uint32_t)r, (uint32_t)(r >> 32));
    }
    // tail (frame_bytes % 8 == 4)
    if ((frame_bytes & 7) && threadIdx.x == 0) {
        uint64_t r = mix64(key + (uint64_t)nq * 0xd1342543de82ef95ull);
        reinterpret_cast<uint32_t *>(d)[2 * nq] = (uint32_t)r;
    }
}

}  // namespace

extern "C" int pfrl_frames_scatter(void *frames, int64_t frame_bytes, const void *src,
                                   const int32_t *slots, int64_t n, void *stream) {
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_frames_scatter, dim3((unsigned)n), dim3(kThreads), 0, (hipStream_t)stream,
                       (uint8_t *)frames, frame_bytes, (const uint8_t *)src, slots);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_frames_synth_u8(void *frames, int64_t frame_bytes, const int32_t *slots,
                                    int64_t n, uint64_t seed, int64_t env_id0, int64_t step,
                                    void *stream) {
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_frames_synth, dim3((unsigned)n), dim3(kThreads), 0, (hipStream_t)stream,
                       (uint8_t *)frames, frame_bytes, slots, (int64_t)0, (int64_t)1, seed, env_id0,
                       step);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_frames_synth_u8_ring(void *frames, int64_t frame_bytes, int64_t n_slots,
                                         int64_t seq0, int64_t n, uint64_t seed, int64_t env_id0,
                                         int64_t step, void *stream) {
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    PFRL_CHECK_ARG(n_slots > 0 && seq0 >= 0 && n <= n_slots, "pfrl_frames_synth_u8_ring: bad ring");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_frames_synth, dim3((unsigned)n), dim3(kThreads), 0, (hipStream_t)stream,
                       (uint8_t *)frames, frame_bytes, (const int32_t *)nullptr, seq0, n_slots, seed,
                       env_id0, step);
    PFRL_LAUNCH_CHECK();
}

// epsilon-greedy on the device: action[i] = choice[i] >= 0 ? choice[i] : greedy[i]
// (pfrl/explorers/epsilon_greedy.py:8-12 with the host's draws in `choice`, -1 = greedy)
template <typename G>
__global__ __launch_bounds__(kThreads) void k_select_actions(const G *__restrict__ greedy,
                                                             const int32_t *__restrict__ choice,
                                                             int64_t *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int32_t c = choice[i];
    out[i] = c >= 0 ? (int64_t)c : (int64_t)greedy[i];
}

extern "C" int pfrl_select_actions(const void *greedy, int greedy_is_i32, const int32_t *choice,
                                   int64_t *out, int64_t n, void *stream) {
    PFRL_CHECK_ARG(greedy && choice && out, "pfrl_select_actions: null argument");
    if (n <= 0) return 0;
    const dim3 grid((unsigned)((n + kThreads - 1) / kThreads)), block(kThreads);
    if (greedy_is_i32)
        hipLaunchKernelGGL(k_select_actions<int32_t>, grid, block, 0, (hipStream_t)stream,
                           (const int32_t *)greedy, choice, out, n);
    else
        hipLaunchKernelGGL(k_select_actions<int64_t>, grid, block, 0, (hipStream_t)stream,
                           (const int64_t *)greedy, choice, out, n);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_batch_states_u8(const void *frames, int64_t frame_bytes, const int32_t *refs,
                                    int64_t n_refs, float divisor, float *out, void *stream) {
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    if (n_refs <= 0) return 0;
    const bool nt = n_refs * frame_bytes * 4 >= pfrl_nt_min_bytes();
    const dim3 grid((unsigned)n_refs), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
    const uint8_t *fr = (const uint8_t *)frames;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    pfrl_profile_events(PFRL_PROFILE_BATCH_STATES_U8, n_refs, &e0, &e1);
    if (divisor == 1.0f) {
        if (nt)
            hipExtLaunchKernelGGL((k_batch_states_u8<false, true>), grid, block, 0, st, e0, e1, 0,
                                  fr, frame_bytes, refs, divisor, out);
        else
            hipExtLaunchKernelGGL((k_batch_states_u8<false, false>), grid, block, 0, st, e0, e1, 0,
                                  fr, frame_bytes, refs, divisor, out);
    } else {
        if (nt)
            hipExtLaunchKernelGGL((k_batch_states_u8<true, true>), grid, block, 0, st, e0, e1, 0,
                                  fr, frame_bytes, refs, divisor, out);
        else
            hipExtLaunchKernelGGL((k_batch_states_u8<true, false>), grid, block, 0, st, e0, e1, 0,
                                  fr, frame_bytes, refs, divisor, out);
    }
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_batch_states_u8_nhwc4(const void *frames, int64_t frame_bytes,
                                          const int32_t *refs, int64_t n_obs, float divisor,
                                          float *out, void *stream) {
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    if (n_obs <= 0) return 0;
    const int tiles = (int)((frame_bytes + pfrl_nhwc::kTilePixels - 1) / pfrl_nhwc::kTilePixels);
    const bool nt = n_obs * 4 * frame_bytes * 4 >= pfrl_nt_min_bytes();
    const dim3 grid((unsigned)(n_obs * tiles)), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
    const uint8_t *fr = (const uint8_t *)frames;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    pfrl_profile_events(PFRL_PROFILE_BATCH_STATES_U8, n_obs * 4, &e0, &e1);
    if (divisor == 1.0f) {
        if (nt)
            hipExtLaunchKernelGGL((k_batch_states_u8_nhwc4<false, true>), grid, block, 0, st, e0,
                                  e1, 0, fr, frame_bytes, refs, divisor, out, tiles);
        else
            hipExtLaunchKernelGGL((k_batch_states_u8_nhwc4<false, false>), grid, block, 0, st, e0,
                                  e1, 0, fr, frame_bytes, refs, divisor, out, tiles);
    } else {
        if (nt)
            hipExtLaunchKernelGGL((k_batch_states_u8_nhwc4<true, true>), grid, block, 0, st, e0,
                                  e1, 0, fr, frame_bytes, refs, divisor, out, tiles);
        else
            hipExtLaunchKernelGGL((k_batch_states_u8_nhwc4<true, false>), grid, block, 0, st, e0,
                                  e1, 0, fr, frame_bytes, refs, divisor, out, tiles);
    }
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_batch_states_u8_raw_nhwc4(const void *frames, int64_t frame_bytes,
                                              const int32_t *refs, int64_t n_obs, void *out,
                                              void *stream) {
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    PFRL_CHECK_ARG(((uintptr_t)out & 15) == 0, "pfrl_batch_states_u8_raw_nhwc4: out must be 16-byte aligned");
    if (n_obs <= 0) return 0;
    const int ndw = (int)(frame_bytes >> 2);
    const int tiles = (ndw + kThreads - 1) / kThreads;
    // (a profile kind of its own: BATCH_STATES_U8 prices 5 bytes per frame byte, the fp32 form;
    // this launch moves 2)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    pfrl_profile_events(PFRL_PROFILE_BATCH_STATES_U8_RAW, n_obs * 4, &e0, &e1);
    hipExtLaunchKernelGGL(k_batch_states_u8_raw_nhwc4, dim3((unsigned)(n_obs * tiles)), dim3(kThreads),
                          0, (hipStream_t)stream, e0, e1, 0, (const uint8_t *)frames, frame_bytes, refs,
                          (uint4 *)out, tiles, ndw);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_batch_states_f32(const void *frames, int64_t frame_bytes, const int32_t *refs,
                                     int64_t n_refs, float *out, void *stream) {
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    if (n_refs <= 0) return 0;
    const bool v16 = (frame_bytes & 15) == 0;
    const int64_t items = n_refs * (frame_bytes / (v16 ? 16 : 4));
    int64_t blocks = (items + kThreads - 1) / kThreads;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (v16)
        hipLaunchKernelGGL(k_batch_states_f32<uint4>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, (const uint8_t *)frames, frame_bytes, refs, n_refs,
                           (uint8_t *)out);
    else
        hipLaunchKernelGGL(k_batch_states_f32<uint32_t>, dim3((unsigned)blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, (const uint8_t *)frames, frame_bytes, refs, n_refs,
                           (uint8_t *)out);
    PFRL_LAUNCH_CHECK();
}
