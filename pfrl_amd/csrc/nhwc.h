// Channels-last emission of a stack of FOUR u8 frames: out[pixel][c] (f32), i.e. the
// memory of an NCHW tensor [4][H][W] in torch.channels_last format.  A channels_last
// network then reads the minibatch as is; PyTorch otherwise converts the NCHW
// minibatch on every forward AND every backward pass (one read + write of the whole
// fp32 batch each: 10 % of PPO's GPU time, profiles/r01e_ppo_kernel_stats.csv).
//
// One 256-thread workgroup converts a tile of 1024 pixels: every lane loads one dword
// (4 pixels) from each of the four frames -- the same coalesced 256 B per wave
// instruction as the planar kernel -- and parks it in LDS; after one barrier lane L
// builds pixel (u*256 + L) for u = 0..3 from four LDS byte reads and stores it as one
// float4, so that each store instruction of a wave still covers 1 KiB of contiguous
// HBM (storing the lane's own 4 pixels instead would write 16 B pieces 64 B apart).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfrl_nhwc {

constexpr int kTilePixels = 1024;

__device__ __forceinline__ int tiles_per_obs(int64_t npix) {
    return (int)((npix + kTilePixels - 1) / kTilePixels);
}

template <bool DIV, bool NT>
__device__ __forceinline__ void convert_tile(const uint8_t *__restrict__ f0,
                                             const uint8_t *__restrict__ f1,
                                             const uint8_t *__restrict__ f2,
                                             const uint8_t *__restrict__ f3,
                                             float4 *__restrict__ dst_obs, int tile, int npix,
                                             float d) {
    __shared__ uint32_t s_raw[4][256];
    const int tid = threadIdx.x;
    const int px0 = tile * kTilePixels;
    const int ndw = min(256, (npix - px0) >> 2);          // dwords of this tile per frame
    const int i = (px0 >> 2) + min(tid, ndw - 1);          // clamped: loads stay unconditional
    const uint32_t w0 = reinterpret_cast<const uint32_t *>(f0)[i];
    const uint32_t w1 = reinterpret_cast<const uint32_t *>(f1)[i];
    const uint32_t w2 = reinterpret_cast<const uint32_t *>(f2)[i];
    const uint32_t w3 = reinterpret_cast<const uint32_t *>(f3)[i];
    s_raw[0][tid] = w0;
    s_raw[1][tid] = w1;
    s_raw[2][tid] = w2;
    s_raw[3][tid] = w3;
    __syncthreads();
    const uint8_t *sb = reinterpret_cast<const uint8_t *>(&s_raw[0][0]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = u * 256 + tid;
        if (px0 + p < npix) {
            float4 o;
            o.x = (float)sb[p];
            o.y = (float)sb[1024 + p];
            o.z = (float)sb[2048 + p];
            o.w = (float)sb[3072 + p];
            if (DIV) {
                o.x = __fdiv_rn(o.x, d);
                o.y = __fdiv_rn(o.y, d);
                o.z = __fdiv_rn(o.z, d);
                o.w = __fdiv_rn(o.w, d);
            }
            float4 *dst = dst_obs + px0 + p;
            if (NT) {
                __builtin_nontemporal_store(o.x, &dst->x);
                __builtin_nontemporal_store(o.y, &dst->y);
                __builtin_nontemporal_store(o.z, &dst->z);
                __builtin_nontemporal_store(o.w, &dst->w);
            } else {
                *dst = o;
            }
        }
    }
}

}  // namespace pfrl_nhwc
