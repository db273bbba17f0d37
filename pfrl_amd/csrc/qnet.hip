// Q-network trunk at minibatch sizes: f32 MFMA implicit-GEMM kernels for gfx950.
//
// Replaces, for the DQN / PPO example networks (pfrl/nn/atari_cnn.py:17-47 and the
// nn.Sequential of examples/atari/train_ppo_ale.py:247-264), the MIOpen / hipBLASLt
// launches behind `activation(layer(h))` and their autograd backward at the batch
// sizes of the update loop (pfrl/agents/dqn.py:316-365: B = 32 per update, 64
// dependent updates per batched env step).  At that size every library kernel is a
// 5-17 us launch for <1 us of work, plus zero-fill and bias / ReLU helper launches.
//
// One tile engine, three problems (all C[M x N] = sum_k A[m][k] * B[k][n] on
// v_mfma_f32_16x16x4_f32: exact f32 fmaf chains, no reduced precision):
//   forward   y[m][co]   = relu(sum_k xcol[m][k] * w[co][k] + b[co])   m = (n, oh, ow), k = (r, s, ci)
//   dgrad     dx[m][ci]  = (sum_{tap, co} dy[m - tap][co] * w[co][tap][ci]) * (a_prev > 0)
//   wgrad     dw[co][k]  = sum_m dy[m][co] * xcol[m][k]                 split over m, partials
// A linear layer is the 1x1 case (H = W = 1).  Activations are NHWC, weights
// [Cout][R][S][Cin] (torch.channels_last memory), so every im2col row is R runs of
// S*Cin contiguous floats: the loaders move 32-float pieces of those runs (one
// float4 per lane, 128 B contiguous per 8 lanes) and never materialise im2col.
//
// Tile engine: 256 threads = 4 waves arranged WM x WN x WK over a BM x BN tile; the
// reduction is staged through LDS in chunks of 32 (double buffered, register
// prefetch of the next chunk, one barrier per chunk).  Two LDS layouts:
//   (R) [row][k], stride 36: operand rows contiguous in k; a lane reads 4 consecutive
//       k with one ds_read_b128 and feeds 4 MFMA steps with them
//   (C) [k][col], stride BT + 4: operand contiguous across the tile dimension
//       (weights in dgrad, both operands in wgrad); 4 ds_read_b32, conflict free
// Both use the same k -> (lane group, step) map k = 16*sc + 4*(lane >> 4) + t.
// MFMA issue is never the bound here (<= 2 us per layer even from one wave per
// SIMD); grids are sized so that >= 256 workgroups exist at B = 32, with split-K
// (partials + one multi-tensor reduce) where the output is small.
#include <type_traits>

#include "common.h"
#include "rms_update.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;
#ifndef PFRL_LDR
#define PFRL_LDR (KC + 4)
#endif
constexpr int LDR = PFRL_LDR;

// Exact unsigned division by a launch constant (Granlund-Montgomery round-up form, any 32-bit
// numerator): the per-thread row -> (image, oh, ow) split of the weight-gradient loader runs once per
// chunk, and a compiler-expanded 32-bit division is ~35 vector instructions each time.
struct FastDiv {
    uint32_t m = 1, s1 = 0, s2 = 0;   // (division by 1)
};
static FastDiv fast_div(uint32_t d) {
    FastDiv f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l < 1 ? 0 : l - 1;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv &f) {
    const uint32_t t = __umulhi(f.m, (uint32_t)n);
    return (int)((t + (((uint32_t)n - t) >> f.s1)) >> f.s2);
}

struct ConvGeom {
    int N, H, W, C, Cout, R, S, ST, OH, OW;
    FastDiv q_ohow, q_ow, q_sc;    // by OH * OW, by OW, by S * C (make_geom)
};
static ConvGeom make_geom(int N, int H, int W, int C, int Cout, int R, int S, int ST) {
    ConvGeom g{N, H, W, C, Cout, R, S, ST, (H - R) / ST + 1, (W - S) / ST + 1};
    g.q_ohow = fast_div((uint32_t)(g.OH > 0 && g.OW > 0 ? g.OH * g.OW : 1));
    g.q_ow = fast_div((uint32_t)(g.OW > 0 ? g.OW : 1));
    g.q_sc = fast_div((uint32_t)(S * C > 0 ? S * C : 1));
    return g;
}

// Loads are issued unconditionally (invalid rows read a valid dummy address and are
// zeroed when the slot is parked in LDS): a predicated load is an exec-masked branch,
// and behind one the compiler drains every outstanding load (s_waitcnt vmcnt(0)),
// which would serialise the chunk pipeline below.
__device__ __forceinline__ float4 ldg4(const float *p) {
    return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ float4 zero_unless(float4 v, bool ok) {
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}
__device__ __forceinline__ float4 relu_mask(float4 v, float4 h) {
    v.x = h.x > 0.f ? v.x : 0.f; v.y = h.y > 0.f ? v.y : 0.f;
    v.z = h.z > 0.f ? v.z : 0.f; v.w = h.w > 0.f ? v.w : 0.f;
    return v;
}

// U8 operand loaders (the first convolution reading a minibatch of u8 NHWC4 pixels, one dword =
// the four stacked frames of a pixel, instead of the fp32 copy the gather would write): the
// feature extractor phi(x) = float32(x) / d (pfrl/utils/batch_states.py:18-36 with the example
// scripts' phi) evaluated where the operand is parked in LDS.  x / d as q = x r, then ONE residual
// step q + (x - q d) r with r = fl(1 / d): correctly rounded for EVERY x in 0..255 when the host has
// checked exactly that for this d (ops.u8_division_exact: all 256 values against IEEE division;
// d = 255 and d = 1 pass) -- the same fp32 numbers k_batch_states_u8* writes with __fdiv_rn.
__device__ __forceinline__ float u8_over(float x, float r, float d) {
    const float q = __fmul_rn(x, r);
    return __fmaf_rn(__fmaf_rn(-q, d, x), r, q);
}
__device__ __forceinline__ float4 u8x4_over(uint32_t w, float r, float d) {
    return make_float4(u8_over((float)(w & 0xffu), r, d), u8_over((float)((w >> 8) & 0xffu), r, d),
                       u8_over((float)((w >> 16) & 0xffu), r, d), u8_over((float)(w >> 24), r, d));
}

// One 16-wide sub-chunk (4 MFMA steps) of the current LDS chunk.
template <int AM, int AN, int P, bool A_R, bool B_R, int LDA, int LDB>
__device__ __forceinline__ void mma_sub(const float *As, const float *Bs, int wm0, int wn0, int sc,
                                        int lane, f32x4 (&acc)[AM][AN][P]) {
    const int i = lane & 15, kq = lane >> 4;
    float a[AM][4], b[AN][4];
#pragma unroll
    for (int am = 0; am < AM; ++am) {
        if (A_R) {
            const float4 v = *reinterpret_cast<const float4 *>(
                &As[(wm0 + 16 * am + i) * LDA + 16 * sc + 4 * kq]);
            a[am][0] = v.x; a[am][1] = v.y; a[am][2] = v.z; a[am][3] = v.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) a[am][t] = As[(16 * sc + 4 * kq + t) * LDA + wm0 + 16 * am + i];
        }
    }
#pragma unroll
    for (int an = 0; an < AN; ++an) {
        if (B_R) {
            const float4 v = *reinterpret_cast<const float4 *>(
                &Bs[(wn0 + 16 * an + i) * LDB + 16 * sc + 4 * kq]);
            b[an][0] = v.x; b[an][1] = v.y; b[an][2] = v.z; b[an][3] = v.w;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) b[an][t] = Bs[(16 * sc + 4 * kq + t) * LDB + wn0 + 16 * an + i];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int am = 0; am < AM; ++am)
#pragma unroll
            for (int an = 0; an < AN; ++an)
                acc[am][an][t % P] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[am][t], b[an][t],
                                                                           acc[am][an][t % P], 0, 0, 0);
}

// Fold the P interleaved accumulators and the WK wave slices of a tile; afterwards the
// waves with wk == 0 hold the result in acc[..][..][0].
template <int AM, int AN, int P, int WK, int NRED>
__device__ __forceinline__ void fold_acc(f32x4 (&acc)[AM][AN][P], f32x4 *red, int tile_wave, int wk,
                                         int lane) {
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
#pragma unroll
            for (int pp = 1; pp < P; ++pp) acc[am][an][0] += acc[am][an][pp];
    if (WK > 1) {
        // wk = 1 .. WK-1 park their tiles, wk = 0 adds them in wave order
        if (wk > 0) {
#pragma unroll
            for (int am = 0; am < AM; ++am)
#pragma unroll
                for (int an = 0; an < AN; ++an)
                    red[(((wk - 1) * NRED + tile_wave) * AM * AN + am * AN + an) * 64 + lane] =
                        acc[am][an][0];
        }
        __syncthreads();
        if (wk == 0) {
            for (int s = 0; s < WK - 1; ++s)
#pragma unroll
                for (int am = 0; am < AM; ++am)
#pragma unroll
                    for (int an = 0; an < AN; ++an)
                        acc[am][an][0] += red[((s * NRED + tile_wave) * AM * AN + am * AN + an) * 64 + lane];
        }
    }
}

// Chunk pipeline.  At B = 32 a workgroup walks only 7-18 chunks of 32 and nothing but
// its own loads hides the ~0.7 us memory round trip; measured on MI355X, a barrier round
// per chunk (LDS write -> barrier -> LDS read -> 4 MFMAs) costs ~0.35 us, far more than
// the MFMAs in it.  So chunks move in STAGES of G: all loads of the next stage (G float4
// per operand piece and thread) are in flight while the current stage computes out of
// LDS, and a stage costs two barriers.  G = 4 for the small latency-bound tiles, 2 for the
// large throughput tiles (LDS budget).  fetch(c, slot) issues the loads of chunk c,
// stash(u, c, slot) parks a landed slot in LDS chunk buffer u, compute(u) runs the MFMAs
// of the chunk in buffer u.  Chunk indices past the end are clamped (re-read, never
// computed), which keeps the loads free of divergent branches.
#ifdef PFRL_QNET_DEBUG
// in-kernel phase clocks of the forward program (tools/qnet_phase.py): thread 0 of every workgroup
// stores wall_clock64 (100 MHz) stamps into g_qstamp[workgroup][k] -- plain stores, no atomics:
// k = 0 kernel entry, 1 loads of the first stage issued, 2 first stage parked in LDS (first data
// landed), 3 pipeline done, 4 accumulators folded, 5 end, 6 / 7 first loop stage: next stage's loads
// issued / its G chunks computed
__device__ unsigned long long g_qstamp[4096][8];
__device__ int g_qreps = 1;
// g_qgrid != 0: only the launch with that many workgroups records (one launch of a whole update)
__device__ int g_qgrid = 0;
#define QSTAMP(k)                                                                                   \
    do {                                                                                            \
        if (threadIdx.x == 0 && q_wg < 4096 &&                                                      \
            (g_qgrid == 0 || (int)(gridDim.x * gridDim.y * gridDim.z) == g_qgrid))                  \
            g_qstamp[q_wg][k] = wall_clock64();                                                     \
    } while (0)
#define QMARK(k, v)                                                                                 \
    do {                                                                                            \
        if (threadIdx.x == 0 && q_wg < 4096 &&                                                      \
            (g_qgrid == 0 || (int)(gridDim.x * gridDim.y * gridDim.z) == g_qgrid))                  \
            g_qstamp[q_wg][k] = (v);                                                                \
    } while (0)
#else
#define QSTAMP(k)
#define QMARK(k, v)
#endif

template <typename Slot, int G, typename Fetch, typename Stash, typename Compute>
__device__ __forceinline__ void run_pipeline(int c0, int c1, Fetch fetch, Stash stash,
                                             Compute compute) {
    if (c0 >= c1) return;
#ifdef PFRL_QNET_DEBUG
    const int q_wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#endif
    Slot slot[G];
#pragma unroll
    for (int u = 0; u < G; ++u) fetch(min(c0 + u, c1 - 1), slot[u]);
    QSTAMP(1);
#pragma unroll
    for (int u = 0; u < G; ++u) stash(u, min(c0 + u, c1 - 1), slot[u]);
    __syncthreads();
    QSTAMP(2);
    for (int cb = c0; cb < c1; cb += G) {
        const bool more = cb + G < c1;
        if (more) {
#pragma unroll
            for (int u = 0; u < G; ++u) fetch(min(cb + G + u, c1 - 1), slot[u]);
        }
#ifdef PFRL_QNET_DEBUG
        if (cb == c0) QSTAMP(6);
#endif
        const int nc = min(G, c1 - cb);
        if (nc == G) {
            // a full stage: no predicate between the chunks, the LDS reads of all G chunks can be
            // issued ahead of the MFMA stream
#pragma unroll
            for (int u = 0; u < G; ++u) compute(u);
        } else {
#pragma unroll
            for (int u = 0; u < G; ++u)
                if (u < nc) compute(u);
        }
#ifdef PFRL_QNET_DEBUG
        if (cb == c0) QSTAMP(7);
#endif
        __syncthreads();
        if (more) {
#pragma unroll
            for (int u = 0; u < G; ++u) stash(u, min(cb + G + u, c1 - 1), slot[u]);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------
// forward: y = act(conv(x, w) + b), also the split-K partial form for linear layers
// ---------------------------------------------------------------------------------
struct FwdArgs {
    const float *x, *w, *bias;
    float *y;
    ConvGeom g;
    int M, K, cps;
    int relu, planar, partial;
    // (TAIL only) input rows split over two tensors: columns [0, K1) from x (row stride K1),
    // [K1, K) from x2 (row stride K - K1) -- cat((obs, action)) without the copy
    const float *x2 = nullptr;
    int K1 = 0;
    // (NOISY only) factorised NoisyNet (pfrl/nn/noisy_linear.py:56-70): w is mu.W, w_sigma is
    // sigma.W, noise the layer's K + Cout unit Gaussians (inputs first), bias_sigma sigma.b.  The
    // B-operand loader forms W = mu + sigma * (f(r_out) f(r_in)) chunk by chunk -- the same three
    // roundings as pfrl_noisy_weights_fwd, so the product is the one on the materialised weights.
    const float *w_sigma = nullptr, *noise = nullptr, *bias_sigma = nullptr;
    // (U8 only) the input as u8 NHWC4 pixels (x is unused), r = fl(1 / d), d: see u8_over
    const uint8_t *xu8 = nullptr;
    float u8_r = 1.f, u8_d = 1.f;
};

__device__ __forceinline__ float noisy_shaped(float r) {
    // |sqrt(|r|)| * sign(r), sign(0) = 0 (csrc/noisy.hip)
    const float s = sqrtf(fabsf(r));
    return r > 0.0f ? s : (r < 0.0f ? -s : r * 0.0f);
}

// TAIL: linear layers (1x1) whose in_features are not a multiple of 32 or whose rows are
// not 16-byte aligned (MLP inputs such as 376 observations, 376 + 17 with the action
// appended): scalar loads, addresses clamped to the row, the overhang zeroed when parked.
template <int BM, int BN, int WM, int WN, int WK, int G, bool TAIL = false, bool NOISY = false,
          bool U8 = false>
__device__ __forceinline__ void fwd_body(const FwdArgs &p, const int bx, const int by, const int bz) {
    static_assert(WM * WN * WK == 4 && WK <= 2, "four waves");
    static_assert(!(TAIL && NOISY), "noisy weights: aligned rows only");
    static_assert(!(U8 && (TAIL || NOISY)), "u8 pixels: convolutions over four stacked frames only");
    constexpr int AM = BM / (16 * WM), AN = BN / (16 * WN);
    constexpr int P = (AM * AN == 1) ? 2 : 1;
    constexpr int NPA = (BM * 8 + 255) / 256, NPB = (BN * 8 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float As[G][BM * LDR];
    __shared__ __attribute__((aligned(16))) float Bs[G][BN * LDR];
    __shared__ f32x4 red[WK > 1 ? (WK - 1) * WM * WN * AM * AN * 64 : 1];
    // (the wave index in an SGPR: everything decided per wave -- which K sub-chunk it multiplies, which
    // operand piece it loads -- becomes a scalar branch instead of an exec-masked one; behind exec-masked
    // branches the compiler emitted ds_read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs per chunk, 258 clocks each)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int m0 = bx * BM, n0 = by * BN;
    const ConvGeom g = p.g;
    const int ohow = g.OH * g.OW;
    const int c0 = bz * p.cps;
    const int c1 = min(c0 + p.cps, TAIL ? (p.K + KC - 1) / KC : p.K / KC);

    struct Slot {
        float4 a[NPA], b[NPB];
        float4 s[NOISY ? NPB : 1], e;    // (NOISY) sigma pieces, this lane's four input-noise values
    };
    const float *ap[NPA], *ap2[NPA], *bp[NPB], *sp[NPB];
    const uint8_t *au[NPA];              // (U8) the piece's pixel in the u8 input
    float fo[NPB];                       // (NOISY) f(r_out) of the piece's weight row
    bool aok[NPA], bok[NPB];
    const int K1 = (TAIL && p.x2 != nullptr) ? p.K1 : p.K;
#pragma unroll
    for (int pp = 0; pp < NPA; ++pp) {
        const int f = tid + 256 * pp, row = f >> 3, q = f & 7;
        const int m = m0 + row;
        const bool ok = row < BM && m < p.M;
        const int mm = ok ? m : 0;
        const int n = fdiv(mm, g.q_ohow), rem = mm - n * ohow;
        const int oh = fdiv(rem, g.q_ow), ow = rem - oh * g.OW;
        if (TAIL) {
            ap[pp] = p.x + (size_t)mm * K1;
            ap2[pp] = p.x2 != nullptr ? p.x2 + (size_t)mm * (p.K - K1) : ap[pp];
        } else {
            const size_t at = ((size_t)(n * g.H + oh * g.ST) * g.W + ow * g.ST) * g.C + 4 * q;
            ap[pp] = p.x + at;
            ap2[pp] = ap[pp];
            au[pp] = p.xu8 + at;          // (one byte per element: the same element offset)
        }
        aok[pp] = ok;
    }
#pragma unroll
    for (int pp = 0; pp < NPB; ++pp) {
        const int f = tid + 256 * pp, row = f >> 3, q = f & 7;
        const int co = n0 + row;
        const bool ok = row < BN && co < g.Cout;
        bp[pp] = p.w + (size_t)(ok ? co : 0) * p.K + (TAIL ? 0 : 4 * q);
        bok[pp] = ok;
        if (NOISY) {
            sp[pp] = p.w_sigma + (size_t)(ok ? co : 0) * p.K + 4 * q;
            fo[pp] = noisy_shaped(p.noise[p.K + (ok ? co : 0)]);
        }
    }
    const int SC = g.S * g.C, WC = g.W * g.C;
    const int kq4 = 4 * (tid & 7);
    // operand pieces smaller than the workgroup (BM = 16: 128 float4) belong to the first waves only:
    // decided per wave, so the others skip the load and the LDS write on a scalar branch
    auto a_on = [&](int pp) { return BM * 8 >= 256 * (pp + 1) || wave * 64 + 256 * pp < BM * 8; };
    auto b_on = [&](int pp) { return BN * 8 >= 256 * (pp + 1) || wave * 64 + 256 * pp < BN * 8; };
    int f_c = c0, f_o = 0, f_ad = 0;
    if (!TAIL) {
        const int r = fdiv(c0 * KC, g.q_sc);
        f_o = c0 * KC - r * SC;
        f_ad = r * WC + f_o;
    }
    auto fetch = [&](int c, Slot &sl) {
        const int k0 = c * KC;
        if (TAIL) {
            const int kl = p.K - 1;
            const int i0 = min(k0 + kq4, kl), i1 = min(k0 + kq4 + 1, kl), i2 = min(k0 + kq4 + 2, kl),
                      i3 = min(k0 + kq4 + 3, kl);
            // (the address is selected, not the load predicated: see ldg4)
#define PFRL_A_AT(pp, i) (*((i) < K1 ? ap[pp] + (i) : ap2[pp] + ((i) - K1)))
#pragma unroll
            for (int pp = 0; pp < NPA; ++pp)
                sl.a[pp] = make_float4(PFRL_A_AT(pp, i0), PFRL_A_AT(pp, i1), PFRL_A_AT(pp, i2),
                                       PFRL_A_AT(pp, i3));
#undef PFRL_A_AT
#pragma unroll
            for (int pp = 0; pp < NPB; ++pp)
                sl.b[pp] = make_float4(bp[pp][i0], bp[pp][i1], bp[pp][i2], bp[pp][i3]);
            return;
        }
        // chunks are asked for in non-decreasing order (the clamp at the end repeats the last one):
        // the (kernel row, offset in the row) of the chunk advances by scalar adds, no division
        if (c > f_c) {
            f_c = c;
            f_o += KC;
            f_ad += KC;
            if (f_o >= SC) {
                f_o -= SC;
                f_ad += WC - SC;
            }
        }
#pragma unroll
        for (int pp = 0; pp < NPA; ++pp)
            if (a_on(pp)) {
                if (U8)   // (C = 4: the float4 piece of the fp32 form is ONE dword here; parked in .x)
                    sl.a[pp].x = __uint_as_float(*reinterpret_cast<const uint32_t *>(au[pp] + f_ad));
                else
                    sl.a[pp] = ldg4(ap[pp] + f_ad);
            }
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp)
            if (b_on(pp)) {
                sl.b[pp] = ldg4(bp[pp] + k0);
                if (NOISY) sl.s[pp] = ldg4(sp[pp] + k0);
            }
        if (NOISY) sl.e = ldg4(p.noise + k0 + kq4);
    };
    auto stash = [&](int buf, int c, const Slot &sl) {
        // (TAIL) elements of this lane that lie inside the row; the overhang is zeroed
        const int left = TAIL ? p.K - (c * KC + kq4) : 4;
#pragma unroll
        for (int pp = 0; pp < NPA; ++pp) {
            const int f = tid + 256 * pp, row = f >> 3, q = f & 7;
            // (rows past M / Cout are never zeroed: a row of A or B only reaches its own row / column
            // of the product, and those are not stored)
            float4 v = sl.a[pp];
            if (U8) v = u8x4_over(__float_as_uint(sl.a[pp].x), p.u8_r, p.u8_d);
            if (TAIL) {
                v.x = left > 0 ? v.x : 0.f; v.y = left > 1 ? v.y : 0.f;
                v.z = left > 2 ? v.z : 0.f; v.w = left > 3 ? v.w : 0.f;
            }
            if (a_on(pp)) *reinterpret_cast<float4 *>(&As[buf][row * LDR + 4 * q]) = v;
        }
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp) {
            const int f = tid + 256 * pp, row = f >> 3, q = f & 7;
            float4 v = sl.b[pp];
            if (TAIL) {
                v.x = left > 0 ? v.x : 0.f; v.y = left > 1 ? v.y : 0.f;
                v.z = left > 2 ? v.z : 0.f; v.w = left > 3 ? v.w : 0.f;
            }
            if (NOISY) {
                // mu + sigma * (f_out * f_in): multiply, multiply, add -- the roundings of k_noisy_fwd
                const float4 sg = sl.s[pp];
                const float f = fo[pp];
                v.x = __fadd_rn(v.x, __fmul_rn(sg.x, __fmul_rn(f, noisy_shaped(sl.e.x))));
                v.y = __fadd_rn(v.y, __fmul_rn(sg.y, __fmul_rn(f, noisy_shaped(sl.e.y))));
                v.z = __fadd_rn(v.z, __fmul_rn(sg.z, __fmul_rn(f, noisy_shaped(sl.e.z))));
                v.w = __fadd_rn(v.w, __fmul_rn(sg.w, __fmul_rn(f, noisy_shaped(sl.e.w))));
            }
            if (b_on(pp)) *reinterpret_cast<float4 *>(&Bs[buf][row * LDR + 4 * q]) = v;
        }
    };

    f32x4 acc[AM][AN][P];
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) acc[am][an][pp] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        // (WK = 2: this wave's K half is an LDS address offset, not a branch)
#pragma unroll
        for (int s = 0; s < 2 / WK; ++s)
            mma_sub<AM, AN, P, true, true, LDR, LDR>(As[buf], Bs[buf], wm * 16 * AM, wn * 16 * AN,
                                                          WK == 2 ? wk : s, lane, acc);
    };
    run_pipeline<Slot, G>(c0, c1, fetch, stash, compute);
#ifdef PFRL_QNET_DEBUG
    const int q_wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#endif
    QSTAMP(3);
    fold_acc<AM, AN, P, WK, WM * WN>(acc, red, wm * WN + wn, wk, lane);
    QSTAMP(4);
    if (!p.planar && n0 + BN <= g.Cout && (g.Cout & 3) == 0) {
        // Row-major output (NHWC rows, split-K slabs) of a tile that lies inside the channel range:
        // the accumulators go through LDS (the chunk buffers are free now) and every thread stores
        // whole float4 pieces of a row -- BN/4 lanes cover a contiguous 4*BN bytes -- instead of 4-byte
        // pieces of four rows behind per-element bounds branches (the 64 x 64 program's epilogue was
        // ~1 000 instructions, as long as its 8-chunk main loop for the first convolution).
        constexpr int LDT = BN + 4;
        static_assert(BM * LDT <= G * BM * LDR, "the tile fits in the A chunk buffers");
        float *tile = &As[0][0];
        __syncthreads();          // every wave is done reading the chunk buffers
        if (wk == 0) {
#pragma unroll
            for (int am = 0; am < AM; ++am)
#pragma unroll
                for (int an = 0; an < AN; ++an) {
                    const int col = wn * 16 * AN + 16 * an + (lane & 15);
                    float bias = p.partial ? 0.f : p.bias[n0 + col];
                    if (NOISY && !p.partial)
                        bias = __fadd_rn(bias, __fmul_rn(p.bias_sigma[n0 + col],
                                                         noisy_shaped(p.noise[p.K + n0 + col])));
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        float v = acc[am][an][0][reg];
                        if (!p.partial) {
                            v = v + bias;
                            if (p.relu) v = fmaxf(v, 0.f);
                        }
                        tile[(wm * 16 * AM + 16 * am + 4 * (lane >> 4) + reg) * LDT + col] = v;
                    }
                }
        }
        __syncthreads();
        constexpr int Q = BN / 4;
        float *yb = p.y + (p.partial ? (size_t)bz * p.M * g.Cout : (size_t)0) + n0;
#pragma unroll
        for (int f = tid; f < BM * Q; f += 256) {
            const int row = f / Q, c4 = 4 * (f - row * Q);
            const int m = m0 + row;
            if (m < p.M)
                *reinterpret_cast<float4 *>(yb + (size_t)m * g.Cout + c4) =
                    *reinterpret_cast<const float4 *>(&tile[row * LDT + c4]);
        }
        return;
    }
    if (wk != 0) return;
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an) {
            const int n = n0 + wn * 16 * AN + 16 * an + (lane & 15);
            if (n >= g.Cout) continue;
            float bias = p.partial ? 0.f : p.bias[n];
            if (NOISY && !p.partial)
                bias = __fadd_rn(bias, __fmul_rn(p.bias_sigma[n], noisy_shaped(p.noise[p.K + n])));
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = m0 + wm * 16 * AM + 16 * am + 4 * (lane >> 4) + reg;
                if (m >= p.M) continue;
                float v = acc[am][an][0][reg];
                if (p.partial) {
                    p.y[((size_t)bz * p.M + m) * g.Cout + n] = v;
                    continue;
                }
                v = v + bias;
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.planar) {
                    const int img = fdiv(m, g.q_ohow), pix = m - img * ohow;
                    p.y[((size_t)img * g.Cout + n) * ohow + pix] = v;
                } else {
                    p.y[(size_t)m * g.Cout + n] = v;
                }
            }
        }
}

template <int BM, int BN, int WM, int WN, int WK, int G, bool TAIL = false, bool NOISY = false,
          bool U8 = false>
__global__ __launch_bounds__(256) void k_conv_fwd(FwdArgs p) {
#ifdef PFRL_QNET_DEBUG
    // g_qreps = 2: the body runs twice and the stamps of the SECOND pass stay -- the same work with
    // the code already fetched (what the instruction fetch of a cold launch costs)
    const int q_wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    for (int rep = g_qreps; rep > 0; --rep) {
        QSTAMP(0);
        fwd_body<BM, BN, WM, WN, WK, G, TAIL, NOISY, U8>(p, blockIdx.x, blockIdx.y, blockIdx.z);
        __syncthreads();
        QSTAMP(5);
    }
#else
    fwd_body<BM, BN, WM, WN, WK, G, TAIL, NOISY, U8>(p, blockIdx.x, blockIdx.y, blockIdx.z);
#endif
}

// Twin launch: two independent problems of the same shape (the twin Q-networks of SAC / TD3,
// pfrl/agents/soft_actor_critic.py:97-110) side by side in one grid, blockIdx.z picks the
// problem.  Inside a captured graph a launch costs ~4 us whatever it computes, and the twins
// are always evaluated on the same inputs one after the other.  No split-K here.
template <int BM, int BN, int WM, int WN, int WK, int G, bool TAIL, bool NOISY = false>
__global__ __launch_bounds__(256) void k_conv_fwd2(FwdArgs p0, FwdArgs p1) {
    const FwdArgs &p = blockIdx.z == 0 ? p0 : p1;
    // (the two problems may differ in out_features: the grid is sized for the wider one)
    if ((int)blockIdx.y * BN >= p.g.Cout) return;
    fwd_body<BM, BN, WM, WN, WK, G, TAIL, NOISY>(p, blockIdx.x, blockIdx.y, 0);
}

// ---------------------------------------------------------------------------------
// The first Nature-DQN convolution (84 x 84 x 4 u8 pixels, 8 x 8 stride 4, 32 channels:
// pfrl/q_functions + examples/atari NatureDQNHead conv1) in DIRECT form.  The implicit-GEMM
// program above parks every input pixel in LDS four times (each pixel lies under 2 x 2 output
// positions) and converts it from u8 four times; at Cout = 32 that operand traffic, not the MFMAs,
// is what its workgroups spend their issue slots on (0.52-0.55 of the f32 MFMA peak).  Here a
// workgroup stages a BAND -- the 20 input rows under 4 output rows of an image, converted once --
// and the MFMA A fragments are read straight out of the staged pixels: unit (row, pixel) = the
// pixel's four channels as one float4, stored as [row][pixel % 4][pixel / 4] so that the 16 lanes
// of a fragment (16 consecutive output columns, one kernel column) read consecutive units, and
// the (kernel row, kernel column) of a fragment is an immediate offset.  The weights never enter
// LDS: a wave owns 16 output channels and keeps their 16 x 256 weights as MFMA B fragments in 64
// VGPRs for its whole (persistent) life.  A workgroup takes two bands at a time -- the same band of
// two consecutive images: 2 x 5 pixel tiles x 2 channel halves = 20 tiles, five per wave, all of
// one channel half -- with the next unit's raw dwords in flight (registers) while it multiplies.
// Per output element the terms enter the accumulator in the order of the tile programs with one
// accumulator per tile (kernel row, half row, channel, column within the half): bit-identical to
// k_conv_fwd<128, 32, ..., U8> / <64, 32, ..., U8>.
// ---------------------------------------------------------------------------------
constexpr int D1_HW = 84, D1_O = 20, D1_BAND_ROWS = 20, D1_AS = 21, D1_RS = 84;
constexpr int D1_BAND_UNITS = D1_BAND_ROWS * D1_RS;       // float4 units of one staged band
constexpr int D1_NQ = (D1_BAND_UNITS + 255) / 256;        // raw dwords per thread and band

// U8 = false: the same kernel on the fp32 NHWC4 minibatch (a pixel = one float4; DQN's acting and target
// passes, pfrl_conv2d_nhwc_fwd): nothing to convert, and the pixels of a unit are loaded at the top of
// its own step instead of a step ahead (14 float4 per thread would not fit beside the fragments at
// three waves per SIMD; the other workgroups of the CU cover the load).
template <bool U8>
__global__ __launch_bounds__(256, 3) void k_conv1_u8_direct(
    const void *__restrict__ xin, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ y, int N, int relu, float u8_r, float u8_d, int units) {
    using Raw = typename std::conditional<U8, uint32_t, float4>::type;
    const Raw *__restrict__ x = static_cast<const Raw *>(xin);
    __shared__ float4 band[2 * D1_BAND_UNITS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, ub = wave >> 1;            // channel half, which of the two bands
    const int i = lane & 15, kq = lane >> 4;
    // B fragments of this wave's 16 channels: [kernel row][half row] x float4 (channels of one pixel)
    float4 bw[8][2];
    {
        const float *wr = w + (size_t)(16 * half + i) * 256 + 4 * kq;
#pragma unroll
        for (int kh = 0; kh < 8; ++kh)
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) bw[kh][sc] = ldg4(wr + 32 * kh + 16 * sc);
    }
    const float4 bias_v = ldg4(bias + 16 * half + 4 * kq);
    // A fragment base (float4 units) of the wave's five pixel tiles: tile j = band pixels 16 j .. 16 j + 15
    int abase[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int px = 16 * j + i, oh = px / D1_O, ow = px - D1_O * oh;
        abase[j] = ub * D1_BAND_UNITS + 4 * oh * D1_RS + kq * D1_AS + ow;
    }
    // where this thread's raw dwords of a band go (dword d = row d / 84, pixel d % 84)
    int sdst[D1_NQ], ssrc[D1_NQ];
#pragma unroll
    for (int q = 0; q < D1_NQ; ++q) {
        const int d = min(tid + 256 * q, D1_BAND_UNITS - 1);
        const int rr = d / D1_HW, px = d - D1_HW * rr;
        ssrc[q] = d;
        sdst[q] = rr * D1_RS + (px & 3) * D1_AS + (px >> 2);
    }
    Raw raw[2][D1_NQ];
    auto fetch = [&](int unit) {
        const int pair = unit / 5, bnd = unit - 5 * pair;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int img = min(2 * pair + b, N - 1);
            const Raw *src = x + (size_t)img * (D1_HW * D1_HW) + bnd * 16 * D1_HW;
#pragma unroll
            for (int q = 0; q < D1_NQ; ++q) raw[b][q] = src[ssrc[q]];
        }
    };
    // One unit.  The loads of the NEXT unit and the stores of this one are unconditional (clamped),
    // and the first unit is peeled below, so every path into the loop header carries the same
    // [14 loads, 5 stores] in flight: the compiler's vmcnt for "this thread's pixels have landed" then
    // does not also wait for the previous unit's stores to be acknowledged.
    auto step = [&](const int unit) {
        if constexpr (U8) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < D1_NQ; ++q)
                    if (tid + 256 * q < D1_BAND_UNITS)
                        band[b * D1_BAND_UNITS + sdst[q]] = u8x4_over(raw[b][q], u8_r, u8_d);
        } else {
            // (band by band: seven float4 in flight per thread, not fourteen)
            const int pair = unit / 5, bnd = unit - 5 * pair;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int img = min(2 * pair + b, N - 1);
                const Raw *src = x + (size_t)img * (D1_HW * D1_HW) + bnd * 16 * D1_HW;
                Raw r[D1_NQ];
#pragma unroll
                for (int q = 0; q < D1_NQ; ++q) r[q] = src[ssrc[q]];
#pragma unroll
                for (int q = 0; q < D1_NQ; ++q)
                    if (tid + 256 * q < D1_BAND_UNITS) band[b * D1_BAND_UNITS + sdst[q]] = r[q];
            }
        }
        __syncthreads();
        if constexpr (U8) fetch(min(unit + (int)gridDim.x, units - 1));
        const int pair = unit / 5, bnd = unit - 5 * pair;
        // (an odd batch: the last unit's second band repeats the last image -- fetch() clamps the same
        // way -- and both waves store the same values to the same rows; unconditional stores also let
        // the compiler count them in vmcnt, so the wait for the next unit's pixels does not wait for
        // this unit's stores to be acknowledged)
        const int img = min(2 * pair + ub, N - 1);
        // (the weights are the MFMA's A operand, the pixels its B operand: a lane then holds four
        // consecutive channels of ONE pixel -- a float4 store -- and the products are the same)
        float *yo = y + ((size_t)img * (D1_O * D1_O) + bnd * 4 * D1_O + i) * 32 + 16 * half + 4 * kq;
        auto tiles = [&](auto nt, const int j0) {
            constexpr int NT = decltype(nt)::value;
            f32x4 acc[NT];
            float4 a[2][NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                a[0][u] = band[abase[j0 + u]];
            }
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                // fragments of the next (kernel row, half row) are read while this one multiplies
                if (st + 1 < 16) {
#pragma unroll
                    for (int u = 0; u < NT; ++u)
                        a[(st + 1) & 1][u] = band[abase[j0 + u] + ((st + 1) >> 1) * D1_RS + ((st + 1) & 1)];
                }
                const float4 b = bw[st >> 1][st & 1];
                const float4 *ac = a[st & 1];
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, ac[u].x, acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, ac[u].y, acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, ac[u].z, acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, ac[u].w, acc[u], 0, 0, 0);
                if (st + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
            }
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                float4 v = make_float4(acc[u][0] + bias_v.x, acc[u][1] + bias_v.y, acc[u][2] + bias_v.z,
                                       acc[u][3] + bias_v.w);
                if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                *reinterpret_cast<float4 *>(yo + (size_t)(16 * (j0 + u)) * 32) = v;
            }
        };
        tiles(std::integral_constant<int, 2>{}, 0);
        tiles(std::integral_constant<int, 2>{}, 2);
        tiles(std::integral_constant<int, 1>{}, 4);
        __syncthreads();
    };
    int unit = blockIdx.x;
    if (unit >= units) return;
    if constexpr (U8) fetch(unit);
    step(unit);
    for (unit += gridDim.x; unit < units; unit += gridDim.x) step(unit);
}

// persistent workgroups of the direct forward kernel: three per CU (what their LDS allows)
static int conv1_direct_slots() {
    static thread_local int slots_dev = -1, slots = 768;
    int devid = 0;
    if (hipGetDevice(&devid) == hipSuccess && devid != slots_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) == hipSuccess && cus > 0)
            slots = 3 * cus;
        slots_dev = devid;
    }
    return slots;
}

// ---------------------------------------------------------------------------------
// dgrad: gradient w.r.t. the layer input, masked by the ReLU of the layer below
// ---------------------------------------------------------------------------------
struct DgradArgs {
    const float *dy, *dymask, *w, *aprev;
    float *dx;
    ConvGeom g;          // forward geometry (H, W = input extents)
    int AH, AW, Mc;      // rows per parity class: n x AH x AW, AH = H / ST
    int TH, TW, K;       // taps per class and dimension; K = TH * TW * Cout
    int permP, permC;    // dx written as [n][p][c] for c*P + p (planar -> NHWC), 0 = off
    FastDiv q_ahw, q_aw, q_st, q_permP;   // by AH * AW, AW, ST, permP
    FastDiv q_c;                          // by C (the class of a merged column)
};

// MERGE: the ST x ST stride-parity classes of a strided convolution side by side in the tile's
// columns (column n = class * C + ci, BN a multiple of C) instead of one class per grid.z slice.
// Every class of an input cell (a, a2) reads the SAME dy taps -- only the weight slice differs --
// so the dy tile (the operand that streams from memory; the weights are cache resident) is loaded
// and parked once for 2 or 4 classes: the second convolution's input gradient (C = 32, four
// classes) becomes a 64- or 128-column problem instead of four 32-column ones.  Same terms in the
// same order per element: bit-identical to the per-class launch.
template <int BM, int BN, int WM, int WN, int WK, int G, bool MERGE = false>
__device__ __forceinline__ void dgrad_body(const DgradArgs &p, const int bx, const int by,
                                           const int bz, float *smem) {
    static_assert(WM * WN * WK == 4 && WK <= 2, "four waves");
    constexpr int AM = BM / (16 * WM), AN = BN / (16 * WN);
    constexpr int P = (AM * AN == 1) ? 2 : 1;
    constexpr int NPA = (BM * 8 + 255) / 256;
    constexpr int QPR = BN / 4, NPB = (32 * QPR + 255) / 256, LDB = BN + 4;
    float(*As)[BM * LDR] = reinterpret_cast<float(*)[BM * LDR]>(smem);
    float(*Bs)[32 * LDB] = reinterpret_cast<float(*)[32 * LDB]>(smem + G * BM * LDR);
    f32x4 *red = reinterpret_cast<f32x4 *>(smem + G * BM * LDR + G * 32 * LDB);
    // (the wave index in an SGPR: everything decided per wave -- which K sub-chunk it multiplies, which
    // operand piece it loads -- becomes a scalar branch instead of an exec-masked one; behind exec-masked
    // branches the compiler emitted ds_read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs per chunk, 258 clocks each)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int m0 = bx * BM, n0 = by * BN;
    const ConvGeom g = p.g;
    const int ph = fdiv(bz, p.q_st), pw = bz - ph * g.ST;     // (per workgroup; MERGE: per column)
    const int ahw = p.AH * p.AW;

    // per-thread part of the dy address (tap (0, 0), first output channel) and of the weight address;
    // what a chunk adds to both is the same for the whole workgroup
    int rah[NPA], raw[NPA];
    size_t abase[NPA];
    bool aok[NPA];
#pragma unroll
    for (int pp = 0; pp < NPA; ++pp) {
        const int f = tid + 256 * pp, row = f >> 3, q = f & 7;
        const int m = m0 + row;
        const bool ok = row < BM && m < p.Mc;
        const int mm = ok ? m : 0;
        const int rn = fdiv(mm, p.q_ahw);
        const int rem = mm - rn * ahw;
        rah[pp] = fdiv(rem, p.q_aw);
        raw[pp] = rem - rah[pp] * p.AW;
        abase[pp] = ((size_t)(rn * g.OH + rah[pp]) * g.OW + raw[pp]) * g.Cout + 4 * q;
        aok[pp] = ok;
    }
    int bkk[NPB], bq4[NPB];
    size_t bbase[NPB];
#pragma unroll
    for (int pp = 0; pp < NPB; ++pp) {
        const int f = tid + 256 * pp;
        bkk[pp] = f / QPR;
        bq4[pp] = 4 * (f - bkk[pp] * QPR);
        if (MERGE) {
            // column -> (class, input channel): the class's (ph, pw) shift of the kernel position
            // is a per-thread constant of the weight address
            const int col = n0 + bq4[pp];
            // (FastDiv: a compiler-expanded division is ~20 vector instructions, and these
            // workgroups only run 4-8 chunks)
            const int cls = fdiv(col, p.q_c), ci = col - cls * g.C;
            const int cph = fdiv(cls, p.q_st), cpw = cls - cph * g.ST;
            bbase[pp] = (size_t)bkk[pp] * g.R * g.S * g.C + (size_t)(cph * g.S + cpw) * g.C + ci;
        } else {
            bbase[pp] = (size_t)bkk[pp] * g.R * g.S * g.C + n0 + bq4[pp];
        }
    }
    struct Slot {
        float4 a[NPA], h[NPA], b[NPB];
        bool ok[NPA];
    };
    const bool has_mask = p.dymask != nullptr;
    // operand pieces smaller than the workgroup belong to the first waves only (scalar branches)
    auto a_on = [&](int pp) { return BM * 8 >= 256 * (pp + 1) || wave * 64 + 256 * pp < BM * 8; };
    auto b_on = [&](int pp) { return 32 * QPR >= 256 * (pp + 1) || wave * 64 + 256 * pp < 32 * QPR; };
    // chunk -> (tap row, tap column, first output channel), advanced by scalar adds: chunks are asked
    // for in non-decreasing order (Cout % 32 == 0: a chunk never straddles two taps)
    int f_c = 0, f_co0 = 0, f_tb = 0, f_tb2 = 0;
    auto fetch = [&](int c, Slot &sl) {
        if (c > f_c) {
            f_c = c;
            f_co0 += KC;
            if (f_co0 >= g.Cout) {
                f_co0 = 0;
                if (++f_tb2 >= p.TW) {
                    f_tb2 = 0;
                    ++f_tb;
                }
            }
        }
        // (uniform, may be "negative": wraps consistently in size_t arithmetic)
        const size_t adelta = (size_t)f_co0 - (size_t)(f_tb * g.OW + f_tb2) * g.Cout;
#pragma unroll
        for (int pp = 0; pp < NPA; ++pp) {
            if (!a_on(pp)) continue;
            const int oh = rah[pp] - f_tb, ow = raw[pp] - f_tb2;
            const bool ok = aok[pp] && oh >= 0 && oh < g.OH && ow >= 0 && ow < g.OW;
            const size_t off = ok ? abase[pp] + adelta : (size_t)0;
            sl.a[pp] = ldg4(p.dy + off);
            if (has_mask) sl.h[pp] = ldg4(p.dymask + off);
            sl.ok[pp] = ok;
        }
        const int r = f_tb * g.ST + (MERGE ? 0 : ph), s = f_tb2 * g.ST + (MERGE ? 0 : pw);
        const size_t bdelta = ((size_t)(f_co0 * g.R + r) * g.S + s) * g.C;
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp)
            if (b_on(pp)) sl.b[pp] = ldg4(p.w + bbase[pp] + bdelta);
    };
    auto stash = [&](int buf, int c, const Slot &sl) {
#pragma unroll
        for (int pp = 0; pp < NPA; ++pp) {
            if (!a_on(pp)) continue;
            const int f = tid + 256 * pp, row = f >> 3, q = f & 7;
            // (taps outside the output are zeroed: they are terms of the sum over K)
            float4 v = zero_unless(sl.a[pp], sl.ok[pp]);
            if (has_mask) v = relu_mask(v, sl.h[pp]);
            *reinterpret_cast<float4 *>(&As[buf][row * LDR + 4 * q]) = v;
        }
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp)
            if (b_on(pp)) *reinterpret_cast<float4 *>(&Bs[buf][bkk[pp] * LDB + bq4[pp]]) = sl.b[pp];
    };

    f32x4 acc[AM][AN][P];
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) acc[am][an][pp] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        // (WK = 2: this wave's K half is an LDS address offset, not a branch)
#pragma unroll
        for (int s = 0; s < 2 / WK; ++s)
            mma_sub<AM, AN, P, true, false, LDR, LDB>(As[buf], Bs[buf], wm * 16 * AM, wn * 16 * AN,
                                                           WK == 2 ? wk : s, lane, acc);
    };
    run_pipeline<Slot, G>(0, p.K / KC, fetch, stash, compute);
    fold_acc<AM, AN, P, WK, WM * WN>(acc, red, wm * WN + wn, wk, lane);
    if (p.permP == 0 && (MERGE || n0 + BN <= g.C) && (g.C & 3) == 0) {
        // NHWC rows of a tile inside the channel range: through LDS (the A chunk buffers are free),
        // so that a thread masks and stores a float4 of one row and the row -> (image, ih, iw) split
        // runs once per 16 bytes instead of once per element
        constexpr int LDT = BN + 4;
        static_assert(BM * LDT <= G * BM * LDR + G * 32 * LDB, "the tile fits in the chunk buffers");
        float *tile = &As[0][0];      // (A buffers, running on into the B buffers behind them)
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int am = 0; am < AM; ++am)
#pragma unroll
                for (int an = 0; an < AN; ++an)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        tile[(wm * 16 * AM + 16 * am + 4 * (lane >> 4) + reg) * LDT + wn * 16 * AN + 16 * an +
                             (lane & 15)] = acc[am][an][0][reg];
        }
        __syncthreads();
        constexpr int Q = BN / 4;
#pragma unroll
        for (int f = tid; f < BM * Q; f += 256) {
            const int row = f / Q, c4 = 4 * (f - row * Q);
            const int m = m0 + row;
            if (m >= p.Mc) continue;
            const int n = fdiv(m, p.q_ahw), rem = m - n * ahw;
            const int a = fdiv(rem, p.q_aw), a2 = rem - a * p.AW;
            int eph = ph, epw = pw, ec = n0 + c4;
            if (MERGE) {
                const int cls = fdiv(ec, p.q_c);
                ec -= cls * g.C;
                eph = fdiv(cls, p.q_st);
                epw = cls - eph * g.ST;
            }
            const int ih = a * g.ST + eph, iw = a2 * g.ST + epw;
            const size_t o = ((size_t)(n * g.H + ih) * g.W + iw) * g.C + ec;
            float4 v = *reinterpret_cast<const float4 *>(&tile[row * LDT + c4]);
            if (p.aprev != nullptr) v = relu_mask(v, *reinterpret_cast<const float4 *>(p.aprev + o));
            *reinterpret_cast<float4 *>(p.dx + o) = v;
        }
        return;
    }
    if (wk != 0) return;
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = m0 + wm * 16 * AM + 16 * am + 4 * (lane >> 4) + reg;
            if (m >= p.Mc) continue;
            const int n = fdiv(m, p.q_ahw), rem = m - n * ahw;
            const int a = fdiv(rem, p.q_aw), a2 = rem - a * p.AW;
            const int ih = a * g.ST + ph, iw = a2 * g.ST + pw;
            const size_t base = ((size_t)(n * g.H + ih) * g.W + iw) * g.C;
#pragma unroll
            for (int an = 0; an < AN; ++an) {
                const int ci = n0 + wn * 16 * AN + 16 * an + (lane & 15);
                if (ci >= g.C) continue;
                float v = acc[am][an][0][reg];
                if (p.aprev != nullptr) v = p.aprev[base + ci] > 0.f ? v : 0.f;
                size_t o = base + ci;
                if (p.permP > 0) {
                    const int c = fdiv(ci, p.q_permP), px = ci - c * p.permP;
                    o = base + (size_t)px * p.permC + c;
                }
                p.dx[o] = v;
            }
        }
}

// LDS floats of a tile program: G chunk buffers per operand + the split-K fold area
constexpr int red_floats(int BM, int BN, int WM, int WN, int WK) {
    return WK > 1 ? (WK - 1) * WM * WN * (BM / (16 * WM)) * (BN / (16 * WN)) * 256 : 4;
}
constexpr int dgrad_smem(int BM, int BN, int WM, int WN, int WK, int G) {
    return G * BM * LDR + G * 32 * (BN + 4) + red_floats(BM, BN, WM, WN, WK);
}
constexpr int wgrad_smem(int BI, int BJ, int WM, int WN, int WK, int G) {
    return G * 32 * (BI + 4) + G * 32 * (BJ + 4) + red_floats(BI, BJ, WM, WN, WK);
}
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int BM, int BN, int WM, int WN, int WK, int G, bool MERGE = false>
__global__ __launch_bounds__(256) void k_conv_dgrad(DgradArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[dgrad_smem(BM, BN, WM, WN, WK, G)];
    dgrad_body<BM, BN, WM, WN, WK, G, MERGE>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// POS: the input gradient of a stride-1 convolution at rollout / update batch sizes, tiled by INPUT
// POSITION: the BM rows of a tile are BM consecutive images at ONE input pixel (a, a2).  Which taps
// (r, s) of that pixel fall inside the output is then a property of the whole tile -- a rectangle
// [tb_lo, tb_hi] x [t2_lo, t2_hi] -- and the others are simply not walked: for the 3 x 3 layer on a
// 9 x 9 input (7 x 7 output) 40 % of the taps of the row-major tiling are padding that costs MFMAs
// and loads for nothing (a corner pixel has 1 valid tap of 9, an interior one 9).  The BM dy rows of a
// tap are one 4 * Cout-byte piece per image (12.5 KB apart): every piece is needed by up to R * S
// positions, so the workgroups of one image block are kept on ONE XCD (blockIdx.x % 8 is the XCD a
// workgroup lands on: the image block is chosen by it, the position by blockIdx.x / 8) and the block's
// dy (BM x OH x OW x Cout floats, 0.8 MB) is read from HBM once and from that XCD's L2 afterwards.
// Same terms in the same order per element as the row-major program, minus exact zeros.
// CLS: a strided layer -- the tile's columns are (parity class, input channel) as in MERGE, the
// position is a cell (a, a2) of the AH x AW grid every class shares, taps step by ST kernel positions.
template <int BM, int BN, int WM, int WN, int WK, int G, bool CLS = false>
__device__ __forceinline__ void dgrad_pos_body(const DgradArgs &p, const int img_blk, const int pos,
                                               const int by, float *smem) {
    static_assert(WM * WN * WK == 4 && WK <= 2, "four waves");
    constexpr int AM = BM / (16 * WM), AN = BN / (16 * WN);
    constexpr int P = (AM * AN == 1) ? 2 : 1;
    constexpr int NPA = (BM * 8 + 255) / 256;
    constexpr int QPR = BN / 4, NPB = (32 * QPR + 255) / 256, LDB = BN + 4;
    float(*As)[BM * LDR] = reinterpret_cast<float(*)[BM * LDR]>(smem);
    float(*Bs)[32 * LDB] = reinterpret_cast<float(*)[32 * LDB]>(smem + G * BM * LDR);
    f32x4 *red = reinterpret_cast<f32x4 *>(smem + G * BM * LDR + G * 32 * LDB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int n0 = by * BN;
    const ConvGeom g = p.g;
    const int a = fdiv(pos, p.q_aw), a2 = pos - a * p.AW;
    // taps whose output pixel (a - tb, a2 - t2) exists
    const int tb_lo = max(0, a - (g.OH - 1)), tb_hi = min(p.TH - 1, a);
    const int t2_lo = max(0, a2 - (g.OW - 1)), t2_hi = min(p.TW - 1, a2);
    const int cpt = g.Cout / KC;
    const int nchunks = (tb_hi - tb_lo + 1) * (t2_hi - t2_lo + 1) * cpt;

    size_t abase[NPA];
    bool aok[NPA];
#pragma unroll
    for (int pp = 0; pp < NPA; ++pp) {
        const int f = tid + 256 * pp, row = f >> 3, q = f & 7;
        const int n = img_blk * BM + row;
        aok[pp] = row < BM && n < g.N;
        abase[pp] = ((size_t)((aok[pp] ? n : 0) * g.OH + a) * g.OW + a2) * g.Cout + 4 * q;
    }
    int bkk[NPB], bq4[NPB];
    size_t bbase[NPB];
#pragma unroll
    for (int pp = 0; pp < NPB; ++pp) {
        const int f = tid + 256 * pp;
        bkk[pp] = f / QPR;
        bq4[pp] = 4 * (f - bkk[pp] * QPR);
        if (CLS) {
            const int col = n0 + bq4[pp];
            // (FastDiv: a compiler-expanded division is ~20 vector instructions, and these
            // workgroups only run 4-8 chunks)
            const int cls = fdiv(col, p.q_c), ci = col - cls * g.C;
            const int cph = fdiv(cls, p.q_st), cpw = cls - cph * g.ST;
            bbase[pp] = (size_t)bkk[pp] * g.R * g.S * g.C + (size_t)(cph * g.S + cpw) * g.C + ci;
        } else {
            bbase[pp] = (size_t)bkk[pp] * g.R * g.S * g.C + n0 + bq4[pp];
        }
    }
    struct Slot {
        float4 a[NPA], h[NPA], b[NPB];
    };
    const bool has_mask = p.dymask != nullptr;
    auto a_on = [&](int pp) { return BM * 8 >= 256 * (pp + 1) || wave * 64 + 256 * pp < BM * 8; };
    auto b_on = [&](int pp) { return 32 * QPR >= 256 * (pp + 1) || wave * 64 + 256 * pp < 32 * QPR; };
    int f_c = 0, f_co0 = 0, f_tb = tb_lo, f_t2 = t2_lo;
    auto fetch = [&](int c, Slot &sl) {
        if (c > f_c) {
            f_c = c;
            f_co0 += KC;
            if (f_co0 >= g.Cout) {
                f_co0 = 0;
                if (++f_t2 > t2_hi) {
                    f_t2 = t2_lo;
                    ++f_tb;
                }
            }
        }
        const size_t adelta = (size_t)f_co0 - (size_t)(f_tb * g.OW + f_t2) * g.Cout;
#pragma unroll
        for (int pp = 0; pp < NPA; ++pp) {
            if (!a_on(pp)) continue;
            const size_t off = abase[pp] + adelta;     // (rows past N read image 0: zeroed when parked)
            sl.a[pp] = ldg4(p.dy + off);
            if (has_mask) sl.h[pp] = ldg4(p.dymask + off);
        }
        const size_t bdelta = ((size_t)(f_co0 * g.R + f_tb * (CLS ? g.ST : 1)) * g.S +
                               f_t2 * (CLS ? g.ST : 1)) * g.C;
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp)
            if (b_on(pp)) sl.b[pp] = ldg4(p.w + bbase[pp] + bdelta);
    };
    auto stash = [&](int buf, int c, const Slot &sl) {
#pragma unroll
        for (int pp = 0; pp < NPA; ++pp) {
            if (!a_on(pp)) continue;
            const int f = tid + 256 * pp, row = f >> 3, q = f & 7;
            float4 v = zero_unless(sl.a[pp], aok[pp]);
            if (has_mask) v = relu_mask(v, sl.h[pp]);
            *reinterpret_cast<float4 *>(&As[buf][row * LDR + 4 * q]) = v;
        }
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp)
            if (b_on(pp)) *reinterpret_cast<float4 *>(&Bs[buf][bkk[pp] * LDB + bq4[pp]]) = sl.b[pp];
    };
    f32x4 acc[AM][AN][P];
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) acc[am][an][pp] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 2 / WK; ++s)
            mma_sub<AM, AN, P, true, false, LDR, LDB>(As[buf], Bs[buf], wm * 16 * AM, wn * 16 * AN,
                                                           WK == 2 ? wk : s, lane, acc);
    };
    run_pipeline<Slot, G>(0, nchunks, fetch, stash, compute);
    fold_acc<AM, AN, P, WK, WM * WN>(acc, red, wm * WN + wn, wk, lane);
    // NHWC rows through LDS: a thread masks and stores a float4 of one image's pixel (a, a2)
    constexpr int LDT = BN + 4;
    static_assert(BM * LDT <= G * BM * LDR + G * 32 * LDB, "the tile fits in the chunk buffers");
    float *tile = &As[0][0];
    __syncthreads();
    if (wk == 0) {
#pragma unroll
        for (int am = 0; am < AM; ++am)
#pragma unroll
            for (int an = 0; an < AN; ++an)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    tile[(wm * 16 * AM + 16 * am + 4 * (lane >> 4) + reg) * LDT + wn * 16 * AN + 16 * an +
                         (lane & 15)] = acc[am][an][0][reg];
    }
    __syncthreads();
    constexpr int Q = BN / 4;
#pragma unroll
    for (int f = tid; f < BM * Q; f += 256) {
        const int row = f / Q, c4 = 4 * (f - row * Q);
        const int n = img_blk * BM + row;
        if (n >= g.N) continue;
        int ih = a, iw = a2, ec = n0 + c4;
        if (CLS) {
            const int cls = fdiv(ec, p.q_c);
            ec -= cls * g.C;
            const int eph = fdiv(cls, p.q_st);
            ih = a * g.ST + eph;
            iw = a2 * g.ST + (cls - eph * g.ST);
        }
        const size_t o = ((size_t)(n * g.H + ih) * g.W + iw) * g.C + ec;
        float4 v = *reinterpret_cast<const float4 *>(&tile[row * LDT + c4]);
        if (p.aprev != nullptr) v = relu_mask(v, *reinterpret_cast<const float4 *>(p.aprev + o));
        *reinterpret_cast<float4 *>(p.dx + o) = v;
    }
}

template <int BM, int BN, int WM, int WN, int WK, int G, bool CLS = false>
__global__ __launch_bounds__(256) void k_conv_dgrad_pos(DgradArgs p, int nblk, int npos) {
    __shared__ __attribute__((aligned(16))) float smem[dgrad_smem(BM, BN, WM, WN, WK, G)];
    // workgroup -> (image block, position): all positions of an image block on one XCD
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int sup = slot / npos, pos = slot - sup * npos;
    const int img_blk = sup * 8 + xcd;
    if (img_blk >= nblk) return;
    dgrad_pos_body<BM, BN, WM, WN, WK, G, CLS>(p, img_blk, pos, blockIdx.y, smem);
}

template <int BM, int BN, int WM, int WN, int WK, int G>
__global__ __launch_bounds__(256) void k_conv_dgrad2(DgradArgs p0, DgradArgs p1) {
    __shared__ __attribute__((aligned(16))) float smem[dgrad_smem(BM, BN, WM, WN, WK, G)];
    dgrad_body<BM, BN, WM, WN, WK, G>(blockIdx.z == 0 ? p0 : p1, blockIdx.x, blockIdx.y, 0, smem);
}

// ---------------------------------------------------------------------------------
// wgrad: dw[co][k] and db[co], reduction over m split across grid.z
// ---------------------------------------------------------------------------------
struct WgradArgs {
    const float *dy, *dymask, *x;
    float *dw, *db;
    long long dw_stride, db_stride;   // between the partials of consecutive splits
    ConvGeom g;
    int M, K, cps;
    const float *x2 = nullptr;   // (TAIL only) second input tensor, as in FwdArgs
    int K1 = 0;
    // (U8 only) the layer input as u8 NHWC4 pixels (x is unused), as in FwdArgs
    const uint8_t *xu8 = nullptr;
    float u8_r = 1.f, u8_d = 1.f;
    // how the large tile programs turn an output row m into its patch offset (see wgrad_body):
    // 0 two divisions per piece and chunk, 1 linear layer (the row itself), 2 table + carried (image, pixel)
    int pix_mode = 0;
};

// TAIL: linear layers (1x1) whose in_features are not a multiple of 32 / whose rows are not
// 16-byte aligned: the x loads are scalar, clamped to the row, the overhang zeroed.
// U8: the layer input is u8 NHWC4 pixels (see u8_over); C = 4, so a float4 piece is one dword.
template <int BI, int BJ, int WM, int WN, int WK, int G, bool TAIL = false, bool U8 = false>
__device__ __forceinline__ void wgrad_body(const WgradArgs &p, const int bx, const int by,
                                           const int bz, float *smem) {
    static_assert(WM * WN * WK == 4 && WK <= 2, "four waves");
    constexpr int AM = BI / (16 * WM), AN = BJ / (16 * WN);
    constexpr int P = (AM * AN == 1) ? 2 : 1;
    constexpr int QA = BI / 4, NPA = (32 * QA + 255) / 256, LDA = BI + 4;
    constexpr int QB = BJ / 4, NPB = (32 * QB + 255) / 256, LDB = BJ + 4;
    float(*As)[32 * LDA] = reinterpret_cast<float(*)[32 * LDA]>(smem);
    float(*Bs)[32 * LDB] = reinterpret_cast<float(*)[32 * LDB]>(smem + G * 32 * LDA);
    f32x4 *red = reinterpret_cast<f32x4 *>(smem + G * 32 * LDA + G * 32 * LDB);
    // (the wave index in an SGPR: everything decided per wave -- which K sub-chunk it multiplies, which
    // operand piece it loads -- becomes a scalar branch instead of an exec-masked one; behind exec-masked
    // branches the compiler emitted ds_read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs per chunk, 258 clocks each)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int i0 = bx * BI, j0 = by * BJ;
    const ConvGeom g = p.g;
    const int ohow = g.OH * g.OW, SC = g.S * g.C, WC = g.W * g.C;
    const int nch_total = (p.M + KC - 1) / KC;
    const int c0 = bz * p.cps;
    const int c1 = min(c0 + p.cps, nch_total);

    int bkk[NPB], bcol[NPB], bq[NPB];
#pragma unroll
    for (int pp = 0; pp < NPB; ++pp) {
        const int f = tid + 256 * pp;
        bkk[pp] = f / QB;
        bq[pp] = f - bkk[pp] * QB;
        const int j = j0 + 4 * bq[pp];
        const int r = fdiv(j, g.q_sc);
        bcol[pp] = r * WC + (j - r * SC);
    }
    // Large tile programs (rollout / update sized): the row -> patch offset of the x loader without
    // divisions in the chunk loop.  The SQ counters put these programs at 2.6 vector instructions per
    // MFMA and, on gfx950, those ADD to the MFMA time; two FastDiv splits per piece and
    // chunk were a third of them.  Mode 2: the offset of pixel (oh, ow) inside an image comes from a
    // table in LDS (OH * OW <= 400 entries, filled once), and every piece carries (image, pixel) of its
    // row from chunk to chunk -- a row advances by 32 per chunk, 32 <= OH * OW, so one conditional wrap.
    // Mode 1 (linear layers): the offset is the row itself.  Same addresses, same values.
    constexpr bool PIXCAP = !TAIL && BI * BJ > 32 * 32;
    __shared__ int pixtab[PIXCAP ? 400 : 1];
    const int pix_mode = PIXCAP ? p.pix_mode : 0;
    const int HWC = g.H * g.W * g.C;
    int pn[NPB], prem[NPB], p_c = c0;
    if (PIXCAP && pix_mode == 2) {
        for (int e = tid; e < ohow; e += 256) {
            const int oh = fdiv(e, g.q_ow), ow = e - oh * g.OW;
            pixtab[e] = (oh * g.ST * g.W + ow * g.ST) * g.C;
        }
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp) {
            const int m = c0 * KC + bkk[pp];
            pn[pp] = fdiv(m, g.q_ohow);
            prem[pp] = m - pn[pp] * ohow;
        }
        __syncthreads();
    }
    struct Slot {
        float4 a[NPA], h[NPA], b[NPB];
        bool ok[NPA];
    };
    const bool has_mask = p.dymask != nullptr;
    // operand pieces smaller than the workgroup belong to the first waves only (scalar branches)
    auto a_on = [&](int pp) { return 32 * QA >= 256 * (pp + 1) || wave * 64 + 256 * pp < 32 * QA; };
    auto b_on = [&](int pp) { return 32 * QB >= 256 * (pp + 1) || wave * 64 + 256 * pp < 32 * QB; };
    int akk[NPA], acol[NPA];
#pragma unroll
    for (int pp = 0; pp < NPA; ++pp) {
        const int f = tid + 256 * pp;
        akk[pp] = f / QA;
        acol[pp] = i0 + 4 * (f - akk[pp] * QA);
    }
    auto fetch = [&](int c, Slot &sl) {
        const int mbase = c * KC;
#pragma unroll
        for (int pp = 0; pp < NPA; ++pp) {
            if (!a_on(pp)) continue;
            const int m = mbase + akk[pp];
            const bool ok = m < p.M && acol[pp] < g.Cout;
            const size_t off = ok ? (size_t)m * g.Cout + acol[pp] : (size_t)0;
            sl.a[pp] = ldg4(p.dy + off);
            if (has_mask) sl.h[pp] = ldg4(p.dymask + off);
            sl.ok[pp] = ok;
        }
        if (PIXCAP && pix_mode == 2 && c > p_c) {
            // (chunks are asked for in non-decreasing order, one step at a time: run_pipeline)
            p_c = c;
#pragma unroll
            for (int pp = 0; pp < NPB; ++pp) {
                prem[pp] += KC;
                const bool wrap = prem[pp] >= ohow;
                prem[pp] -= wrap ? ohow : 0;
                pn[pp] += wrap ? 1 : 0;
            }
        }
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp) {
            if (!b_on(pp)) continue;
            const int m = mbase + bkk[pp];
            const int mm = m < p.M ? m : 0;
            if (PIXCAP && pix_mode != 0) {
                size_t at;
                if (pix_mode == 1) {
                    at = (size_t)mm * g.C + bcol[pp];
                } else {
                    const bool in = m < p.M;        // (rows past M read row 0, as below)
                    at = (size_t)(in ? pn[pp] : 0) * HWC + pixtab[in ? prem[pp] : 0] + bcol[pp];
                }
                if (U8)
                    sl.b[pp].x = __uint_as_float(*reinterpret_cast<const uint32_t *>(p.xu8 + at));
                else
                    sl.b[pp] = ldg4(p.x + at);
                continue;
            }
            if (TAIL) {
                const int K1 = p.x2 != nullptr ? p.K1 : p.K;
                const float *row = p.x + (size_t)mm * K1;
                const float *row2 = p.x2 != nullptr ? p.x2 + (size_t)mm * (p.K - K1) : row;
                const int j = j0 + 4 * bq[pp], kl = p.K - 1;
                float e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = min(j + u, kl);
                    e[u] = *(i < K1 ? row + i : row2 + (i - K1));
                }
                sl.b[pp] = make_float4(e[0], e[1], e[2], e[3]);
                continue;
            }
            const int n = fdiv(mm, g.q_ohow), rem = mm - n * ohow;
            const int oh = fdiv(rem, g.q_ow), ow = rem - oh * g.OW;
            const size_t at = ((size_t)(n * g.H + oh * g.ST) * g.W + ow * g.ST) * g.C + bcol[pp];
            if (U8)
                sl.b[pp].x = __uint_as_float(*reinterpret_cast<const uint32_t *>(p.xu8 + at));
            else
                sl.b[pp] = ldg4(p.x + at);
        }
    };
    auto stash = [&](int buf, int c, const Slot &sl) {
#pragma unroll
        for (int pp = 0; pp < NPA; ++pp) {
            if (!a_on(pp)) continue;
            // rows past M are zeroed in dy only: the x rows beside them are real rows (row 0),
            // and 0 * x adds nothing to the sum over m
            float4 v = zero_unless(sl.a[pp], sl.ok[pp]);
            if (has_mask) v = relu_mask(v, sl.h[pp]);
            *reinterpret_cast<float4 *>(&As[buf][akk[pp] * LDA + (acol[pp] - i0)]) = v;
        }
#pragma unroll
        for (int pp = 0; pp < NPB; ++pp) {
            if (!b_on(pp)) continue;
            float4 v = sl.b[pp];
            if (U8) v = u8x4_over(__float_as_uint(sl.b[pp].x), p.u8_r, p.u8_d);
            if (TAIL) {
                const int left = p.K - (j0 + 4 * bq[pp]);
                v.x = left > 0 ? v.x : 0.f; v.y = left > 1 ? v.y : 0.f;
                v.z = left > 2 ? v.z : 0.f; v.w = left > 3 ? v.w : 0.f;
            }
            *reinterpret_cast<float4 *>(&Bs[buf][bkk[pp] * LDB + 4 * bq[pp]]) = v;
        }
    };

    f32x4 acc[AM][AN][P];
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
#pragma unroll
            for (int pp = 0; pp < P; ++pp) acc[am][an][pp] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool do_bias = by == 0 && p.db != nullptr;

    auto compute = [&](int buf) {
        // (WK = 2: this wave's K half is an LDS address offset, not a branch)
#pragma unroll
        for (int s = 0; s < 2 / WK; ++s)
            mma_sub<AM, AN, P, false, false, LDA, LDB>(As[buf], Bs[buf], wm * 16 * AM, wn * 16 * AN,
                                                            WK == 2 ? wk : s, lane, acc);
        if (do_bias && tid < BI) {
#pragma unroll 8
            for (int kk = 0; kk < 32; ++kk) bsum += As[buf][kk * LDA + tid];
        }
    };
    run_pipeline<Slot, G>(c0, c1, fetch, stash, compute);
    if (do_bias && tid < BI && i0 + tid < g.Cout)
        p.db[(size_t)bz * p.db_stride + i0 + tid] = bsum;
    fold_acc<AM, AN, P, WK, WM * WN>(acc, red, wm * WN + wn, wk, lane);
    if (wk != 0) return;
    float *dw = p.dw + (size_t)bz * p.dw_stride;
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an) {
            const int j = j0 + wn * 16 * AN + 16 * an + (lane & 15);
            if (j >= p.K) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = i0 + wm * 16 * AM + 16 * am + 4 * (lane >> 4) + reg;
                if (co < g.Cout) dw[(size_t)co * p.K + j] = acc[am][an][0][reg];
            }
        }
}

template <int BI, int BJ, int WM, int WN, int WK, int G, bool TAIL = false, bool U8 = false>
__global__ __launch_bounds__(256) void k_conv_wgrad(WgradArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[wgrad_smem(BI, BJ, WM, WN, WK, G)];
    wgrad_body<BI, BJ, WM, WN, WK, G, TAIL, U8>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// ---------------------------------------------------------------------------------
// Weight gradient of the first Nature convolution on u8 pixels, direct form (see k_conv1_u8_direct).
// The tile program <32, 256, ..., U8> rebuilds the 32 x 256 patch matrix of every chunk of 32
// output pixels in LDS: each input pixel is loaded, converted and parked four times.  Here a
// workgroup (one split-K slab = a range of chunks, as before) keeps a RING of converted input
// rows -- indexed by the global row number (image * 84 + row) mod 24, every row staged once as the
// chunks walk down the images -- and the MFMA B fragments (16 consecutive patch elements of one
// kernel row = 4 pixels x 4 channels, contiguous in the input row) are read straight from it;
// rows 0..3 of the ring are mirrored behind row 23 so that the 8 kernel rows of a patch are
// always consecutive (immediate offsets).  Within a row, bit 4 of the float offset is flipped by
// bit 6: the lanes of a fragment whose pixels lie 4 output columns apart (64 floats) then hit
// different banks.  dy chunks are parked as in the tile program.  Per element of dW the terms
// enter the accumulator in the tile program's order (chunk, half chunk, t, and 4 k + t inside
// the MFMA), the bias gradient in pixel order: bit-identical slabs.
// ---------------------------------------------------------------------------------
// (a row holds 84 pixels and the bit flip sends pixels 80..83 to 84..87; 96 pixels = 384 floats per row
// keep bits 6..7 of a byte offset inside the row, which is what compute() flips)
constexpr int W1_RING = 24, W1_ROWS = 28, W1_RSF = 384, W1_LDA = 36, W1_NQ = 4;

__global__ __launch_bounds__(256, 3) void k_conv1_u8_wgrad_direct(
    const float *__restrict__ dy, const float *__restrict__ dymask, const uint32_t *__restrict__ x,
    float *__restrict__ dw, float *__restrict__ db, long long dw_stride, long long db_stride, int N,
    int M, int cps, float u8_r, float u8_d) {
    __shared__ __attribute__((aligned(16))) float ring[W1_ROWS * W1_RSF];
    __shared__ __attribute__((aligned(16))) float As[2][32 * W1_LDA];
    // per chunk parity and pixel group (half chunk, kq): byte offset in the ring of the group's first
    // patch column, see group_entry()
    __shared__ int gtab[2][8];
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const int nch = (M + 31) / 32;
    const int c0 = blockIdx.x * cps, c1 = min(c0 + cps, nch);
    if (c0 >= c1) return;
    const int last_row = N * D1_HW - 1;
    const bool has_mask = dymask != nullptr;
    // first input row (global numbering) of the patches of global output row G = pixel / 20
    auto patch_row = [](int G) { return 4 * (G + G / D1_O); };
    // rows [.., need(c)) must be staged before chunk c is multiplied
    auto need = [&](int c) { return patch_row(min(32 * c + 31, M - 1) / D1_O) + 8; };

    // this thread's dwords of a run of rows: dword d = tid + 256 q -> row d / 84, pixel d % 84
    int rr[W1_NQ], rpx[W1_NQ], rdst[W1_NQ];
#pragma unroll
    for (int q = 0; q < W1_NQ; ++q) {
        const int d = tid + 256 * q;
        rr[q] = d / D1_HW;
        rpx[q] = d - D1_HW * rr[q];
        rdst[q] = 4 * (rpx[q] ^ (((rpx[q] >> 4) & 1) << 2));
    }
    uint32_t raw[W1_NQ];
    auto fetch_rows = [&](int lo) {      // (rows past the run are loaded too, clamped: never parked)
#pragma unroll
        for (int q = 0; q < W1_NQ; ++q)
            raw[q] = x[(size_t)min(lo + rr[q], last_row) * D1_HW + rpx[q]];
    };
    auto stash_rows = [&](int lo, int hi) {
        const int lo24 = lo % W1_RING, nd = (hi - lo) * D1_HW;
#pragma unroll
        for (int q = 0; q < W1_NQ; ++q) {
            if (tid + 256 * q >= nd) continue;
            int idx = lo24 + rr[q];
            idx -= idx >= W1_RING ? W1_RING : 0;
            const float4 v = u8x4_over(raw[q], u8_r, u8_d);
            *reinterpret_cast<float4 *>(&ring[idx * W1_RSF + rdst[q]]) = v;
            if (idx < W1_ROWS - W1_RING)
                *reinterpret_cast<float4 *>(&ring[(idx + W1_RING) * W1_RSF + rdst[q]]) = v;
        }
    };
    // dy chunk: thread -> (pixel tid / 8, channels 4 (tid % 8) ..)
    float4 dv, dh;
    bool dok;
    auto fetch_dy = [&](int c) {
        const int m = 32 * c + (tid >> 3);
        dok = m < M;
        const size_t off = dok ? (size_t)m * 32 + 4 * (tid & 7) : (size_t)0;
        dv = ldg4(dy + off);
        if (has_mask) dh = ldg4(dymask + off);
    };
    auto stash_dy = [&](int buf) {
        float4 v = zero_unless(dv, dok);
        if (has_mask) v = relu_mask(v, dh);
        *reinterpret_cast<float4 *>(&As[buf][(tid >> 3) * W1_LDA + 4 * (tid & 7)]) = v;
    };

    f32x4 acc[2][4];
#pragma unroll
    for (int am = 0; am < 2; ++am)
#pragma unroll
        for (int an = 0; an < 4; ++an) acc[am][an] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool do_bias = db != nullptr;

    // Group g = 4 sc + kq of chunk c: the four pixels 32 c + 4 g .. + 3 lie in ONE output row
    // (20 % 4 == 0); pixels past M (a ragged last chunk) read the last pixel group: their dy is zero.
    // The entry is the ring byte offset of patch column block u = ow0 of kernel row 0, with the
    // block's flip bit f = (ow0 / 4) & 1 already applied: block ow0 + t is then the entry ^ 64 t for
    // t < 4, and block ow0 + 4 (flip bit !f) is (entry ^ 64) + 256.  Eight threads of wave 1 fill the
    // table of the NEXT chunk while the others park its operands: two divisions per chunk instead
    // of two per lane (vector instructions are not hidden behind the MFMAs: they add to them).
    auto group_entry = [&](int c, int g) {
        const int px0 = min(32 * c + 4 * g, M - 4);
        const int G = px0 / D1_O, ow0 = px0 - D1_O * G;
        return 4 * ((patch_row(G) % W1_RING) * W1_RSF + 16 * ow0 + 16 * ((ow0 >> 2) & 1));
    };
    const int lane_off = 4 * i + 4 * 2 * wn * W1_RSF;       // this lane's column, this wave's kernel rows
    const char *ringb = reinterpret_cast<const char *>(ring);
    auto compute = [&](int c) {
        const float *Ab = As[c & 1];
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
            const int P = gtab[c & 1][4 * sc + kq] + lane_off;
            const int blk[5] = {P, P ^ 64, P ^ 128, P ^ 192, (P ^ 64) + 256};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *b0 = reinterpret_cast<const float *>(ringb + blk[t]);
                const float *b1 = reinterpret_cast<const float *>(ringb + blk[t + 1]);
                const float *ap = Ab + (16 * sc + 4 * kq + t) * W1_LDA + i;
                const float a0 = ap[0], a1 = ap[16];
                const float b[4] = {b0[0], b1[0], b0[W1_RSF], b1[W1_RSF]};
#pragma unroll
                for (int am = 0; am < 2; ++am)
#pragma unroll
                    for (int an = 0; an < 4; ++an)
                        acc[am][an] = __builtin_amdgcn_mfma_f32_16x16x4f32(am ? a1 : a0, b[an],
                                                                           acc[am][an], 0, 0, 0);
            }
        }
        if (do_bias && tid < 32) {
#pragma unroll 8
            for (int kk = 0; kk < 32; ++kk) bsum += Ab[kk * W1_LDA + tid];
        }
    };

    // prologue: the rows of the first chunk (up to 20: two runs), its dy, and the loads of the second
    int staged = patch_row(32 * c0 / D1_O);
    for (const int n0 = need(c0); staged < n0;) {
        const int hi = min(staged + 12, n0);
        fetch_rows(staged);
        stash_rows(staged, hi);
        staged = hi;
    }
    fetch_dy(c0);
    stash_dy(c0 & 1);
    if (tid >= 64 && tid < 72) gtab[c0 & 1][tid - 64] = group_entry(c0, tid - 64);
    int lo1 = staged, hi1 = staged;
    if (c0 + 1 < c1) {
        hi1 = max(staged, need(c0 + 1));
        fetch_rows(lo1);
        fetch_dy(c0 + 1);
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
        if (c + 1 < c1) {
            stash_rows(lo1, hi1);
            stash_dy((c + 1) & 1);
            if (tid >= 64 && tid < 72) gtab[(c + 1) & 1][tid - 64] = group_entry(c + 1, tid - 64);
            staged = hi1;
        }
        if (c + 2 < c1) {
            lo1 = staged;
            hi1 = max(staged, need(c + 2));
            fetch_rows(lo1);
            fetch_dy(c + 2);
        }
        compute(c);
        __syncthreads();
    }
    float *dwp = dw + (size_t)blockIdx.x * dw_stride;
#pragma unroll
    for (int am = 0; am < 2; ++am)
#pragma unroll
        for (int an = 0; an < 4; ++an)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                dwp[(size_t)(16 * am + 4 * kq + reg) * 256 + 64 * wn + 16 * an + i] = acc[am][an][reg];
    if (do_bias && tid < 32) db[(size_t)blockIdx.x * db_stride + tid] = bsum;
}

template <int BI, int BJ, int WM, int WN, int WK, int G, bool TAIL>
__global__ __launch_bounds__(256) void k_conv_wgrad2(WgradArgs p0, WgradArgs p1, int nz) {
    __shared__ __attribute__((aligned(16))) float smem[wgrad_smem(BI, BJ, WM, WN, WK, G)];
    const bool second = (int)blockIdx.z >= nz;
    wgrad_body<BI, BJ, WM, WN, WK, G, TAIL>(second ? p1 : p0, blockIdx.x, blockIdx.y,
                                            second ? blockIdx.z - nz : blockIdx.z, smem);
}

// The LAST backward launch of a minibatch-sized update (the first layer's weight gradient: a few
// hundred latency-bound workgroups walking a long reduction) with optimizer blocks riding in it.
// By then the gradients of the layers above are final, and their parameters are not read again in
// this update; the hidden layer alone is 95 % of the example network's parameters, so most of the
// optimizer's 47 MB of streaming runs under this launch's load latency instead of in a launch of
// its own.  Workgroups [0, nw) run the weight-gradient tile program, the rest one RMSprop chunk
// (1 024 elements) each: the arithmetic of pfrl_rmsprop_step, element by element.
// (Tried and measured slower, 89.7 vs 87.2 us per update: the hidden layer riding WHOLE -- its
// weight-gradient tiles formed in this launch and stepped from the accumulators, so that the 6.4 MB
// gradient is never written or read.  The layer's own launch, left with the input gradient, got
// 1.6 us shorter; this one 3 us longer: 32 x 32 tiles touch parameter and state in 128-byte pieces
// 12.5 KB apart, against 1 KB contiguous per wave for the chunks below.)
constexpr int RIDE_MAX = 8;
constexpr int RIDE_CHUNK = 1024;
// A riding tensor's gradient is either a finished dense tensor (n_slabs == 0) or -- round 6 -- the
// split-K slabs a backward launch left (n_slabs >= 1, slab s at g + s * slab_stride): summed here
// in the order of k_rmsprop_fused / k_splitk_reduce (0 + slab 0 + slab 1 + ..., eight in flight),
// so that a step that rides is the step the optimizer launch would have made, bit for bit.
struct RideArgs {
    float *p[RIDE_MAX], *sq[RIDE_MAX], *ga[RIDE_MAX];
    const float *g[RIDE_MAX];
    long long numel[RIDE_MAX];
    long long slab_stride[RIDE_MAX];
    int n_slabs[RIDE_MAX];
    int block_end[RIDE_MAX];
    int n;
    float lr, alpha, eps, weight_decay;
};
struct WgradGrid {
    int nw, wgx, wgy;
    FastDiv q_wgx, q_wgy;
};

template <bool CENTERED>
__device__ __forceinline__ void ride_block(const RideArgs &r, const int b, const int tid) {
    int t = 0;
    while (t < r.n - 1 && b >= r.block_end[t]) ++t;
    const long long i0 = (long long)(b - (t == 0 ? 0 : r.block_end[t - 1])) * RIDE_CHUNK + 4 * tid;
    const long long n = r.numel[t];
    if (i0 >= n) return;
    const float oma = __fsub_rn(1.0f, r.alpha);
    const int S = r.n_slabs[t];
    const float *__restrict__ src = r.g[t] + i0;
    if (i0 + 4 <= n) {
        float4 gv;
        if (S == 0) {
            gv = *reinterpret_cast<const float4 *>(src);
        } else {
            const long long st = r.slab_stride[t];
            gv = make_float4(0.f, 0.f, 0.f, 0.f);
            int k = 0;
            for (; k + 8 <= S; k += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(src + (long long)(k + u) * st);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    gv.x = __fadd_rn(gv.x, v[u].x); gv.y = __fadd_rn(gv.y, v[u].y);
                    gv.z = __fadd_rn(gv.z, v[u].z); gv.w = __fadd_rn(gv.w, v[u].w);
                }
            }
            if (k < S) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = *reinterpret_cast<const float4 *>(src + (long long)min(k + u, S - 1) * st);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k + u < S) {
                        gv.x = __fadd_rn(gv.x, v[u].x); gv.y = __fadd_rn(gv.y, v[u].y);
                        gv.z = __fadd_rn(gv.z, v[u].z); gv.w = __fadd_rn(gv.w, v[u].w);
                    }
            }
        }
        float4 pv = *reinterpret_cast<float4 *>(r.p[t] + i0);
        float4 sv = *reinterpret_cast<float4 *>(r.sq[t] + i0);
        float4 mv = CENTERED ? *reinterpret_cast<float4 *>(r.ga[t] + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
        rms_update<CENTERED>(pv.x, gv.x, sv.x, mv.x, r.lr, r.alpha, oma, r.eps, r.weight_decay);
        rms_update<CENTERED>(pv.y, gv.y, sv.y, mv.y, r.lr, r.alpha, oma, r.eps, r.weight_decay);
        rms_update<CENTERED>(pv.z, gv.z, sv.z, mv.z, r.lr, r.alpha, oma, r.eps, r.weight_decay);
        rms_update<CENTERED>(pv.w, gv.w, sv.w, mv.w, r.lr, r.alpha, oma, r.eps, r.weight_decay);
        *reinterpret_cast<float4 *>(r.p[t] + i0) = pv;
        *reinterpret_cast<float4 *>(r.sq[t] + i0) = sv;
        if (CENTERED) *reinterpret_cast<float4 *>(r.ga[t] + i0) = mv;
        return;
    }
    // (tail of a tensor whose element count is not a multiple of four: a head's bias)
    for (long long i = i0; i < n; ++i) {
        float g = 0.f;
        if (S == 0) {
            g = r.g[t][i];
        } else {
            for (int k = 0; k < S; ++k) g = __fadd_rn(g, r.g[t][(long long)k * r.slab_stride[t] + i]);
        }
        float pi = r.p[t][i], si = r.sq[t][i], mi = CENTERED ? r.ga[t][i] : 0.f;
        rms_update<CENTERED>(pi, g, si, mi, r.lr, r.alpha, oma, r.eps, r.weight_decay);
        r.p[t][i] = pi;
        r.sq[t][i] = si;
        if (CENTERED) r.ga[t][i] = mi;
    }
}

template <int BI, int BJ, int WM, int WN, int WK, int G, bool CENTERED>
__global__ __launch_bounds__(256) void k_conv_wgrad_ride(int nw, WgradArgs p, WgradGrid wg, RideArgs r) {
    __shared__ __attribute__((aligned(16))) float smem[wgrad_smem(BI, BJ, WM, WN, WK, G)];
    const int b = blockIdx.x;
    if (b < nw) {
        const int q = fdiv(b, wg.q_wgx), bx = b - q * wg.wgx;
        const int bz = fdiv(q, wg.q_wgy);
        wgrad_body<BI, BJ, WM, WN, WK, G>(p, bx, q - bz * wg.wgy, bz, smem);
    } else {
        ride_block<CENTERED>(r, b - nw, threadIdx.x);
    }
}

// Input gradient and weight gradient of one layer in ONE launch: both consume the same dy
// and neither depends on the other.  At B = 32 each is a few hundred latency-bound
// workgroups that leave most of every CU idle; side by side they overlap almost fully and
// one launch boundary (and one cold start of the load pipeline) disappears per layer.  A
// fork / join inside a captured graph costs ~8 us on this stack, so the fusion is by grid:
// workgroups [0, n_dgrad) run the dgrad tile program, the rest the wgrad one.
// workgroup -> (program, tile) of the fused launch, with the divisors prepared on the host
struct BwdGrid {
    int nd, dgx, dgy, wgx, wgy;
    FastDiv q_dgx, q_dgy, q_wgx, q_wgy;
};
static BwdGrid make_bwd_grid(int dgx, int dgy, int dgz, int wgx, int wgy) {
    BwdGrid b;
    b.nd = dgx * dgy * dgz;
    b.dgx = dgx; b.dgy = dgy; b.wgx = wgx; b.wgy = wgy;
    b.q_dgx = fast_div((uint32_t)dgx); b.q_dgy = fast_div((uint32_t)dgy);
    b.q_wgx = fast_div((uint32_t)wgx); b.q_wgy = fast_div((uint32_t)wgy);
    return b;
}

template <int DM, int DN, int DWM, int DWN, int DWK, int DG_, int WI, int WWM, int WWN, int WWK>
// (nd first: a leading scalar is preloaded into an SGPR at wave launch -- the build passes
// -amdgpu-kernarg-preload-count -- so the program split needs no kernarg round trip of its own.
// Fetching the whole argument block with one vector load + v_readlane instead of the compiler's
// 2-3 dependent scalar batches was tried and measured 3 us slower per update.)
__global__ __launch_bounds__(256) void k_conv_bwd(int nd, DgradArgs d, WgradArgs w, BwdGrid bg) {
    // one LDS area, laid out by whichever tile program this workgroup runs
    __shared__ __attribute__((aligned(16))) float
        smem[cmax(dgrad_smem(DM, DN, DWM, DWN, DWK, DG_), wgrad_smem(WI, 32, WWM, WWN, WWK, 4))];
    const int b = blockIdx.x;
#ifdef PFRL_QNET_DEBUG
    // (tools/bwd_phase.py) 0 entry, 4 tile program (1 dgrad, 2 wgrad), 5 end; 1 2 6 7 from run_pipeline
    const int q_wg = blockIdx.x;
#endif
    QSTAMP(0);
    QMARK(4, b < nd ? 1ull : 2ull);
    if (b < nd) {
        const int r = fdiv(b, bg.q_dgx), bx = b - r * bg.dgx;
        const int bz = fdiv(r, bg.q_dgy);
        dgrad_body<DM, DN, DWM, DWN, DWK, DG_>(d, bx, r - bz * bg.dgy, bz, smem);
    } else {
        const int c = b - nd;
        const int r = fdiv(c, bg.q_wgx), bx = c - r * bg.wgx;
        const int bz = fdiv(r, bg.q_wgy);
        wgrad_body<WI, 32, WWM, WWN, WWK, 4>(w, bx, r - bz * bg.wgy, bz, smem);
    }
#ifdef PFRL_QNET_DEBUG
    __syncthreads();
#endif
    QSTAMP(5);
}

// k_conv_bwd with the optimizer steps of FINISHED tensors riding as a third kind of workgroup
// (round 6): blocks [nd + nw, grid) run ride_block.  By the time a layer's backward launch starts,
// the slabs of the layer ABOVE are complete and its parameters have been read for the last time in
// this update (by that layer's own input-gradient workgroups, one launch earlier), so its RMSprop
// step -- slab sums included -- needs no launch of its own.
template <int DM, int DN, int DWM, int DWN, int DWK, int DG_, int WI, int WWM, int WWN, int WWK, bool CENTERED>
__global__ __launch_bounds__(256) void k_conv_bwd_ride(int nd, int nw, DgradArgs d, WgradArgs w, BwdGrid bg,
                                                       RideArgs r) {
    __shared__ __attribute__((aligned(16))) float
        smem[cmax(dgrad_smem(DM, DN, DWM, DWN, DWK, DG_), wgrad_smem(WI, 32, WWM, WWN, WWK, 4))];
    const int b = blockIdx.x;
    if (b < nd) {
        const int q = fdiv(b, bg.q_dgx), bx = b - q * bg.dgx;
        const int bz = fdiv(q, bg.q_dgy);
        dgrad_body<DM, DN, DWM, DWN, DWK, DG_>(d, bx, q - bz * bg.dgy, bz, smem);
    } else if (b < nd + nw) {
        const int c = b - nd;
        const int q = fdiv(c, bg.q_wgx), bx = c - q * bg.wgx;
        const int bz = fdiv(q, bg.q_wgy);
        wgrad_body<WI, 32, WWM, WWN, WWK, 4>(w, bx, q - bz * bg.wgy, bz, smem);
    } else {
        ride_block<CENTERED>(r, b - nd - nw, threadIdx.x);
    }
}

template <int DM, int DN, int DWM, int DWN, int DWK, int DG_, int WI, int WWM, int WWN, int WWK>
__global__ __launch_bounds__(256) void k_conv_bwd2(DgradArgs d0, WgradArgs w0, DgradArgs d1, WgradArgs w1,
                                                   BwdGrid bg) {
    __shared__ __attribute__((aligned(16))) float
        smem[cmax(dgrad_smem(DM, DN, DWM, DWN, DWK, DG_), wgrad_smem(WI, 32, WWM, WWN, WWK, 4))];
    const bool second = blockIdx.y != 0;
    const int b = blockIdx.x;
    const int nd = bg.nd;
    if (b < nd) {
        const int by = fdiv(b, bg.q_dgx);
        dgrad_body<DM, DN, DWM, DWN, DWK, DG_>(second ? d1 : d0, b - by * bg.dgx, by, 0, smem);
    } else {
        const int c = b - nd;
        const int r = fdiv(c, bg.q_wgx), bx = c - r * bg.wgx;
        const int bz = fdiv(r, bg.q_wgy);
        wgrad_body<WI, 32, WWM, WWN, WWK, 4>(second ? w1 : w0, bx, r - bz * bg.wgy, bz, smem);
    }
}

// ---------------------------------------------------------------------------------
// split-K fold: out[e] = act(sum_s part[s][e] + bias[e % ncol]), several tensors per launch
// ---------------------------------------------------------------------------------
constexpr int RED_MAX = 12;
constexpr int RED_CHUNK = 1024;

struct RedArgs {
    const float *part[RED_MAX];
    float *out[RED_MAX];
    const float *bias[RED_MAX];
    // (NoisyNet layers) bias = bias + bias_sigma * f(bias_noise): the layer's sigma.b and the
    // out_features Gaussians of its draw, combined as pfrl_noisy_weights_fwd combines them
    const float *bias_sigma[RED_MAX], *bias_noise[RED_MAX];
    long long stride[RED_MAX];
    int n[RED_MAX], splits[RED_MAX], ncol[RED_MAX], relu[RED_MAX];
    int block_end[RED_MAX];
    int ntask;
};

__global__ __launch_bounds__(256) void k_splitk_reduce(RedArgs a) {
    int t = 0;
    const int b = blockIdx.x;
    while (t < a.ntask - 1 && b >= a.block_end[t]) ++t;
    const int first = t == 0 ? 0 : a.block_end[t - 1];
    const int e = (b - first) * RED_CHUNK + threadIdx.x * 4;
    const int n = a.n[t];
    if (e >= n) return;
    const float *part = a.part[t];
    const long long stride = a.stride[t];
    const int S = a.splits[t];
    if (e + 4 <= n) {
        // 8 partial slabs in flight per thread: the slabs were written by the previous
        // launch and come from memory, one round trip each if read one after the other
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 8 <= S; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = *reinterpret_cast<const float4 *>(part + (long long)(k + u) * stride + e);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
            }
        }
        {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = *reinterpret_cast<const float4 *>(part + (long long)min(k + u, S - 1) * stride + e);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k + u < S) {
                    s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
                }
        }
        if (a.bias[t] != nullptr) {
            const int col = e % a.ncol[t];   // ncol % 4 == 0: the four lanes stay in one row
            float4 bb = *reinterpret_cast<const float4 *>(a.bias[t] + col);
            if (a.bias_sigma[t] != nullptr) {
                const float4 bs = *reinterpret_cast<const float4 *>(a.bias_sigma[t] + col);
                const float *rn = a.bias_noise[t] + col;
                bb.x = __fadd_rn(bb.x, __fmul_rn(bs.x, noisy_shaped(rn[0])));
                bb.y = __fadd_rn(bb.y, __fmul_rn(bs.y, noisy_shaped(rn[1])));
                bb.z = __fadd_rn(bb.z, __fmul_rn(bs.z, noisy_shaped(rn[2])));
                bb.w = __fadd_rn(bb.w, __fmul_rn(bs.w, noisy_shaped(rn[3])));
            }
            s.x += bb.x; s.y += bb.y; s.z += bb.z; s.w += bb.w;
        }
        if (a.relu[t]) {
            s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
        }
        *reinterpret_cast<float4 *>(a.out[t] + e) = s;
    } else {
        for (int u = e; u < n; ++u) {
            float s = part[u];
            for (int k = 1; k < S; ++k) s += part[k * stride + u];
            if (a.bias[t] != nullptr) {
                const int col = u % a.ncol[t];
                float bb = a.bias[t][col];
                if (a.bias_sigma[t] != nullptr)
                    bb = __fadd_rn(bb, __fmul_rn(a.bias_sigma[t][col], noisy_shaped(a.bias_noise[t][col])));
                s += bb;
            }
            if (a.relu[t]) s = fmaxf(s, 0.f);
            a.out[t][u] = s;
        }
    }
}

// First level of a two-level fold: out[g][e] = sum of the slabs [g * per, min((g + 1) * per, S)) in
// slab order, for every group g in ONE launch (blockIdx.y = g).  A tensor with thousands of short
// slabs -- the first convolution's weight gradient at rollout size: 1 600 - 4 096 slabs of 33 KB --
// is otherwise folded by n / 1024 = 9 workgroups walking all slabs, eight at a time: 93 us for
// 52 MB at B = 2 048 (profiles/r06_ppo_rank_shape.txt); two levels are ~50 x 9 workgroups, then 9.
__global__ __launch_bounds__(256) void k_splitk_group(const float *__restrict__ part, long long stride, int n,
                                                      int S, int per, float *__restrict__ out,
                                                      long long out_stride) {
    const int e = blockIdx.x * RED_CHUNK + threadIdx.x * 4;
    if (e >= n) return;
    const int s0 = blockIdx.y * per, s1 = min(S, s0 + per);
    float *o = out + (long long)blockIdx.y * out_stride + e;
    if (e + 4 <= n) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = s0;
        for (; k + 8 <= s1; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = *reinterpret_cast<const float4 *>(part + (long long)(k + u) * stride + e);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
            }
        }
        if (k < s1) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = *reinterpret_cast<const float4 *>(part + (long long)min(k + u, s1 - 1) * stride + e);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k + u < s1) {
                    s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
                }
        }
        *reinterpret_cast<float4 *>(o) = s;
    } else {
        for (int u = e; u < n; ++u) {
            float acc = 0.f;
            for (int k = s0; k < s1; ++k) acc += part[(long long)k * stride + u];
            out[(long long)blockIdx.y * out_stride + u] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------
// narrow linear head (out_features <= 16), e.g. Linear(512, n_actions)
// ---------------------------------------------------------------------------------
constexpr int SMALL_N = 16;
constexpr int SMALL_BWD_N = 64;   // (the backward kernel also covers the 2 x action_size policy heads)

// y[m][n] = sum_k x[m][k] w[n][k] + b[n]; one workgroup per row, every load of a pass
// (4 k per thread: 4 of x, 4 N of w) issued before the first use
// (blockIdx.y picks one of up to two independent heads: the twin launch)
struct SmallFwdArgs {
    const float *x[2], *w[2], *bias[2];
    float *y[2];
};

// (the row product shared by k_linear_small_fwd and k_dqn_act_head: thread n < N returns
// sum_k x[m][k] w[n][k], other threads 0; `part` is the workgroup's [4][N] LDS scratch)
template <int N>
__device__ __forceinline__ float small_fwd_row(const float *__restrict__ x, const float *__restrict__ w,
                                               const int K, const int m, float (*part)[N]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 1024) {
        float xv[4], wv[N][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + tid + 256 * u;
            const int kk = k < K ? k : K - 1;
            xv[u] = x[(size_t)m * K + kk];
#pragma unroll
            for (int n = 0; n < N; ++n) wv[n][u] = w[(size_t)n * K + kk];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xs = (k0 + tid + 256 * u) < K ? xv[u] : 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = fmaf(xs, wv[n][u], acc[n]);
        }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float v = acc[n];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) part[wave][n] = v;
    }
    __syncthreads();
    if (tid < N) return ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
    return 0.f;
}

template <int N>
__global__ __launch_bounds__(256) void k_linear_small_fwd(SmallFwdArgs a, int K) {
    __shared__ float part[4][N];
    const float *__restrict__ bias = a.bias[blockIdx.y];
    float *__restrict__ y = a.y[blockIdx.y];
    const int m = blockIdx.x, tid = threadIdx.x;
    float v = small_fwd_row<N>(a.x[blockIdx.y], a.w[blockIdx.y], K, m, part);
    if (tid < N) {
        if (bias != nullptr) v += bias[tid];
        y[(size_t)m * N + tid] = v;
    }
}

// The Q head on the acting path + greedy action + the host's epsilon-greedy decision, one launch
// (see pfrl_dqn_act_head in include/pfrl_amd.h).  The action values are those of
// k_linear_small_fwd bit for bit; the first maximum wins, as numpy / torch argmax on the host.
template <int N>
__global__ __launch_bounds__(256) void k_dqn_act_head(const float *__restrict__ h, const float *__restrict__ w,
                                                      const float *__restrict__ bias,
                                                      const int32_t *__restrict__ choice, float *__restrict__ q,
                                                      int64_t *__restrict__ greedy, int64_t *__restrict__ action,
                                                      int K) {
    __shared__ float part[4][N];
    __shared__ float qrow[N];
    const int m = blockIdx.x, tid = threadIdx.x;
    // (requested before the row product: nothing below waits for it until the very end)
    const int ch = (tid == 0 && choice != nullptr) ? choice[m] : -1;
    float v = small_fwd_row<N>(h, w, K, m, part);
    if (tid < N) {
        if (bias != nullptr) v += bias[tid];
        qrow[tid] = v;
        if (q != nullptr) q[(size_t)m * N + tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0;
        float bv = qrow[0];
#pragma unroll
        for (int n = 1; n < N; ++n) {
            const float c = qrow[n];
            // (a NaN compares as the maximum, as in torch / numpy argmax)
            if (c > bv || (c != c && bv == bv)) {
                bv = c;
                best = n;
            }
        }
        if (greedy != nullptr) greedy[m] = best;
        if (action != nullptr) action[m] = ch >= 0 ? ch : best;
    }
}

// dx[m][k] = sum_n dy[m][n] w[n][k];  dw[n][k] = sum_m dy[m][n] x[m][k];  db[n] = sum_m dy[m][n]
// One launch, two kinds of workgroup.  The first ceil(K / 32) own dw (and db): 32 k columns
// x 8 slices of the batch per workgroup, each thread walks its rows m = slice, slice + 8, ..
// eight loads at a time, the slices are folded through LDS.  (A column per thread over the
// whole batch is M / 8 dependent memory round trips: 30 us at M = 256.)  The others own dx
// for 8 batch rows x 256 k columns each.
struct SmallBwdArgs {
    const float *dy[2], *x[2], *w[2];
    float *dx[2], *dw[2], *db[2];
};

// N = compile-time bound of the width (registers, unrolling, the LDS row stride), NR <= N the width
// itself: widths up to 16 are instantiated exactly, 17..64 (the 2 x action_size policy heads) in
// steps of 8 with the LDS copy of dy zero-padded to N columns -- so that no loop over n or over rows
// carries a branch (a guard per FMA made every LDS read wait for itself: 50 us for 256 x 34).
template <int N>
__global__ __launch_bounds__(256) void k_linear_small_bwd(SmallBwdArgs a, int M, int K, int n_dw, int NR) {
    extern __shared__ float sdy[];   // [M][N], columns >= NR zero
    const float *__restrict__ dy = a.dy[blockIdx.y];
    const float *__restrict__ x = a.x[blockIdx.y];
    const float *__restrict__ w = a.w[blockIdx.y];
    float *__restrict__ dx = a.dx[blockIdx.y];
    float *__restrict__ dw = a.dw[blockIdx.y];
    float *__restrict__ db = a.db[blockIdx.y];
    __shared__ float red[8][32][N + 1];
    const int tid = threadIdx.x;
    // Everything a workgroup needs from memory -- its share of dy for the LDS copy AND its own x rows
    // / w rows -- is requested before anything is waited for: after a kernel boundary a global load is
    // a ~2 us round trip here (the producer ran on other XCDs).
    const bool dw_role = (int)blockIdx.x < n_dw;
    constexpr int DYL = 40;            // M * NR <= M * N <= 10 240 = 256 threads x 40
    constexpr int XPRE = 4;            // x rows preloaded: 4 trips of 8 (M <= 256), the rest in the loop
    float dv[DYL];
    const int ndl = (M * NR + 255) / 256;   // (uniform)
    // (unconditional loads on clamped addresses, eight per uniform block: a load inside its own
    // branch gets its own s_waitcnt and the batch serialises)
#pragma unroll
    for (int g8 = 0; g8 < DYL / 8; ++g8)
        if (8 * g8 < ndl) {
#pragma unroll
            for (int u = 8 * g8; u < 8 * g8 + 8; ++u) dv[u] = dy[min(tid + 256 * u, M * NR - 1)];
        }
    const int kc = tid & 31, slice = tid >> 5;
    const int kx = (K + 255) / 256;
    const int r = dw_role ? 0 : blockIdx.x - n_dw;
    const int k = dw_role ? blockIdx.x * 32 + kc : (r % kx) * 256 + tid;
    const bool valid = k < K;
    const int kk = valid ? k : K - 1;
    float xv[XPRE][8], wk[N];
    if (dw_role) {
#pragma unroll
        for (int q = 0; q < XPRE; ++q)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = slice + 64 * q + 8 * u;
                xv[q][u] = x[(size_t)min(m, M - 1) * K + kk];
            }
    } else {
#pragma unroll
        for (int n = 0; n < N; ++n) wk[n] = w[(size_t)min(n, NR - 1) * K + kk];
#pragma unroll
        for (int n = 0; n < N; ++n) wk[n] = n < NR ? wk[n] : 0.f;
    }
#pragma unroll
    for (int g8 = 0; g8 < DYL / 8; ++g8)
        if (8 * g8 < ndl) {
#pragma unroll
            for (int u = 8 * g8; u < 8 * g8 + 8; ++u) {
                const int e = tid + 256 * u;
                if (e < M * NR) {
                    const int row = N == 1 ? e : e / NR;
                    sdy[row * N + (e - row * NR)] = dv[u];
                }
            }
        }
    if (NR < N) {      // (uniform) the pad columns
        const int pad = N - NR;
        for (int e = tid; e < M * pad; e += 256) {
            const int row = e / pad;
            sdy[row * N + NR + (e - row * pad)] = 0.f;
        }
    }
    __syncthreads();
    if (dw_role) {
        if (blockIdx.x == 0 && tid < NR && db != nullptr) {
            float s = 0.f;
            for (int m0 = 0; m0 < M; m0 += 8) {   // (loads first, then the adds in row order)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = sdy[min(m0 + u, M - 1) * N + tid];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (m0 + u < M) s += v[u];
            }
            db[tid] = s;
        }
        float gw[N];
#pragma unroll
        for (int n = 0; n < N; ++n) gw[n] = 0.f;
#pragma unroll
        for (int q = 0; q < XPRE; ++q) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = slice + 64 * q + 8 * u;
                const float xm = m < M ? xv[q][u] : 0.f;          // (rows past M add 0)
                const float *__restrict__ row = sdy + min(m, M - 1) * N;
#pragma unroll
                for (int n = 0; n < N; ++n) gw[n] = fmaf(row[n], xm, gw[n]);
            }
        }
        for (int m0 = slice + 64 * XPRE; m0 < M; m0 += 64) {
            float xl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xl[u] = x[(size_t)min(m0 + 8 * u, M - 1) * K + kk];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = m0 + 8 * u;
                const float xm = m < M ? xl[u] : 0.f;
                const float *__restrict__ row = sdy + min(m, M - 1) * N;
#pragma unroll
                for (int n = 0; n < N; ++n) gw[n] = fmaf(row[n], xm, gw[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < N; ++n) red[slice][kc][n] = gw[n];
        __syncthreads();
        // 32 columns x NR outputs folded by the threads, 256 at a time
        for (int e = tid; e < 32 * NR; e += 256) {
            const int c = e & 31, n = e >> 5;
            float s = red[0][c][n];
#pragma unroll
            for (int sl = 1; sl < 8; ++sl) s += red[sl][c][n];
            const int ko = blockIdx.x * 32 + c;
            if (ko < K) dw[(size_t)n * K + ko] = s;
        }
    } else {
        const int m0 = (r / kx) * 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + u;
            const float *__restrict__ row = sdy + min(m, M - 1) * N;
            float g = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) g = fmaf(row[n], wk[n], g);
            if (m < M && valid) dx[(size_t)m * K + k] = g;
        }
    }
}

// PFRL_QNET_FWD / _DGRAD / _WGRAD = tile program id: measurement hook of tools/layer_bench.py
// (-1 = the rule below decides).  Read per call; a launch is microseconds, getenv nanoseconds.
int prog_override(const char *name) {
    const char *e = getenv(name);
    return e != nullptr && *e ? atoi(e) : -1;
}

bool geom_ok(const ConvGeom &g) {
    return g.N > 0 && g.C > 0 && g.Cout > 0 && g.R > 0 && g.S > 0 && g.ST > 0 &&
           g.H >= g.R && g.W >= g.S && g.OH == (g.H - g.R) / g.ST + 1 &&
           g.OW == (g.W - g.S) / g.ST + 1 && (g.S * g.C) % KC == 0 && g.C % 4 == 0;
}

}  // namespace

// ===================================================================================
// C ABI
// ===================================================================================

// Batch size the forward tile programs are PLANNED for (0: the batch of the call).  A caller that
// evaluates a few rows of a large batch on their own -- PPO's V(next_state) for the rows that are
// not also a state of the rollout, pfrl_amd/agents/ppo.py::_next_values_exact -- sets it to the
// size of the large batch's chunks: every output row is then computed by the SAME tile program
// (same K order, same wave split) as in the large batch, i.e. bit for bit the same value, because
// no tile program mixes rows.  Per host thread; see pfrl_qnet_plan_images().
static thread_local int g_plan_images = 0;

extern "C" int pfrl_qnet_plan_images(int32_t images) {
    g_plan_images = images > 0 ? images : 0;
    return 0;
}

// rows the tile program is chosen for: the call's own, or the planned batch's if that is larger
static long long plan_rows(long long M, int images) {
    if (g_plan_images <= images || images <= 0) return M;
    return M / images * g_plan_images;
}

// tile program of the forward kernel for a problem (see the switch in pfrl_conv2d_nhwc_fwd)
static int fwd_program(const FwdArgs &a, int Cout, unsigned z, long long Mplan = 0) {
    const long long Mp = Mplan > 0 ? Mplan : a.M;
    auto blocks = [&](int bm, int bn) { return ((Mp + bm - 1) / bm) * ((Cout + bn - 1) / bn) * z; };
    const int force = prog_override("PFRL_QNET_FWD");
    int prog;
    if (Cout % 32 != 0) prog = blocks(64, 16) >= 512 ? 0 : 1;        // narrow outputs (16 channels)
    else if (Cout % 64 == 0 && a.K >= 2048 && blocks(128, 64) >= 1024) prog = 7;
    else if (Cout % 64 != 0 && blocks(128, 32) >= 16384) prog = 8;
    else if (Cout % 64 == 0 && blocks(64, 64) >= 1024) prog = 2;
    else if (blocks(64, 32) >= 1024) prog = 3;
    else if (blocks(32, 32) >= 384) prog = 4;
    else prog = a.cps >= 12 ? 5 : 6;   // few workgroups, long reduction: stages of 8 chunks
    if (force >= 0 && force <= 8) {
        const bool narrow = Cout % 32 != 0, wide = force == 2 || force == 7;
        if (narrow ? force <= 1 : (force >= 2 && (!wide || Cout % 64 == 0))) prog = force;
    }
    return prog;
}

// y = act(x W^T + b) of a factorised NoisyNet layer (pfrl/nn/noisy_linear.py:56-70) WITHOUT its
// perturbed weights ever existing: W = mu_w + sigma_w * (f(r_out) f(r_in)^T), b = mu_b + sigma_b *
// f(r_out), r = the layer's K + N unit Gaussians (inputs first), formed by the B-operand loader of
// the forward kernel / the bias of its epilogue.  Same tile programs, same summation order and the
// same roundings in W as pfrl_noisy_weights_fwd followed by pfrl_linear_fwd: bit-identical
// output.  Minibatch-sized problems (the programs of up to 32 x 32 tiles); K % 32 == 0.
// splits > 1 writes partials [splits][M][N] for pfrl_splitk_reduce_noisy.
extern "C" int pfrl_linear_noisy_fwd(const float *x, const float *mu_w, const float *sigma_w,
                                     const float *mu_b, const float *sigma_b, const float *r, float *y,
                                     int32_t M, int32_t K, int32_t N, int32_t relu, int32_t splits,
                                     void *stream) {
    PFRL_CHECK_ARG(M >= 1 && N >= 1 && K >= KC && K % KC == 0 && splits >= 1,
                   "pfrl_linear_noisy_fwd: in_features must be a multiple of 32");
    PFRL_CHECK_ARG(x && mu_w && sigma_w && r && y, "pfrl_linear_noisy_fwd: null pointer");
    const uintptr_t bits = (uintptr_t)x | (uintptr_t)mu_w | (uintptr_t)sigma_w | (uintptr_t)r;
    PFRL_CHECK_ARG((bits & 15) == 0, "pfrl_linear_noisy_fwd: 16-byte aligned operands");
    FwdArgs a;
    a.x = x; a.w = mu_w; a.bias = mu_b; a.y = y;
    a.w_sigma = sigma_w; a.noise = r; a.bias_sigma = sigma_b;
    a.g = make_geom(M, 1, 1, K, N, 1, 1, 1);
    a.M = M;
    a.K = K;
    const int nch = K / KC;
    a.cps = (nch + splits - 1) / splits;
    a.relu = relu; a.planar = 0; a.partial = splits > 1;
    PFRL_CHECK_ARG(a.partial || (mu_b != nullptr && sigma_b != nullptr),
                   "pfrl_linear_noisy_fwd: both biases required");
    hipStream_t st = (hipStream_t)stream;
    const unsigned z = (unsigned)splits;
#define FWDN(BM, BN, WM, WN, WK, G)                                                                  \
    hipLaunchKernelGGL((k_conv_fwd<BM, BN, WM, WN, WK, G, false, true>),                             \
                       dim3((M + BM - 1) / BM, (N + BN - 1) / BN, z), dim3(256), 0, st, a)
    switch (fwd_program(a, N, z)) {
        case 1: FWDN(32, 16, 2, 1, 2, 4); break;
        case 4: FWDN(32, 32, 2, 2, 1, 4); break;
        case 5: FWDN(16, 32, 1, 2, 2, 8); break;
        case 6: FWDN(16, 32, 1, 2, 2, 4); break;
        default:
            pfrl_set_error("pfrl_linear_noisy_fwd: problem outside the minibatch-sized tile programs");
            return PFRL_ERR_ARG;
    }
#undef FWDN
    PFRL_LAUNCH_CHECK();
}

// Two NoisyNet layers of one network side by side in ONE launch: the advantage and value streams of
// the distributional dueling head (pfrl/q_functions/dueling_dqn.py:93-118: a_stream on the first
// half of the hidden activations, v_stream on the second).  Problem t reads rows of x[t] with row
// stride x_row_stride (a multiple of K: the halves of h are addressed in place, no copy) and
// writes y[t] [M, N[t]].  Both problems must fall into the narrow-output tile program (N % 32 != 0,
// few workgroups), which is also what the single-layer entry picks for them: bit-identical to two
// pfrl_linear_noisy_fwd calls on contiguous copies of the halves.
extern "C" int pfrl_linear_noisy_fwd_pair(const float *const *x, int32_t x_row_stride,
                                          const float *const *mu_w, const float *const *sigma_w,
                                          const float *const *mu_b, const float *const *sigma_b,
                                          const float *const *r, float *const *y, int32_t M, int32_t K,
                                          const int32_t *N, int32_t relu, void *stream) {
    PFRL_CHECK_ARG(M >= 1 && K >= KC && K % KC == 0 && x_row_stride >= K && x_row_stride % K == 0,
                   "pfrl_linear_noisy_fwd_pair: in_features % 32, row stride a multiple of in_features");
    FwdArgs a[2];
    int ny = 0;
    for (int t = 0; t < 2; ++t) {
        PFRL_CHECK_ARG(x[t] && mu_w[t] && sigma_w[t] && mu_b[t] && sigma_b[t] && r[t] && y[t] && N[t] >= 1,
                       "pfrl_linear_noisy_fwd_pair: null pointer");
        const uintptr_t bits = (uintptr_t)x[t] | (uintptr_t)mu_w[t] | (uintptr_t)sigma_w[t] | (uintptr_t)r[t];
        PFRL_CHECK_ARG((bits & 15) == 0, "pfrl_linear_noisy_fwd_pair: 16-byte aligned operands");
        a[t].x = x[t]; a[t].w = mu_w[t]; a[t].bias = mu_b[t]; a[t].y = y[t];
        a[t].w_sigma = sigma_w[t]; a[t].noise = r[t]; a[t].bias_sigma = sigma_b[t];
        // rows of K floats at a stride of x_row_stride: a 1 x 1 convolution over [M, 1, W, K] that
        // steps over the W - 1 other pieces of each row
        const int Wd = x_row_stride / K;
        a[t].g = make_geom(M, 1, Wd, K, N[t], 1, 1, Wd);
        a[t].M = M;
        a[t].K = K;
        a[t].cps = K / KC;
        a[t].relu = relu; a[t].planar = 0; a[t].partial = 0;
        PFRL_CHECK_ARG(fwd_program(a[t], N[t], 1) == 1,
                       "pfrl_linear_noisy_fwd_pair: both layers must be narrow-output minibatch problems");
        ny = ny > (N[t] + 15) / 16 ? ny : (N[t] + 15) / 16;
    }
    hipLaunchKernelGGL((k_conv_fwd2<32, 16, 2, 1, 2, 4, false, true>), dim3((M + 31) / 32, ny, 2), dim3(256),
                       0, (hipStream_t)stream, a[0], a[1]);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_conv2d_nhwc_fwd(const float *x, const float *w, const float *bias, float *y,
                                    int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t R,
                                    int32_t S, int32_t stride, int32_t relu, int32_t planar_out,
                                    int32_t splits, void *stream) {
    const ConvGeom g = make_geom(N, H, W, C, Cout, R, S, stride);
    PFRL_CHECK_ARG(geom_ok(g), "pfrl_conv2d_nhwc_fwd: unsupported geometry (need S*C % 32 == 0)");
    PFRL_CHECK_ARG(splits >= 1, "pfrl_conv2d_nhwc_fwd: splits >= 1");
    FwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.g = g;
    a.M = N * g.OH * g.OW;
    a.K = R * S * C;
    const int nch = a.K / KC;
    a.cps = (nch + splits - 1) / splits;
    a.relu = relu; a.planar = planar_out; a.partial = splits > 1;
    PFRL_CHECK_ARG(a.partial || bias != nullptr, "pfrl_conv2d_nhwc_fwd: bias required");
    hipStream_t st = (hipStream_t)stream;
    const unsigned z = (unsigned)splits;
    {
        // the Nature first layer on the fp32 NHWC4 minibatch wherever a one-accumulator tile program
        // would run (the acting and target passes of the replay agents): the direct kernel, same bits
        const int prog = fwd_program(a, Cout, z, plan_rows(a.M, N));
        if ((prog == 3 || prog == 8) && splits == 1 && C == 4 && H == D1_HW && W == D1_HW && Cout == 32 &&
            R == 8 && S == 8 && stride == 4 && !planar_out && prog_override("PFRL_CONV1_DIRECT") != 0 &&
            (((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)y) & 15) == 0) {
            const int units = (N + 1) / 2 * 5, slots = conv1_direct_slots();
            hipLaunchKernelGGL(k_conv1_u8_direct<false>, dim3(units < slots ? units : slots), dim3(256), 0,
                               st, static_cast<const void *>(x), w, bias, y, N, relu, 1.f, 1.f, units);
            PFRL_LAUNCH_CHECK();
        }
    }
#define FWD(BM, BN, WM, WN, WK, G)                                                                   \
    hipLaunchKernelGGL((k_conv_fwd<BM, BN, WM, WN, WK, G>),                                          \
                       dim3((a.M + BM - 1) / BM, (Cout + BN - 1) / BN, z), dim3(256), 0, st, a)
    // Rollout- and update-sized batches (M in the millions of rows).  Measured on MI355X at B = 16384
    // (tools/layer_bench.py --sweep, profiles/r04_layer_sweep.txt): what separates the programs is the
    // instruction overhead per MFMA (per-chunk address arithmetic, LDS traffic, barriers: 2.0 VALU per
    // MFMA for 64 x 64, 3.1 for 64 x 32 by SQ_INSTS_*), not bytes: 128 x 32 beats 64 x 32 for the first
    // convolution (1241 vs 1436 us), 128 x 64 beats 64 x 64 only for the long reduction of the linear
    // layer (485 vs 535 us) and loses below ~1000 workgroups.
    switch (fwd_program(a, Cout, z, plan_rows(a.M, N))) {
        case 0: FWD(64, 16, 4, 1, 1, 2); break;
        case 1: FWD(32, 16, 2, 1, 2, 4); break;
        case 2: FWD(64, 64, 2, 2, 1, 2); break;
        case 3: FWD(64, 32, 2, 2, 1, 2); break;
        case 4: FWD(32, 32, 2, 2, 1, 4); break;
        case 5: FWD(16, 32, 1, 2, 2, 8); break;
        case 6: FWD(16, 32, 1, 2, 2, 4); break;
        case 7: FWD(128, 64, 2, 2, 1, 2); break;
        default: FWD(128, 32, 4, 1, 1, 2); break;
    }
#undef FWD
    PFRL_LAUNCH_CHECK();
}

// pfrl_conv2d_nhwc_fwd for an input of u8 NHWC4 pixels (C = 4: one dword per pixel, e.g. the four
// stacked 84 x 84 frames of an Atari observation as pfrl_batch_states_u8_raw_nhwc4 gathers them):
// y = act(conv(float32(x) / divisor, w) + b) with the division done where the operand enters LDS
// (u8_over) -- the same tile program, the same fp32 operands and the same summation order as the
// fp32 entry on the gathered fp32 minibatch, so the output is bit-identical, while the minibatch
// costs a quarter of the bytes to write and to read.  Rollout- / update-sized batches of layers
// with Cout % 32 == 0 and Cout % 64 != 0 (the 64 x 32, 32 x 32 and 128 x 32 programs); the caller
// has checked the divisor (ops.u8_division_exact).  No split-K.
extern "C" int pfrl_conv2d_u8nhwc4_fwd(const uint8_t *x, float divisor, const float *w,
                                       const float *bias, float *y, int32_t N, int32_t H, int32_t W,
                                       int32_t Cout, int32_t R, int32_t S, int32_t stride,
                                       int32_t relu, int32_t planar_out, void *stream) {
    const ConvGeom g = make_geom(N, H, W, 4, Cout, R, S, stride);
    PFRL_CHECK_ARG(geom_ok(g), "pfrl_conv2d_u8nhwc4_fwd: unsupported geometry (need S*4 % 32 == 0)");
    PFRL_CHECK_ARG(x != nullptr && bias != nullptr && divisor > 0.f,
                   "pfrl_conv2d_u8nhwc4_fwd: null input / bias, or divisor <= 0");
    FwdArgs a;
    a.x = nullptr; a.xu8 = x; a.u8_d = divisor; a.u8_r = 1.0f / divisor;
    a.w = w; a.bias = bias; a.y = y; a.g = g;
    a.M = N * g.OH * g.OW;
    a.K = R * S * 4;
    a.cps = a.K / KC;
    a.relu = relu; a.planar = planar_out; a.partial = 0;
    hipStream_t st = (hipStream_t)stream;
    const int prog = fwd_program(a, Cout, 1, plan_rows(a.M, N));
    // the Nature first layer wherever a one-accumulator tile program would run: the direct kernel
    // (same bits; PFRL_CONV1_DIRECT=0 keeps the tile programs, for the comparison tests)
    const int direct = prog_override("PFRL_CONV1_DIRECT");
    if (direct != 0 && (prog == 3 || prog == 8) && H == D1_HW && W == D1_HW && Cout == 32 && R == 8 &&
        S == 8 && stride == 4 && !planar_out && ((uintptr_t)x & 3) == 0 &&
        (((uintptr_t)w | (uintptr_t)bias | (uintptr_t)y) & 15) == 0) {
        const int units = (N + 1) / 2 * 5, slots = conv1_direct_slots();
        hipLaunchKernelGGL(k_conv1_u8_direct<true>, dim3(units < slots ? units : slots), dim3(256), 0, st,
                           static_cast<const void *>(x), w, bias, y, N, relu, a.u8_r, a.u8_d, units);
        PFRL_LAUNCH_CHECK();
    }
#define FWDU(BM, BN, WM, WN, WK, G)                                                                  \
    hipLaunchKernelGGL((k_conv_fwd<BM, BN, WM, WN, WK, G, false, false, true>),                      \
                       dim3((a.M + BM - 1) / BM, (Cout + BN - 1) / BN, 1), dim3(256), 0, st, a)
    switch (prog) {
        case 3: FWDU(64, 32, 2, 2, 1, 2); break;
        case 4: FWDU(32, 32, 2, 2, 1, 4); break;
        case 8: FWDU(128, 32, 4, 1, 1, 2); break;
        default:
            pfrl_set_error("pfrl_conv2d_u8nhwc4_fwd: problem outside the u8 tile programs "
                           "(Cout % 32 == 0, Cout % 64 != 0, >= 384 tiles of 32 x 32)");
            return PFRL_ERR_ARG;
    }
#undef FWDU
    PFRL_LAUNCH_CHECK();
}

// y = act(x w^T + b) for any in_features: the 1x1 case of the forward kernel, with the TAIL
// loaders when K is not a multiple of 32 (rows need no alignment then).  splits > 1 writes
// partials [splits][M][N] for pfrl_splitk_reduce, as pfrl_conv2d_nhwc_fwd does.
extern "C" int pfrl_linear_fwd(const float *x, const float *w, const float *bias, float *y, int32_t M,
                               int32_t K, int32_t N, int32_t relu, int32_t splits, void *stream) {
    PFRL_CHECK_ARG(M >= 1 && K >= 1 && N >= 1 && splits >= 1, "pfrl_linear_fwd: empty problem");
    if (K % KC == 0)
        return pfrl_conv2d_nhwc_fwd(x, w, bias, y, M, 1, 1, K, N, 1, 1, 1, relu, 0, splits, stream);
    FwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.g = make_geom(M, 1, 1, K, N, 1, 1, 1);
    a.M = M;
    a.K = K;
    const int nch = (K + KC - 1) / KC;
    a.cps = (nch + splits - 1) / splits;
    a.relu = relu; a.planar = 0; a.partial = splits > 1;
    PFRL_CHECK_ARG(a.partial || bias != nullptr, "pfrl_linear_fwd: bias required");
    hipStream_t st = (hipStream_t)stream;
    const unsigned z = (unsigned)splits;
#define FWDT(BM, BN, WM, WN, WK, G)                                                                  \
    hipLaunchKernelGGL((k_conv_fwd<BM, BN, WM, WN, WK, G, true>),                                    \
                       dim3((M + BM - 1) / BM, (N + BN - 1) / BN, z), dim3(256), 0, st, a)
    if (N % 32 != 0 && N <= 16) FWDT(32, 16, 2, 1, 2, 4);
    else if ((long long)((M + 31) / 32) * ((N + 31) / 32) * z >= 384) FWDT(32, 32, 2, 2, 1, 4);
    else FWDT(16, 32, 1, 2, 2, 4);
#undef FWDT
    PFRL_LAUNCH_CHECK();
}

static int make_dgrad_args(DgradArgs &a, const float *dy, const float *dy_mask, const float *w,
                           const float *a_prev, float *dx, int32_t N, int32_t H, int32_t W, int32_t C,
                           int32_t Cout, int32_t R, int32_t S, int32_t stride, int32_t perm_p,
                           int32_t perm_c) {
    const ConvGeom g = make_geom(N, H, W, C, Cout, R, S, stride);
    PFRL_CHECK_ARG(g.OH >= 1 && g.OW >= 1 && Cout % KC == 0 && C % 16 == 0 && R % stride == 0 &&
                       S % stride == 0 && H % stride == 0 && W % stride == 0,
                   "pfrl_conv2d_nhwc_bwd_data: unsupported geometry");
    PFRL_CHECK_ARG(perm_p == 0 || (H == 1 && W == 1 && perm_p * perm_c == C),
                   "pfrl_conv2d_nhwc_bwd_data: bad permutation");
    a.dy = dy; a.dymask = dy_mask; a.w = w; a.aprev = a_prev; a.dx = dx; a.g = g;
    a.AH = H / stride; a.AW = W / stride;
    a.Mc = N * a.AH * a.AW;
    a.TH = R / stride; a.TW = S / stride;
    a.K = a.TH * a.TW * Cout;
    a.permP = perm_p; a.permC = perm_c;
    a.q_ahw = fast_div((uint32_t)(a.AH * a.AW));
    a.q_aw = fast_div((uint32_t)a.AW);
    a.q_st = fast_div((uint32_t)stride);
    a.q_c = fast_div((uint32_t)C);
    a.q_permP = fast_div((uint32_t)(perm_p > 0 ? perm_p : 1));
    return 0;
}

static int make_wgrad_args(WgradArgs &a, const float *dy, const float *dy_mask, const float *x,
                           float *dw_part, float *db_part, int64_t dw_stride, int64_t db_stride,
                           int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t R,
                           int32_t S, int32_t stride, int32_t splits) {
    const ConvGeom g = make_geom(N, H, W, C, Cout, R, S, stride);
    PFRL_CHECK_ARG(geom_ok(g) && Cout % 16 == 0, "pfrl_conv2d_nhwc_bwd_weight: unsupported geometry");
    PFRL_CHECK_ARG(splits >= 1, "pfrl_conv2d_nhwc_bwd_weight: splits >= 1");
    a.dy = dy; a.dymask = dy_mask; a.x = x; a.dw = dw_part; a.db = db_part;
    a.dw_stride = dw_stride; a.db_stride = db_stride; a.g = g;
    a.M = N * g.OH * g.OW;
    a.K = R * S * C;
    const int nch = (a.M + KC - 1) / KC;
    a.cps = (nch + splits - 1) / splits;
    const int ohow = g.OH * g.OW;
    a.pix_mode = prog_override("PFRL_WGRAD_PIX") == 0 ? 0
                 : (H == 1 && W == 1 && R == 1 && S == 1 && stride == 1) ? 1
                 : (ohow >= KC && ohow <= 400) ? 2 : 0;
    return 0;
}

// tile program of the input-gradient kernel for a problem: 0 = <64,64> 1 = <64,32> (throughput
// tiles, stages of 2), 2 = <32,32,G4>, 3 = <16,32,G8>, 4 = <16,32,G4>, 5 = <32,16,G4>
static int dgrad_program(const DgradArgs &a, int z) {
    const int C = a.g.C;
    auto blocks = [&](int bm, int bn) { return (long long)((a.Mc + bm - 1) / bm) * (C / bn) * z; };
    if (C % 32 != 0) return 5;
    // (128-row programs measured SLOWER here at every size -- 188 VGPRs, two waves per SIMD -- and
    // were dropped: conv2 1580 vs 1343 us, conv3 1036 vs 960 us at B = 16384)
    const int force = prog_override("PFRL_QNET_DGRAD");
    const int st = a.g.ST, ncol = st * st * C;
    // 6 = <64,64,MERGE>, 7 = <64,128,MERGE>: the parity classes of a strided layer in one tile
    const bool merge64 = st > 1 && 64 % C == 0 && ncol % 64 == 0;
    const bool merge128 = st > 1 && 128 % C == 0 && ncol % 128 == 0;
    if (force >= 0 && force <= 4 && (C % 64 == 0 || force != 0)) return force;
    if ((force == 6 && merge64) || (force == 7 && merge128)) return force;
    // 8 = <64,64,POS>: stride-1 layers with more than one tap, tiled by input position
    const bool pos_ok = st == 1 && a.TH * a.TW > 1 && C % 64 == 0 && a.permP == 0;
    if (force == 8 && pos_ok) return 8;
    if (force < 0 && pos_ok && a.g.N >= 1024) return 8;
    // 9 = <64,64,POS,CLS>: strided layers by cell of the class grid, classes side by side
    const bool poscls_ok = merge64 && a.permP == 0;
    if (force == 9 && poscls_ok) return 9;
    if (force < 0 && poscls_ok && a.g.N >= 1024) return 9;
    // (64 x 64 measured ahead of 64 x 128: 1219 vs 1262 us against 1345 per class, conv2 at B = 16384)
    if (force < 0 && merge64 && (long long)((a.Mc + 63) / 64) * (ncol / 64) >= 2048) return 6;
    if (C % 64 == 0 && blocks(64, 64) >= 1024) return 0;
    if (blocks(64, 32) >= 1024) return 1;
    if (blocks(32, 32) >= 384) return 2;
    // few workgroups: 32 input channels per workgroup (whole 128 B lines of the weight rows),
    // long reductions in stages of 8 chunks
    return a.K / KC >= 12 ? 3 : 4;
}

extern "C" int pfrl_conv2d_nhwc_bwd_data(const float *dy, const float *dy_mask, const float *w,
                                         const float *a_prev, float *dx, int32_t N, int32_t H, int32_t W,
                                         int32_t C, int32_t Cout, int32_t R, int32_t S, int32_t stride,
                                         int32_t perm_p, int32_t perm_c, void *stream) {
    DgradArgs a;
    if (int rc = make_dgrad_args(a, dy, dy_mask, w, a_prev, dx, N, H, W, C, Cout, R, S, stride, perm_p,
                                 perm_c))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    const unsigned z = (unsigned)(stride * stride);
#define DG(BM, BN, WM, WN, WK, G)                                                                    \
    hipLaunchKernelGGL((k_conv_dgrad<BM, BN, WM, WN, WK, G>), dim3((a.Mc + BM - 1) / BM, C / BN, z), \
                       dim3(256), 0, st, a)
#define DGM(BM, BN, WM, WN, WK, G)                                                                   \
    hipLaunchKernelGGL((k_conv_dgrad<BM, BN, WM, WN, WK, G, true>),                                  \
                       dim3((a.Mc + BM - 1) / BM, (z * C) / BN, 1), dim3(256), 0, st, a)
    switch (dgrad_program(a, (int)z)) {
        case 8: {
            const int nblk = (N + 63) / 64, npos = a.AH * a.AW;
            hipLaunchKernelGGL((k_conv_dgrad_pos<64, 64, 2, 2, 1, 2>),
                               dim3((unsigned)(((nblk + 7) / 8) * 8 * npos), C / 64, 1), dim3(256), 0, st,
                               a, nblk, npos);
            break;
        }
        case 9: {
            const int nblk = (N + 63) / 64, npos = a.AH * a.AW;
            hipLaunchKernelGGL((k_conv_dgrad_pos<64, 64, 2, 2, 1, 2, true>),
                               dim3((unsigned)(((nblk + 7) / 8) * 8 * npos), (z * C) / 64, 1), dim3(256), 0,
                               st, a, nblk, npos);
            break;
        }
        case 6: DGM(64, 64, 2, 2, 1, 2); break;
        case 7: DGM(64, 128, 2, 2, 1, 2); break;
        case 0: DG(64, 64, 2, 2, 1, 2); break;
        case 1: DG(64, 32, 2, 2, 1, 2); break;
        case 2: DG(32, 32, 2, 2, 1, 4); break;
        case 3: DG(16, 32, 1, 2, 2, 8); break;
        case 4: DG(16, 32, 1, 2, 2, 4); break;
        default: DG(32, 16, 2, 1, 2, 4); break;
    }
#undef DGM
#undef DG
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_conv2d_nhwc_bwd_weight(const float *dy, const float *dy_mask, const float *x,
                                           float *dw_part, float *db_part, int64_t dw_stride,
                                           int64_t db_stride, int32_t N, int32_t H, int32_t W, int32_t C,
                                           int32_t Cout, int32_t R, int32_t S, int32_t stride,
                                           int32_t splits, void *stream) {
    WgradArgs a;
    if (int rc = make_wgrad_args(a, dy, dy_mask, x, dw_part, db_part, dw_stride, db_stride, N, H, W, C,
                                 Cout, R, S, stride, splits))
        return rc;
    hipStream_t st = (hipStream_t)stream;
    // From a few thousand rows up: large output tiles.  A 32 x 32 tile is ONE MFMA tile per wave, so
    // every chunk's loads, address arithmetic, LDS traffic and barriers are paid per 8 MFMAs (6.2 VALU
    // instructions per MFMA by SQ_INSTS_*); 64 x 128 is eight tiles per wave.  Measured (tools/
    // layer_bench.py --sweep): conv1 2062 -> 1290 us (32 x 256), conv2 1358 -> 934 (64 x 128), conv3
    // 844 -> 669 (64 x 64), linear 670 -> 552 at B = 16384; also ahead at B = 512 and 2048.  The
    // minibatch-sized launches of the replay agents (M < 16384 rows) keep the 32 x 32 program and with
    // it their bit-identity tests; the large programs sum in a different order (one accumulator
    // instead of two interleaved ones).
    // 0 = <32,32,G4>, 1 = <16,32,G4>, 2 = <64,64>, 3 = <64,128>, 4 = <32,128>, 5 = <32,256>
    int prog = Cout % 32 == 0 ? 0 : 1;
    if (a.M >= 16384 && Cout % 32 == 0) {
        if (Cout % 64 == 0 && a.K % 128 == 0 && a.M >= 262144) prog = 3;
        else if (Cout % 64 == 0 && a.K % 64 == 0) prog = 2;
        else if (a.K % 256 == 0) prog = 5;
        else if (a.K % 128 == 0) prog = 4;
    }
    const int force = prog_override("PFRL_QNET_WGRAD");
    if (force >= 0 && Cout % 32 == 0 && force != 1) {
        const int bi = (force == 2 || force == 3) ? 64 : 32;
        const int bj = force == 0 ? 32 : force == 2 ? 64 : force == 5 ? 256 : 128;
        if (Cout % bi == 0 && a.K % bj == 0) prog = force;
    }
#define WG(BI, BJ, WM, WN, WK, G)                                                                    \
    hipLaunchKernelGGL((k_conv_wgrad<BI, BJ, WM, WN, WK, G>), dim3(Cout / BI, a.K / BJ, splits),    \
                       dim3(256), 0, st, a)
    switch (prog) {
        case 0: WG(32, 32, 2, 2, 1, 4); break;
        case 1: WG(16, 32, 1, 2, 2, 4); break;
        case 2: WG(64, 64, 2, 2, 1, 2); break;
        case 3: WG(64, 128, 2, 2, 1, 2); break;
        case 4: WG(32, 128, 1, 4, 1, 2); break;
        default: WG(32, 256, 1, 4, 1, 2); break;
    }
#undef WG
    PFRL_LAUNCH_CHECK();
}

// pfrl_conv2d_nhwc_bwd_weight for a layer whose input is u8 NHWC4 pixels (see
// pfrl_conv2d_u8nhwc4_fwd): dw = dy^T (float32(x) / divisor), the same tile program and summation
// order as the fp32 entry -- bit-identical partials.  Cout % 32 == 0 and Cout % 64 != 0.
extern "C" int pfrl_conv2d_u8nhwc4_bwd_weight(const float *dy, const float *dy_mask, const uint8_t *x,
                                              float divisor, float *dw_part, float *db_part,
                                              int64_t dw_stride, int64_t db_stride, int32_t N,
                                              int32_t H, int32_t W, int32_t Cout, int32_t R, int32_t S,
                                              int32_t stride, int32_t splits, void *stream) {
    WgradArgs a;
    if (int rc = make_wgrad_args(a, dy, dy_mask, nullptr, dw_part, db_part, dw_stride, db_stride, N, H,
                                 W, 4, Cout, R, S, stride, splits))
        return rc;
    PFRL_CHECK_ARG(x != nullptr && divisor > 0.f && Cout % 32 == 0 && Cout % 64 != 0,
                   "pfrl_conv2d_u8nhwc4_bwd_weight: null input, divisor <= 0, or Cout outside the "
                   "u8 tile programs (Cout % 32 == 0, Cout % 64 != 0)");
    a.xu8 = x; a.u8_d = divisor; a.u8_r = 1.0f / divisor;
    hipStream_t st = (hipStream_t)stream;
    // (the rule of pfrl_conv2d_nhwc_bwd_weight for these Cout: keep in step)
    int prog = 0;
    if (a.M >= 16384) prog = a.K % 256 == 0 ? 5 : (a.K % 128 == 0 ? 4 : 0);
    // the Nature first layer where the 32 x 256 tile program would run: the direct kernel (same slabs)
    if (prog == 5 && prog_override("PFRL_CONV1_DIRECT") != 0 && H == D1_HW && W == D1_HW && Cout == 32 &&
        R == 8 && S == 8 && stride == 4 && ((uintptr_t)x & 3) == 0 &&
        (((uintptr_t)dy | (uintptr_t)dy_mask) & 15) == 0) {
        hipLaunchKernelGGL(k_conv1_u8_wgrad_direct, dim3(splits), dim3(256), 0, st, dy, dy_mask,
                           reinterpret_cast<const uint32_t *>(x), dw_part, db_part, (long long)dw_stride,
                           (long long)db_stride, N, a.M, a.cps, a.u8_r, a.u8_d);
        PFRL_LAUNCH_CHECK();
    }
#define WGU(BI, BJ, WM, WN, WK, G)                                                                   \
    hipLaunchKernelGGL((k_conv_wgrad<BI, BJ, WM, WN, WK, G, false, true>),                           \
                       dim3(Cout / BI, a.K / BJ, splits), dim3(256), 0, st, a)
    switch (prog) {
        case 0: WGU(32, 32, 2, 2, 1, 4); break;
        case 4: WGU(32, 128, 1, 4, 1, 2); break;
        default: WGU(32, 256, 1, 4, 1, 2); break;
    }
#undef WGU
    PFRL_LAUNCH_CHECK();
}

// Optimizer steps waiting for the NEXT backward launch of this host thread to carry them
// (pfrl_ride_set): consumed -- and cleared -- by pfrl_conv2d_nhwc_bwd / pfrl_conv2d_nhwc_bwd_weight_ride.
struct PendingRide {
    RideArgs r;
    int n = 0, blocks = 0, centered = 0;
};
static thread_local PendingRide g_ride;

extern "C" int pfrl_ride_set(int32_t n, float *const *param, const float *const *grad_src,
                             float *const *square_avg, float *const *grad_avg, const int64_t *numel,
                             const int32_t *n_slabs, const int64_t *slab_stride, float lr, float alpha,
                             float eps, float weight_decay, int centered) {
    g_ride.n = 0;
    g_ride.blocks = 0;
    if (n == 0) return 0;
    PFRL_CHECK_ARG(n >= 1 && n <= RIDE_MAX && param && grad_src && square_avg && numel && n_slabs &&
                       slab_stride && (!centered || grad_avg),
                   "pfrl_ride_set: 1..8 riding tensors");
    RideArgs &r = g_ride.r;
    int blocks = 0;
    for (int i = 0; i < RIDE_MAX; ++i) {
        const int j = i < n ? i : 0;
        r.p[i] = param[j]; r.g[i] = grad_src[j]; r.sq[i] = square_avg[j];
        r.ga[i] = centered ? grad_avg[j] : nullptr;
        r.numel[i] = i < n ? numel[j] : 0;
        r.n_slabs[i] = n_slabs[j];
        r.slab_stride[i] = slab_stride[j];
        if (i < n) {
            const uintptr_t bits = (uintptr_t)r.p[i] | (uintptr_t)r.g[i] | (uintptr_t)r.sq[i] |
                                   (uintptr_t)r.ga[i] | (uintptr_t)(r.slab_stride[i] * 4);
            PFRL_CHECK_ARG(r.p[i] && r.g[i] && r.sq[i] && (bits & 15) == 0 && r.numel[i] > 0 &&
                               r.numel[i] < (1ll << 40) && r.n_slabs[i] >= 0,
                           "pfrl_ride_set: riding tensors and slab strides must be 16-byte aligned");
            blocks += (int)((r.numel[i] + RIDE_CHUNK - 1) / RIDE_CHUNK);
        }
        r.block_end[i] = blocks;
    }
    r.n = n;
    r.lr = lr; r.alpha = alpha; r.eps = eps; r.weight_decay = weight_decay;
    g_ride.n = n;
    g_ride.blocks = blocks;
    g_ride.centered = centered;
    return 0;
}

// pfrl_conv2d_nhwc_bwd_weight with RMSprop steps of OTHER parameters riding in the launch
// (k_conv_wgrad_ride): n_ride <= 4 tensors, each with its finished gradient tensor grad[i];
// 16-byte aligned, numel % 4 == 0.  The caller guarantees that nothing later in the stream order
// of this update reads those parameters or gradients before the launch completes.
extern "C" int pfrl_conv2d_nhwc_bwd_weight_ride(
    const float *dy, const float *dy_mask, const float *x, float *dw_part, float *db_part,
    int64_t dw_stride, int64_t db_stride, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout,
    int32_t R, int32_t S, int32_t stride, int32_t splits, int32_t n_ride, float *const *ride_param,
    const float *const *ride_grad, float *const *ride_square_avg, float *const *ride_grad_avg,
    const int64_t *ride_numel, float lr, float alpha, float eps, float weight_decay, int centered,
    void *stream) {
    WgradArgs a;
    if (int rc = make_wgrad_args(a, dy, dy_mask, x, dw_part, db_part, dw_stride, db_stride, N, H, W, C,
                                 Cout, R, S, stride, splits))
        return rc;
    PFRL_CHECK_ARG(n_ride >= 1 && n_ride <= RIDE_MAX && ride_param && ride_grad && ride_square_avg &&
                       ride_numel && (!centered || ride_grad_avg),
                   "pfrl_conv2d_nhwc_bwd_weight_ride: 1..8 riding tensors");
    // (more steps may be waiting from pfrl_ride_set: they join this launch's own)
    const PendingRide pend = g_ride;
    g_ride.n = 0;
    g_ride.blocks = 0;
    PFRL_CHECK_ARG(n_ride + pend.n <= RIDE_MAX && (pend.n == 0 || pend.centered == centered),
                   "pfrl_conv2d_nhwc_bwd_weight_ride: too many riding tensors with the pending set");
    RideArgs r;
    int blocks = 0;
    const int n_all = n_ride + pend.n;
    for (int i = 0; i < RIDE_MAX; ++i) {
        if (i >= n_ride && i < n_all) {
            const int q = i - n_ride;
            r.p[i] = pend.r.p[q]; r.g[i] = pend.r.g[q]; r.sq[i] = pend.r.sq[q]; r.ga[i] = pend.r.ga[q];
            r.numel[i] = pend.r.numel[q]; r.n_slabs[i] = pend.r.n_slabs[q];
            r.slab_stride[i] = pend.r.slab_stride[q];
            blocks += (int)((r.numel[i] + RIDE_CHUNK - 1) / RIDE_CHUNK);
            r.block_end[i] = blocks;
            continue;
        }
        const int j = i < n_ride ? i : 0;
        r.p[i] = ride_param[j]; r.g[i] = ride_grad[j]; r.sq[i] = ride_square_avg[j];
        r.ga[i] = centered ? ride_grad_avg[j] : nullptr;
        r.numel[i] = i < n_ride ? ride_numel[j] : 0;
        r.n_slabs[i] = 0;
        r.slab_stride[i] = 0;
        if (i < n_ride) {
            const uintptr_t bits = (uintptr_t)r.p[i] | (uintptr_t)r.g[i] | (uintptr_t)r.sq[i] |
                                   (uintptr_t)r.ga[i];
            PFRL_CHECK_ARG(r.p[i] && r.g[i] && r.sq[i] && (bits & 15) == 0 && r.numel[i] > 0 &&
                               r.numel[i] % 4 == 0 && r.numel[i] < (1ll << 40),
                           "pfrl_conv2d_nhwc_bwd_weight_ride: riding tensors must be 16-byte aligned, "
                           "numel % 4 == 0");
            blocks += (int)((r.numel[i] + RIDE_CHUNK - 1) / RIDE_CHUNK);
        }
        r.block_end[i] = blocks;
    }
    r.n = n_all;
    r.lr = lr; r.alpha = alpha; r.eps = eps; r.weight_decay = weight_decay;
    const bool w32 = Cout % 32 == 0;
    WgradGrid wg;
    wg.wgx = w32 ? Cout / 32 : Cout / 16;
    wg.wgy = a.K / 32;
    wg.nw = wg.wgx * wg.wgy * splits;
    wg.q_wgx = fast_div((uint32_t)wg.wgx);
    wg.q_wgy = fast_div((uint32_t)wg.wgy);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(wg.nw + blocks));
#define RIDE(BI, WM, WN, WK, CEN)                                                                    \
    hipLaunchKernelGGL((k_conv_wgrad_ride<BI, 32, WM, WN, WK, 4, CEN>), grid, dim3(256), 0, st, wg.nw, \
                       a, wg, r)
    if (w32) {
        if (centered) RIDE(32, 2, 2, 1, true);
        else RIDE(32, 2, 2, 1, false);
    } else {
        if (centered) RIDE(16, 1, 2, 2, true);
        else RIDE(16, 1, 2, 2, false);
    }
#undef RIDE
    PFRL_LAUNCH_CHECK();
}

// dw[n][k] = sum_m (dy (.) mask)[m][n] x[m][k], db[n]: the weight gradient of a linear layer
// with any in_features (the TAIL loaders when K is not a multiple of 32).  Partials as
// pfrl_conv2d_nhwc_bwd_weight writes them.
extern "C" int pfrl_linear_bwd_weight(const float *dy, const float *dy_mask, const float *x,
                                      float *dw_part, float *db_part, int64_t dw_stride,
                                      int64_t db_stride, int32_t M, int32_t K, int32_t N, int32_t splits,
                                      void *stream) {
    if (K % KC == 0)
        return pfrl_conv2d_nhwc_bwd_weight(dy, dy_mask, x, dw_part, db_part, dw_stride, db_stride, M, 1,
                                           1, K, N, 1, 1, 1, splits, stream);
    PFRL_CHECK_ARG(M >= 1 && K >= 1 && N >= 16 && N % 16 == 0 && splits >= 1,
                   "pfrl_linear_bwd_weight: out_features must be a multiple of 16");
    WgradArgs a;
    a.dy = dy; a.dymask = dy_mask; a.x = x; a.dw = dw_part; a.db = db_part;
    a.dw_stride = dw_stride; a.db_stride = db_stride;
    a.g = make_geom(M, 1, 1, K, N, 1, 1, 1);
    a.M = M;
    a.K = K;
    const int nch = (M + KC - 1) / KC;
    a.cps = (nch + splits - 1) / splits;
    hipStream_t st = (hipStream_t)stream;
    const unsigned gy = (unsigned)((K + 31) / 32);
    if (N % 32 == 0)
        hipLaunchKernelGGL((k_conv_wgrad<32, 32, 2, 2, 1, 4, true>), dim3(N / 32, gy, splits), dim3(256),
                           0, st, a);
    else
        hipLaunchKernelGGL((k_conv_wgrad<16, 32, 1, 2, 2, 4, true>), dim3(N / 16, gy, splits), dim3(256),
                           0, st, a);
    PFRL_LAUNCH_CHECK();
}

// Both gradients of one layer in one launch (k_conv_bwd).  The dgrad arguments describe the
// layer as pfrl_conv2d_nhwc_bwd_data does, the wgrad arguments as pfrl_conv2d_nhwc_bwd_weight;
// dy / dy_mask are shared.  Minibatch-sized problems only (the small tile programs); larger
// ones return PFRL_ERR_ARG and the caller issues the two launches.
extern "C" int pfrl_conv2d_nhwc_bwd(const float *dy, const float *dy_mask, const float *w,
                                    const float *a_prev, const float *x, float *dx, float *dw_part,
                                    float *db_part, int64_t dw_stride, int64_t db_stride, int32_t N,
                                    int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t R, int32_t S,
                                    int32_t stride, int32_t perm_p, int32_t perm_c, int32_t splits,
                                    void *stream) {
    DgradArgs d;
    WgradArgs wa;
    if (int rc = make_dgrad_args(d, dy, dy_mask, w, a_prev, dx, N, H, W, C, Cout, R, S, stride, perm_p,
                                 perm_c))
        return rc;
    if (int rc = make_wgrad_args(wa, dy, dy_mask, x, dw_part, db_part, dw_stride, db_stride, N, H, W, C,
                                 Cout, R, S, stride, splits))
        return rc;
    const int z = stride * stride;
    const int prog = dgrad_program(d, z);
    const PendingRide pend = g_ride;      // (consumed here, whatever happens below)
    g_ride.n = 0;
    g_ride.blocks = 0;
    PFRL_CHECK_ARG(prog >= 2 && prog <= 5, "pfrl_conv2d_nhwc_bwd: problem too large for the fused launch");
    hipStream_t st = (hipStream_t)stream;
    const bool w32 = Cout % 32 == 0;
    const int wgx = w32 ? Cout / 32 : Cout / 16, wgy = wa.K / 32;
    const int nw = wgx * wgy * splits;
    if (pend.n > 0) {
        // optimizer steps of finished tensors ride in this launch (pfrl_ride_set)
        PFRL_CHECK_ARG(w32 && prog >= 2 && prog <= 4,
                       "pfrl_conv2d_nhwc_bwd: no riding form of this tile program (nothing was launched)");
#define BWDR(DM, DN, DWM, DWN, DWK, DGG)                                                             \
    do {                                                                                             \
        const int dgx = (d.Mc + DM - 1) / DM, dgy = C / DN;                                          \
        const int nd = dgx * dgy * z;                                                                \
        const unsigned grid = (unsigned)(nd + nw + pend.blocks);                                     \
        if (pend.centered)                                                                           \
            hipLaunchKernelGGL((k_conv_bwd_ride<DM, DN, DWM, DWN, DWK, DGG, 32, 2, 2, 1, true>),     \
                               dim3(grid), dim3(256), 0, st, nd, nw, d, wa,                          \
                               make_bwd_grid(dgx, dgy, z, wgx, wgy), pend.r);                        \
        else                                                                                         \
            hipLaunchKernelGGL((k_conv_bwd_ride<DM, DN, DWM, DWN, DWK, DGG, 32, 2, 2, 1, false>),    \
                               dim3(grid), dim3(256), 0, st, nd, nw, d, wa,                          \
                               make_bwd_grid(dgx, dgy, z, wgx, wgy), pend.r);                        \
    } while (0)
        if (prog == 2) BWDR(32, 32, 2, 2, 1, 4);
        else BWDR(16, 32, 1, 2, 2, 4);
#undef BWDR
        PFRL_LAUNCH_CHECK();
    }
#define BWD(DM, DN, DWM, DWN, DWK, DGG)                                                              \
    do {                                                                                             \
        const int dgx = (d.Mc + DM - 1) / DM, dgy = C / DN;                                          \
        const unsigned grid = (unsigned)(dgx * dgy * z + nw);                                        \
        if (w32)                                                                                     \
            hipLaunchKernelGGL((k_conv_bwd<DM, DN, DWM, DWN, DWK, DGG, 32, 2, 2, 1>), dim3(grid),    \
                               dim3(256), 0, st, dgx * dgy * z, d, wa,                               \
                               make_bwd_grid(dgx, dgy, z, wgx, wgy));                                \
        else                                                                                         \
            hipLaunchKernelGGL((k_conv_bwd<DM, DN, DWM, DWN, DWK, DGG, 16, 1, 2, 2>), dim3(grid),    \
                               dim3(256), 0, st, dgx * dgy * z, d, wa,                               \
                               make_bwd_grid(dgx, dgy, z, wgx, wgy));                                \
    } while (0)
    switch (prog) {
        case 2: BWD(32, 32, 2, 2, 1, 4); break;
        case 3:   // (stages of 4 here: with stages of 8 the input-gradient workgroups finish 1.7 us
                  // sooner, but at 55 KB of LDS two workgroups fill a CU and the weight-gradient
                  // workgroups queue behind them: the launch measured 2 us LONGER, tools/bwd_phase.py)
        case 4: BWD(16, 32, 1, 2, 2, 4); break;
        default: BWD(32, 16, 2, 1, 2, 4); break;
    }
#undef BWD
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_splitk_reduce(int32_t n_tasks, const float *const *host_part, float *const *host_out,
                                  const float *const *host_bias, const int64_t *host_stride,
                                  const int32_t *host_n, const int32_t *host_splits,
                                  const int32_t *host_ncol, const int32_t *host_relu, void *stream) {
    return pfrl_splitk_reduce_noisy(n_tasks, host_part, host_out, host_bias, nullptr, nullptr, host_stride,
                                    host_n, host_splits, host_ncol, host_relu, stream);
}

extern "C" int pfrl_splitk_reduce_noisy(int32_t n_tasks, const float *const *host_part,
                                        float *const *host_out, const float *const *host_bias,
                                        const float *const *host_bias_sigma,
                                        const float *const *host_bias_noise, const int64_t *host_stride,
                                        const int32_t *host_n, const int32_t *host_splits,
                                        const int32_t *host_ncol, const int32_t *host_relu, void *stream) {
    PFRL_CHECK_ARG(n_tasks >= 0 && n_tasks <= RED_MAX, "pfrl_splitk_reduce: too many tensors");
    if (n_tasks == 0) return 0;
    RedArgs a;
    int blocks = 0;
    for (int t = 0; t < n_tasks; ++t) {
        a.part[t] = host_part[t];
        a.out[t] = host_out[t];
        a.bias[t] = host_bias ? host_bias[t] : nullptr;
        a.bias_sigma[t] = host_bias_sigma ? host_bias_sigma[t] : nullptr;
        a.bias_noise[t] = host_bias_noise ? host_bias_noise[t] : nullptr;
        PFRL_CHECK_ARG(a.bias_sigma[t] == nullptr || (a.bias[t] != nullptr && a.bias_noise[t] != nullptr),
                       "pfrl_splitk_reduce_noisy: sigma bias needs the mu bias and the noise");
        a.stride[t] = host_stride[t];
        a.n[t] = host_n[t];
        a.splits[t] = host_splits[t];
        a.ncol[t] = host_ncol ? host_ncol[t] : 4;
        a.relu[t] = host_relu ? host_relu[t] : 0;
        PFRL_CHECK_ARG(a.bias[t] == nullptr || a.ncol[t] % 4 == 0, "pfrl_splitk_reduce: ncol % 4");
        PFRL_CHECK_ARG(a.stride[t] % 4 == 0, "pfrl_splitk_reduce: stride % 4");
        blocks += (host_n[t] + RED_CHUNK - 1) / RED_CHUNK;
        a.block_end[t] = blocks;
    }
    a.ntask = n_tasks;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(k_splitk_reduce, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_splitk_group(const float *part, int64_t stride, int32_t n, int32_t splits,
                                 int32_t groups, float *out, int64_t out_stride, void *stream) {
    PFRL_CHECK_ARG(part && out && n >= 1 && splits >= 1 && groups >= 1 && groups <= splits &&
                       stride % 4 == 0 && out_stride % 4 == 0 && out_stride >= n,
                   "pfrl_splitk_group: bad arguments (strides % 4, groups <= splits)");
    const int per = (splits + groups - 1) / groups;
    PFRL_CHECK_ARG((long long)per * (groups - 1) < splits, "pfrl_splitk_group: an empty group");
    hipLaunchKernelGGL(k_splitk_group, dim3((n + RED_CHUNK - 1) / RED_CHUNK, groups), dim3(256), 0,
                       (hipStream_t)stream, part, (long long)stride, n, splits, per, out,
                       (long long)out_stride);
    PFRL_LAUNCH_CHECK();
}

#define SMALL_DISPATCH(N, CALL)                                            \
    switch (N) {                                                           \
        case 1: CALL(1); break;   case 2: CALL(2); break;   case 3: CALL(3); break;   \
        case 4: CALL(4); break;   case 5: CALL(5); break;   case 6: CALL(6); break;   \
        case 7: CALL(7); break;   case 8: CALL(8); break;   case 9: CALL(9); break;   \
        case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
        case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; \
        default: CALL(16); break;                                          \
    }

static int small_fwd_launch(const SmallFwdArgs &a, int twins, int32_t M, int32_t K, int32_t N,
                            void *stream) {
    PFRL_CHECK_ARG(N >= 1 && N <= SMALL_N && M >= 1 && K >= 1, "pfrl_linear_small_fwd: N <= 16");
#define CALL_FWD(NN)                                                                             \
    hipLaunchKernelGGL(k_linear_small_fwd<NN>, dim3(M, twins), dim3(256), 0, (hipStream_t)stream, a, K)
    SMALL_DISPATCH(N, CALL_FWD)
#undef CALL_FWD
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_linear_small_fwd(const float *x, const float *w, const float *bias, float *y,
                                     int32_t M, int32_t K, int32_t N, void *stream) {
    SmallFwdArgs a{{x, nullptr}, {w, nullptr}, {bias, nullptr}, {y, nullptr}};
    return small_fwd_launch(a, 1, M, K, N, stream);
}

extern "C" int pfrl_dqn_act_head(const float *h, const float *w, const float *bias, const int32_t *choice,
                                 float *q, int64_t *greedy, int64_t *action, int32_t M, int32_t K,
                                 int32_t N, void *stream) {
    PFRL_CHECK_ARG(N >= 1 && N <= SMALL_N && M >= 1 && K >= 1, "pfrl_dqn_act_head: 1 <= N <= 16");
    PFRL_CHECK_ARG(h && w && (q || greedy || action), "pfrl_dqn_act_head: null argument");
#define CALL_ACT(NN)                                                                             \
    hipLaunchKernelGGL(k_dqn_act_head<NN>, dim3(M), dim3(256), 0, (hipStream_t)stream, h, w, bias, \
                       choice, q, greedy, action, K)
    SMALL_DISPATCH(N, CALL_ACT)
#undef CALL_ACT
    PFRL_LAUNCH_CHECK();
}

static int small_bwd_launch(const SmallBwdArgs &a, int twins, bool want_dx, bool want_dw, int32_t M,
                            int32_t K, int32_t N, void *stream) {
    const int NP = N <= SMALL_N ? N : (N + 7) / 8 * 8;   // the instantiated width (LDS row stride)
    PFRL_CHECK_ARG(N >= 1 && N <= SMALL_BWD_N && M >= 1 && K >= 1 && (size_t)M * NP * 4 <= 40 * 1024,
                   "pfrl_linear_small_bwd: N <= 64, M * (N rounded up to 8 above 16) <= 10240");
    PFRL_CHECK_ARG(want_dw || want_dx, "pfrl_linear_small_bwd: nothing to compute");
    const int n_dw = want_dw ? (K + 31) / 32 : 0;   // dw == NULL: input gradient only
    const dim3 grid(n_dw + (want_dx ? ((K + 255) / 256) * ((M + 7) / 8) : 0), twins);
#define CALL_BWD(NN)                                                                              \
    hipLaunchKernelGGL(k_linear_small_bwd<NN>, grid, dim3(256), (size_t)M * NP * sizeof(float),  \
                       (hipStream_t)stream, a, M, K, n_dw, N)
    if (N <= SMALL_N) {
        SMALL_DISPATCH(N, CALL_BWD)
    } else if (N <= 24) CALL_BWD(24);
    else if (N <= 32) CALL_BWD(32);
    else if (N <= 40) CALL_BWD(40);
    else if (N <= 48) CALL_BWD(48);
    else if (N <= 56) CALL_BWD(56);
    else CALL_BWD(64);
#undef CALL_BWD
    PFRL_LAUNCH_CHECK();
}

namespace {

// Widths 17..64 (the 2 x action_size policy head of SAC, Linear(256, 34)): both gradients on the
// matrix cores.  dy is staged ONCE into LDS, zero-padded to [M16][16 NT] (row stride + 1 against
// bank conflicts), and serves as an MFMA operand of both products:
//   dw[n][k] = sum_m dy[m][n] x[m][k]   A = dy^T (row n = lane & 15, m = 4 step + lane >> 4, from LDS)
//                                       B = x    (one global load per step, straight into the operand)
//              a wave = one 16-column tile of k x all n x one half of the batch; the halves meet in
//              LDS; one more "tile" whose B is all ones yields db = sum_m dy[m][n]
//   dx[m][k] = sum_n dy[m][n] w[n][k]   A = dy (row m), B = w: a wave = one k tile x half the m tiles
// Every global load of a wave is issued before the LDS copy is waited for: the launch costs one memory
// round trip.  18.4 us for the scalar narrow-head kernel at 256 x 34 (LDS-read bound), 15.6 for the
// library's three launches.
template <int NT>
__global__ __launch_bounds__(256) void k_linear_narrow_bwd(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ w,
    float *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db, int M, int K, int NR, int n_dw) {
    constexpr int NP = 16 * NT, LD = NP + 1;
    extern __shared__ float sdy[];               // [M16][LD]
    __shared__ float comb[2][NT][64][4];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r16 = lane & 15, q = lane >> 4;
    const int M16 = (M + 15) / 16 * 16, KT = (K + 15) / 16;
    constexpr int DYL = 64;                      // M * NR <= 256 threads x 64
    float dv[DYL];
    const int ndl = (M * NR + 255) / 256;        // (uniform)
    // (loads are UNCONDITIONAL on clamped addresses and masked by value afterwards: a load inside
    // a branch gets its own s_waitcnt, which serialises the whole batch -- 15 us instead of 7)
#pragma unroll
    for (int g8 = 0; g8 < DYL / 8; ++g8)
        if (8 * g8 < ndl) {                      // (uniform, eight loads per block)
#pragma unroll
            for (int u = 8 * g8; u < 8 * g8 + 8; ++u) dv[u] = dy[min(tid + 256 * u, M * NR - 1)];
        }
    const bool dw_role = (int)blockIdx.x < n_dw;
    // ---- this wave's own operand loads, before anything is waited for
    constexpr int BMAX = 32;                     // B operand registers: 32 steps of the batch half / 4 NT steps of n
    float b[BMAX];
    int tile, mh = 0, Mh = 0, mt_lo = 0, mt_hi = 0;
    bool ones = false, idle = false;
    if (dw_role) {
        tile = blockIdx.x * 2 + (wave >> 1);
        mh = wave & 1;
        Mh = M16 / 2;
        ones = tile == KT;
        idle = tile > KT;
        const int k = min(tile * 16 + r16, K - 1);
        const bool kin = tile * 16 + r16 < K;
        float xl[BMAX];
#pragma unroll
        for (int u = 0; u < BMAX; ++u) xl[u] = x[(size_t)min(mh * Mh + 4 * u + q, M - 1) * K + k];
#pragma unroll
        for (int u = 0; u < BMAX; ++u) {
            const int m = mh * Mh + 4 * u + q;
            const bool in = 4 * u < Mh && m < M && !idle;
            b[u] = ones ? (in ? 1.f : 0.f) : ((in && kin) ? xl[u] : 0.f);
        }
    } else {
        const int KG = (KT + 3) / 4, MT = M16 / 16, MS = MT >= 2 ? 2 : 1;
        const int idx = blockIdx.x - n_dw;
        tile = (idx % KG) * 4 + wave;
        idle = tile >= KT;
        const int ms = idx / KG, MTh = (MT + MS - 1) / MS;
        mt_lo = ms * MTh;
        mt_hi = min(MT, mt_lo + MTh);
        const int k = min(tile * 16 + r16, K - 1);
        const bool kin = tile * 16 + r16 < K && !idle;
        float wl[4 * NT];
#pragma unroll
        for (int u = 0; u < 4 * NT; ++u) wl[u] = w[(size_t)min(4 * u + q, NR - 1) * K + k];
#pragma unroll
        for (int u = 0; u < 4 * NT; ++u) b[u] = (kin && 4 * u + q < NR) ? wl[u] : 0.f;
    }
    // ---- the LDS copy of dy: zero, then scatter
    for (int e = tid; e < M16 * LD; e += 256) sdy[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int g8 = 0; g8 < DYL / 8; ++g8)
        if (8 * g8 < ndl) {
#pragma unroll
            for (int u = 8 * g8; u < 8 * g8 + 8; ++u) {
                const int e = tid + 256 * u;
                if (e < M * NR) {
                    const int row = e / NR;
                    sdy[row * LD + (e - row * NR)] = dv[u];
                }
            }
        }
    __syncthreads();
    if (dw_role) {
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (batches beyond the first 32 steps -- M16 > 256 -- load as they go)
        for (int s0 = 0; 4 * s0 < Mh; s0 += BMAX) {
            if (s0 > 0) {
                const int k = min(tile * 16 + r16, K - 1);
                const bool kin = tile * 16 + r16 < K;
                float xl[BMAX];
#pragma unroll
                for (int u = 0; u < BMAX; ++u) xl[u] = x[(size_t)min(mh * Mh + 4 * (s0 + u) + q, M - 1) * K + k];
#pragma unroll
                for (int u = 0; u < BMAX; ++u) {
                    const int m = mh * Mh + 4 * (s0 + u) + q;
                    const bool in = 4 * (s0 + u) < Mh && m < M && !idle;
                    b[u] = ones ? (in ? 1.f : 0.f) : ((in && kin) ? xl[u] : 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < BMAX; ++u) {
                if (4 * (s0 + u) >= Mh) continue;        // (uniform)
                const int m = mh * Mh + 4 * (s0 + u) + q;
                const float *__restrict__ row = sdy + m * LD + r16;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(row[16 * t], b[u], acc[t], 0, 0, 0);
            }
        }
        if (mh == 1) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) comb[wave >> 1][t][lane][i] = acc[t][i];
        }
        __syncthreads();
        if (mh == 0 && !idle) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[t][i] + comb[wave >> 1][t][lane][i];
                    const int n = 16 * t + 4 * q + i, k = tile * 16 + r16;   // accumulator: row = 4 q + i, column = r16
                    if (n < NR) {
                        if (ones) {
                            if (r16 == 0 && db != nullptr) db[n] = v;
                        } else if (k < K) {
                            dw[(size_t)n * K + k] = v;
                        }
                    }
                }
        }
    } else if (!idle) {
        for (int mt = mt_lo; mt < mt_hi; ++mt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float *__restrict__ row = sdy + (mt * 16 + r16) * LD + q;
#pragma unroll
            for (int u = 0; u < 4 * NT; ++u)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(row[4 * u], b[u], acc, 0, 0, 0);
            const int k = tile * 16 + r16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mt * 16 + 4 * q + i;
                if (m < M && k < K) dx[(size_t)m * K + k] = acc[i];
            }
        }
    }
}

}  // namespace

extern "C" int pfrl_linear_small_bwd(const float *dy, const float *x, const float *w, float *dx,
                                     float *dw, float *db, int32_t M, int32_t K, int32_t N,
                                     void *stream) {
    if (N > SMALL_N && dw != nullptr && getenv("PFRL_NARROW_BWD_SCALAR") == nullptr) {
        // widths 17..64 with weight gradients wanted: the MFMA kernel
        const int NT = (N + 15) / 16, M16 = (M + 15) / 16 * 16;
        const size_t lds = (size_t)M16 * (16 * NT + 1) * sizeof(float);
        PFRL_CHECK_ARG(N <= SMALL_BWD_N && M >= 1 && K >= 1 && lds <= 64 * 1024 && (size_t)M * N <= 256 * 64,
                       "pfrl_linear_small_bwd: N <= 64, M16 * (16 ceil(N / 16) + 1) floats of LDS <= 64 KB");
        const int KT = (K + 15) / 16, n_dw = (KT + 1 + 1) / 2;
        const int n_dx = dx != nullptr ? ((KT + 3) / 4) * (M16 / 16 >= 2 ? 2 : 1) : 0;
        const dim3 grid(n_dw + n_dx);
#define CALL_NARROW(NTT)                                                                                  \
    hipLaunchKernelGGL(k_linear_narrow_bwd<NTT>, grid, dim3(256), lds, (hipStream_t)stream, dy, x, w, dx, dw, db, \
                       M, K, N, n_dw)
        if (NT == 2) CALL_NARROW(2);
        else if (NT == 3) CALL_NARROW(3);
        else CALL_NARROW(4);
#undef CALL_NARROW
        PFRL_LAUNCH_CHECK();
    }
    SmallBwdArgs a{{dy, nullptr}, {x, nullptr}, {w, nullptr}, {dx, nullptr}, {dw, nullptr}, {db, nullptr}};
    return small_bwd_launch(a, 1, dx != nullptr, dw != nullptr, M, K, N, stream);
}

// ===================================================================================
// Twin launches: the same layer of two independent networks in one grid (host arrays of
// two device pointers each).  Fixed small tile programs: these are minibatch-sized problems.
// ===================================================================================
namespace {

// dx[m][j] = sum_t sum_n (dy_t (.) mask_t)[m][n] * w_t[n][col0 + j], j < ncol: the gradient
// w.r.t. the last `ncol` input columns of the twins' first layer (the action appended to the
// observation), the only input gradient the SAC / TD3 policy loss needs.  One workgroup per
// 4 rows, a wave per row; lanes walk n, every lane keeps ncol partial sums.
constexpr int DXT_MAX = 32;

struct TwinDxArgs {
    const float *dy[2], *mask[2], *w[2];
};

__global__ __launch_bounds__(256) void k_twin_input_grad(TwinDxArgs a, int ldw, int col0, int ncol,
                                                         float *__restrict__ dx, int M, int N) {
    extern __shared__ float ws[];   // [2][N][ncol]: the weight columns, staged once per workgroup
    const int per = N * ncol, tot = 2 * per;
    // every thread's share of the staging loads in flight at once (up to 60 KB: 60 per thread)
    for (int e0 = threadIdx.x; e0 < tot; e0 += 256 * 20) {
        float v[20];
#pragma unroll
        for (int u = 0; u < 20; ++u) {
            const int e = min(e0 + 256 * u, tot - 1);
            const int t = e >= per, r = e - t * per;
            const int n = r / ncol, j = r - n * ncol;
            v[u] = a.w[t][(size_t)n * ldw + col0 + j];
        }
#pragma unroll
        for (int u = 0; u < 20; ++u)
            if (e0 + 256 * u < tot) ws[e0 + 256 * u] = v[u];
    }
    const int m = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int mm = m < M ? m : M - 1;
    // every lane's dy (ReLU-masked) values of both networks, loads first
    constexpr int NMAX = 8;   // N <= 512
    float g[2][NMAX];
    const int nu = (N + 63) / 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float *__restrict__ dy = a.dy[t] + (size_t)mm * N;
        const float *__restrict__ mk = a.mask[t] != nullptr ? a.mask[t] + (size_t)mm * N : dy;
        float d[NMAX], h[NMAX];
#pragma unroll
        for (int u = 0; u < NMAX; ++u) {
            if (u < nu) {   // (uniform)
                const int n = min(lane + 64 * u, N - 1);
                d[u] = dy[n];
                h[u] = mk[n];
            } else {
                d[u] = 0.f;
                h[u] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < NMAX; ++u) {
            const bool in = lane + 64 * u < N;
            g[t][u] = (in && (a.mask[t] == nullptr || h[u] > 0.f)) ? d[u] : 0.f;
        }
    }
    __syncthreads();
    float acc[DXT_MAX];
#pragma unroll
    for (int j = 0; j < DXT_MAX; ++j) acc[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NMAX; ++u) {
            if (64 * u >= N) continue;   // (uniform)
            const int n = min(lane + 64 * u, N - 1);
            const float *wr = ws + (size_t)(t * N + n) * ncol;
#pragma unroll
            for (int j = 0; j < DXT_MAX; ++j)
                if (j < ncol) acc[j] = fmaf(g[t][u], wr[j], acc[j]);
        }
#pragma unroll
    for (int j = 0; j < DXT_MAX; ++j) {
        if (j < ncol) {   // (uniform; the loop stays unrolled so that acc lives in registers)
            float v = acc[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0 && m < M) dx[(size_t)m * ncol + j] = v;
        }
    }
}

// The same product on the matrix cores, for N % 8 == 0: [16 rows] x [2N] (the masked dy of both
// networks side by side) times [2N] x [32] (the weight columns, zero beyond ncol).  A workgroup
// per 16 rows, its NW waves an equal share of the 2N reduction each: every operand element is ONE
// global load straight into the MFMA operand layout (A: row = lane & 15, k = lane >> 4; B: k =
// lane >> 4, column = lane & 15) -- no LDS staging of the weights, no index divisions, all loads
// independent -- and the partial tiles are summed through LDS in wave order.  NW = waves per
// workgroup: after a kernel boundary a global load is a ~2 us round trip on this part (the producer
// ran on other XCDs), so what a launch this small costs is its number of DEPENDENT load batches:
// with 16 waves (N % 32 == 0) a wave's 32 loads per lane are one batch for N = 256.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_twin_input_grad_mfma(TwinDxArgs a, int ldw, int col0, int ncol,
                                                                  float *__restrict__ dx, int M, int N) {
    __shared__ float part[NW][16][33];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r16 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int mrow = min(m0 + r16, M - 1);
    const int KW = (2 * N) / NW;                // this wave's share of the reduction (a multiple of 16)
    const bool two = ncol > 16;                 // (uniform)
    const bool c0 = r16 < ncol, c1 = 16 + r16 < ncol;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // k order inside a block of 16: MFMA step i of the block takes k = 16 j + 4 kq + i from lane
    // (r16, kq) for BOTH operands (any bijection serves a reduction), so that a lane's four A
    // elements of a block are ONE 16-byte load along its row (a 4-byte load per step dragged 16
    // cache lines through the L1 per instruction: 12.8 us; this: see profiles/r04_sac_update_timeline.txt).
    for (int s0 = 0; s0 < KW; s0 += 32) {       // two blocks of 16 k per trip, loads first
        float4 g[2], h[2];
        float b0[2][4], b1[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kb = wave * KW + min(s0 + 16 * j, KW - 16) + 4 * kq;   // (clamped: the tail repeats, masked below)
            const int t = kb >= N, n = kb - t * N;
            g[j] = *reinterpret_cast<const float4 *>(a.dy[t] + (size_t)mrow * N + n);
            h[j] = a.mask[t] != nullptr ? *reinterpret_cast<const float4 *>(a.mask[t] + (size_t)mrow * N + n)
                                        : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float *__restrict__ wr = a.w[t] + (size_t)(n + i) * ldw + col0;
                b0[j][i] = wr[min(r16, ncol - 1)];          // (unconditional, masked below)
                b1[j][i] = wr[min(16 + r16, ncol - 1)];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (s0 + 16 * j >= KW) continue;    // (uniform)
            const float gv[4] = {g[j].x, g[j].y, g[j].z, g[j].w}, hv[4] = {h[j].x, h[j].y, h[j].z, h[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float av = hv[i] > 0.f ? gv[i] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, c0 ? b0[j][i] : 0.f, acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, c1 ? b1[j][i] : 0.f, acc1, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {               // accumulator layout: row = 4 (lane >> 4) + i, column = lane & 15
        part[wave][kq * 4 + i][r16] = acc0[i];
        part[wave][kq * 4 + i][16 + r16] = acc1[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * 32; e += NW * 64) {
        const int row = e >> 5, j = e & 31;
        if (j < ncol && m0 + row < M) {
            float v = part[0][row][j];
#pragma unroll
            for (int q = 1; q < NW; ++q) v += part[q][row][j];
            dx[(size_t)(m0 + row) * ncol + j] = v;
        }
    }
}

}  // namespace

extern "C" int pfrl_linear_fwd_twin(const float *const *x, const float *const *x2, int32_t K1,
                                    const float *const *w, const float *const *bias, float *const *y,
                                    int32_t M, int32_t K, int32_t N, int32_t relu, void *stream) {
    PFRL_CHECK_ARG(M >= 1 && K >= 1 && N >= 32 && N % 32 == 0, "pfrl_linear_fwd_twin: out_features % 32");
    PFRL_CHECK_ARG(x2 == nullptr || (K1 >= 1 && K1 < K), "pfrl_linear_fwd_twin: bad column split");
    FwdArgs a[2];
    for (int t = 0; t < 2; ++t) {
        PFRL_CHECK_ARG(x[t] && w[t] && bias[t] && y[t], "pfrl_linear_fwd_twin: null pointer");
        a[t].x = x[t]; a[t].w = w[t]; a[t].bias = bias[t]; a[t].y = y[t];
        a[t].g = make_geom(M, 1, 1, K, N, 1, 1, 1);
        a[t].M = M;
        a[t].K = K;
        a[t].cps = (K + KC - 1) / KC;
        a[t].relu = relu; a[t].planar = 0; a[t].partial = 0;
        if (x2 != nullptr) {
            a[t].x2 = x2[t];
            a[t].K1 = K1;
        }
    }
    const dim3 grid((M + 15) / 16, N / 32, 2);
    hipStream_t st = (hipStream_t)stream;
    if (K % KC == 0 && x2 == nullptr)
        hipLaunchKernelGGL((k_conv_fwd2<16, 32, 1, 2, 2, 4, false>), grid, dim3(256), 0, st, a[0], a[1]);
    else
        hipLaunchKernelGGL((k_conv_fwd2<16, 32, 1, 2, 2, 4, true>), grid, dim3(256), 0, st, a[0], a[1]);
    PFRL_LAUNCH_CHECK();
}

// Gradients of a twin layer.  dx == NULL: weight gradients only (any in_features);
// dw_part == NULL: input gradients only; both given: one launch for all four (needs
// in_features % 32 == 0).  Partials as pfrl_conv2d_nhwc_bwd_weight writes them.
extern "C" int pfrl_linear_bwd_twin(const float *const *dy, const float *const *dy_mask,
                                    const float *const *w, const float *const *x,
                                    const float *const *x2, int32_t K1, float *const *dx,
                                    float *const *dw_part, float *const *db_part, int64_t dw_stride,
                                    int64_t db_stride, int32_t M, int32_t K, int32_t N, int32_t splits,
                                    void *stream) {
    PFRL_CHECK_ARG(M >= 1 && K >= 1 && N >= 32 && N % 32 == 0 && splits >= 1,
                   "pfrl_linear_bwd_twin: out_features % 32");
    PFRL_CHECK_ARG(dx != nullptr || dw_part != nullptr, "pfrl_linear_bwd_twin: nothing to compute");
    hipStream_t st = (hipStream_t)stream;
    DgradArgs d[2];
    WgradArgs wa[2];
    for (int t = 0; t < 2; ++t) {
        const float *mk = dy_mask != nullptr ? dy_mask[t] : nullptr;
        if (dx != nullptr) {
            PFRL_CHECK_ARG(K % KC == 0, "pfrl_linear_bwd_twin: input gradient needs in_features % 32");
            if (int rc = make_dgrad_args(d[t], dy[t], mk, w[t], nullptr, dx[t], M, 1, 1, K, N, 1, 1, 1, 0, 0))
                return rc;
        }
        if (dw_part != nullptr) {
            wa[t].dy = dy[t]; wa[t].dymask = mk; wa[t].x = x[t];
            wa[t].dw = dw_part[t]; wa[t].db = db_part != nullptr ? db_part[t] : nullptr;
            wa[t].dw_stride = dw_stride; wa[t].db_stride = db_stride;
            wa[t].g = make_geom(M, 1, 1, K, N, 1, 1, 1);
            wa[t].M = M;
            wa[t].K = K;
            const int nch = (M + KC - 1) / KC;
            wa[t].cps = (nch + splits - 1) / splits;
            if (x2 != nullptr) {
                PFRL_CHECK_ARG(dx == nullptr && K1 >= 1 && K1 < K, "pfrl_linear_bwd_twin: bad column split");
                wa[t].x2 = x2[t];
                wa[t].K1 = K1;
            }
        }
    }
    const int wgx = N / 32, wgy = (K + 31) / 32;
    if (dx != nullptr && dw_part != nullptr) {
        const int dgx = (M + 15) / 16, dgy = K / 32;
        hipLaunchKernelGGL((k_conv_bwd2<16, 32, 1, 2, 2, 4, 32, 2, 2, 1>),
                           dim3(dgx * dgy + wgx * wgy * splits, 2), dim3(256), 0, st, d[0], wa[0], d[1],
                           wa[1], make_bwd_grid(dgx, dgy, 1, wgx, wgy));
    } else if (dx != nullptr) {
        hipLaunchKernelGGL((k_conv_dgrad2<16, 32, 1, 2, 2, 4>), dim3((M + 15) / 16, K / 32, 2), dim3(256),
                           0, st, d[0], d[1]);
    } else if (K % KC == 0 && x2 == nullptr) {
        hipLaunchKernelGGL((k_conv_wgrad2<32, 32, 2, 2, 1, 4, false>), dim3(wgx, wgy, 2 * splits),
                           dim3(256), 0, st, wa[0], wa[1], splits);
    } else {
        hipLaunchKernelGGL((k_conv_wgrad2<32, 32, 2, 2, 1, 4, true>), dim3(wgx, wgy, 2 * splits),
                           dim3(256), 0, st, wa[0], wa[1], splits);
    }
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_linear_small_fwd_twin(const float *const *x, const float *const *w,
                                          const float *const *bias, float *const *y, int32_t M, int32_t K,
                                          int32_t N, void *stream) {
    SmallFwdArgs a{{x[0], x[1]}, {w[0], w[1]}, {bias[0], bias[1]}, {y[0], y[1]}};
    return small_fwd_launch(a, 2, M, K, N, stream);
}

extern "C" int pfrl_linear_small_bwd_twin(const float *const *dy, const float *const *x,
                                          const float *const *w, float *const *dx, float *const *dw,
                                          float *const *db, int32_t M, int32_t K, int32_t N, void *stream) {
    SmallBwdArgs a{{dy[0], dy[1]}, {x[0], x[1]}, {w[0], w[1]},
                   {dx ? dx[0] : nullptr, dx ? dx[1] : nullptr},
                   {dw ? dw[0] : nullptr, dw ? dw[1] : nullptr},
                   {db ? db[0] : nullptr, db ? db[1] : nullptr}};
    return small_bwd_launch(a, 2, dx != nullptr, dw != nullptr, M, K, N, stream);
}

extern "C" int pfrl_twin_input_grad(const float *const *dy, const float *const *dy_mask,
                                    const float *const *w, int32_t ldw, int32_t col0, int32_t ncol,
                                    float *dx, int32_t M, int32_t N, void *stream) {
    PFRL_CHECK_ARG(M >= 1 && N >= 1 && N <= 512 && ncol >= 1 && ncol <= DXT_MAX && col0 >= 0 &&
                       col0 + ncol <= ldw && (size_t)2 * N * ncol * sizeof(float) <= 60 * 1024,
                   "pfrl_twin_input_grad: at most 32 columns, N <= 512, 2*N*ncol floats of LDS");
    TwinDxArgs a{{dy[0], dy[1]},
                 {dy_mask ? dy_mask[0] : nullptr, dy_mask ? dy_mask[1] : nullptr},
                 {w[0], w[1]}};
    // (a wave's share is whole blocks of 16 k that do not straddle the two networks)
    if (N % 128 == 0 && getenv("PFRL_TWIN_DX_STAGED") == nullptr) {
        hipLaunchKernelGGL(k_twin_input_grad_mfma<16>, dim3((M + 15) / 16), dim3(1024), 0, (hipStream_t)stream,
                           a, ldw, col0, ncol, dx, M, N);
        PFRL_LAUNCH_CHECK();
    }
    if (N % 32 == 0 && getenv("PFRL_TWIN_DX_STAGED") == nullptr) {
        hipLaunchKernelGGL(k_twin_input_grad_mfma<4>, dim3((M + 15) / 16), dim3(256), 0, (hipStream_t)stream, a,
                           ldw, col0, ncol, dx, M, N);
        PFRL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_twin_input_grad, dim3((M + 3) / 4), dim3(256),
                       (size_t)2 * N * ncol * sizeof(float), (hipStream_t)stream, a, ldw, col0, ncol, dx, M,
                       N);
    PFRL_LAUNCH_CHECK();
}

#ifdef PFRL_QNET_DEBUG
extern "C" int pfrl_qnet_debug_reset() {
    void *p;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_qstamp)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * 4096 * 8);
}
extern "C" int pfrl_qnet_debug_set_grid(int workgroups) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_qgrid), &workgroups, sizeof(int));
}
extern "C" int pfrl_qnet_debug_set_reps(int reps) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_qreps), &reps, sizeof(int));
}
extern "C" int pfrl_qnet_debug_read(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qstamp), sizeof(unsigned long long) * 4096 * 8);
}
#endif
