// Transition table / n-step entry kernels and the fused batch_experiences
// (pfrl/replay_buffer.py:157-212): one launch resolves the sampled entries,
// collapses the n-step scalars and gathers state + next_state stacks into
// fp32 minibatches.
//
// Roofline: HBM.  Algorithmic bytes per sampled entry =
//   2 * k * frame_bytes read + 2 * k * 4 * frame_bytes written (u8 frames)
// plus < 0.1 % metadata.
#include <hip/hip_ext.h>

#include <stdlib.h>

#include <vector>

#include "common.h"
#include "nhwc.h"

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 8;

// Optional in-library timing of the fused gather: when enabled, every launch
// carries a start/stop hipEvent pair attached to the dispatch itself
// (hipExtLaunchKernelGGL), i.e. the kernel's own begin/end timestamps on the
// stream it runs on -- what bench.py reports as roofline.avg_launch_us.
struct TimedLaunch {
    hipEvent_t start, stop;
    int64_t units;
    int32_t kind;
};
bool g_profile = false;
std::vector<TimedLaunch> g_timed;

struct GammaPow {
    double g[PFRL_MAX_NSTEP + 1];
};

template <typename ActT>
__global__ __launch_bounds__(kThreads) void k_table_append(pfrl_table_t tab, int64_t n_rows,
                                                           const int32_t *__restrict__ t_slots,
                                                           const int32_t *__restrict__ state_ref,
                                                           const int32_t *__restrict__ next_ref,
                                                           const ActT *__restrict__ action,
                                                           const double *__restrict__ reward,
                                                           const uint8_t *__restrict__ terminal) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_rows) return;
    const int64_t s = t_slots[i];
    for (int j = 0; j < tab.k; ++j) {
        tab.t_state_ref[s * tab.k + j] = state_ref[i * tab.k + j];
        tab.t_next_ref[s * tab.k + j] = next_ref[i * tab.k + j];
    }
    const int ad = tab.act_dim > 0 ? tab.act_dim : 1;
    ActT *dst = reinterpret_cast<ActT *>(tab.t_action);
    for (int j = 0; j < ad; ++j) dst[s * ad + j] = action[i * ad + j];
    tab.t_reward[s] = reward[i];
    tab.t_terminal[s] = terminal[i];
}

__global__ __launch_bounds__(kThreads) void k_entries_append(pfrl_table_t tab, int64_t n_rows,
                                                             const int32_t *__restrict__ e_slots,
                                                             const int32_t *__restrict__ tids,
                                                             const int32_t *__restrict__ lens) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_rows) return;
    const int64_t s = e_slots[i];
    for (int j = 0; j < tab.n; ++j) tab.e_tids[s * tab.n + j] = tids[i * tab.n + j];
    tab.e_len[s] = lens[i];
}

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

// MODE 0: u8 / divisor, 1: u8 cast only, 2: f32 copy.
// NT: non-temporal stores for launches whose output (hundreds of MB) cannot stay
// in L2/MALL anyway: +11 % on the 578 MB step-fused gather (5.45 vs 4.92 TB/s).
template <int MODE, bool NT>
__device__ __forceinline__ void move_frame(const uint8_t *__restrict__ src8,
                                           uint8_t *__restrict__ dst8, int64_t frame_bytes,
                                           float d) {
    const int tid = threadIdx.x;
    if (MODE == 2) {
        if ((frame_bytes & 15) == 0) {
            const uint4 *s = reinterpret_cast<const uint4 *>(src8);
            uint4 *o = reinterpret_cast<uint4 *>(dst8);
            const int nv = (int)(frame_bytes >> 4);
            for (int i = tid; i < nv; i += kThreads) o[i] = s[i];
        } else {
            const uint32_t *s = reinterpret_cast<const uint32_t *>(src8);
            uint32_t *o = reinterpret_cast<uint32_t *>(dst8);
            const int nv = (int)(frame_bytes >> 2);
            for (int i = tid; i < nv; i += kThreads) o[i] = s[i];
        }
        return;
    }
    const uint32_t *src = reinterpret_cast<const uint32_t *>(src8);
    float4 *dst = reinterpret_cast<float4 *>(dst8);
    const int nd = (int)(frame_bytes >> 2);
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
                const float4 o = (MODE == 0) ? cvt_div(w[u], d) : cvt_only(w[u]);
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

// grid.x = 2*B*k frame blocks followed by ceil(B/256) scalar blocks.
template <int MODE, typename ActT, bool NT>
__global__ __launch_bounds__(kThreads) void k_batch_experiences(
    pfrl_table_t tab, const uint8_t *__restrict__ frames, int64_t frame_bytes, float divisor,
    const int32_t *__restrict__ entry_slots, int64_t B, GammaPow gp, uint8_t *__restrict__ out_state,
    uint8_t *__restrict__ out_next, ActT *__restrict__ out_action, float *__restrict__ out_reward,
    float *__restrict__ out_terminal, float *__restrict__ out_discount) {
    const int64_t nf = B * tab.k;
    const int64_t f = blockIdx.x;
    const int64_t out_frame_bytes = (MODE == 2) ? frame_bytes : 4 * frame_bytes;
    if (f < 2 * nf) {
        const bool is_next = f >= nf;
        const int64_t ff = is_next ? f - nf : f;
        const int64_t b = ff / tab.k;
        const int j = (int)(ff - b * tab.k);
        const int64_t e = entry_slots[b];
        int64_t slot;
        if (!is_next) {
            const int64_t t0 = tab.e_tids[e * tab.n];
            slot = tab.t_state_ref[t0 * tab.k + j];
        } else {
            const int len = tab.e_len[e];
            const int64_t tl = tab.e_tids[e * tab.n + len - 1];
            slot = tab.t_next_ref[tl * tab.k + j];
        }
        uint8_t *dst = (is_next ? out_next : out_state) + ff * out_frame_bytes;
        move_frame<MODE, NT>(frames + slot * frame_bytes, dst, frame_bytes, divisor);
        return;
    }
    // scalar collapse
    const int64_t b = (f - 2 * nf) * kThreads + threadIdx.x;
    if (b >= B) return;
    const int64_t e = entry_slots[b];
    const int len = tab.e_len[e];
    double acc = 0.0;
    bool any = false;
    for (int i = 0; i < len; ++i) {
        const int64_t t = tab.e_tids[e * tab.n + i];
        acc = __dadd_rn(acc, __dmul_rn(gp.g[i], tab.t_reward[t]));
        any |= tab.t_terminal[t] != 0;
    }
    out_reward[b] = (float)acc;
    out_terminal[b] = any ? 1.0f : 0.0f;
    out_discount[b] = (float)gp.g[len];
    const int64_t t0 = tab.e_tids[e * tab.n];
    const int ad = tab.act_dim > 0 ? tab.act_dim : 1;
    const ActT *src = reinterpret_cast<const ActT *>(tab.t_action);
    for (int i = 0; i < ad; ++i) out_action[b * ad + i] = src[t0 * ad + i];
}

// Ragged gather of sampled EPISODES (pfrl/replay_buffer.py:219-287 batch_recurrent_experiences
// over pfrl/replay_buffers/episodic.py:48-85 sample_episodes): an episode is a run of consecutive
// one-transition entries of the entry ring, (first entry slot, length); the launch gets the
// sampled windows sorted by descending length and emits
//   state / next_state  episode-major: the rows of episode 0, then episode 1, ... (what the
//                       reference builds per episode with batch_states and the agent then packs)
//   action / reward / is_state_terminal / discount  time-major packed: all episodes' step 0,
//                       then every episode longer than 1 at step 1, ... (flatten_sequences_time_first)
// ep_row0[e] = first episode-major row of episode e, row_start[t] = first time-major row of step t
// (row_start[T] = rows).  MODE as k_batch_experiences.
struct EpisodeArgs {
    const int32_t *ep_first;   // [n_eps] entry slot of the window's first transition
    const int32_t *ep_row0;    // [n_eps + 1]
    const int32_t *row_start;  // [T + 1]
    int32_t n_eps, T;
    int64_t rows, E;
};

__device__ __forceinline__ int upper_index(const int32_t *__restrict__ a, int n, int64_t v) {
    // largest i in [0, n) with a[i] <= v  (a ascending, a[0] <= v)
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a[mid] <= v) lo = mid; else hi = mid - 1;
    }
    return lo;
}

template <int MODE, typename ActT>
__global__ __launch_bounds__(kThreads) void k_batch_episodes(
    pfrl_table_t tab, const uint8_t *__restrict__ frames, int64_t frame_bytes, float divisor,
    EpisodeArgs ea, float gamma, uint8_t *__restrict__ out_state, uint8_t *__restrict__ out_next,
    ActT *__restrict__ out_action, float *__restrict__ out_reward, float *__restrict__ out_terminal,
    float *__restrict__ out_discount) {
    const int64_t nf = ea.rows * tab.k;
    const int64_t f = blockIdx.x;
    const int64_t out_frame_bytes = MODE == 2 ? frame_bytes : 4 * frame_bytes;
    if (f < 2 * nf) {
        const bool is_next = f >= nf;
        const int64_t ff = is_next ? f - nf : f;
        const int64_t row = ff / tab.k;
        const int j = (int)(ff - row * tab.k);
        const int e = upper_index(ea.ep_row0, ea.n_eps, row);
        const int64_t es = ((int64_t)ea.ep_first[e] + (row - ea.ep_row0[e])) % ea.E;
        const int64_t t = tab.e_tids[es * tab.n];
        const int64_t slot = is_next ? tab.t_next_ref[t * tab.k + j] : tab.t_state_ref[t * tab.k + j];
        uint8_t *dst = (is_next ? out_next : out_state) + ff * out_frame_bytes;
        move_frame<MODE, false>(frames + slot * frame_bytes, dst, frame_bytes, divisor);
        return;
    }
    const int64_t r = (f - 2 * nf) * kThreads + threadIdx.x;
    if (r >= ea.rows) return;
    const int step = upper_index(ea.row_start, ea.T, r);
    const int b = (int)(r - ea.row_start[step]);          // b-th longest episode, at this step
    const int64_t es = ((int64_t)ea.ep_first[b] + step) % ea.E;
    const int64_t t = tab.e_tids[es * tab.n];
    out_reward[r] = (float)tab.t_reward[t];
    out_terminal[r] = tab.t_terminal[t] != 0 ? 1.0f : 0.0f;
    out_discount[r] = gamma;
    const int ad = tab.act_dim > 0 ? tab.act_dim : 1;
    const ActT *src = reinterpret_cast<const ActT *>(tab.t_action);
    for (int i = 0; i < ad; ++i) out_action[r * ad + i] = src[t * ad + i];
}

// Persistent form of the frame part: P workgroups walk the 2*B*k output frames
// with stride P.  The slot of the NEXT frame is resolved (three dependent scalar
// loads) and its dwords are requested before the current frame's stores are
// issued, so neither the index chain nor the HBM read latency is exposed between
// frames, and no workgroup launch/drain happens per 7 KB of input.
// Requires frame_bytes / 4 <= kThreads * kUnroll (one pass per frame) and u8 input.
template <int MODE, typename ActT, bool NT>
__global__ __launch_bounds__(kThreads) void k_batch_experiences_persist(
    pfrl_table_t tab, const uint8_t *__restrict__ frames, int64_t frame_bytes, float divisor,
    const int32_t *__restrict__ entry_slots, int64_t B, GammaPow gp, uint8_t *__restrict__ out_state,
    uint8_t *__restrict__ out_next, ActT *__restrict__ out_action, float *__restrict__ out_reward,
    float *__restrict__ out_terminal, float *__restrict__ out_discount, int P) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= P) {
        // scalar collapse (same as k_batch_experiences)
        const int64_t b = (int64_t)(blockIdx.x - P) * kThreads + tid;
        if (b >= B) return;
        const int64_t e = entry_slots[b];
        const int len = tab.e_len[e];
        double acc = 0.0;
        bool any = false;
        for (int i = 0; i < len; ++i) {
            const int64_t t = tab.e_tids[e * tab.n + i];
            acc = __dadd_rn(acc, __dmul_rn(gp.g[i], tab.t_reward[t]));
            any |= tab.t_terminal[t] != 0;
        }
        out_reward[b] = (float)acc;
        out_terminal[b] = any ? 1.0f : 0.0f;
        out_discount[b] = (float)gp.g[len];
        const int64_t t0 = tab.e_tids[e * tab.n];
        const int ad = tab.act_dim > 0 ? tab.act_dim : 1;
        const ActT *src = reinterpret_cast<const ActT *>(tab.t_action);
        for (int i = 0; i < ad; ++i) out_action[b * ad + i] = src[t0 * ad + i];
        return;
    }
    const int64_t nf = B * tab.k;
    const int64_t total = 2 * nf;
    const int nd = (int)(frame_bytes >> 2);
    const int64_t out_frame_bytes = 4 * frame_bytes;

    auto resolve = [&](int64_t f, const uint32_t *&src, float4 *&dst) {
        const bool is_next = f >= nf;
        const int64_t ff = is_next ? f - nf : f;
        const int64_t b = ff / tab.k;
        const int j = (int)(ff - b * tab.k);
        const int64_t e = entry_slots[b];
        int64_t slot;
        if (!is_next) {
            slot = tab.t_state_ref[(int64_t)tab.e_tids[e * tab.n] * tab.k + j];
        } else {
            const int len = tab.e_len[e];
            slot = tab.t_next_ref[(int64_t)tab.e_tids[e * tab.n + len - 1] * tab.k + j];
        }
        src = reinterpret_cast<const uint32_t *>(frames + slot * frame_bytes);
        dst = reinterpret_cast<float4 *>((is_next ? out_next : out_state) + ff * out_frame_bytes);
    };
    auto load = [&](const uint32_t *src, uint32_t (&w)[kUnroll]) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int i = u * kThreads + tid;
            w[u] = src[min(i, nd - 1)];   // unconditional: all loads in one basic block
        }
    };
    auto store = [&](float4 *dst, const uint32_t (&w)[kUnroll]) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int i = u * kThreads + tid;
            if (i < nd) {
                const float4 o = (MODE == 0) ? cvt_div(w[u], divisor) : cvt_only(w[u]);
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
    };

    int64_t f = blockIdx.x;
    if (f >= total) return;
    const uint32_t *src;
    float4 *dst;
    uint32_t w[kUnroll];
    resolve(f, src, dst);
    load(src, w);
    for (;;) {
        const int64_t fn = f + P;
        if (fn >= total) {
            store(dst, w);
            break;
        }
        const uint32_t *src_n;
        float4 *dst_n;
        uint32_t wn[kUnroll];
        resolve(fn, src_n, dst_n);
        load(src_n, wn);
        store(dst, w);
        f = fn;
        dst = dst_n;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) w[u] = wn[u];
    }
}

// Channels-last form (k == 4, u8 frames): grid.x = 2*B*tiles observation-tile blocks
// followed by ceil(B/256) scalar blocks; see nhwc.h for the tile scheme.
template <int MODE, typename ActT, bool NT>
__global__ __launch_bounds__(kThreads) void k_batch_experiences_nhwc4(
    pfrl_table_t tab, const uint8_t *__restrict__ frames, int64_t frame_bytes, float divisor,
    const int32_t *__restrict__ entry_slots, int64_t B, GammaPow gp, uint8_t *__restrict__ out_state,
    uint8_t *__restrict__ out_next, ActT *__restrict__ out_action, float *__restrict__ out_reward,
    float *__restrict__ out_terminal, float *__restrict__ out_discount, int tiles) {
    const int64_t nt_blocks = 2 * B * tiles;
    // scalar-collapse workgroups (serial chains of dependent loads) come FIRST in the grid so
    // that they run under the streaming tiles instead of forming the tail of the launch
    const int64_t scalar_blocks = (int64_t)gridDim.x - nt_blocks;
    if ((int64_t)blockIdx.x >= scalar_blocks) {
        const int64_t tb = (int64_t)blockIdx.x - scalar_blocks;
        const int64_t ot = tb / tiles;                       // observation index in [0, 2B)
        const int tile = (int)(tb - ot * tiles);
        const bool is_next = ot >= B;
        const int64_t b = is_next ? ot - B : ot;
        const int64_t e = entry_slots[b];
        const int32_t *r;
        if (!is_next) {
            r = tab.t_state_ref + (int64_t)tab.e_tids[e * tab.n] * 4;
        } else {
            const int len = tab.e_len[e];
            r = tab.t_next_ref + (int64_t)tab.e_tids[e * tab.n + len - 1] * 4;
        }
        const uint8_t *f0 = frames + (int64_t)r[0] * frame_bytes;
        const uint8_t *f1 = frames + (int64_t)r[1] * frame_bytes;
        const uint8_t *f2 = frames + (int64_t)r[2] * frame_bytes;
        const uint8_t *f3 = frames + (int64_t)r[3] * frame_bytes;
        float4 *dst = reinterpret_cast<float4 *>((is_next ? out_next : out_state) +
                                                 b * 16 * frame_bytes);
        pfrl_nhwc::convert_tile<MODE == 0, NT>(f0, f1, f2, f3, dst, tile, (int)frame_bytes,
                                               divisor);
        return;
    }
    const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (b >= B) return;
    const int64_t e = entry_slots[b];
    const int len = tab.e_len[e];
    double acc = 0.0;
    bool any = false;
    for (int i = 0; i < len; ++i) {
        const int64_t t = tab.e_tids[e * tab.n + i];
        acc = __dadd_rn(acc, __dmul_rn(gp.g[i], tab.t_reward[t]));
        any |= tab.t_terminal[t] != 0;
    }
    out_reward[b] = (float)acc;
    out_terminal[b] = any ? 1.0f : 0.0f;
    out_discount[b] = (float)gp.g[len];
    const int64_t t0 = tab.e_tids[e * tab.n];
    const int ad = tab.act_dim > 0 ? tab.act_dim : 1;
    const ActT *src = reinterpret_cast<const ActT *>(tab.t_action);
    for (int i = 0; i < ad; ++i) out_action[b * ad + i] = src[t0 * ad + i];
}

int pfrl_gather_persist_blocks() {
    static const int v = [] {
        const char *e = getenv("PFRL_GATHER_PERSIST");
        return e != nullptr && *e ? atoi(e) : 0;
    }();
    return v;
}

}  // namespace

extern "C" int pfrl_table_append(const pfrl_table_t *tab, int64_t n_rows, const int32_t *t_slots,
                                 const int32_t *state_ref, const int32_t *next_ref,
                                 const void *action, const double *reward, const uint8_t *terminal,
                                 void *stream) {
    PFRL_CHECK_ARG(tab && tab->k >= 1 && tab->k <= PFRL_MAX_STACK, "bad table.k");
    if (n_rows <= 0) return 0;
    const unsigned blocks = (unsigned)((n_rows + kThreads - 1) / kThreads);
    if (tab->act_dim > 0)
        hipLaunchKernelGGL(k_table_append<float>, dim3(blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, *tab, n_rows, t_slots, state_ref, next_ref,
                           (const float *)action, reward, terminal);
    else
        hipLaunchKernelGGL(k_table_append<int64_t>, dim3(blocks), dim3(kThreads), 0,
                           (hipStream_t)stream, *tab, n_rows, t_slots, state_ref, next_ref,
                           (const int64_t *)action, reward, terminal);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_entries_append(const pfrl_table_t *tab, int64_t n_rows, const int32_t *e_slots,
                                   const int32_t *tids, const int32_t *lens, void *stream) {
    PFRL_CHECK_ARG(tab && tab->n >= 1 && tab->n <= PFRL_MAX_NSTEP, "bad table.n");
    if (n_rows <= 0) return 0;
    const unsigned blocks = (unsigned)((n_rows + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(k_entries_append, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, *tab,
                       n_rows, e_slots, tids, lens);
    PFRL_LAUNCH_CHECK();
}

// f32 vector observations of a few KB (MuJoCo-shaped: 376 floats): one WAVE per output frame
// instead of one workgroup (94 of 256 lanes busy, and a three-deep index chain per
// workgroup).  Each wave resolves kFPW frames, issues all their loads, then stores; the
// scalar collapse rides in extra workgroups as above.  frame_bytes % 16 == 0.
// Measured on the 14 336-entry launch of the SAC step (88.5 MB): kFPW = 8 / 4 / 2 / 1 ->
// 24.2 / 22.1-23.4 / 21.9 / 20.8 us: more, shorter waves hide the index chain better than more
// loads per wave.
constexpr int kFPW = 1;

template <typename ActT>
__global__ __launch_bounds__(kThreads) void k_batch_experiences_f32_small(
    pfrl_table_t tab, const uint8_t *__restrict__ frames, int64_t frame_bytes,
    const int32_t *__restrict__ entry_slots, int64_t B, GammaPow gp, uint8_t *__restrict__ out_state,
    uint8_t *__restrict__ out_next, ActT *__restrict__ out_action, float *__restrict__ out_reward,
    float *__restrict__ out_terminal, float *__restrict__ out_discount, int frame_blocks) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the scalar collapse (a serial chain of dependent loads per thread, 17 strided action
    // floats) goes FIRST in the grid: dispatched last it would be the tail of the launch
    const int scalar_blocks = (int)gridDim.x - frame_blocks;
    if ((int)blockIdx.x < scalar_blocks) {
        const int64_t b = (int64_t)blockIdx.x * kThreads + tid;
        if (b >= B) return;
        const int64_t e = entry_slots[b];
        const int len = tab.e_len[e];
        double acc = 0.0;
        bool any = false;
        for (int i = 0; i < len; ++i) {
            const int64_t t = tab.e_tids[e * tab.n + i];
            acc = __dadd_rn(acc, __dmul_rn(gp.g[i], tab.t_reward[t]));
            any |= tab.t_terminal[t] != 0;
        }
        out_reward[b] = (float)acc;
        out_terminal[b] = any ? 1.0f : 0.0f;
        out_discount[b] = (float)gp.g[len];
        const int64_t t0 = tab.e_tids[e * tab.n];
        const int ad = tab.act_dim > 0 ? tab.act_dim : 1;
        const ActT *src = reinterpret_cast<const ActT *>(tab.t_action);
        for (int i = 0; i < ad; ++i) out_action[b * ad + i] = src[t0 * ad + i];
        return;
    }
    const int64_t nf = B * tab.k, total = 2 * nf;
    const int nv = (int)(frame_bytes >> 4);            // uint4 per frame
    const int64_t f0 = ((int64_t)(blockIdx.x - scalar_blocks) * 4 + wave) * kFPW;
    const uint4 *src[kFPW];
    uint4 *dst[kFPW];
#pragma unroll
    for (int u = 0; u < kFPW; ++u) {
        const int64_t f = f0 + u < total ? f0 + u : total - 1;
        const bool is_next = f >= nf;
        const int64_t ff = is_next ? f - nf : f;
        const int64_t b = ff / tab.k;
        const int j = (int)(ff - b * tab.k);
        const int64_t e = entry_slots[b];
        int64_t slot;
        if (!is_next) {
            slot = tab.t_state_ref[(int64_t)tab.e_tids[e * tab.n] * tab.k + j];
        } else {
            const int len = tab.e_len[e];
            slot = tab.t_next_ref[(int64_t)tab.e_tids[e * tab.n + len - 1] * tab.k + j];
        }
        src[u] = reinterpret_cast<const uint4 *>(frames + slot * frame_bytes);
        dst[u] = reinterpret_cast<uint4 *>((is_next ? out_next : out_state) + ff * frame_bytes);
    }
    for (int base = 0; base < nv; base += 128) {
        uint4 v[kFPW][2];
#pragma unroll
        for (int u = 0; u < kFPW; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[u][h] = src[u][min(base + h * 64 + lane, nv - 1)];
#pragma unroll
        for (int u = 0; u < kFPW; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = base + h * 64 + lane;
                if (i < nv && f0 + u < total) dst[u][i] = v[u][h];
            }
    }
}

template <int MODE, bool NT>
static void launch_be2(const pfrl_table_t *tab, const void *frames, int64_t frame_bytes,
                       float divisor, const int32_t *entry_slots, int64_t B, const GammaPow &gp,
                       float *out_state, float *out_next_state, void *out_action,
                       float *out_reward, float *out_terminal, float *out_discount,
                       hipStream_t stream) {
    const unsigned blocks = (unsigned)(2 * B * tab->k + (B + kThreads - 1) / kThreads);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    pfrl_profile_events(PFRL_PROFILE_BATCH_EXPERIENCES, B, &e0, &e1);
    if constexpr (MODE == 2) {
        if ((frame_bytes & 15) == 0 && frame_bytes <= 8192 && 2 * B * tab->k >= 4096) {
            const int fb = (int)((2 * B * tab->k + 4 * kFPW - 1) / (4 * kFPW));
            const unsigned gb = (unsigned)(fb + (B + kThreads - 1) / kThreads);
            if (tab->act_dim > 0)
                hipExtLaunchKernelGGL((k_batch_experiences_f32_small<float>), dim3(gb), dim3(kThreads),
                                      0, stream, e0, e1, 0, *tab, (const uint8_t *)frames, frame_bytes,
                                      entry_slots, B, gp, (uint8_t *)out_state,
                                      (uint8_t *)out_next_state, (float *)out_action, out_reward,
                                      out_terminal, out_discount, fb);
            else
                hipExtLaunchKernelGGL((k_batch_experiences_f32_small<int64_t>), dim3(gb),
                                      dim3(kThreads), 0, stream, e0, e1, 0, *tab,
                                      (const uint8_t *)frames, frame_bytes, entry_slots, B, gp,
                                      (uint8_t *)out_state, (uint8_t *)out_next_state,
                                      (int64_t *)out_action, out_reward, out_terminal, out_discount,
                                      fb);
            return;
        }
    }
    if constexpr (MODE != 2) {
        const int P = pfrl_gather_persist_blocks();
        if (P > 0 && (frame_bytes >> 2) <= kThreads * kUnroll && 2 * B * tab->k > P) {
            const unsigned pb = (unsigned)(P + (B + kThreads - 1) / kThreads);
            if (tab->act_dim > 0)
                hipExtLaunchKernelGGL((k_batch_experiences_persist<MODE, float, NT>), dim3(pb),
                                      dim3(kThreads), 0, stream, e0, e1, 0, *tab,
                                      (const uint8_t *)frames, frame_bytes, divisor, entry_slots, B,
                                      gp, (uint8_t *)out_state, (uint8_t *)out_next_state,
                                      (float *)out_action, out_reward, out_terminal, out_discount,
                                      P);
            else
                hipExtLaunchKernelGGL((k_batch_experiences_persist<MODE, int64_t, NT>), dim3(pb),
                                      dim3(kThreads), 0, stream, e0, e1, 0, *tab,
                                      (const uint8_t *)frames, frame_bytes, divisor, entry_slots, B,
                                      gp, (uint8_t *)out_state, (uint8_t *)out_next_state,
                                      (int64_t *)out_action, out_reward, out_terminal,
                                      out_discount, P);
            return;
        }
    }
    if (tab->act_dim > 0)
        hipExtLaunchKernelGGL((k_batch_experiences<MODE, float, NT>), dim3(blocks), dim3(kThreads),
                              0, stream, e0, e1, 0, *tab, (const uint8_t *)frames, frame_bytes,
                              divisor, entry_slots, B, gp, (uint8_t *)out_state,
                              (uint8_t *)out_next_state, (float *)out_action, out_reward,
                              out_terminal, out_discount);
    else
        hipExtLaunchKernelGGL((k_batch_experiences<MODE, int64_t, NT>), dim3(blocks),
                              dim3(kThreads), 0, stream, e0, e1, 0, *tab, (const uint8_t *)frames,
                              frame_bytes, divisor, entry_slots, B, gp, (uint8_t *)out_state,
                              (uint8_t *)out_next_state, (int64_t *)out_action, out_reward,
                              out_terminal, out_discount);
}

template <int MODE>
static void launch_be(const pfrl_table_t *tab, const void *frames, int64_t frame_bytes,
                      float divisor, const int32_t *entry_slots, int64_t B, const GammaPow &gp,
                      float *out_state, float *out_next_state, void *out_action, float *out_reward,
                      float *out_terminal, float *out_discount, hipStream_t stream) {
    // output of this launch (both stacks); beyond ~128 MB it cannot be cache resident
    const int64_t out_bytes = 2 * B * tab->k * frame_bytes * (MODE == 2 ? 1 : 4);
    if (MODE != 2 && out_bytes >= pfrl_nt_min_bytes())
        launch_be2<MODE, true>(tab, frames, frame_bytes, divisor, entry_slots, B, gp, out_state,
                               out_next_state, out_action, out_reward, out_terminal, out_discount,
                               stream);
    else
        launch_be2<MODE, false>(tab, frames, frame_bytes, divisor, entry_slots, B, gp, out_state,
                                out_next_state, out_action, out_reward, out_terminal, out_discount,
                                stream);
}

extern "C" int pfrl_batch_experiences(const pfrl_table_t *tab, const void *frames,
                                      int64_t frame_bytes, int frame_is_f32, float divisor,
                                      const int32_t *entry_slots, int64_t B,
                                      const double *host_gamma_pow, float *out_state,
                                      float *out_next_state, void *out_action, float *out_reward,
                                      float *out_terminal, float *out_discount, void *stream) {
    PFRL_CHECK_ARG(tab && tab->k >= 1 && tab->k <= PFRL_MAX_STACK, "bad table.k");
    PFRL_CHECK_ARG(tab->n >= 1 && tab->n <= PFRL_MAX_NSTEP, "bad table.n");
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    if (B <= 0) return 0;
    GammaPow gp;
    for (int i = 0; i <= tab->n; ++i) gp.g[i] = host_gamma_pow[i];
    hipStream_t s = (hipStream_t)stream;
    if (frame_is_f32)
        launch_be<2>(tab, frames, frame_bytes, divisor, entry_slots, B, gp, out_state,
                     out_next_state, out_action, out_reward, out_terminal, out_discount, s);
    else if (divisor == 1.0f)
        launch_be<1>(tab, frames, frame_bytes, divisor, entry_slots, B, gp, out_state,
                     out_next_state, out_action, out_reward, out_terminal, out_discount, s);
    else
        launch_be<0>(tab, frames, frame_bytes, divisor, entry_slots, B, gp, out_state,
                     out_next_state, out_action, out_reward, out_terminal, out_discount, s);
    PFRL_LAUNCH_CHECK();
}

int64_t pfrl_nt_min_bytes() {
    static const int64_t v = [] {
        const char *e = getenv("PFRL_NT_MIN_BYTES");
        return e != nullptr && *e ? (int64_t)atoll(e) : ((int64_t)32 << 20);
    }();
    return v;
}

// Shared with frames.hip (declared in common.h): hands out an event pair for one
// dispatch when profiling is on, nullptrs otherwise.
void pfrl_profile_events(int kind, int64_t units, hipEvent_t *start, hipEvent_t *stop) {
    *start = *stop = nullptr;
    if (!g_profile) return;
    if (hipEventCreate(start) != hipSuccess || hipEventCreate(stop) != hipSuccess) {
        *start = *stop = nullptr;
        return;
    }
    g_timed.push_back({*start, *stop, units, (int32_t)kind});
}

extern "C" int pfrl_profile_enable(int on) {
    g_profile = on != 0;
    return 0;
}

// Waits for the timed launches, writes their durations (microseconds) and entry
// counts and kinds, frees the events.  Returns the number of launches written.
extern "C" int64_t pfrl_profile_collect(double *out_us, int64_t *out_units, int32_t *out_kind,
                                        int64_t cap) {
    int64_t n = 0;
    for (auto &t : g_timed) {
        (void)hipEventSynchronize(t.stop);
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess && n < cap) {
            out_us[n] = (double)ms * 1e3;
            out_units[n] = t.units;
            out_kind[n] = t.kind;
            ++n;
        }
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    g_timed.clear();
    return n;
}

template <int MODE, bool NT>
static void launch_be_nhwc4(const pfrl_table_t *tab, const void *frames, int64_t frame_bytes,
                            float divisor, const int32_t *entry_slots, int64_t B,
                            const GammaPow &gp, float *out_state, float *out_next_state,
                            void *out_action, float *out_reward, float *out_terminal,
                            float *out_discount, hipStream_t stream) {
    const int tiles = (int)((frame_bytes + pfrl_nhwc::kTilePixels - 1) / pfrl_nhwc::kTilePixels);
    const unsigned blocks = (unsigned)(2 * B * tiles + (B + kThreads - 1) / kThreads);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    pfrl_profile_events(PFRL_PROFILE_BATCH_EXPERIENCES, B, &e0, &e1);
    if (tab->act_dim > 0)
        hipExtLaunchKernelGGL((k_batch_experiences_nhwc4<MODE, float, NT>), dim3(blocks),
                              dim3(kThreads), 0, stream, e0, e1, 0, *tab, (const uint8_t *)frames,
                              frame_bytes, divisor, entry_slots, B, gp, (uint8_t *)out_state,
                              (uint8_t *)out_next_state, (float *)out_action, out_reward,
                              out_terminal, out_discount, tiles);
    else
        hipExtLaunchKernelGGL((k_batch_experiences_nhwc4<MODE, int64_t, NT>), dim3(blocks),
                              dim3(kThreads), 0, stream, e0, e1, 0, *tab, (const uint8_t *)frames,
                              frame_bytes, divisor, entry_slots, B, gp, (uint8_t *)out_state,
                              (uint8_t *)out_next_state, (int64_t *)out_action, out_reward,
                              out_terminal, out_discount, tiles);
}

extern "C" int pfrl_batch_experiences_nhwc4(const pfrl_table_t *tab, const void *frames,
                                            int64_t frame_bytes, float divisor,
                                            const int32_t *entry_slots, int64_t B,
                                            const double *host_gamma_pow, float *out_state,
                                            float *out_next_state, void *out_action,
                                            float *out_reward, float *out_terminal,
                                            float *out_discount, void *stream) {
    PFRL_CHECK_ARG(tab && tab->k == 4, "pfrl_batch_experiences_nhwc4: stacks of 4 frames only");
    PFRL_CHECK_ARG(tab->n >= 1 && tab->n <= PFRL_MAX_NSTEP, "bad table.n");
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    if (B <= 0) return 0;
    GammaPow gp;
    for (int i = 0; i <= tab->n; ++i) gp.g[i] = host_gamma_pow[i];
    hipStream_t s = (hipStream_t)stream;
    const bool nt = 2 * B * 16 * frame_bytes >= pfrl_nt_min_bytes();
    if (divisor == 1.0f) {
        if (nt)
            launch_be_nhwc4<1, true>(tab, frames, frame_bytes, divisor, entry_slots, B, gp,
                                     out_state, out_next_state, out_action, out_reward,
                                     out_terminal, out_discount, s);
        else
            launch_be_nhwc4<1, false>(tab, frames, frame_bytes, divisor, entry_slots, B, gp,
                                      out_state, out_next_state, out_action, out_reward,
                                      out_terminal, out_discount, s);
    } else {
        if (nt)
            launch_be_nhwc4<0, true>(tab, frames, frame_bytes, divisor, entry_slots, B, gp,
                                     out_state, out_next_state, out_action, out_reward,
                                     out_terminal, out_discount, s);
        else
            launch_be_nhwc4<0, false>(tab, frames, frame_bytes, divisor, entry_slots, B, gp,
                                      out_state, out_next_state, out_action, out_reward,
                                      out_terminal, out_discount, s);
    }
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_batch_episodes(const pfrl_table_t *tab, const void *frames, int64_t frame_bytes,
                                   int frames_are_f32, float divisor, const int32_t *ep_first,
                                   const int32_t *ep_row0, const int32_t *row_start, int32_t n_eps,
                                   int32_t T, int64_t rows, int64_t entry_ring, float gamma,
                                   void *out_state, void *out_next_state, void *out_action,
                                   float *out_reward, float *out_terminal, float *out_discount,
                                   void *stream) {
    PFRL_CHECK_ARG(tab && tab->n >= 1 && tab->k >= 1 && tab->k <= PFRL_MAX_STACK, "bad table");
    PFRL_CHECK_ARG(frame_bytes > 0 && (frame_bytes & 3) == 0, "frame_bytes must be a multiple of 4");
    PFRL_CHECK_ARG(n_eps >= 1 && T >= 1 && rows >= 1 && entry_ring >= 1 && ep_first && ep_row0 &&
                       row_start, "pfrl_batch_episodes: bad episode table");
    EpisodeArgs ea{ep_first, ep_row0, row_start, n_eps, T, rows, entry_ring};
    const int64_t frame_blocks = 2 * rows * tab->k;
    const dim3 grid((unsigned)(frame_blocks + (rows + kThreads - 1) / kThreads)), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
    const uint8_t *fr = (const uint8_t *)frames;
    uint8_t *os = (uint8_t *)out_state, *on = (uint8_t *)out_next_state;
#define EP_LAUNCH(MODE, ACT)                                                                     \
    hipLaunchKernelGGL((k_batch_episodes<MODE, ACT>), grid, block, 0, st, *tab, fr, frame_bytes, \
                       divisor, ea, gamma, os, on, (ACT *)out_action, out_reward, out_terminal,  \
                       out_discount)
    const int mode = frames_are_f32 ? 2 : (divisor == 1.0f ? 1 : 0);
    if (tab->act_dim > 0) {
        if (mode == 2) EP_LAUNCH(2, float); else if (mode == 1) EP_LAUNCH(1, float); else EP_LAUNCH(0, float);
    } else {
        if (mode == 2) EP_LAUNCH(2, int64_t); else if (mode == 1) EP_LAUNCH(1, int64_t); else EP_LAUNCH(0, int64_t);
    }
#undef EP_LAUNCH
    PFRL_LAUNCH_CHECK();
}
