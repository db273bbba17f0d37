// Prioritized-replay sum/min trees (pfrl/collections/prioritized.py:135-323)
// as per-level rings of NEP-50 tagged nodes in HBM.
//
// Every node value is a pure function of its two children
//   sum: ((0 + left) + right) over the children present   (prioritized.py:140-151)
//   min: left unless right < left
// so a batch of leaf writes followed by a bottom-up re-reduction of the touched
// paths yields exactly the state the reference reaches with its sequential
// _write calls.  Structure changes (frame doubling / halving / re-rooting,
// prioritized.py:207-242) are integer bookkeeping done by the host and handed
// over in the pfrl_tree_t descriptor; the host starts a new launch whenever
// the frame changes, so all leaves of one launch share one frame.
//
// These kernels are latency-bound (B sequential 20-level descents), not
// bandwidth-bound: ~27 KB of tree traffic per DQN update.
#include <stdlib.h>

#include "common.h"
#include "powf_glibc.h"

namespace {

constexpr int kMaxBatch = 1024;

__device__ __forceinline__ int64_t node_idx(const pfrl_tree_t &T, int l, int64_t x) {
    const int sh = T.log2_smax - l;
    const int64_t M = sh > 0 ? ((int64_t)1 << sh) : 1;
    const int64_t q = (x - T.origin[l]) >> l;
    return T.level_off[l] + (q & (M - 1));
}

__device__ __forceinline__ TV reduce_sum(TV l, TV r) {
    const bool lp = l.t != PFRL_TAG_ABSENT, rp = r.t != PFRL_TAG_ABSENT;
    const TV both = tv_add(l, r);
    const TV one = lp ? l : r;   // also the absent (0, tag 0) result when neither is present
    return (lp && rp) ? both : mk_tv(one.t != PFRL_TAG_ABSENT ? one.v : 0.0, one.t);
}

__device__ __forceinline__ TV reduce_min(TV l, TV r) {
    const bool lp = l.t != PFRL_TAG_ABSENT, rp = r.t != PFRL_TAG_ABSENT;
    const TV m = tv_lt(r, l) ? r : l;
    const TV one = lp ? l : r;
    return (lp && rp) ? m : mk_tv(one.t != PFRL_TAG_ABSENT ? one.v : 0.0, one.t);
}

// Re-reduce the ancestors of leaf x (levels 1..log2_size) in both trees.
// Must be called by every thread of the (single) workgroup.
__device__ void repair_paths(const pfrl_tree_t &T, bool active, int64_t x) {
    const int L = T.log2_size;
    if (active && (x < T.base || x >= T.base + ((int64_t)1 << L))) active = false;
    for (int l = 1; l <= L; ++l) {
        __threadfence_block();
        __syncthreads();
        if (active) {
            const int64_t half = (int64_t)1 << (l - 1);
            const int64_t xl = x - ((x - T.origin[l]) & (((int64_t)1 << l) - 1));
            const int64_t il = node_idx(T, l - 1, xl);
            const int64_t ir = node_idx(T, l - 1, xl + half);
            const int64_t ip = node_idx(T, l, x);
            TV a = mk_tv(T.sum_val[il], T.sum_tag[il]);
            TV b = mk_tv(T.sum_val[ir], T.sum_tag[ir]);
            TV s = reduce_sum(a, b);
            T.sum_val[ip] = s.v;
            T.sum_tag[ip] = (uint8_t)s.t;
            a = mk_tv(T.min_val[il], T.min_tag[il]);
            b = mk_tv(T.min_val[ir], T.min_tag[ir]);
            TV m = reduce_min(a, b);
            T.min_val[ip] = m.v;
            T.min_tag[ip] = (uint8_t)m.t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// repair_paths without a memory round trip per level (round 5).  repair_paths() walks the levels
// between barriers and every level READS the two children from memory that the level below has
// just STORED: 21 levels x (barrier + store -> fence -> load through L2) = 28 us for a minibatch's
// priorities plus the pending appends, on the chain the next minibatch's draws wait for.  Here a
// thread carries its path node in registers.  What it needs per level is the SIBLING: either
// untouched by this launch -- then its value is in memory already, and the siblings of ALL levels
// are requested up front, one round trip for the whole walk -- or on another thread's path, and
// then that thread publishes its node per level in a small LDS hash table keyed by the node's
// index (three tables in rotation: insert, barrier, look up; the table of two levels ago is
// cleared meanwhile).  Same typed reductions on the same operands as repair_paths: the nodes come
// out bit-identical; threads whose paths merge compute (and store) identical parents.
// Launches of up to kFastThreads = 128 leaves on frames of up to 2^kFastLevels = 2^22 leaves;
// anything else takes repair_paths.
// ---------------------------------------------------------------------------------------------
constexpr int kFastThreads = 128;     // largest launch of the fast kernels
constexpr int kFastLevels = 22;

template <int SLOTS>
struct PathTab {
    int key[SLOTS];
    double sv[SLOTS], mv[SLOTS];
    int tags[SLOTS];      // sum tag | min tag << 8
};

// LDS of the fast kernels (dynamic; 38 KB for launches of up to 64 leaves, 76 KB up to 128 -- the
// smaller the better: the workgroup has to find a CU with that much LDS free beside the backward
// pass running on the other stream): the three tables (2 slots per thread), the prefetched
// siblings [level][thread] and the scratch arrays of set_priorities_leaves
template <int THREADS>
struct FastLds {
    static constexpr int kHashSlots = 2 * THREADS;
    PathTab<2 * THREADS> tab[3];
    double sib_sv[kFastLevels][THREADS], sib_mv[kFastLevels][THREADS];
    int sib_tg[kFastLevels][THREADS];
    double s_v[THREADS];
    int64_t s_x[THREADS];
    uint8_t s_t[THREADS];
};

__device__ __forceinline__ int path_hash(int key) { return (int)(((unsigned)key * 2654435761u) >> 20); }

template <int THREADS>
__device__ void repair_paths_hashed(const pfrl_tree_t &T, bool active, int64_t x, FastLds<THREADS> &S) {
    constexpr int kHashSlots = 2 * THREADS;
    using PathTab = PathTab<2 * THREADS>;
    const int L = T.log2_size;
    const int tid = threadIdx.x;
    if (active && (x < T.base || x >= T.base + ((int64_t)1 << L))) active = false;
    if (!active) x = T.base;
    for (int i = tid; i < 3 * kHashSlots; i += blockDim.x) S.tab[i / kHashSlots].key[i % kHashSlots] = -1;
    __threadfence_block();
    __syncthreads();            // every leaf of this launch is stored; tables empty
    // my leaf as it stands now (a later write of the launch may have replaced what I stored) and
    // the sibling of my path node at every level: all requested before anything is waited for,
    // then parked in LDS (the level loop below is rolled: its 22 typed bodies unrolled are 12 700
    // instructions of once-executed straight-line code, and a cold launch pays an instruction
    // fetch stall per line -- DESIGN_LOG 2a -- while register arrays indexed by a run-time level
    // end up in scratch memory)
    const int64_t ileaf = node_idx(T, 0, x);
    const double leaf_sv = T.sum_val[ileaf], leaf_mv = T.min_val[ileaf];
    const int leaf_st = T.sum_tag[ileaf], leaf_mt = T.min_tag[ileaf];
    if (tid < THREADS) {        // (a wider launch -- the fused sampler -- has its leaves in the first THREADS threads)
        double sv[kFastLevels], mv[kFastLevels];
        int tg[kFastLevels];
#pragma unroll
        for (int l = 0; l < kFastLevels; ++l) {
            const int ll = l < L ? l : 0;               // (levels past the root: a valid dummy address)
            const int64_t half = (int64_t)1 << ll;
            const int64_t xp = x - ((x - T.origin[ll + 1]) & (2 * half - 1));   // parent's span start
            const int64_t xme = x - ((x - T.origin[ll]) & (half - 1));           // my node's span start
            const int64_t is = node_idx(T, ll, xme == xp ? xp + half : xp);
            sv[l] = T.sum_val[is];
            mv[l] = T.min_val[is];
            tg[l] = (int)T.sum_tag[is] | ((int)T.min_tag[is] << 8);
        }
#pragma unroll
        for (int l = 0; l < kFastLevels; ++l) {
            S.sib_sv[l][tid] = sv[l];
            S.sib_mv[l][tid] = mv[l];
            S.sib_tg[l][tid] = tg[l];
        }
    }
    TV cs = mk_tv(leaf_sv, leaf_st), cm = mk_tv(leaf_mv, leaf_mt);
#pragma unroll 1
    for (int l = 0; l < L; ++l) {                    // (L is uniform: the barriers are taken by all)
        PathTab &h = S.tab[l % 3];
        const int64_t half = (int64_t)1 << l;
        const int64_t xp = x - ((x - T.origin[l + 1]) & (2 * half - 1));
        const int64_t xme = x - ((x - T.origin[l]) & (half - 1));
        const bool left = xme == xp;
        if (active) {
            const int key = (int)node_idx(T, l, xme);
            int s_ = path_hash(key) & (kHashSlots - 1);
            while (true) {
                const int old = atomicCAS(&h.key[s_], -1, key);
                if (old == -1 || old == key) break;      // (same key: a merged path, same values)
                s_ = (s_ + 1) & (kHashSlots - 1);
            }
            h.sv[s_] = cs.v;
            h.mv[s_] = cm.v;
            h.tags[s_] = cs.t | (cm.t << 8);
        }
        __threadfence_block();
        __syncthreads();
        if (active) {
            const int tg = S.sib_tg[l][tid];
            TV ss = mk_tv(S.sib_sv[l][tid], tg & 255), sm = mk_tv(S.sib_mv[l][tid], tg >> 8);
            const int skey = (int)node_idx(T, l, left ? xp + half : xp);
            int s_ = path_hash(skey) & (kHashSlots - 1);
            while (true) {
                const int k_ = h.key[s_];
                if (k_ == -1) break;
                if (k_ == skey) {                        // the sibling is on another thread's path
                    ss = mk_tv(h.sv[s_], h.tags[s_] & 255);
                    sm = mk_tv(h.mv[s_], h.tags[s_] >> 8);
                    break;
                }
                s_ = (s_ + 1) & (kHashSlots - 1);
            }
            cs = left ? reduce_sum(cs, ss) : reduce_sum(ss, cs);
            cm = left ? reduce_min(cm, sm) : reduce_min(sm, cm);
            const int64_t ip = node_idx(T, l + 1, x);
            T.sum_val[ip] = cs.v;
            T.sum_tag[ip] = (uint8_t)cs.t;
            T.min_val[ip] = cm.v;
            T.min_tag[ip] = (uint8_t)cm.t;
        }
        // the table of level l - 1 (read for the last time before the barrier above) is the one
        // level l + 2 inserts into, a barrier from now
        PathTab &old_tab = S.tab[(l + 2) % 3];
        for (int i = tid; i < kHashSlots; i += blockDim.x) old_tab.key[i] = -1;
    }
}

__global__ __launch_bounds__(kMaxBatch) void k_tree_write(pfrl_tree_t T, int64_t n,
                                                          const int64_t *__restrict__ x,
                                                          const double *__restrict__ val,
                                                          const uint8_t *__restrict__ tag,
                                                          const uint8_t *__restrict__ use_maxp) {
    const int i = threadIdx.x;
    const bool active = i < n;
    int64_t xi = 0;
    if (active) {
        xi = x[i];
        TV p;
        if (use_maxp && use_maxp[i])
            p = mk_tv(*T.maxp_val, *T.maxp_tag);
        else
            p = mk_tv(val[i], tag[i]);
        const int64_t il = node_idx(T, 0, xi);
        T.sum_val[il] = p.v;
        T.sum_tag[il] = (uint8_t)p.t;
        T.min_val[il] = p.v;
        T.min_tag[il] = (uint8_t)p.t;
    }
    repair_paths(T, active, xi);
}

// TreeQueue._write on the SUM tree only, for a batch of distinct leaves: what
// SumTreeQueue.uniform_sample does to the leaves sample_n_k picked (prioritized.py:278-292:
// _write(ix, 0.0) one after the other, the previous values returned) and what both samplers do
// afterwards with remove=False (:289-291, :308-310: _write(ix, val) puts them back).  The min
// tree is never touched by a sampler.  Node values are pure functions of the leaves, so the
// leaf stores followed by one bottom-up re-reduction of the touched paths give the state the
// sequential writes give; old_val / old_tag (may be NULL) receive what the leaves held.
__global__ __launch_bounds__(kMaxBatch) void k_tree_write_sum(pfrl_tree_t T, int64_t n,
                                                              const int64_t *__restrict__ x,
                                                              const double *__restrict__ val,
                                                              const uint8_t *__restrict__ tag,
                                                              double *__restrict__ old_val,
                                                              uint8_t *__restrict__ old_tag) {
    const int i = threadIdx.x;
    bool active = i < n;
    int64_t xi = 0;
    if (active) {
        xi = x[i];
        const int64_t il = node_idx(T, 0, xi);
        if (old_val != nullptr) {
            old_val[i] = T.sum_val[il];
            old_tag[i] = T.sum_tag[il];
        }
        T.sum_val[il] = val != nullptr ? val[i] : 0.0;
        T.sum_tag[il] = tag != nullptr ? tag[i] : (uint8_t)PFRL_TAG_PY;
    }
    const int L = T.log2_size;
    if (active && (xi < T.base || xi >= T.base + ((int64_t)1 << L))) active = false;
    for (int l = 1; l <= L; ++l) {
        __threadfence_block();
        __syncthreads();
        if (active) {
            const int64_t half = (int64_t)1 << (l - 1);
            const int64_t xl = xi - ((xi - T.origin[l]) & (((int64_t)1 << l) - 1));
            const int64_t il = node_idx(T, l - 1, xl);
            const int64_t ir = node_idx(T, l - 1, xl + half);
            const int64_t ip = node_idx(T, l, xi);
            const TV a = mk_tv(T.sum_val[il], T.sum_tag[il]);
            const TV b = mk_tv(T.sum_val[ir], T.sum_tag[ir]);
            const TV sum = reduce_sum(a, b);
            T.sum_val[ip] = sum.v;
            T.sum_tag[ip] = (uint8_t)sum.t;
        }
    }
}

// Shared body of set_last_priority: typed max_priority scan, last-occurrence
// de-duplication, leaf writes into both trees.  Returns whether this thread's leaf
// takes part in the path repair that has to follow.
__device__ bool set_priorities_leaves(const pfrl_tree_t &T, int64_t B, const int64_t *x, TV p,
                                      int dedupe, double *s_v, uint8_t *s_t, int64_t *s_x,
                                      int64_t &xi) {
    const int i = threadIdx.x;
    const bool in = i < B;
    xi = 0;
    if (in) {
        xi = x[i];
        s_v[i] = p.v;
        s_t[i] = (uint8_t)p.t;
        s_x[i] = xi;
    }
    __syncthreads();
    if (i == 0) {
        // prioritized.py:114  self.max_priority = max(self.max_priority, p)
        TV m = mk_tv(*T.maxp_val, *T.maxp_tag);
        for (int j = 0; j < B; ++j) {
            TV pj = mk_tv(s_v[j], s_t[j]);
            if (tv_lt(m, pj)) m = pj;
        }
        *T.maxp_val = m.v;
        *T.maxp_tag = (uint8_t)m.t;
    }
    bool active = in;
    if (in && dedupe) {
        for (int j = i + 1; j < B; ++j)
            if (s_x[j] == xi) {
                active = false;
                break;
            }
    }
    if (active) {
        const int64_t il = node_idx(T, 0, xi);
        T.sum_val[il] = p.v;
        T.sum_tag[il] = (uint8_t)p.t;
        T.min_val[il] = p.v;
        T.min_tag[il] = (uint8_t)p.t;
    }
    return active;
}

__device__ void set_priorities_tail(const pfrl_tree_t &T, int64_t B, const int64_t *x, TV p,
                                    int dedupe, double *s_v, uint8_t *s_t, int64_t *s_x) {
    int64_t xi;
    const bool active = set_priorities_leaves(T, B, x, p, dedupe, s_v, s_t, s_x, xi);
    repair_paths(T, active, xi);
}

__global__ __launch_bounds__(kMaxBatch) void k_tree_set_priorities(pfrl_tree_t T, int64_t B,
                                                                   const int64_t *__restrict__ x,
                                                                   const double *__restrict__ val,
                                                                   const uint8_t *__restrict__ tag,
                                                                   int dedupe) {
    __shared__ double s_v[kMaxBatch];
    __shared__ uint8_t s_t[kMaxBatch];
    __shared__ int64_t s_x[kMaxBatch];
    const int i = threadIdx.x;
    TV p = mk_tv(0.0, PFRL_TAG_PY);
    if (i < B) p = mk_tv(val[i], tag[i]);
    set_priorities_tail(T, B, x, p, dedupe, s_v, s_t, s_x);
}

struct ErrCfg {
    int has_min, has_max;
    float error_min, error_max;
    double pri_at_min, pri_at_max, eps, alpha;
    int pow_mode;   // PFRL_POW_*
};

// pfrl/replay_buffers/prioritized.py:47-55 with np.float32 errors
__device__ __forceinline__ TV priority_of_error(const ErrCfg &c, float e) {
    if (c.has_min && !(e > c.error_min)) return mk_tv(c.pri_at_min, PFRL_TAG_PY);
    if (c.has_max && !(e < c.error_max)) return mk_tv(c.pri_at_max, PFRL_TAG_PY);
    const float s = __fadd_rn(e, (float)c.eps);
    // np.float32 ** float -> powf(s, (float)alpha) of the host's libm: glibc's powf is
    // not correctly rounded (1 ulp off in ~0.05 % of inputs), so it is restated
    // operation by operation (powf_glibc.h) -- the leaves are the numbers NumPy gives.
    float r;
    if (c.pow_mode == PFRL_POW_GLIBC_FMA)
        r = pfrl_powf::powf_glibc<true>(s, (float)c.alpha);
    else if (c.pow_mode == PFRL_POW_GLIBC)
        r = pfrl_powf::powf_glibc<false>(s, (float)c.alpha);
    else   // PFRL_POW_CORRECTLY_ROUNDED (rounds 1 and 2)
        r = (float)pow((double)s, (double)(float)c.alpha);
    return mk_tv((double)r, PFRL_TAG_F32);
}

__global__ __launch_bounds__(kMaxBatch) void k_tree_update_errors(pfrl_tree_t T, int64_t B,
                                                                  const int64_t *__restrict__ x,
                                                                  const float *__restrict__ err,
                                                                  ErrCfg c, int dedupe) {
    __shared__ double s_v[kMaxBatch];
    __shared__ uint8_t s_t[kMaxBatch];
    __shared__ int64_t s_x[kMaxBatch];
    const int i = threadIdx.x;
    TV p = mk_tv(0.0, PFRL_TAG_PY);
    if (i < B) p = priority_of_error(c, err[i]);
    set_priorities_tail(T, B, x, p, dedupe, s_v, s_t, s_x);
}

// update_errors of one minibatch followed by the leaf writes recorded since (appends at
// max_priority, pops) as ONE launch and ONE bottom-up repair over the union of the touched
// paths.  Sequentially (prioritized.py:107-116, then :39-54) the priorities are set first --
// max_priority included, which the appended leaves then take -- and a later write to the same
// leaf wins; the node values are pure functions of the leaves, so one repair after both sets
// of leaf stores gives the state the two launches gave.  Threads [0, B) carry the errors,
// [B, B + n) the writes; both under the same frame (the host launches separately otherwise).
__global__ __launch_bounds__(kMaxBatch) void k_tree_update_errors_write(
    pfrl_tree_t T, int64_t B, const int64_t *__restrict__ x, const float *__restrict__ err,
    ErrCfg c, int dedupe, int64_t n, const int64_t *__restrict__ wx,
    const double *__restrict__ wval, const uint8_t *__restrict__ wtag,
    const uint8_t *__restrict__ wuse_maxp) {
    __shared__ double s_v[kMaxBatch];
    __shared__ uint8_t s_t[kMaxBatch];
    __shared__ int64_t s_x[kMaxBatch];
    const int i = threadIdx.x;
    TV p = mk_tv(0.0, PFRL_TAG_PY);
    if (i < B) p = priority_of_error(c, err[i]);
    int64_t xi;
    bool active = set_priorities_leaves(T, B, x, p, dedupe, s_v, s_t, s_x, xi);
    __threadfence_block();
    __syncthreads();             // max_priority and the minibatch's leaves are in place
    const int64_t k = (int64_t)i - B;
    if (k >= 0 && k < n) {
        xi = wx[k];
        TV q;
        if (wuse_maxp && wuse_maxp[k])
            q = mk_tv(*T.maxp_val, *T.maxp_tag);
        else
            q = mk_tv(wval[k], wtag[k]);
        const int64_t il = node_idx(T, 0, xi);
        T.sum_val[il] = q.v;
        T.sum_tag[il] = (uint8_t)q.t;
        T.min_val[il] = q.v;
        T.min_tag[il] = (uint8_t)q.t;
        active = true;
    }
    repair_paths(T, active, xi);
}

// One wave; lane 0 performs the B sequentially dependent draws
// (prioritized.py:294-312), the wave then evaluates probabilities / weights.
__global__ __launch_bounds__(64) void k_tree_sample(pfrl_tree_t T, int64_t B,
                                                    const double *__restrict__ u01,
                                                    int64_t *__restrict__ out_x,
                                                    double *__restrict__ out_pri,
                                                    uint8_t *__restrict__ out_pri_tag,
                                                    double *__restrict__ out_prob,
                                                    float *__restrict__ out_weight,
                                                    double *__restrict__ out_total,
                                                    uint8_t *__restrict__ out_total_tag,
                                                    double *__restrict__ out_min_prob,
                                                    int normalize, double beta, int64_t slot_mod,
                                                    int32_t *__restrict__ out_slot) {
    __shared__ double sib_v[PFRL_MAX_LEVELS];
    __shared__ uint8_t sib_t[PFRL_MAX_LEVELS];
    __shared__ uint8_t went_right[PFRL_MAX_LEVELS];
    __shared__ double s_total_v, s_min_v;
    __shared__ int s_total_t, s_min_t;
    const int L = T.log2_size;
    if (threadIdx.x == 0) {
        const int64_t iroot = node_idx(T, L, T.base);
        TV total = mk_tv(T.sum_val[iroot], T.sum_tag[iroot]);
        TV minv = mk_tv(T.min_val[iroot], T.min_tag[iroot]);
        s_total_v = total.v;
        s_total_t = total.t;
        s_min_v = minv.v;
        s_min_t = minv.t;
        for (int64_t i = 0; i < B; ++i) {
            TV root = mk_tv(T.sum_val[iroot], T.sum_tag[iroot]);
            // np.random.uniform(0.0, root) = 0.0 + (root - 0.0) * u
            TV pos = mk_tv(__dadd_rn(0.0, __dmul_rn(root.v, u01[i])), PFRL_TAG_PY);
            int64_t x = T.base;
            for (int l = L; l >= 1; --l) {
                const int64_t half = (int64_t)1 << (l - 1);
                const int64_t il = node_idx(T, l - 1, x);
                const int64_t ir = node_idx(T, l - 1, x + half);
                TV lc = mk_tv(T.sum_val[il], T.sum_tag[il]);
                TV rc = mk_tv(T.sum_val[ir], T.sum_tag[ir]);
                TV left = lc.t != PFRL_TAG_ABSENT ? lc : mk_tv(0.0, PFRL_TAG_PY);
                if (tv_lt(pos, left)) {
                    sib_v[l] = rc.v;
                    sib_t[l] = (uint8_t)rc.t;
                    went_right[l] = 0;
                } else {
                    pos = tv_sub(pos, left);
                    x += half;
                    sib_v[l] = lc.v;
                    sib_t[l] = (uint8_t)lc.t;
                    went_right[l] = 1;
                }
            }
            const int64_t ileaf = node_idx(T, 0, x);
            out_x[i] = x;
            out_pri[i] = T.sum_val[ileaf];
            out_pri_tag[i] = T.sum_tag[ileaf];
            // _write(ix, 0.0): zero the leaf, re-reduce the path (sum tree only)
            T.sum_val[ileaf] = 0.0;
            T.sum_tag[ileaf] = PFRL_TAG_PY;
            TV cur = mk_tv(0.0, PFRL_TAG_PY);
            for (int l = 1; l <= L; ++l) {
                TV sib = mk_tv(sib_v[l], sib_t[l]);
                cur = went_right[l] ? reduce_sum(sib, cur) : reduce_sum(cur, sib);
                const int64_t ip = node_idx(T, l, x);
                T.sum_val[ip] = cur.v;
                T.sum_tag[ip] = (uint8_t)cur.t;
            }
            __threadfence_block();
        }
        *out_total = s_total_v;
        *out_total_tag = (uint8_t)s_total_t;
    }
    __syncthreads();
    // probabilities (prioritized.py:80-83 with uniform_ratio = 0)
    const TV total = mk_tv(s_total_v, s_total_t);
    double local_min = __builtin_huge_val();
    for (int64_t i = threadIdx.x; i < B; i += 64) {
        TV pr = tv_add(mk_tv(0.0, PFRL_TAG_PY), tv_div(mk_tv(out_pri[i], out_pri_tag[i]), total));
        out_prob[i] = pr.v;
        local_min = fmin(local_min, pr.v);
    }
    for (int off = 32; off > 0; off >>= 1) local_min = fmin(local_min, __shfl_xor(local_min, off));
    double min_prob = tv_div(mk_tv(s_min_v, s_min_t), total).v;
    if (threadIdx.x == 0) *out_min_prob = min_prob;
    // weights (pfrl/replay_buffers/prioritized.py:57-66)
    if (normalize == 1) min_prob = local_min;
    for (int64_t i = threadIdx.x; i < B; i += 64) {
        const double p = out_prob[i];
        double w;
        if (normalize)
            w = pow(p / min_prob, -beta);
        else
            w = pow((double)T.length * p, -beta);
        out_weight[i] = (float)w;
        if (out_slot) out_slot[i] = (int32_t)(out_x[i] % slot_mod);
    }
}

// ---------------------------------------------------------------------------
// LDS-staged sampler.  The B draws are sequentially dependent, so the cost is
// a latency chain; this version shortens every link of it:
//   * the top of the tree (levels L .. r, r = min(L, 9)) is copied once into an
//     LDS heap (<= 8192 nodes, 72 KB) and stays authoritative for the whole
//     launch: 12 of the 21 levels of a 1M-leaf tree descend at LDS latency;
//   * the 2^(r+1)-1 nodes below the chosen level-r node are fetched by the 64
//     lanes in ONE parallel round trip into a second LDS heap, so the bottom
//     r levels also descend at LDS latency (instead of r dependent HBM/L2
//     round trips);
//   * the zero-and-repair pass runs on the LDS copies and the touched path is
//     written back to HBM by one lane per level.
// Arithmetic and visiting order are exactly those of k_tree_sample.
// ---------------------------------------------------------------------------
#ifdef PFRL_TREE_DEBUG
__device__ unsigned long long g_dbg[8];
#define DBG_T(k) do { if (lane == 0) { unsigned long long t__ = wall_clock64(); g_dbg[k] += t__ - t_prev; t_prev = t__; } } while (0)
#else
#define DBG_T(k)
#endif

constexpr int kBotLevels = 9;   // r: bottom subtree root level
constexpr int kMaxTopLog2 = 13; // top heap holds levels L..r, at most 13 levels

__global__ __launch_bounds__(64) void k_tree_sample_lds(
    pfrl_tree_t T, int64_t B, const double *__restrict__ u01, int64_t *__restrict__ out_x,
    double *__restrict__ out_pri, uint8_t *__restrict__ out_pri_tag, double *__restrict__ out_prob,
    float *__restrict__ out_weight, double *__restrict__ out_total,
    uint8_t *__restrict__ out_total_tag, double *__restrict__ out_min_prob, int normalize,
    double beta, int64_t slot_mod, int32_t *__restrict__ out_slot) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int L = T.log2_size;
    const int r = L < kBotLevels ? L : kBotLevels;
    const int top_levels = L - r + 1;           // levels L..r  -> depths 0..top_levels-1
    const int top_n = 1 << top_levels;          // heap indices 1..top_n-1
    const int bot_n = 1 << (r + 1);             // heap indices 1..bot_n-1 (levels r..0)
    double *top_v = reinterpret_cast<double *>(smem);
    double *bot_v = top_v + top_n;
    uint8_t *top_t = reinterpret_cast<uint8_t *>(bot_v + bot_n);
    uint8_t *bot_t = top_t + top_n;
    __shared__ double s_total_v, s_min_v;
    __shared__ int s_total_t, s_min_t;
    // per-level addressing constants (kernel-argument arrays indexed per lane would
    // otherwise become memory loads in the hot loops)
    __shared__ int64_t lv_off[PFRL_MAX_LEVELS], lv_org[PFRL_MAX_LEVELS], lv_mask[PFRL_MAX_LEVELS];
    const int lane = threadIdx.x;
    if (lane <= L) {
        const int sh = T.log2_smax - lane;
        lv_off[lane] = T.level_off[lane];
        lv_org[lane] = T.origin[lane];
        lv_mask[lane] = (sh > 0 ? ((int64_t)1 << sh) : 1) - 1;
    }
    __syncthreads();
#define NODE_AT(l, x) (lv_off[l] + ((((x) - lv_org[l]) >> (l)) & lv_mask[l]))

    // stage the top of the sum tree (loads batched 16 deep per lane)
    for (int h0 = 1; h0 < top_n; h0 += 64 * 16) {
        double v[16];
        uint8_t tg[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int h = h0 + k * 64 + lane;
            if (h < top_n) {
                const int d = 31 - __clz(h);
                const int l = L - d;
                const int64_t gi = NODE_AT(l, T.base + ((int64_t)(h - (1 << d)) << l));
                v[k] = T.sum_val[gi];
                tg[k] = T.sum_tag[gi];
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int h = h0 + k * 64 + lane;
            if (h < top_n) {
                top_v[h] = v[k];
                top_t[h] = tg[k];
            }
        }
    }
    if (lane == 0) {
        const int64_t iroot = NODE_AT(L, T.base);
        s_total_v = T.sum_val[iroot];
        s_total_t = T.sum_tag[iroot];
        s_min_v = T.min_val[iroot];
        s_min_t = T.min_tag[iroot];
    }
    __syncthreads();

#ifdef PFRL_TREE_DEBUG
    unsigned long long t_prev = wall_clock64();
    if (lane == 0) for (int k = 0; k < 8; ++k) g_dbg[k] = 0;
#endif
    for (int64_t i = 0; i < B; ++i) {
        // siblings met on the way down, kept in registers for the repair pass
        double sv[kMaxTopLog2 + kBotLevels];
        int st[kMaxTopLog2 + kBotLevels];
        bool went_right[kMaxTopLog2 + kBotLevels];
        // ---- descend the top heap (every lane computes the same path) ----
        TV root = mk_tv(top_v[1], top_t[1]);
        TV pos = mk_tv(__dadd_rn(0.0, __dmul_rn(root.v, u01[i])), PFRL_TAG_PY);
        int h = 1;
        // Two levels per LDS round trip: the children AND the four grandchildren of h
        // are requested together (12 independent LDS reads), then both decisions are
        // taken in registers.  The dependent chain is what this kernel pays for, and
        // an LDS read (~100 ns) is most of a level.
#pragma unroll
        for (int d = 0; d < kMaxTopLog2 - 1; d += 2) {
            if (d < top_levels - 1) {
                const bool two = d + 1 < top_levels - 1;       // uniform
                const int gbase = two ? 4 * h : 2 * h;          // always inside the heap
                const double cv0 = top_v[2 * h], cv1 = top_v[2 * h + 1];
                const int ct0 = top_t[2 * h], ct1 = top_t[2 * h + 1];
                const double gv0 = top_v[gbase], gv1 = top_v[gbase + 1];
                const double gv2 = top_v[gbase + (two ? 2 : 0)], gv3 = top_v[gbase + (two ? 3 : 1)];
                const int gt0 = top_t[gbase], gt1 = top_t[gbase + 1];
                const int gt2 = top_t[gbase + (two ? 2 : 0)], gt3 = top_t[gbase + (two ? 3 : 1)];
                {
                    TV lc = mk_tv(cv0, ct0), rc = mk_tv(cv1, ct1);
                    TV left = lc.t != PFRL_TAG_ABSENT ? lc : mk_tv(0.0, PFRL_TAG_PY);
                    const bool go_left = tv_lt(pos, left);
                    if (!go_left) pos = tv_sub(pos, left);
                    sv[d] = go_left ? rc.v : lc.v;
                    st[d] = go_left ? rc.t : lc.t;
                    went_right[d] = !go_left;
                    h = 2 * h + (go_left ? 0 : 1);
                }
                if (d + 1 < kMaxTopLog2 - 1 && two) {
                    const bool was_left = !went_right[d];
                    TV lc = mk_tv(was_left ? gv0 : gv2, was_left ? gt0 : gt2);
                    TV rc = mk_tv(was_left ? gv1 : gv3, was_left ? gt1 : gt3);
                    TV left = lc.t != PFRL_TAG_ABSENT ? lc : mk_tv(0.0, PFRL_TAG_PY);
                    const bool go_left = tv_lt(pos, left);
                    if (!go_left) pos = tv_sub(pos, left);
                    sv[d + 1] = go_left ? rc.v : lc.v;
                    st[d + 1] = go_left ? rc.t : lc.t;
                    went_right[d + 1] = !go_left;
                    h = 2 * h + (go_left ? 0 : 1);
                }
            }
        }
        const int64_t x0 = T.base + ((int64_t)(h - (top_n >> 1)) << r);
        DBG_T(0);
        // ---- fan-out: the whole subtree below in one parallel round trip ----
        // level l of the subtree is 2^(r-l) consecutive ring slots: lanes read
        // consecutive addresses (coalesced), all loads issued before any use.
        {
            double v[8 + 4 + 2 + kBotLevels - 2];
            uint8_t tg[8 + 4 + 2 + kBotLevels - 2];
            // slot numbering below is a compile-time function of (l, k): the
            // arrays stay in registers
            int n_ld = 0;
#pragma unroll
            for (int l = 0; l <= kBotLevels; ++l) {
                const int lr = l <= r ? l : r;               // clamp (uniform)
                const int cnt = l <= r ? (1 << (r - l)) : 0;
                const int64_t q0 = (x0 - lv_org[lr]) >> lr;
                const int64_t off = lv_off[lr], mask = lv_mask[lr];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k * 64 < (1 << (kBotLevels - l))) {
                        // unconditional load from a clamped (always valid) slot: keeps
                        // all loads in one basic block so they are issued back to back
                        const int j = k * 64 + lane;
                        const int64_t gi = off + ((q0 + (j < cnt ? j : 0)) & mask);
                        v[n_ld] = T.sum_val[gi];
                        tg[n_ld] = T.sum_tag[gi];
                        ++n_ld;
                    }
                }
            }
            n_ld = 0;
#pragma unroll
            for (int l = 0; l <= kBotLevels; ++l) {
                const int cnt = l <= r ? (1 << (r - l)) : 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k * 64 < (1 << (kBotLevels - l))) {
                        const int j = k * 64 + lane;
                        if (j < cnt) {
                            bot_v[cnt + j] = v[n_ld];   // heap index of (level l, j)
                            bot_t[cnt + j] = tg[n_ld];
                        }
                        ++n_ld;
                    }
                }
            }
        }
        __syncthreads();
        DBG_T(1);
        // ---- descend the bottom heap ----
        int g = 1;
#pragma unroll
        for (int d = 0; d < kBotLevels; d += 2) {
            if (d < r) {
                const bool two = d + 1 < r;                     // uniform
                const int gbase = two ? 4 * g : 2 * g;
                const double cv0 = bot_v[2 * g], cv1 = bot_v[2 * g + 1];
                const int ct0 = bot_t[2 * g], ct1 = bot_t[2 * g + 1];
                const double gv0 = bot_v[gbase], gv1 = bot_v[gbase + 1];
                const double gv2 = bot_v[gbase + (two ? 2 : 0)], gv3 = bot_v[gbase + (two ? 3 : 1)];
                const int gt0 = bot_t[gbase], gt1 = bot_t[gbase + 1];
                const int gt2 = bot_t[gbase + (two ? 2 : 0)], gt3 = bot_t[gbase + (two ? 3 : 1)];
                {
                    TV lc = mk_tv(cv0, ct0), rc = mk_tv(cv1, ct1);
                    TV left = lc.t != PFRL_TAG_ABSENT ? lc : mk_tv(0.0, PFRL_TAG_PY);
                    const bool go_left = tv_lt(pos, left);
                    if (!go_left) pos = tv_sub(pos, left);
                    sv[kMaxTopLog2 + d] = go_left ? rc.v : lc.v;
                    st[kMaxTopLog2 + d] = go_left ? rc.t : lc.t;
                    went_right[kMaxTopLog2 + d] = !go_left;
                    g = 2 * g + (go_left ? 0 : 1);
                }
                if (d + 1 < kBotLevels && two) {
                    const bool was_left = !went_right[kMaxTopLog2 + d];
                    TV lc = mk_tv(was_left ? gv0 : gv2, was_left ? gt0 : gt2);
                    TV rc = mk_tv(was_left ? gv1 : gv3, was_left ? gt1 : gt3);
                    TV left = lc.t != PFRL_TAG_ABSENT ? lc : mk_tv(0.0, PFRL_TAG_PY);
                    const bool go_left = tv_lt(pos, left);
                    if (!go_left) pos = tv_sub(pos, left);
                    sv[kMaxTopLog2 + d + 1] = go_left ? rc.v : lc.v;
                    st[kMaxTopLog2 + d + 1] = go_left ? rc.t : lc.t;
                    went_right[kMaxTopLog2 + d + 1] = !go_left;
                    g = 2 * g + (go_left ? 0 : 1);
                }
            }
        }
        const int64_t x = x0 + (g - (bot_n >> 1));
        const double leaf_v = bot_v[g];
        const uint8_t leaf_t = bot_t[g];
        __syncthreads();
        DBG_T(2);
        // ---- zero the leaf, repair the path from the remembered siblings ----
        TV cur = mk_tv(0.0, PFRL_TAG_PY);
        if (lane == 0) {
            out_x[i] = x;
            out_pri[i] = leaf_v;
            out_pri_tag[i] = leaf_t;
            bot_v[g] = 0.0;
            bot_t[g] = PFRL_TAG_PY;
        }
        int c = g;
#pragma unroll
        for (int d = kBotLevels - 1; d >= 0; --d) {
            if (d < r) {
                TV sib = mk_tv(sv[kMaxTopLog2 + d], st[kMaxTopLog2 + d]);
                cur = reduce_sum(cur, sib);   // IEEE add commutes, so operand order is immaterial
                c >>= 1;
                if (lane == 0) {
                    bot_v[c] = cur.v;
                    bot_t[c] = (uint8_t)cur.t;
                }
            }
        }
        c = h;
        if (lane == 0) {
            top_v[c] = cur.v;
            top_t[c] = (uint8_t)cur.t;
        }
#pragma unroll
        for (int d = kMaxTopLog2 - 2; d >= 0; --d) {
            if (d < top_levels - 1) {
                TV sib = mk_tv(sv[d], st[d]);
                cur = reduce_sum(cur, sib);
                c >>= 1;
                if (lane == 0) {
                    top_v[c] = cur.v;
                    top_t[c] = (uint8_t)cur.t;
                }
            }
        }
        __syncthreads();
        DBG_T(3);
        // ---- write the touched path back to HBM: one lane per level ----
        if (lane <= L) {
            const int l = lane;
            double v;
            uint8_t tg;
            if (l <= r) {
                const int gg = g >> l;          // ancestor of the leaf at level l
                v = bot_v[gg];
                tg = bot_t[gg];
            } else {
                const int hh = h >> (l - r);
                v = top_v[hh];
                tg = top_t[hh];
            }
            const int64_t gi = NODE_AT(l, x);
            T.sum_val[gi] = v;
            T.sum_tag[gi] = tg;
        }
        __syncthreads();
        DBG_T(4);
    }
#undef NODE_AT
    if (lane == 0) {
        *out_total = s_total_v;
        *out_total_tag = (uint8_t)s_total_t;
    }
    __threadfence_block();
    __syncthreads();
    const TV total = mk_tv(s_total_v, s_total_t);
    double local_min = __builtin_huge_val();
    for (int64_t i = lane; i < B; i += 64) {
        TV pr = tv_add(mk_tv(0.0, PFRL_TAG_PY), tv_div(mk_tv(out_pri[i], out_pri_tag[i]), total));
        out_prob[i] = pr.v;
        local_min = fmin(local_min, pr.v);
    }
    for (int off = 32; off > 0; off >>= 1) local_min = fmin(local_min, __shfl_xor(local_min, off));
    double min_prob = tv_div(mk_tv(s_min_v, s_min_t), total).v;
    if (lane == 0) *out_min_prob = min_prob;
    if (normalize == 1) min_prob = local_min;
    for (int64_t i = lane; i < B; i += 64) {
        const double p = out_prob[i];
        double w;
        if (normalize)
            w = pow(p / min_prob, -beta);
        else
            w = pow((double)T.length * p, -beta);
        out_weight[i] = (float)w;
        if (out_slot) out_slot[i] = (int32_t)(out_x[i] % slot_mod);
    }
}


// ---------------------------------------------------------------------------
// Lean sampler (round 4).  Same arithmetic, same visiting order, same results as
// k_tree_sample; what changes is the number of instructions on the chain.  The
// kernel is ONE wave, and a wave issues at most one instruction -- of any kind --
// every four clocks: k_tree_sample_lds spends ~70 instructions per level on
// branch-free typed arithmetic (both candidate types evaluated, selects) and
// ~50 per level of the repair, i.e. it is bound by instruction issue, not by
// LDS latency (in-kernel clocks: 157 ns per level).
//
//   * descent: almost every level is an np.float32 operation -- the position is
//     np.float32 after its first subtraction, and NEP 50 makes f32 (op) python
//     float a float32 operation on the float32-rounded values; an absent left child
//     is the Python float 0.0.  A level whose operands do not fit that (np.float64
//     anywhere, Python float against Python float) takes the general typed step.
//     The position is carried as (p32 = float32 of its value, p64 = its value
//     while it is not np.float32); only the left child of a node is read, two
//     levels per LDS round trip (the left children of both children requested
//     with it).
//   * siblings: once the path is known, lane l reads the sibling of level l from
//     the heaps -- one parallel round trip instead of selects at every level.
//   * repair (prioritized.py:304-308: _write(ix, 0.0)): a chain of L typed additions
//     cur = cur + sibling, bottom-up.  Result types only ever widen along it
//     (PY -> F32 -> F64), so the chain is at most three runs of plain adds (f64,
//     f32, f64) whose boundaries two ballots over the sibling tags give; the running
//     sum reads lane l's sibling with v_readlane.  Lane l keeps the node of level l
//     and stores it to the LDS top heap and to HBM itself.
//   * the draws' uniforms are loaded 64 at a time, one per lane (a scalar load per
//     draw was a memory round trip on the chain); level constants come out of the
//     owning lane's registers.
// Tried first and dropped (profiles/r04_per_sampler_phases.txt): six levels per round
// with lane j speculating the path whose decisions are the bits of j -- bit-exact, and
// slower (2.6 us for the 12 top levels against 1.9): the per-round bookkeeping (64
// paths' loads, sibling selects, the winner's stores, broadcasts) is ~500 instructions.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int64_t readlane_i64(int64_t v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffff), l);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((int64_t)hi << 32) | (int64_t)lo;
}

__device__ __forceinline__ float readlane_f32(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// One level of _find (prioritized.py:262-276) given the left child (lv, lt) of the current node.
// Position: value p64 / type pt while pt != F32, value p32 when pt == F32; p32 == (float)value
// always.  Returns true when the reference goes right.
__device__ __forceinline__ bool find_level(double lv, int lt, double &p64, float &p32, int &pt) {
    const bool absent = lt == PFRL_TAG_ABSENT;
    const bool lean = (pt == PFRL_TAG_F32 && lt != PFRL_TAG_F64) ||
                      (pt == PFRL_TAG_PY && lt == PFRL_TAG_F32);
    bool go_left;
    if (lean) {
        const float l32 = absent ? 0.0f : (float)lv;
        go_left = p32 < l32;
        const float sub = __fsub_rn(p32, l32);
        p32 = go_left ? p32 : sub;
        pt = (go_left || absent) ? pt : PFRL_TAG_F32;
    } else {
        const TV pos = mk_tv(pt == PFRL_TAG_F32 ? (double)p32 : p64, pt);
        const TV left = mk_tv(absent ? 0.0 : lv, absent ? PFRL_TAG_PY : lt);
        go_left = tv_lt(pos, left);
        const TV sub = tv_sub(pos, left);
        p64 = go_left ? p64 : sub.v;
        pt = go_left ? pt : sub.t;
        p32 = go_left ? p32 : (float)sub.v;
    }
    return !go_left;
}

// n levels down an LDS heap from node h (children of node i: 2i, 2i + 1); returns the node reached.
// Two levels per LDS round trip: the left child of h and the left children of both its children
// are requested together.  While the position is np.float32 and none of the three is np.float64
// (one test for the pair of levels) both levels are plain float compare / subtract / select;
// anything else goes through find_level.
__device__ __forceinline__ int find_down(const double *hv, const uint8_t *ht, int h, int n,
                                         double &p64, float &p32, int &pt) {
    int d = 0;
    for (; d + 1 < n; d += 2) {
        const double lv0 = hv[2 * h], lva = hv[4 * h], lvb = hv[4 * h + 2];
        const int lt0 = ht[2 * h], lta = ht[4 * h], ltb = ht[4 * h + 2];
        bool r0, r1;
        if (pt == PFRL_TAG_F32 && lt0 != PFRL_TAG_F64 && lta != PFRL_TAG_F64 && ltb != PFRL_TAG_F64) {
            const float l0 = lt0 == PFRL_TAG_ABSENT ? 0.0f : (float)lv0;
            r0 = !(p32 < l0);
            const float s0 = __fsub_rn(p32, l0);
            p32 = r0 ? s0 : p32;
            const double lv1 = r0 ? lvb : lva;
            const int lt1 = r0 ? ltb : lta;
            const float l1 = lt1 == PFRL_TAG_ABSENT ? 0.0f : (float)lv1;
            r1 = !(p32 < l1);
            const float s1 = __fsub_rn(p32, l1);
            p32 = r1 ? s1 : p32;
        } else {
            r0 = find_level(lv0, lt0, p64, p32, pt);
            r1 = find_level(r0 ? lvb : lva, r0 ? ltb : lta, p64, p32, pt);
        }
        h = 4 * h + (r0 ? 2 : 0) + (r1 ? 1 : 0);
    }
    if (d < n) {
        const bool r0 = find_level(hv[2 * h], ht[2 * h], p64, p32, pt);
        h = 2 * h + (r0 ? 1 : 0);
    }
    return h;
}

__global__ __launch_bounds__(64) void k_tree_sample_lean(
    pfrl_tree_t T, int64_t B, const double *__restrict__ u01, int64_t *__restrict__ out_x,
    double *__restrict__ out_pri, uint8_t *__restrict__ out_pri_tag, double *__restrict__ out_prob,
    float *__restrict__ out_weight, double *__restrict__ out_total,
    uint8_t *__restrict__ out_total_tag, double *__restrict__ out_min_prob, int normalize,
    double beta, int64_t slot_mod, int32_t *__restrict__ out_slot) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int L = T.log2_size;
    const int r = L < kBotLevels ? L : kBotLevels;
    const int top_levels = L - r + 1;           // levels L..r  -> depths 0..top_levels-1
    const int top_n = 1 << top_levels;          // heap indices 1..top_n-1
    const int bot_n = 1 << (r + 1);             // heap indices 1..bot_n-1 (levels r..0)
    double *top_v = reinterpret_cast<double *>(smem);
    double *bot_v = top_v + top_n;
    uint8_t *top_t = reinterpret_cast<uint8_t *>(bot_v + bot_n);
    uint8_t *bot_t = top_t + top_n;
    __shared__ double s_total_v, s_min_v;
    __shared__ int s_total_t, s_min_t;
    __shared__ int64_t lv_off[PFRL_MAX_LEVELS], lv_org[PFRL_MAX_LEVELS], lv_mask[PFRL_MAX_LEVELS];
    const int lane = threadIdx.x;
    if (lane <= L) {
        const int sh = T.log2_smax - lane;
        lv_off[lane] = T.level_off[lane];
        lv_org[lane] = T.origin[lane];
        lv_mask[lane] = (sh > 0 ? ((int64_t)1 << sh) : 1) - 1;
    }
    if (lane == 0) {
        // heap index 0 is never a node; the pair (0, 1) is never requested either, but keep it defined
        top_v[0] = 0.0;
        top_t[0] = 0;
        bot_v[0] = 0.0;
        bot_t[0] = 0;
    }
    __syncthreads();
#define NODE_AT(l, x) (lv_off[l] + ((((x) - lv_org[l]) >> (l)) & lv_mask[l]))
    // my own level's addressing constants (lane l owns level l in the repair / write-back)
    const int myl = lane <= L ? lane : 0;
    const int64_t my_off = lv_off[myl], my_org = lv_org[myl], my_mask = lv_mask[myl];

    // stage the top of the sum tree (loads batched 16 deep per lane)
    for (int h0 = 1; h0 < top_n; h0 += 64 * 16) {
        double v[16];
        uint8_t tg[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int h = h0 + k * 64 + lane;
            if (h < top_n) {
                const int d = 31 - __clz(h);
                const int l = L - d;
                const int64_t gi = NODE_AT(l, T.base + ((int64_t)(h - (1 << d)) << l));
                v[k] = T.sum_val[gi];
                tg[k] = T.sum_tag[gi];
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int h = h0 + k * 64 + lane;
            if (h < top_n) {
                top_v[h] = v[k];
                top_t[h] = tg[k];
            }
        }
    }
    if (lane == 0) {
        const int64_t iroot = NODE_AT(L, T.base);
        s_total_v = T.sum_val[iroot];
        s_total_t = T.sum_tag[iroot];
        s_min_v = T.min_val[iroot];
        s_min_t = T.min_tag[iroot];
    }
    __syncthreads();

#ifdef PFRL_TREE_DEBUG
    unsigned long long t_prev = wall_clock64();
    if (lane == 0) for (int k = 0; k < 8; ++k) g_dbg[k] = 0;
#endif
    double my_u = 0.0;
    for (int64_t i = 0; i < B; ++i) {
        // the draws of this launch, 64 at a time, one per lane (a scalar load per draw would be
        // a memory round trip on the chain)
        if ((i & 63) == 0) my_u = i + lane < B ? u01[i + lane] : 0.0;
        const double u = readlane_f64(my_u, (int)(i & 63));
        // ---- top heap: L - r levels ----
        const TV root = mk_tv(top_v[1], top_t[1]);
        // np.random.uniform(0.0, root) = 0.0 + (root - 0.0) * u
        double p64 = __dadd_rn(0.0, __dmul_rn(root.v, u));
        float p32 = (float)p64;
        int pt = PFRL_TAG_PY;
        const int h = find_down(top_v, top_t, 1, L - r, p64, p32, pt);
        const int64_t x0 = T.base + ((int64_t)(h - (top_n >> 1)) << r);
        DBG_T(0);
        // ---- fan-out: the whole subtree below in one parallel round trip ----
        {
            double v[8 + 4 + 2 + kBotLevels - 2];
            uint8_t tg[8 + 4 + 2 + kBotLevels - 2];
            int n_ld = 0;
#pragma unroll
            for (int l = 0; l <= kBotLevels; ++l) {
                const int lr = l <= r ? l : r;               // clamp (uniform)
                const int cnt = l <= r ? (1 << (r - l)) : 0;
                // level constants from the lane that owns the level: scalar registers, and the
                // slot of the subtree's first node of the level is scalar arithmetic
                const int64_t org = readlane_i64(my_org, lr);
                const int64_t off = readlane_i64(my_off, lr), mask = readlane_i64(my_mask, lr);
                const int64_t q0 = (x0 - org) >> lr;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k * 64 < (1 << (kBotLevels - l))) {
                        const int j = k * 64 + lane;
                        const int64_t gi = off + ((q0 + (j < cnt ? j : 0)) & mask);
                        v[n_ld] = T.sum_val[gi];
                        tg[n_ld] = T.sum_tag[gi];
                        ++n_ld;
                    }
                }
            }
            n_ld = 0;
#pragma unroll
            for (int l = 0; l <= kBotLevels; ++l) {
                const int cnt = l <= r ? (1 << (r - l)) : 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k * 64 < (1 << (kBotLevels - l))) {
                        const int j = k * 64 + lane;
                        if (j < cnt) {
                            bot_v[cnt + j] = v[n_ld];   // heap index of (level l, j)
                            bot_t[cnt + j] = tg[n_ld];
                        }
                        ++n_ld;
                    }
                }
            }
        }
        __syncthreads();
        DBG_T(1);
        // ---- bottom heap: r levels ----
        const int g = find_down(bot_v, bot_t, 1, r, p64, p32, pt);
        const int64_t x = x0 + (g - (bot_n >> 1));
        const double leaf_v = bot_v[g];
        const uint8_t leaf_t = bot_t[g];
        // ---- the siblings of the path: lane l reads the one of level l ----
        int st = PFRL_TAG_ABSENT;
        double sv = 0.0;
        if (lane < L) {
            const bool below = lane < r;
            const int node = below ? (g >> lane) : (h >> (lane - r));     // path node of level `lane`
            const double *sib_v = below ? bot_v : top_v;
            const uint8_t *sib_t = below ? bot_t : top_t;
            st = sib_t[node ^ 1];
            sv = sib_v[node ^ 1];
        }
        if (st == PFRL_TAG_ABSENT) sv = 0.0;      // (an absent child adds nothing; its slot may hold anything)
        DBG_T(2);
        if (lane == 0) {
            out_x[i] = x;
            out_pri[i] = leaf_v;
            out_pri_tag[i] = leaf_t;
        }
        // ---- zero the leaf, re-reduce the path: lane l holds the sibling of level l ----
        const float sv32 = (float)sv;
        const unsigned long long m2 = __ballot(st >= PFRL_TAG_F32);
        const unsigned long long m3 = __ballot(st == PFRL_TAG_F64);
        const int l2 = m2 ? __builtin_ctzll(m2) : L;    // first sibling that makes the sum f32 (or f64)
        const int l3 = m3 ? __builtin_ctzll(m3) : L;    // first sibling that makes it f64
        double c64 = 0.0, mine64 = 0.0;
        float c32 = 0.0f, mine32 = 0.0f;
        int j = 0;
        for (; j < l2; ++j) {                 // Python floats: f64 adds
            c64 = __dadd_rn(c64, readlane_f64(sv, j));
            if (lane == j + 1) mine64 = c64;
        }
        if (l2 < l3) {
            c32 = (float)c64;
            for (; j < l3; ++j) {             // np.float32 result type: f32 adds of the f32 operands
                c32 = __fadd_rn(c32, readlane_f32(sv32, j));
                if (lane == j + 1) mine32 = c32;
            }
            c64 = (double)c32;
        }
        for (; j < L; ++j) {                  // np.float64 result type
            c64 = __dadd_rn(c64, readlane_f64(sv, j));
            if (lane == j + 1) mine64 = c64;
        }
        DBG_T(3);
        // ---- lane l owns the node of level l: top heap + HBM ----
        if (lane <= L) {
            const int l = lane;
            double v;
            int tg;
            if (l == 0) {
                v = 0.0;
                tg = PFRL_TAG_PY;
            } else {
                tg = l > l3 ? PFRL_TAG_F64 : (l > l2 ? PFRL_TAG_F32 : PFRL_TAG_PY);
                v = tg == PFRL_TAG_F32 ? (double)mine32 : mine64;
            }
            if (l >= r) {
                const int hh = h >> (l - r);
                top_v[hh] = v;
                top_t[hh] = (uint8_t)tg;
            }
            const int64_t gi = my_off + (((x - my_org) >> l) & my_mask);
            T.sum_val[gi] = v;
            T.sum_tag[gi] = (uint8_t)tg;
        }
        __syncthreads();          // stores visible to the next draw's fan-out, heaps settled
        DBG_T(4);
    }
#undef NODE_AT
    if (lane == 0) {
        *out_total = s_total_v;
        *out_total_tag = (uint8_t)s_total_t;
    }
    __threadfence_block();
    __syncthreads();
    const TV total = mk_tv(s_total_v, s_total_t);
    double local_min = __builtin_huge_val();
    for (int64_t i = lane; i < B; i += 64) {
        TV pr = tv_add(mk_tv(0.0, PFRL_TAG_PY), tv_div(mk_tv(out_pri[i], out_pri_tag[i]), total));
        out_prob[i] = pr.v;
        local_min = fmin(local_min, pr.v);
    }
    for (int off = 32; off > 0; off >>= 1) local_min = fmin(local_min, __shfl_xor(local_min, off));
    double min_prob = tv_div(mk_tv(s_min_v, s_min_t), total).v;
    if (lane == 0) *out_min_prob = min_prob;
    if (normalize == 1) min_prob = local_min;
    for (int64_t i = lane; i < B; i += 64) {
        const double p = out_prob[i];
        double w;
        if (normalize)
            w = pow(p / min_prob, -beta);
        else
            w = pow((double)T.length * p, -beta);
        out_weight[i] = (float)w;
        if (out_slot) out_slot[i] = (int32_t)(out_x[i] % slot_mod);
    }
}


// ---------------------------------------------------------------------------
// Lean sampler with a prefetching second wave.  What is left on the chain of
// k_tree_sample_lean is the fetch of the 1023-node subtree under the level-9
// node a draw reaches (1.7 us of 5.5, almost all of it memory latency), and it
// cannot start before the draw's own top descent has finished.  It can be
// PREDICTED, though: draw i + 1 starts from root_i * u, and root_i differs from
// the root before draw i by one leaf out of a million, so a descent of the top
// heap as it stands while draw i is still in flight ends in the right level-9
// node almost always.  Wave 1 does that -- predicts draw i + 1 while wave 0 works
// on draw i, and loads that subtree into the other of two LDS buffers.  Wave 0
// descends the top heap exactly, as before; if it arrives where wave 1 predicted
// AND no earlier draw of this launch went through that subtree (its zeroed leaf
// would be missing from a fetch issued before or beside the write-back), the
// subtree is already in LDS; otherwise it fetches it itself, exactly as
// k_tree_sample_lean does.  The prediction decides where bytes are fetched from
// early, never what is computed: results are those of k_tree_sample bit for bit.
// One workgroup barrier per draw (both waves, top of the loop).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {
    // LDS traffic of ONE wave executes in program order; this only keeps the compiler from
    // moving accesses across (no workgroup barrier: the other wave is elsewhere)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the 2^(r+1) - 1 nodes under the level-r node whose first leaf is x0, in two halves: the loads
// (kSubLoads values + tags per lane, all requested before anything waits) and the stores into an
// LDS heap (bv, bt).  Lane l of the calling wave holds the addressing constants of level l.
constexpr int kSubLoads = 8 + 4 + 2 + kBotLevels - 2;

__device__ __forceinline__ void subtree_loads(const pfrl_tree_t &T, int64_t x0, int r, int lane,
                                              int64_t my_off, int64_t my_org, int64_t my_mask,
                                              double (&v)[kSubLoads], uint8_t (&tg)[kSubLoads]) {
    int n_ld = 0;
#pragma unroll
    for (int l = 0; l <= kBotLevels; ++l) {
        const int lr = l <= r ? l : r;               // clamp (uniform)
        const int cnt = l <= r ? (1 << (r - l)) : 0;
        const int64_t org = readlane_i64(my_org, lr);
        const int64_t off = readlane_i64(my_off, lr), mask = readlane_i64(my_mask, lr);
        const int64_t q0 = (x0 - org) >> lr;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k * 64 < (1 << (kBotLevels - l))) {
                const int j = k * 64 + lane;
                const int64_t gi = off + ((q0 + (j < cnt ? j : 0)) & mask);
                v[n_ld] = T.sum_val[gi];
                tg[n_ld] = T.sum_tag[gi];
                ++n_ld;
            }
        }
    }
}

__device__ __forceinline__ void subtree_stores(int r, int lane, const double (&v)[kSubLoads],
                                               const uint8_t (&tg)[kSubLoads], double *bv, uint8_t *bt) {
    int n_ld = 0;
#pragma unroll
    for (int l = 0; l <= kBotLevels; ++l) {
        const int cnt = l <= r ? (1 << (r - l)) : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k * 64 < (1 << (kBotLevels - l))) {
                const int j = k * 64 + lane;
                if (j < cnt) {
                    bv[cnt + j] = v[n_ld];   // heap index of (level l, j)
                    bt[cnt + j] = tg[n_ld];
                }
                ++n_ld;
            }
        }
    }
}

__device__ __forceinline__ void fetch_subtree(const pfrl_tree_t &T, int64_t x0, int r, int lane,
                                              int64_t my_off, int64_t my_org, int64_t my_mask,
                                              double *bv, uint8_t *bt) {
    double v[kSubLoads];
    uint8_t tg[kSubLoads];
    subtree_loads(T, x0, r, lane, my_off, my_org, my_mask, v, tg);
    subtree_stores(r, lane, v, tg, bv, bt);
}

// the level-r node a draw with uniform u would reach on the top heap as it stands
__device__ __forceinline__ int predict_level_r(const double *top_v, const uint8_t *top_t, int n, double u) {
    double p64 = __dadd_rn(0.0, __dmul_rn(top_v[1], u));
    float p32 = (float)p64;
    int pt = PFRL_TAG_PY;
    return find_down(top_v, top_t, 1, n, p64, p32, pt);
}

__device__ __forceinline__ void tree_sample_lean2_body(
    const pfrl_tree_t &T, int64_t B, const double *__restrict__ u01, int64_t *__restrict__ out_x,
    double *__restrict__ out_pri, uint8_t *__restrict__ out_pri_tag, double *__restrict__ out_prob,
    float *__restrict__ out_weight, double *__restrict__ out_total,
    uint8_t *__restrict__ out_total_tag, double *__restrict__ out_min_prob, int normalize,
    double beta, int64_t slot_mod, int32_t *__restrict__ out_slot, unsigned char *smem) {
    const int L = T.log2_size;
    const int r = L < kBotLevels ? L : kBotLevels;
    const int top_levels = L - r + 1;           // levels L..r  -> depths 0..top_levels-1
    const int top_n = 1 << top_levels;          // heap indices 1..top_n-1
    const int bot_n = 1 << (r + 1);             // heap indices 1..bot_n-1 (levels r..0)
    double *top_v = reinterpret_cast<double *>(smem);
    double *bot_v2 = top_v + top_n;             // two subtree buffers
    uint8_t *top_t = reinterpret_cast<uint8_t *>(bot_v2 + 2 * bot_n);
    uint8_t *bot_t2 = top_t + top_n;
    __shared__ double s_total_v, s_min_v;
    __shared__ int s_total_t, s_min_t;
    __shared__ int s_pred_h[2];
    __shared__ int64_t lv_off[PFRL_MAX_LEVELS], lv_org[PFRL_MAX_LEVELS], lv_mask[PFRL_MAX_LEVELS];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
#ifdef PFRL_TREE_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
#ifdef PFRL_TREE_DEBUG
    // phase clocks of wave 0 (tools/per_dbg2.py): [0] prologue, [1] the draws, [2] epilogue
    unsigned long long t_dbg0 = wall_clock64(), t_dbg1 = 0, t_dbg2 = 0;
#endif
    if (tid <= L) {
        const int sh = T.log2_smax - tid;
        lv_off[tid] = T.level_off[tid];
        lv_org[tid] = T.origin[tid];
        lv_mask[tid] = (sh > 0 ? ((int64_t)1 << sh) : 1) - 1;
    }
    if (tid < 2) s_pred_h[tid] = -1;
    __syncthreads();
#define NODE_AT(l, x) (lv_off[l] + ((((x) - lv_org[l]) >> (l)) & lv_mask[l]))
    const int myl = lane <= L ? lane : 0;
    const int64_t my_off = lv_off[myl], my_org = lv_org[myl], my_mask = lv_mask[myl];
    // stage the top of the sum tree, both waves (loads batched 16 deep per thread)
    for (int h0 = 1; h0 < top_n; h0 += 128 * 16) {
        double v[16];
        uint8_t tg[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            // (unconditional loads from a clamped, always valid node: a load inside a branch is
            // waited for before the next one is issued -- 32 round trips per batch instead of one)
            const int hh = h0 + k * 128 + tid;
            const int h = hh < top_n ? hh : top_n - 1;
            const int d = 31 - __clz(h);
            const int l = L - d;
            const int64_t gi = NODE_AT(l, T.base + ((int64_t)(h - (1 << d)) << l));
            v[k] = T.sum_val[gi];
            tg[k] = T.sum_tag[gi];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int h = h0 + k * 128 + tid;
            if (h < top_n) {
                top_v[h] = v[k];
                top_t[h] = tg[k];
            }
        }
    }
    if (tid == 0) {
        const int64_t iroot = NODE_AT(L, T.base);
        s_total_v = T.sum_val[iroot];
        s_total_t = T.sum_tag[iroot];
        s_min_v = T.min_val[iroot];
        s_min_t = T.min_tag[iroot];
    }
#undef NODE_AT
    double my_u = 0.0;
    int myh = -1;       // wave 0, lane j: the level-r node draw j went through (first 64 draws)
#ifdef PFRL_TREE_DEBUG
    __syncthreads();
    t_dbg1 = wall_clock64();
#endif
    if (wave == 1) my_u = lane < B ? u01[lane] : 0.0;      // the prefetcher looks at draws 1..63 only
    // Both waves pass B workgroup barriers, one per draw (wave 0 at the top of its loop, wave 1 at
    // the top of its own): draw i - 1 is complete and the subtree predicted for draw i is in LDS.
    if (wave == 0) {
        // (the uniforms are loaded 64 draws at a time OUTSIDE the loop over those draws: a load
        // anywhere inside makes every iteration wait for the wave's vector-memory counter, i.e.
        // for the previous draw's write-back stores to reach memory -- 1-3 us beside a busy GPU)
        for (int64_t i0 = 0; i0 < B; i0 += 64) {
          my_u = i0 + lane < B ? u01[i0 + lane] : 0.0;
          const int64_t i_end = i0 + 64 < B ? i0 + 64 : B;
          // (first use of the loaded register here, so that the wait for it is here too)
          double u_next = readlane_f64(my_u, 0);
          for (int64_t i = i0; i < i_end; ++i) {
            __syncthreads();
            const double u = u_next;
            u_next = readlane_f64(my_u, (int)((i + 1 - i0) & 63));
            // ---- top heap: L - r levels ----
            const double rootv = top_v[1];
            // np.random.uniform(0.0, root) = 0.0 + (root - 0.0) * u
            double p64 = __dadd_rn(0.0, __dmul_rn(rootv, u));
            float p32 = (float)p64;
            int pt = PFRL_TAG_PY;
            const int h = find_down(top_v, top_t, 1, L - r, p64, p32, pt);
            const int64_t x0 = T.base + ((int64_t)(h - (top_n >> 1)) << r);
            double *bot_v = bot_v2 + (i & 1) * bot_n;
            uint8_t *bot_t = bot_t2 + (i & 1) * bot_n;
            bool prefetched = false;
            if (i > 0 && i < 64) {
                const bool seen = __ballot(lane < (int)i && myh == h) != 0;
                prefetched = !seen && s_pred_h[i & 1] == h;
            }
            if (lane == (int)(i & 63)) myh = h;
            if (!prefetched) {
                fetch_subtree(T, x0, r, lane, my_off, my_org, my_mask, bot_v, bot_t);
                // vmcnt(0) HERE, on the rare path: the last stores of the fetch sit behind a lane
                // mask, so without it the compiler waits for "possibly pending" loads after the
                // join -- on every draw, and the counter it waits for also holds the previous
                // draw's write-back stores (1-3 us to reach memory beside a busy GPU)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                wave_lds_fence();
            }
            // ---- bottom heap: r levels ----
            const int g = find_down(bot_v, bot_t, 1, r, p64, p32, pt);
            const int64_t x = x0 + (g - (bot_n >> 1));
            const double leaf_v = bot_v[g];
            const uint8_t leaf_t = bot_t[g];
            // ---- the siblings of the path: lane l reads the one of level l ----
            int st = PFRL_TAG_ABSENT;
            double sv = 0.0;
            if (lane < L) {
                const bool below = lane < r;
                const int node = below ? (g >> lane) : (h >> (lane - r));
                const double *sib_v = below ? bot_v : top_v;
                const uint8_t *sib_t = below ? bot_t : top_t;
                st = sib_t[node ^ 1];
                sv = sib_v[node ^ 1];
            }
            if (st == PFRL_TAG_ABSENT) sv = 0.0;
            if (lane == 0) {
                out_x[i] = x;
                out_pri[i] = leaf_v;
                out_pri_tag[i] = leaf_t;
            }
            // ---- zero the leaf, re-reduce the path ----
            const float sv32 = (float)sv;
            const unsigned long long m2 = __ballot(st >= PFRL_TAG_F32);
            const unsigned long long m3 = __ballot(st == PFRL_TAG_F64);
            const int l2 = m2 ? __builtin_ctzll(m2) : L;
            const int l3 = m3 ? __builtin_ctzll(m3) : L;
            double c64 = 0.0, mine64 = 0.0;
            float c32 = 0.0f, mine32 = 0.0f;
            int j = 0;
            for (; j < l2; ++j) {
                c64 = __dadd_rn(c64, readlane_f64(sv, j));
                if (lane == j + 1) mine64 = c64;
            }
            if (l2 < l3) {
                c32 = (float)c64;
                for (; j < l3; ++j) {
                    c32 = __fadd_rn(c32, readlane_f32(sv32, j));
                    if (lane == j + 1) mine32 = c32;
                }
                c64 = (double)c32;
            }
            for (; j < L; ++j) {
                c64 = __dadd_rn(c64, readlane_f64(sv, j));
                if (lane == j + 1) mine64 = c64;
            }
            wave_lds_fence();           // (sibling reads of the top heap before its nodes are rewritten)
            // ---- lane l owns the node of level l: top heap + HBM ----
            if (lane <= L) {
                const int l = lane;
                double v;
                int tg;
                if (l == 0) {
                    v = 0.0;
                    tg = PFRL_TAG_PY;
                } else {
                    tg = l > l3 ? PFRL_TAG_F64 : (l > l2 ? PFRL_TAG_F32 : PFRL_TAG_PY);
                    v = tg == PFRL_TAG_F32 ? (double)mine32 : mine64;
                }
                if (l >= r) {
                    const int hh = h >> (l - r);
                    top_v[hh] = v;
                    top_t[hh] = (uint8_t)tg;
                }
                const int64_t gi = my_off + (((x - my_org) >> l) & my_mask);
                T.sum_val[gi] = v;
                T.sum_tag[gi] = (uint8_t)tg;
            }
          }
        }
    } else {
        // ---- wave 1: where will the next draws go?  (the top heap may change under these
        // descents: any outcome is a valid level-r node, and wave 0 checks it.)  The subtree of
        // draw i + 1 must be in LDS when iteration i ends; beside the backward / optimizer launches
        // of the other stream a fetch takes longer than a draw, so its loads are requested one
        // iteration earlier and stay in flight across the barrier (the registers that receive
        // them are written in ONE place per iteration: a merge of two definitions made the
        // compiler copy them, i.e. wait for them, at the top of the loop).
        const int last = (int)((B < 64 ? B : 64) - 1);       // draws beyond are wave 0's own business
        double pf_v[kSubLoads];
        uint8_t pf_t[kSubLoads];
        int pend_h;
        __syncthreads();                                     // (iteration 0: the top heap is staged)
        {
            const int k1 = 1 < last ? 1 : last;
            pend_h = predict_level_r(top_v, top_t, L - r, readlane_f64(my_u, k1));
            subtree_loads(T, T.base + ((int64_t)(pend_h - (top_n >> 1)) << r), r, lane, my_off, my_org,
                          my_mask, pf_v, pf_t);
            if (1 <= last) {
                subtree_stores(r, lane, pf_v, pf_t, bot_v2 + bot_n, bot_t2 + bot_n);
                if (lane == 0) s_pred_h[1] = pend_h;
            }
            const int k2 = 2 < last ? 2 : last;
            pend_h = predict_level_r(top_v, top_t, L - r, readlane_f64(my_u, k2));
            subtree_loads(T, T.base + ((int64_t)(pend_h - (top_n >> 1)) << r), r, lane, my_off, my_org,
                          my_mask, pf_v, pf_t);
        }
        for (int64_t i = 1; i < B; ++i) {
            __syncthreads();
            if (i + 1 <= last) {
                subtree_stores(r, lane, pf_v, pf_t, bot_v2 + ((i + 1) & 1) * bot_n,
                               bot_t2 + ((i + 1) & 1) * bot_n);
                if (lane == 0) s_pred_h[(i + 1) & 1] = pend_h;
            }
            // (always: past the last prefetched draw the same subtree is requested again and
            // nobody reads it)
            const int k2 = i + 2 < last ? (int)(i + 2) : last;
            pend_h = predict_level_r(top_v, top_t, L - r, readlane_f64(my_u, k2));
            subtree_loads(T, T.base + ((int64_t)(pend_h - (top_n >> 1)) << r), r, lane, my_off, my_org,
                          my_mask, pf_v, pf_t);
        }
    }
    __threadfence_block();
    __syncthreads();
#ifdef PFRL_TREE_DEBUG
    t_dbg2 = wall_clock64();
#endif
    if (wave == 0) {
        if (lane == 0) {
            *out_total = s_total_v;
            *out_total_tag = (uint8_t)s_total_t;
        }
        const TV total = mk_tv(s_total_v, s_total_t);
        double local_min = __builtin_huge_val();
        for (int64_t i = lane; i < B; i += 64) {
            TV pr = tv_add(mk_tv(0.0, PFRL_TAG_PY), tv_div(mk_tv(out_pri[i], out_pri_tag[i]), total));
            out_prob[i] = pr.v;
            local_min = fmin(local_min, pr.v);
        }
        for (int off = 32; off > 0; off >>= 1) local_min = fmin(local_min, __shfl_xor(local_min, off));
        double min_prob = tv_div(mk_tv(s_min_v, s_min_t), total).v;
        if (lane == 0) *out_min_prob = min_prob;
        if (normalize == 1) min_prob = local_min;
        for (int64_t i = lane; i < B; i += 64) {
            const double p = out_prob[i];
            double w;
            if (normalize)
                w = pow(p / min_prob, -beta);
            else
                w = pow((double)T.length * p, -beta);
            out_weight[i] = (float)w;
            if (out_slot) out_slot[i] = (int32_t)(out_x[i] % slot_mod);
        }
    }
#ifdef PFRL_TREE_DEBUG
    if (tid == 0) {
        const unsigned long long t3 = wall_clock64();
        g_dbg[0] = (t_dbg1 - t_dbg0) * (unsigned long long)B;     // (per_dbg2 divides by B)
        g_dbg[1] = (t_dbg2 - t_dbg1) * (unsigned long long)B;
        g_dbg[2] = (t3 - t_dbg2) * (unsigned long long)B;
        g_dbg[3] = g_dbg[4] = 0;
    }
#endif
}

__global__ __launch_bounds__(128) void k_tree_sample_lean2(
    pfrl_tree_t T, int64_t B, const double *__restrict__ u01, int64_t *__restrict__ out_x,
    double *__restrict__ out_pri, uint8_t *__restrict__ out_pri_tag, double *__restrict__ out_prob,
    float *__restrict__ out_weight, double *__restrict__ out_total,
    uint8_t *__restrict__ out_total_tag, double *__restrict__ out_min_prob, int normalize,
    double beta, int64_t slot_mod, int32_t *__restrict__ out_slot) {
    extern __shared__ __align__(16) unsigned char smem[];
    tree_sample_lean2_body(T, B, u01, out_x, out_pri, out_pri_tag, out_prob, out_weight, out_total,
                           out_total_tag, out_min_prob, normalize, beta, slot_mod, out_slot, smem);
}

}  // namespace

#ifdef PFRL_TREE_DEBUG
extern "C" int pfrl_tree_debug_read(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 8);
}
#endif

extern "C" int pfrl_tree_write(const pfrl_tree_t *tree, int64_t n, const int64_t *x,
                               const double *val, const uint8_t *tag, const uint8_t *use_maxp,
                               void *stream) {
    PFRL_CHECK_ARG(tree && n <= kMaxBatch, "pfrl_tree_write: n must be <= 1024");
    if (n <= 0) return 0;
    int threads = (int)((n + 63) / 64 * 64);
    hipLaunchKernelGGL(k_tree_write, dim3(1), dim3(threads), 0, (hipStream_t)stream, *tree, n, x,
                       val, tag, use_maxp);
    PFRL_LAUNCH_CHECK();
}

// k_tree_update_errors_write for launches of up to kFastThreads (128) leaves: the same leaf stores, then
// repair_paths_hashed instead of repair_paths.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_tree_update_errors_write_fast(
    pfrl_tree_t T, int64_t B, const int64_t *__restrict__ x, const float *__restrict__ err,
    ErrCfg c, int dedupe, int64_t n, const int64_t *__restrict__ wx,
    const double *__restrict__ wval, const uint8_t *__restrict__ wtag,
    const uint8_t *__restrict__ wuse_maxp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fast_lds_raw[];
    FastLds<THREADS> &S = *reinterpret_cast<FastLds<THREADS> *>(fast_lds_raw);
    const int i = threadIdx.x;
    TV p = mk_tv(0.0, PFRL_TAG_PY);
    if (i < B) p = priority_of_error(c, err[i]);
    int64_t xi;
    bool active = set_priorities_leaves(T, B, x, p, dedupe, S.s_v, S.s_t, S.s_x, xi);
    __threadfence_block();
    __syncthreads();             // max_priority and the minibatch's leaves are in place
    const int64_t k = (int64_t)i - B;
    if (k >= 0 && k < n) {
        xi = wx[k];
        TV q;
        if (wuse_maxp && wuse_maxp[k])
            q = mk_tv(*T.maxp_val, *T.maxp_tag);
        else
            q = mk_tv(wval[k], wtag[k]);
        const int64_t il = node_idx(T, 0, xi);
        T.sum_val[il] = q.v;
        T.sum_tag[il] = (uint8_t)q.t;
        T.min_val[il] = q.v;
        T.min_tag[il] = (uint8_t)q.t;
        active = true;
    }
    repair_paths_hashed(T, active, xi, S);
}

// The priority update of minibatch k, the leaf writes recorded since AND the draws of minibatch
// k + 1 as ONE launch (VERDICT r4 next #1b): what separates the TD errors from the next minibatch on
// the replay stream was update launch -> boundary -> sampler launch -> its prologue; here the
// 128 threads of the sampler first run k_tree_update_errors_write_fast's body (leaves in the first
// 64 threads, hashed path repair), then -- one barrier later -- the sampler proper on the repaired
// tree.  Same arithmetic in the same order as the two launches: bit-identical trees and draws.
__global__ __launch_bounds__(128) void k_tree_update_errors_write_sample(
    pfrl_tree_t T, int64_t Be, const int64_t *__restrict__ x, const float *__restrict__ err,
    ErrCfg c, int dedupe, int64_t n, const int64_t *__restrict__ wx,
    const double *__restrict__ wval, const uint8_t *__restrict__ wtag,
    const uint8_t *__restrict__ wuse_maxp, int64_t B, const double *__restrict__ u01,
    int64_t *__restrict__ out_x, double *__restrict__ out_pri, uint8_t *__restrict__ out_pri_tag,
    double *__restrict__ out_prob, float *__restrict__ out_weight, double *__restrict__ out_total,
    uint8_t *__restrict__ out_total_tag, double *__restrict__ out_min_prob, int normalize,
    double beta, int64_t slot_mod, int32_t *__restrict__ out_slot, int sampler_lds) {
    extern __shared__ __align__(16) unsigned char fused_smem[];
    FastLds<64> &S = *reinterpret_cast<FastLds<64> *>(fused_smem + sampler_lds);
    const int i = threadIdx.x;
    TV p = mk_tv(0.0, PFRL_TAG_PY);
    if (i < Be) p = priority_of_error(c, err[i]);
    int64_t xi;
    bool active = set_priorities_leaves(T, Be, x, p, dedupe, S.s_v, S.s_t, S.s_x, xi);
    __threadfence_block();
    __syncthreads();             // max_priority and the minibatch's leaves are in place
    const int64_t k = (int64_t)i - Be;
    if (k >= 0 && k < n) {
        xi = wx[k];
        TV q;
        if (wuse_maxp && wuse_maxp[k])
            q = mk_tv(*T.maxp_val, *T.maxp_tag);
        else
            q = mk_tv(wval[k], wtag[k]);
        const int64_t il = node_idx(T, 0, xi);
        T.sum_val[il] = q.v;
        T.sum_tag[il] = (uint8_t)q.t;
        T.min_val[il] = q.v;
        T.min_tag[il] = (uint8_t)q.t;
        active = true;
    }
    repair_paths_hashed<64>(T, active, xi, S);
    __threadfence_block();
    __syncthreads();             // the repaired tree is what the draws see
    tree_sample_lean2_body(T, B, u01, out_x, out_pri, out_pri_tag, out_prob, out_weight, out_total,
                           out_total_tag, out_min_prob, normalize, beta, slot_mod, out_slot, fused_smem);
}

extern "C" int pfrl_tree_write_sum(const pfrl_tree_t *tree, int64_t n, const int64_t *x,
                                   const double *val, const uint8_t *tag, double *old_val,
                                   uint8_t *old_tag, void *stream) {
    PFRL_CHECK_ARG(tree && n <= kMaxBatch, "pfrl_tree_write_sum: n must be <= 1024");
    PFRL_CHECK_ARG((val == nullptr) == (tag == nullptr) && (old_val == nullptr) == (old_tag == nullptr),
                   "pfrl_tree_write_sum: value / tag arrays come in pairs");
    if (n <= 0) return 0;
    int threads = (int)((n + 63) / 64 * 64);
    hipLaunchKernelGGL(k_tree_write_sum, dim3(1), dim3(threads), 0, (hipStream_t)stream, *tree, n, x,
                       val, tag, old_val, old_tag);
    PFRL_LAUNCH_CHECK();
}

// hipFuncSetAttribute is per DEVICE: a process that drives trees on a second device must set the
// dynamic-LDS limit there too.  One bit per device ordinal per call site.
static bool attr_needed(unsigned long long *seen) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    const unsigned long long bit = 1ull << dev;
    if (*seen & bit) return false;
    *seen |= bit;
    return true;
}

extern "C" int pfrl_tree_sample(const pfrl_tree_t *tree, int64_t B, const double *u01,
                                int64_t *out_x, double *out_pri, uint8_t *out_pri_tag,
                                double *out_prob, float *out_weight, double *out_total,
                                uint8_t *out_total_tag, double *out_min_prob, int normalize,
                                double beta, int64_t slot_mod, int32_t *out_slot, void *stream) {
    PFRL_CHECK_ARG(tree && B >= 0, "pfrl_tree_sample: bad args");
    PFRL_CHECK_ARG(tree->length >= B, "pfrl_tree_sample: fewer items than requested");
    if (B == 0) return 0;
    const int L = tree->log2_size;
    const int r = L < kBotLevels ? L : kBotLevels;
    // PFRL_TREE_SAMPLE: "prefetch" (default) = lean sampler + prefetching wave, "lean" = without it,
    // "lds" = the round-2 sampler, "global" = global-memory descent
    static int mode = -1;
    if (mode < 0) {
        const char *e = getenv("PFRL_TREE_SAMPLE");
        const char *old = getenv("PFRL_TREE_SAMPLE_LDS");
        mode = 3;
        if (e && e[0] == 'l' && e[1] == 'e') mode = 2;
        if (e && e[0] == 'l' && e[1] == 'd') mode = 1;
        if ((e && e[0] == 'g') || (old && old[0] == '0')) mode = 0;
    }
    if (mode && L - r + 1 <= kMaxTopLog2) {
        const size_t top_n = (size_t)1 << (L - r + 1), bot_n = (size_t)1 << (r + 1);
        const size_t lds = (top_n + bot_n) * (sizeof(double) + 1);
        const size_t lds2 = (top_n + 2 * bot_n) * (sizeof(double) + 1);
        static unsigned long long attr_seen = 0;
        if (attr_needed(&attr_seen)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tree_sample_lds),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tree_sample_lean),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tree_sample_lean2),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
        }
        if (mode == 3)
            hipLaunchKernelGGL(k_tree_sample_lean2, dim3(1), dim3(128), lds2, (hipStream_t)stream,
                               *tree, B, u01, out_x, out_pri, out_pri_tag, out_prob, out_weight,
                               out_total, out_total_tag, out_min_prob, normalize, beta,
                               slot_mod > 0 ? slot_mod : 1, out_slot);
        else if (mode == 2)
            hipLaunchKernelGGL(k_tree_sample_lean, dim3(1), dim3(64), lds, (hipStream_t)stream,
                               *tree, B, u01, out_x, out_pri, out_pri_tag, out_prob, out_weight,
                               out_total, out_total_tag, out_min_prob, normalize, beta,
                               slot_mod > 0 ? slot_mod : 1, out_slot);
        else
            hipLaunchKernelGGL(k_tree_sample_lds, dim3(1), dim3(64), lds, (hipStream_t)stream, *tree,
                               B, u01, out_x, out_pri, out_pri_tag, out_prob, out_weight, out_total,
                               out_total_tag, out_min_prob, normalize, beta,
                               slot_mod > 0 ? slot_mod : 1, out_slot);
    } else {
        hipLaunchKernelGGL(k_tree_sample, dim3(1), dim3(64), 0, (hipStream_t)stream, *tree, B, u01,
                           out_x, out_pri, out_pri_tag, out_prob, out_weight, out_total,
                           out_total_tag, out_min_prob, normalize, beta,
                           slot_mod > 0 ? slot_mod : 1, out_slot);
    }
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_tree_update_errors_f32(const pfrl_tree_t *tree, int64_t B, const int64_t *x,
                                           const float *err, int has_min, float error_min,
                                           double pri_at_min, int has_max, float error_max,
                                           double pri_at_max, double eps, double alpha, int dedupe,
                                           int pow_mode, void *stream) {
    PFRL_CHECK_ARG(tree && B <= kMaxBatch, "pfrl_tree_update_errors_f32: B must be <= 1024");
    PFRL_CHECK_ARG(pow_mode >= 0 && pow_mode <= 2, "pfrl_tree_update_errors_f32: bad pow_mode");
    if (B <= 0) return 0;
    ErrCfg c;
    c.has_min = has_min;
    c.has_max = has_max;
    c.error_min = error_min;
    c.error_max = error_max;
    c.pri_at_min = pri_at_min;
    c.pri_at_max = pri_at_max;
    c.eps = eps;
    c.alpha = alpha;
    c.pow_mode = pow_mode;
    int threads = (int)((B + 63) / 64 * 64);
    hipLaunchKernelGGL(k_tree_update_errors, dim3(1), dim3(threads), 0, (hipStream_t)stream, *tree,
                       B, x, err, c, dedupe);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_tree_update_errors_write_f32(
    const pfrl_tree_t *tree, int64_t B, const int64_t *x, const float *err, int has_min,
    float error_min, double pri_at_min, int has_max, float error_max, double pri_at_max, double eps,
    double alpha, int dedupe, int pow_mode, int64_t n, const int64_t *wx, const double *wval,
    const uint8_t *wtag, const uint8_t *wuse_maxp, void *stream) {
    PFRL_CHECK_ARG(tree && B >= 0 && n >= 0 && B + n <= kMaxBatch,
                   "pfrl_tree_update_errors_write_f32: B + n must be <= 1024");
    PFRL_CHECK_ARG(pow_mode >= 0 && pow_mode <= 2, "pfrl_tree_update_errors_write_f32: bad pow_mode");
    PFRL_CHECK_ARG(n == 0 || (wx && wval && wtag), "pfrl_tree_update_errors_write_f32: null write list");
    if (B + n <= 0) return 0;
    ErrCfg c;
    c.has_min = has_min;
    c.has_max = has_max;
    c.error_min = error_min;
    c.error_max = error_max;
    c.pri_at_min = pri_at_min;
    c.pri_at_max = pri_at_max;
    c.eps = eps;
    c.alpha = alpha;
    c.pow_mode = pow_mode;
    int threads = (int)((B + n + 63) / 64 * 64);
    // PFRL_TREE_REPAIR=levels: the level-by-level repair for every launch (A/B, tests)
    const char *repair_env = getenv("PFRL_TREE_REPAIR");     // (read per call: tests switch it)
    const bool hashed = !(repair_env != nullptr && repair_env[0] == 'l');
    if (hashed && B + n <= kFastThreads && tree->log2_size <= kFastLevels && tree->log2_size >= 1) {
        static unsigned long long attr_seen = 0;
        if (attr_needed(&attr_seen)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tree_update_errors_write_fast<128>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(FastLds<128>));
        }
        if (B + n <= 64)
            hipLaunchKernelGGL(k_tree_update_errors_write_fast<64>, dim3(1), dim3(64), sizeof(FastLds<64>),
                               (hipStream_t)stream, *tree, B, x, err, c, dedupe, n, wx, wval, wtag, wuse_maxp);
        else
            hipLaunchKernelGGL(k_tree_update_errors_write_fast<128>, dim3(1), dim3(128), sizeof(FastLds<128>),
                               (hipStream_t)stream, *tree, B, x, err, c, dedupe, n, wx, wval, wtag, wuse_maxp);
    }
    else
        hipLaunchKernelGGL(k_tree_update_errors_write, dim3(1), dim3(threads), 0, (hipStream_t)stream,
                           *tree, B, x, err, c, dedupe, n, wx, wval, wtag, wuse_maxp);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_tree_update_errors_write_sample(
    const pfrl_tree_t *tree, int64_t Be, const int64_t *x, const float *err, int has_min,
    float error_min, double pri_at_min, int has_max, float error_max, double pri_at_max, double eps,
    double alpha, int dedupe, int pow_mode, int64_t n, const int64_t *wx, const double *wval,
    const uint8_t *wtag, const uint8_t *wuse_maxp, int64_t B, const double *u01, int64_t *out_x,
    double *out_pri, uint8_t *out_pri_tag, double *out_prob, float *out_weight, double *out_total,
    uint8_t *out_total_tag, double *out_min_prob, int normalize, double beta, int64_t slot_mod,
    int32_t *out_slot, void *stream) {
    PFRL_CHECK_ARG(tree && Be >= 1 && n >= 0 && Be + n <= 64 && B >= 1,
                   "pfrl_tree_update_errors_write_sample: at most 64 priorities + writes");
    PFRL_CHECK_ARG(pow_mode >= 0 && pow_mode <= 2, "pfrl_tree_update_errors_write_sample: bad pow_mode");
    PFRL_CHECK_ARG(n == 0 || (wx && wval && wtag), "pfrl_tree_update_errors_write_sample: null write list");
    const int L = tree->log2_size;
    const int r = L < kBotLevels ? L : kBotLevels;
    PFRL_CHECK_ARG(L >= 1 && L <= kFastLevels && L - r + 1 <= kMaxTopLog2,
                   "pfrl_tree_update_errors_write_sample: tree frame outside the fused kernel");
    ErrCfg c;
    c.has_min = has_min;
    c.has_max = has_max;
    c.error_min = error_min;
    c.error_max = error_max;
    c.pri_at_min = pri_at_min;
    c.pri_at_max = pri_at_max;
    c.eps = eps;
    c.alpha = alpha;
    c.pow_mode = pow_mode;
    const size_t top_n = (size_t)1 << (L - r + 1), bot_n = (size_t)1 << (r + 1);
    const size_t lds2 = ((top_n + 2 * bot_n) * (sizeof(double) + 1) + 15) & ~(size_t)15;
    const size_t total = lds2 + sizeof(FastLds<64>);
    static unsigned long long attr_seen = 0;
    if (attr_needed(&attr_seen)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_tree_update_errors_write_sample),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    }
    PFRL_CHECK_ARG(total <= 156 * 1024, "pfrl_tree_update_errors_write_sample: LDS");
    hipLaunchKernelGGL(k_tree_update_errors_write_sample, dim3(1), dim3(128), total, (hipStream_t)stream,
                       *tree, Be, x, err, c, dedupe, n, wx, wval, wtag, wuse_maxp, B, u01, out_x, out_pri,
                       out_pri_tag, out_prob, out_weight, out_total, out_total_tag, out_min_prob, normalize,
                       beta, slot_mod > 0 ? slot_mod : 1, out_slot, (int)lds2);
    PFRL_LAUNCH_CHECK();
}

// the priority transform's power on plain arrays (what k_tree_update_errors evaluates per
// leaf): parity tests sweep it over millions of float32 values
__global__ __launch_bounds__(256) void k_powf(int pow_mode, const float *__restrict__ x, float alpha,
                                              float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float s = x[i];
    float r;
    if (pow_mode == PFRL_POW_GLIBC_FMA)
        r = pfrl_powf::powf_glibc<true>(s, alpha);
    else if (pow_mode == PFRL_POW_GLIBC)
        r = pfrl_powf::powf_glibc<false>(s, alpha);
    else
        r = (float)pow((double)s, (double)alpha);
    out[i] = r;
}

extern "C" int pfrl_powf_device(int pow_mode, const float *x, float alpha, float *out, int64_t n,
                                void *stream) {
    PFRL_CHECK_ARG(x && out && n >= 0, "pfrl_powf_device: null argument");
    PFRL_CHECK_ARG(pow_mode >= 0 && pow_mode <= 2, "pfrl_powf_device: bad pow_mode");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_powf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pow_mode, x, alpha, out, n);
    PFRL_LAUNCH_CHECK();
}

extern "C" int pfrl_tree_set_priorities(const pfrl_tree_t *tree, int64_t B, const int64_t *x,
                                        const double *val, const uint8_t *tag, int dedupe,
                                        void *stream) {
    PFRL_CHECK_ARG(tree && B <= kMaxBatch, "pfrl_tree_set_priorities: B must be <= 1024");
    if (B <= 0) return 0;
    int threads = (int)((B + 63) / 64 * 64);
    hipLaunchKernelGGL(k_tree_set_priorities, dim3(1), dim3(threads), 0, (hipStream_t)stream, *tree,
                       B, x, val, tag, dedupe);
    PFRL_LAUNCH_CHECK();
}
