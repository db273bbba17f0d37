// Host-side native helpers of the data path (no kernels in this file): the probe that picks
// the device restatement of libm's powf, and the step planner that walks the reference's
// NumPy draws for a whole batched env step.
#include <math.h>
#include <stdint.h>

#include "common.h"
#include <vector>
#include "powf_glibc.h"

// One pinned-host -> device transfer on `stream` (hipMemcpyAsync): the staging rings' control
// traffic (a few hundred bytes of indices / draws per call).  Through torch the same copy is two
// slices + copy_ + dispatch, ~15 us of host time; Rainbow makes four such uploads per update.
extern "C" int pfrl_h2d_async(void *dst, const void *host_src, int64_t nbytes, void *stream) {
    PFRL_CHECK_ARG(dst && host_src && nbytes >= 0, "pfrl_h2d_async: null argument");
    if (nbytes == 0) return 0;
    hipError_t e = hipMemcpyAsync(dst, host_src, (size_t)nbytes, hipMemcpyHostToDevice,
                                  (hipStream_t)stream);
    if (e != hipSuccess) {
        pfrl_set_error(hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" int pfrl_powf_host(int pow_mode, const float *host_x, float alpha, float *host_out,
                              int64_t n) {
    PFRL_CHECK_ARG(host_x && host_out && n >= 0, "pfrl_powf_host: null argument");
    PFRL_CHECK_ARG(pow_mode == PFRL_POW_GLIBC || pow_mode == PFRL_POW_GLIBC_FMA,
                   "pfrl_powf_host: pow_mode must be PFRL_POW_GLIBC or PFRL_POW_GLIBC_FMA");
    if (pow_mode == PFRL_POW_GLIBC_FMA)
        for (int64_t i = 0; i < n; ++i) host_out[i] = pfrl_powf::powf_glibc<true>(host_x[i], alpha);
    else
        for (int64_t i = 0; i < n; ++i) host_out[i] = pfrl_powf::powf_glibc<false>(host_x[i], alpha);
    return 0;
}

extern "C" int pfrl_powf_host_variant(float alpha, int64_t n_probe) {
    // inputs: a 64-bit LCG over the float32 values of (2^-20, 2) -- the priority transform's
    // domain is (eps, error_max + eps] -- with the caller's alpha and a few common ones
    const float alphas[] = {alpha, 0.5f, 0.6f, 0.7f};
    bool ok_plain = true, ok_fma = true;
    uint64_t state = 0x9E3779B97F4A7C15ull;
    for (int64_t i = 0; i < n_probe && (ok_plain || ok_fma); ++i) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t bits = 0x35800000u + (uint32_t)((state >> 33) % (0x40000000u - 0x35800000u));
        const float x = pfrl_powf::as_f32(bits);
        const float a = alphas[i & 3];
        volatile float xv = x, av = a;          // keep the call to libm (no constant folding)
        const uint32_t want = pfrl_powf::as_u32(powf(xv, av));
        if (ok_plain && pfrl_powf::as_u32(pfrl_powf::powf_glibc<false>(x, a)) != want) ok_plain = false;
        if (ok_fma && pfrl_powf::as_u32(pfrl_powf::powf_glibc<true>(x, a)) != want) ok_fma = false;
    }
    if (ok_fma) return PFRL_POW_GLIBC_FMA;
    if (ok_plain) return PFRL_POW_GLIBC;
    return -1;
}

// ---------------------------------------------------------------------------------------
// Step planner: the reference's walk through NumPy's legacy global stream for one batched
// env step, in native code.
//
// What "identical seeds" means is defined by host draws on np.random's global MT19937
// stream, in this order per batched step (pfrl/agents/dqn.py:490-549):
//   batch_act      per env: rand() -> [< epsilon: random_action_func() = randint(n_actions)]
//                  (pfrl/explorers/epsilon_greedy.py:8-12)
//   batch_observe  per env, in env order: append; when an update is due (len >= replay_start_size
//                  and t % update_interval == 0): n_times_update x sample_n_k(len, B)
//                  (pfrl/replay_buffer.py:329-356, pfrl/utils/random.py:4-28)
// The draws are made on NumPy's OWN generator through the function table it publishes for
// exactly this purpose (numpy/random/bitgen.h `bitgen_t`, reachable from Python as
// np.random.mtrand._rand._bit_generator.ctypes.bit_generator): same state object, same stream
// position before and after as the Python loop, no restated generator.  What is restated is how
// RandomState turns raw words into numbers (numpy/random/src/distributions/distributions.c):
//   rand()            = next_double
//   randint(0, n)     = masked rejection on 32-bit words for n - 1 <= 0xffffffff
//                       (random_bounded_uint64_fill, use_masked = 1), on 64-bit words above.
// ---------------------------------------------------------------------------------------
struct npy_bitgen {   // numpy/random/bitgen.h
    void *state;
    uint64_t (*next_uint64)(void *st);
    uint32_t (*next_uint32)(void *st);
    double (*next_double)(void *st);
    uint64_t (*next_raw)(void *st);
};

static inline uint64_t gen_mask64(uint64_t max) {
    uint64_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    mask |= mask >> 32;
    return mask;
}

// RandomState.randint(0, n) for n >= 1 (scalar and array forms draw the same way)
struct BoundedDraw {
    uint64_t rng, mask;
    explicit BoundedDraw(uint64_t n) : rng(n - 1), mask(gen_mask64(n - 1)) {}
    inline uint64_t operator()(npy_bitgen *bg) const {
        if (rng == 0) return 0;                        // no word consumed
        if (rng <= 0xFFFFFFFFull) {
            if (rng == 0xFFFFFFFFull) return bg->next_uint32(bg->state);
            uint32_t v;
            do v = bg->next_uint32(bg->state) & (uint32_t)mask; while (v > rng);
            return v;
        }
        if (rng == 0xFFFFFFFFFFFFFFFFull) return bg->next_uint64(bg->state);
        uint64_t v;
        do v = bg->next_uint64(bg->state) & mask; while (v > rng);
        return v;
    }
};

// pfrl/utils/random.py:4-28, sparse regime (3 k < n); out[0..k) = the k distinct indices,
// scratch = int64[2 k].  Same draws, same repair walk, same refill points.
static void sample_n_k_sparse(npy_bitgen *bg, int64_t n, int k, int64_t *result) {
    const BoundedDraw draw((uint64_t)n);
    for (int i = 0; i < 2 * k; ++i) result[i] = (int64_t)draw(bg);
    int j = k;
    for (int i = 0; i < k; ++i) {
        int64_t x = result[i];
        for (;;) {
            bool seen = false;
            for (int s = 0; s < i; ++s)
                if (result[s] == x) {
                    seen = true;
                    break;
                }
            if (!seen) break;
            x = result[i] = result[j];
            ++j;
            if (j == 2 * k) {
                for (int s = k; s < 2 * k; ++s) result[s] = (int64_t)draw(bg);
                j = k;
            }
        }
    }
}

extern "C" int pfrl_plan_sample_n_k(void *bitgen, int64_t n, int32_t k, int64_t *host_out) {
    PFRL_CHECK_ARG(bitgen && host_out, "pfrl_plan_sample_n_k: null argument");
    PFRL_CHECK_ARG(k >= 1 && k <= 4096 && 3 * (int64_t)k < n,
                   "pfrl_plan_sample_n_k: sparse regime only (3 k < n, k <= 4096)");
    int64_t scratch[2 * 4096];
    sample_n_k_sparse((npy_bitgen *)bitgen, n, k, scratch);
    for (int i = 0; i < k; ++i) host_out[i] = scratch[i];
    return 0;
}

extern "C" int pfrl_plan_eps_greedy(void *bitgen, int64_t n_envs, double epsilon, int64_t n_actions,
                                    int32_t *host_choice) {
    PFRL_CHECK_ARG(bitgen && host_choice && n_actions >= 1 && n_actions <= 0x7fffffff,
                   "pfrl_plan_eps_greedy: bad argument");
    npy_bitgen *bg = (npy_bitgen *)bitgen;
    const BoundedDraw draw((uint64_t)n_actions);
    for (int64_t i = 0; i < n_envs; ++i) {
        // select_action_epsilon_greedily: rand() < epsilon ? random_action_func() : greedy
        if (bg->next_double(bg->state) < epsilon)
            host_choice[i] = (int32_t)draw(bg);
        else
            host_choice[i] = -1;
    }
    return 0;
}

static inline int64_t align16(int64_t x) { return (x + 15) & ~(int64_t)15; }

extern "C" int64_t pfrl_plan_dqn_range(const pfrl_host_store_t *st, void *bitgen, int64_t m,
                                       const int32_t *s_refs, const int64_t *s_min_seq,
                                       const int32_t *n_refs, const int64_t *n_min_seq,
                                       const double *reward, const uint8_t *done, int64_t t0,
                                       int64_t replay_start, int64_t update_interval,
                                       int32_t n_times_update, int32_t B, int64_t oldest_live_fseq,
                                       int64_t *counters, uint8_t *host_block, int64_t block_bytes,
                                       int64_t *offs) {
    if (!(st && bitgen && counters && host_block && offs && m >= 1 && B >= 1 && B <= 4096 &&
          st->n == 1 && st->k >= 1 && st->k <= PFRL_MAX_STACK)) {
        pfrl_set_error("pfrl_plan_dqn_range: bad argument (one-step entries only)");
        return PFRL_ERR_ARG;
    }
    npy_bitgen *bg = (npy_bitgen *)bitgen;
    const int k = st->k;
    const int64_t n_trans0 = counters[0], n_entries0 = counters[1], head0 = counters[2];
    const int64_t maxlen = st->maxlen;
    // number of updates of this range and the regime of its first draw, BEFORE anything moves
    int64_t U = 0;
    for (int64_t j = 0; j < m; ++j) {
        const int64_t total = n_entries0 + 1 + j;
        int64_t head = head0;
        if (maxlen >= 0 && total - maxlen > head) head = total - maxlen;
        const int64_t len = total - head;
        if (maxlen < 0 && len > st->bound) {
            pfrl_set_error("unbounded ReplayBuffer exceeded its device allocation");
            return PFRL_PLAN_OVERFLOW;
        }
        if (len >= replay_start && (t0 + j + 1) % update_interval == 0) {
            if (3 * (int64_t)B >= len) return PFRL_PLAN_DENSE;   // choice(replace=False): Python
            U += n_times_update;
        }
    }
    // staging block layout
    int64_t o = 0;
    offs[0] = o; o = align16(o + 4 * m);            // t_slots  int32 [m]
    offs[1] = o; o = align16(o + 4 * m * k);        // state_ref int32 [m][k]
    offs[2] = o; o = align16(o + 4 * m * k);        // next_ref int32 [m][k]
    offs[3] = o; o = align16(o + 8 * m);            // reward f64 [m]
    offs[4] = o; o = align16(o + m);                // terminal u8 [m]
    offs[5] = o; o = align16(o + 4 * m);            // e_slots int32 [m]
    offs[6] = o; o = align16(o + 4 * m);            // e_tids int32 [m][1]
    offs[7] = o; o = align16(o + 4 * m);            // e_len int32 [m]
    offs[8] = o; o = align16(o + 4 * U * B);        // sampled entry slots int32 [U][B]
    offs[9] = o;                                    // bytes used
    if (o > block_bytes) {
        pfrl_set_error("pfrl_plan_dqn_range: staging block too small");
        return PFRL_ERR_ARG;
    }
    int32_t *b_tslot = (int32_t *)(host_block + offs[0]);
    int32_t *b_sref = (int32_t *)(host_block + offs[1]);
    int32_t *b_nref = (int32_t *)(host_block + offs[2]);
    double *b_rew = (double *)(host_block + offs[3]);
    uint8_t *b_term = host_block + offs[4];
    int32_t *b_eslot = (int32_t *)(host_block + offs[5]);
    int32_t *b_etid = (int32_t *)(host_block + offs[6]);
    int32_t *b_elen = (int32_t *)(host_block + offs[7]);
    int32_t *b_samp = (int32_t *)(host_block + offs[8]);
    int64_t scratch[2 * 4096];
    int64_t u = 0, head = head0;
    for (int64_t j = 0; j < m; ++j) {
        // ReplayBuffer.append with num_steps == 1 (pfrl/replay_buffers/replay_buffer.py:33-62):
        // the transition and its one-transition entry
        const int64_t tid = n_trans0 + j, tslot = tid % st->R;
        const int64_t seq = n_entries0 + j, eslot = seq % st->E;
        const int64_t mf = s_min_seq[j] < n_min_seq[j] ? s_min_seq[j] : n_min_seq[j];
        b_tslot[j] = (int32_t)tslot;
        for (int c = 0; c < k; ++c) {
            b_sref[j * k + c] = st->h_state_ref[tslot * k + c] = s_refs[j * k + c];
            b_nref[j * k + c] = st->h_next_ref[tslot * k + c] = n_refs[j * k + c];
        }
        b_rew[j] = st->h_reward[tslot] = reward[j];
        b_term[j] = st->h_terminal[tslot] = done[j] ? 1 : 0;
        st->h_min_fseq[tslot] = mf;
        b_eslot[j] = (int32_t)eslot;
        b_etid[j] = (int32_t)tslot;
        b_elen[j] = 1;
        st->h_e_tids[eslot] = tid;
        st->h_e_len[eslot] = 1;
        st->h_e_min_fseq[eslot] = mf;
        // RandomAccessQueue with maxlen: the oldest entry leaves when the queue is full
        const int64_t total = seq + 1;
        if (maxlen >= 0 && total - maxlen > head) head = total - maxlen;
        const int64_t len = total - head;
        if (len >= replay_start && (t0 + j + 1) % update_interval == 0) {
            for (int r = 0; r < n_times_update; ++r, ++u) {
                sample_n_k_sparse(bg, len, B, scratch);
                for (int i = 0; i < B; ++i) {
                    const int64_t es = (head + scratch[i]) % st->E;
                    if (st->h_e_min_fseq[es] < oldest_live_fseq) {
                        pfrl_set_error("frame ring too small: a sampled transition references a "
                                       "frame the ring has already wrapped past");
                        // the stream has moved: the caller must raise, not fall back
                        counters[0] = n_trans0 + j + 1;
                        counters[1] = n_entries0 + j + 1;
                        counters[2] = head;
                        return PFRL_PLAN_FRAME_RING;
                    }
                    b_samp[u * B + i] = (int32_t)es;
                }
            }
        }
    }
    counters[0] = n_trans0 + m;
    counters[1] = n_entries0 + m;
    counters[2] = head;
    return U;
}

// Synthetic Atari-shaped env: rewards in {-1, 0, +1} w.p. (0.05, 0.90, 0.05) and
// done w.p. p_done as pure functions of (seed, env, t) -- the same splitmix-style hash as
// pfrl_amd/envs/synthetic.py (`reward_done_stream`), which stays the definition and is what
// tests compare this with.
static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01_of(uint64_t seed, uint64_t env, uint64_t t, uint64_t stream) {
    const uint64_t key = mix64(seed ^ mix64(env * 0x9E3779B97F4A7C15ull + t));
    const uint64_t r = mix64(key + stream * 0xD1342543DE82EF95ull);
    return (double)(r >> 11) * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------------------------------
// PPO's minibatch order: random.sample(range(n), k=n) on Python's `random` module
// (pfrl/agents/ppo.py:247-257 -> _yield_minibatches).  CPython (Lib/random.py, 3.9 - 3.12):
//   pool = list(range(n)); for i in range(n): j = _randbelow(n - i); result[i] = pool[j];
//   pool[j] = pool[n - i - 1]                      (k == n always takes the pool branch)
//   _randbelow(m): k = m.bit_length(); r = getrandbits(k); while r >= m: r = getrandbits(k)
//   getrandbits(k <= 32) = genrand_uint32() >> (32 - k)           (Modules/_randommodule.c)
// `random` publishes no function table, so the generator itself is restated (MT19937, the
// reference implementation CPython embeds) and runs on the module's state: 624 words + index as
// random.getstate() hands them out; the caller writes the advanced state back with setstate().
// At n = 65 536 the interpreter spends 10 - 20 ms per permutation, four per PPO update.
// ---------------------------------------------------------------------------------------
static inline uint32_t mt_next(uint32_t *mt, uint32_t &idx) {
    constexpr int N = 624, M = 397;
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    if (idx >= N) {
        int kk;
        uint32_t y;
        for (kk = 0; kk < N - M; ++kk) {
            y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        for (; kk < N - 1; ++kk) {
            y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
        mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

extern "C" int pfrl_pyrandom_permutation(uint32_t *state625, int64_t n, int64_t *host_out) {
    PFRL_CHECK_ARG(state625 && host_out && n >= 1 && n <= 0x7fffffffll,
                   "pfrl_pyrandom_permutation: 1 <= n < 2^31");
    PFRL_CHECK_ARG(state625[624] <= 624, "pfrl_pyrandom_permutation: bad generator index");
    uint32_t idx = state625[624];
    std::vector<int64_t> pool((size_t)n);
    for (int64_t i = 0; i < n; ++i) pool[(size_t)i] = i;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t m = (uint32_t)(n - i);
        const int bits = 32 - __builtin_clz(m);          // m.bit_length(), m >= 1
        uint32_t r;
        do r = mt_next(state625, idx) >> (32 - bits); while (r >= m);
        host_out[i] = pool[r];
        pool[r] = pool[(size_t)(n - i - 1)];
    }
    state625[624] = idx;
    return 0;
}

extern "C" int pfrl_synth_reward_done(uint64_t seed, int64_t env_id0, int64_t n, int64_t t,
                                      double p_done, double *host_reward, uint8_t *host_done) {
    PFRL_CHECK_ARG(host_reward && host_done && n >= 0, "pfrl_synth_reward_done: null argument");
    for (int64_t i = 0; i < n; ++i) {
        const double u = u01_of(seed, (uint64_t)(env_id0 + i), (uint64_t)t, 1);
        host_reward[i] = u < 0.05 ? -1.0 : (u < 0.10 ? 1.0 : 0.0);
        host_done[i] = u01_of(seed, (uint64_t)(env_id0 + i), (uint64_t)t, 2) < p_done ? 1 : 0;
    }
    return 0;
}
