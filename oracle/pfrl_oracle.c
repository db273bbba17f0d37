/*
 * pfrl_oracle.c -- CPU restatement of the reference (pfnet/pfrl v0.4.0) replay
 * data path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; nothing under pfrl_amd/ does.  It restates, in plain C,
 * the algorithms of
 *
 *   pfrl/collections/prioritized.py          (TreeQueue / SumTreeQueue /
 *                                             MinTreeQueue / PrioritizedBuffer)
 *   pfrl/collections/random_access_queue.py  (FIFO with maxlen)
 *   pfrl/replay_buffers/replay_buffer.py     (n-step windows per env_id)
 *   pfrl/replay_buffers/prioritized.py       (PriorityWeightError)
 *   pfrl/replay_buffer.py:157-212            (batch_experiences)
 *   pfrl/utils/batch_states.py:18-36 + the examples' phi (u8 -> f32 / 255)
 *   pfrl/agents/ppo.py:36-47                 (GAE reverse scan)
 *   pfrl/agents/ppo.py:476-478,494-495       (advantage standardisation)
 *   pfrl/agents/a2c.py:150-167               (A2C return scan)
 *   pfrl/agents/categorical_dqn.py:7-104,150-204 + categorical_double_dqn.py:25-47
 *                                            (C51 projection and loss)
 *
 * Parity pinning: every function here is checked against golden vectors that
 * tests/golden/make_golden.py recorded by running the reference itself
 * (tests/test_oracle_golden.py).
 *
 * Scalars carry a NumPy-2 (NEP 50) type tag because the reference, as executed
 * with NumPy 2.2, mixes Python floats (weak f64), np.float32 and np.float64
 * inside the priority trees (hard part 1):
 *     tag 0 = absent (an empty list node), 1 = Python float, 2 = np.float32,
 *     tag 3 = np.float64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define T_NONE 0
#define T_PY 1
#define T_F32 2
#define T_F64 3

typedef struct {
    double v;
    int t;
} tv;

static tv mk(double v, int t) {
    tv x;
    x.v = v;
    x.t = t;
    return x;
}

/* NEP 50 result type of a binary op between two present scalars. */
static int res_type(int a, int b) {
    if (a == T_F64 || b == T_F64) return T_F64;
    if (a == T_F32 || b == T_F32) return T_F32;
    return T_PY;
}

static tv tv_add(tv a, tv b) {
    int t = res_type(a.t, b.t);
    if (t == T_F32) {
        volatile float x = (float)a.v, y = (float)b.v;
        volatile float r = x + y;
        return mk((double)r, t);
    }
    return mk(a.v + b.v, t);
}

static tv tv_sub(tv a, tv b) {
    int t = res_type(a.t, b.t);
    if (t == T_F32) {
        volatile float x = (float)a.v, y = (float)b.v;
        volatile float r = x - y;
        return mk((double)r, t);
    }
    return mk(a.v - b.v, t);
}

static tv tv_div(tv a, tv b) {
    int t = res_type(a.t, b.t);
    if (t == T_F32) {
        volatile float x = (float)a.v, y = (float)b.v;
        volatile float r = x / y;
        return mk((double)r, t);
    }
    return mk(a.v / b.v, t);
}

/* a < b with NEP 50 operand conversion. */
static int tv_lt(tv a, tv b) {
    if (res_type(a.t, b.t) == T_F32) return (float)a.v < (float)b.v;
    return a.v < b.v;
}

/* ------------------------------------------------------------------------ *
 * TreeQueue: pfrl/collections/prioritized.py:135-242.  A node is a Python
 * list that is either empty ([]), or [left, right, value].  Here a node is a
 * pool slot; present==0 plays the role of the empty list.
 * ------------------------------------------------------------------------ */
typedef struct {
    int l, r; /* pool ids of the child list objects, 0 = none allocated */
    tv val;
    int present;
} node_t;

typedef struct {
    node_t *pool;
    int pool_len, pool_cap;
    int *free_ids;
    int n_free, free_cap;
    int root; /* 0 = no root */
    long ixl, ixr;
    long length;
    int op; /* 0 = sum, 1 = min */
} tree_t;

static int node_alloc(tree_t *t) {
    int id;
    if (t->n_free > 0) {
        id = t->free_ids[--t->n_free];
    } else {
        if (t->pool_len == t->pool_cap) {
            t->pool_cap = t->pool_cap ? t->pool_cap * 2 : 1024;
            t->pool = (node_t *)realloc(t->pool, sizeof(node_t) * t->pool_cap);
        }
        id = t->pool_len++;
    }
    t->pool[id].l = t->pool[id].r = 0;
    t->pool[id].present = 0;
    t->pool[id].val = mk(0.0, T_NONE);
    return id;
}

static void node_free(tree_t *t, int id) {
    if (id == 0) return;
    if (t->n_free == t->free_cap) {
        t->free_cap = t->free_cap ? t->free_cap * 2 : 1024;
        t->free_ids = (int *)realloc(t->free_ids, sizeof(int) * t->free_cap);
    }
    t->free_ids[t->n_free++] = id;
}

static void node_free_rec(tree_t *t, int id) {
    if (id == 0) return;
    node_free_rec(t, t->pool[id].l);
    node_free_rec(t, t->pool[id].r);
    node_free(t, id);
}

static void tree_init(tree_t *t, int op) {
    memset(t, 0, sizeof(*t));
    t->op = op;
    node_alloc(t); /* slot 0 is the "no node" sentinel */
}

static void tree_destroy(tree_t *t) {
    free(t->pool);
    free(t->free_ids);
}

/* op([children present]) -- prioritized.py:140-151 (_reduce).
 * sum([l, r]) is ((0 + l) + r); min([l, r]) returns l unless r < l. */
static void node_reduce(tree_t *t, int id) {
    node_t *n = &t->pool[id];
    int lp = n->l && t->pool[n->l].present;
    int rp = n->r && t->pool[n->r].present;
    if (!lp && !rp) {
        /* del node[:] -- both child lists are empty, drop them */
        node_free(t, n->l);
        node_free(t, n->r);
        n->l = n->r = 0;
        n->present = 0;
        n->val = mk(0.0, T_NONE);
        return;
    }
    if (t->op == 0) {
        if (lp && rp)
            n->val = tv_add(t->pool[n->l].val, t->pool[n->r].val);
        else
            n->val = lp ? t->pool[n->l].val : t->pool[n->r].val;
    } else {
        if (lp && rp)
            n->val = tv_lt(t->pool[n->r].val, t->pool[n->l].val) ? t->pool[n->r].val
                                                                   : t->pool[n->l].val;
        else
            n->val = lp ? t->pool[n->l].val : t->pool[n->r].val;
    }
}

/* prioritized.py:154-180 (_write).  value.t == T_NONE means value=None.
 * Returns the previous leaf value (T_NONE if the leaf was absent). */
static tv node_write(tree_t *t, long ixl, long ixr, int id, long key, tv value) {
    tv ret;
    if (ixr - ixl == 1) {
        node_t *n = &t->pool[id];
        ret = n->present ? n->val : mk(0.0, T_NONE);
        if (value.t == T_NONE) {
            n->present = 0;
            n->val = mk(0.0, T_NONE);
        } else {
            n->present = 1;
            n->val = value;
        }
        return ret;
    }
    if (!t->pool[id].present) { /* _expand */
        int a = node_alloc(t);
        int b = node_alloc(t);
        t->pool[id].l = a;
        t->pool[id].r = b;
        t->pool[id].present = 1;
        t->pool[id].val = mk(0.0, T_NONE);
    }
    {
        /* floor division, bounds may be negative: prioritized.py:172 */
        long s = ixl + ixr;
        long ixc = (s >= 0) ? s / 2 : -((-s + 1) / 2);
        if (key < ixc)
            ret = node_write(t, ixl, ixc, t->pool[id].l, key, value);
        else
            ret = node_write(t, ixc, ixr, t->pool[id].r, key, value);
    }
    node_reduce(t, id);
    return ret;
}

static tv tree_write(tree_t *t, long ix, tv value) {
    return node_write(t, t->ixl, t->ixr, t->root, ix, value);
}

/* prioritized.py:207-223 (TreeQueue.append) */
static void tree_append(tree_t *t, tv value) {
    if (t->length == 0) {
        t->root = node_alloc(t);
        t->pool[t->root].present = 1;
        t->pool[t->root].val = value;
        t->ixl = 0;
        t->ixr = 1;
        t->length = 1;
        return;
    }
    if (t->ixr == t->length) {
        int nr = node_alloc(t);
        int empty = node_alloc(t);
        t->pool[nr].l = t->root;
        t->pool[nr].r = empty;
        t->pool[nr].present = 1;
        t->pool[nr].val = t->pool[t->root].val;
        t->root = nr;
        t->ixr += t->ixr - t->ixl;
    }
    tree_write(t, t->length, value);
    t->length += 1;
}

/* prioritized.py:225-242 (TreeQueue.popleft) */
static tv tree_popleft(tree_t *t) {
    tv ret = tree_write(t, 0, mk(0.0, T_NONE));
    t->ixl -= 1;
    t->ixr -= 1;
    t->length -= 1;
    if (t->length == 0) {
        node_free_rec(t, t->root);
        t->root = 0;
        return ret;
    }
    {
        long s = t->ixl + t->ixr;
        long ixc = (s >= 0) ? s / 2 : -((-s + 1) / 2);
        if (ixc == 0) {
            int old = t->root;
            int right = t->pool[old].r;
            node_free_rec(t, t->pool[old].l);
            node_free(t, old);
            t->root = right;
            t->ixl = ixc;
        }
    }
    return ret;
}

/* prioritized.py:245-258 (_find) */
static long node_find(tree_t *t, long ixl, long ixr, int id, tv pos) {
    while (ixr - ixl != 1) {
        node_t *n = &t->pool[id];
        long s = ixl + ixr;
        long ixc = (s >= 0) ? s / 2 : -((-s + 1) / 2);
        tv left = (n->l && t->pool[n->l].present) ? t->pool[n->l].val : mk(0.0, T_PY);
        if (tv_lt(pos, left)) {
            ixr = ixc;
            id = n->l;
        } else {
            pos = tv_sub(pos, left);
            ixl = ixc;
            id = n->r;
        }
    }
    return ixl;
}

/* ------------------------------------------------------------------------ *
 * PrioritizedBuffer: pfrl/collections/prioritized.py:21-123
 * ------------------------------------------------------------------------ */
typedef struct {
    long capacity; /* <0 = None */
    tree_t sums, mins;
    tv max_priority;
    int flag_wait;
    long *sampled;
    long n_sampled, sampled_cap;
    /* data deque: payload ids */
    int64_t *data;
    long data_head, data_len, data_cap;
} pbuf_t;

void *orc_pbuf_create(long capacity) {
    pbuf_t *b = (pbuf_t *)calloc(1, sizeof(pbuf_t));
    b->capacity = capacity;
    tree_init(&b->sums, 0);
    tree_init(&b->mins, 1);
    b->max_priority = mk(1.0, T_PY); /* initial_max_priority=1.0, :26 */
    b->data_cap = 1024;
    b->data = (int64_t *)malloc(sizeof(int64_t) * b->data_cap);
    return b;
}

void orc_pbuf_destroy(void *h) {
    pbuf_t *b = (pbuf_t *)h;
    tree_destroy(&b->sums);
    tree_destroy(&b->mins);
    free(b->sampled);
    free(b->data);
    free(b);
}

long orc_pbuf_len(void *h) { return ((pbuf_t *)h)->data_len; }

static void data_push(pbuf_t *b, int64_t x) {
    if (b->data_len == b->data_cap) {
        long nc = b->data_cap * 2, i;
        int64_t *nd = (int64_t *)malloc(sizeof(int64_t) * nc);
        for (i = 0; i < b->data_len; i++) nd[i] = b->data[(b->data_head + i) % b->data_cap];
        free(b->data);
        b->data = nd;
        b->data_cap = nc;
        b->data_head = 0;
    }
    b->data[(b->data_head + b->data_len) % b->data_cap] = x;
    b->data_len++;
}

static int64_t data_get(pbuf_t *b, long i) { return b->data[(b->data_head + i) % b->data_cap]; }

/* :50-54 popleft */
int64_t orc_pbuf_popleft(void *h) {
    pbuf_t *b = (pbuf_t *)h;
    int64_t x;
    tree_popleft(&b->sums);
    tree_popleft(&b->mins);
    x = b->data[b->data_head];
    b->data_head = (b->data_head + 1) % b->data_cap;
    b->data_len--;
    return x;
}

/* :39-48 append; ptag==T_NONE -> priority=None -> max_priority */
void orc_pbuf_append(void *h, int64_t payload, double pval, int ptag) {
    pbuf_t *b = (pbuf_t *)h;
    tv p;
    if (b->capacity >= 0 && b->data_len == b->capacity) orc_pbuf_popleft(h);
    p = (ptag == T_NONE) ? b->max_priority : mk(pval, ptag);
    data_push(b, payload);
    tree_append(&b->sums, p);
    tree_append(&b->mins, p);
}

/* :56-105 sample with uniform_ratio == 0 (the only value the replay buffers
 * use, replay_buffers/prioritized.py:119).  u01[i] are the doubles that
 * np.random.uniform(0.0, root[2]) would consume: pos = 0.0 + root * u
 * (numpy legacy uniform = low + (high-low)*next_double).
 * Outputs: logical indices, payload ids, removed priorities (value+tag),
 * probabilities, total (value+tag) and min_prob (as double). */
void orc_pbuf_sample(void *h, long n, const double *u01, long *out_idx, int64_t *out_payload,
                     double *out_pri, int *out_pri_tag, double *out_prob, double *out_total,
                     int *out_total_tag, double *out_min_prob) {
    pbuf_t *b = (pbuf_t *)h;
    tv total = b->sums.length ? b->sums.pool[b->sums.root].val : mk(0.0, T_PY);
    tv minv = b->mins.length ? b->mins.pool[b->mins.root].val : mk(INFINITY, T_F64);
    tv min_prob = tv_div(minv, total);
    long i;
    if (n > b->sampled_cap) {
        b->sampled_cap = n;
        b->sampled = (long *)realloc(b->sampled, sizeof(long) * n);
    }
    for (i = 0; i < n; i++) {
        /* prioritized.py:301-306 */
        tv root = b->sums.pool[b->sums.root].val;
        double pos = 0.0 + root.v * u01[i];
        long ix = node_find(&b->sums, b->sums.ixl, b->sums.ixr, b->sums.root, mk(pos, T_PY));
        tv val = tree_write(&b->sums, ix, mk(0.0, T_PY));
        out_idx[i] = ix;
        out_pri[i] = val.v;
        out_pri_tag[i] = val.t;
        b->sampled[i] = ix;
    }
    b->n_sampled = n;
    for (i = 0; i < n; i++) {
        /* :80-83 with uniform_ratio = 0 (int): 0/len + (1-0)*pri/total */
        tv pri = mk(out_pri[i], out_pri_tag[i]);
        tv pr = tv_add(mk(0.0, T_PY), tv_div(pri, total));
        out_prob[i] = pr.v;
        out_payload[i] = data_get(b, out_idx[i]);
    }
    *out_total = total.v;
    *out_total_tag = total.t;
    *out_min_prob = min_prob.v;
    b->flag_wait = 1;
}

/* :107-116 set_last_priority */
void orc_pbuf_set_last_priority(void *h, long n, const double *pval, const int *ptag) {
    pbuf_t *b = (pbuf_t *)h;
    long i;
    for (i = 0; i < n; i++) {
        tv p = mk(pval[i], ptag[i]);
        tree_write(&b->sums, b->sampled[i], p);
        tree_write(&b->mins, b->sampled[i], p);
        /* max(self.max_priority, p): returns p only if p > max_priority */
        if (tv_lt(b->max_priority, p)) b->max_priority = p;
    }
    b->flag_wait = 0;
    b->n_sampled = 0;
}

void orc_pbuf_stats(void *h, double *vals, int *tags, long *ints) {
    pbuf_t *b = (pbuf_t *)h;
    tv s = b->sums.length ? b->sums.pool[b->sums.root].val : mk(0.0, T_PY);
    tv m = b->mins.length ? b->mins.pool[b->mins.root].val : mk(INFINITY, T_F64);
    vals[0] = s.v;
    tags[0] = s.t;
    vals[1] = m.v;
    tags[1] = m.t;
    vals[2] = b->max_priority.v;
    tags[2] = b->max_priority.t;
    ints[0] = b->data_len;
    ints[1] = b->sums.ixl;
    ints[2] = b->sums.ixr;
}

/* Dump one level of a tree: depth d below the root (d=0 root).  Writes
 * (ixr-ixl)>>(log2size-d) ... in frame order; absent nodes get tag 0.
 * which = 0 sums, 1 mins.  Returns the number of nodes written. */
static void dump_rec(tree_t *t, int id, long lo, long hi, long width, long base, double *v,
                     int *tg) {
    if (hi - lo == width) {
        long j = (lo - base) / width;
        if (id && t->pool[id].present) {
            v[j] = t->pool[id].val.v;
            tg[j] = t->pool[id].val.t;
        } else {
            v[j] = 0.0;
            tg[j] = T_NONE;
        }
        return;
    }
    {
        long s = lo + hi;
        long c = (s >= 0) ? s / 2 : -((-s + 1) / 2);
        int l = (id && t->pool[id].present) ? t->pool[id].l : 0;
        int r = (id && t->pool[id].present) ? t->pool[id].r : 0;
        dump_rec(t, l, lo, c, width, base, v, tg);
        dump_rec(t, r, c, hi, width, base, v, tg);
    }
}

long orc_pbuf_dump_level(void *h, int which, long width, double *v, int *tg) {
    pbuf_t *b = (pbuf_t *)h;
    tree_t *t = which ? &b->mins : &b->sums;
    if (t->length == 0) return 0;
    dump_rec(t, t->root, t->ixl, t->ixr, width, t->ixl, v, tg);
    return (t->ixr - t->ixl) / width;
}

/* ------------------------------------------------------------------------ *
 * n-step replay bookkeeping: pfrl/replay_buffers/replay_buffer.py:24-76 and
 * pfrl/collections/random_access_queue.py (FIFO, maxlen -> popleft on
 * overflow, :81-84).  Transitions are integers (ids into the caller's table);
 * an entry is a list of 1..n transition ids.
 * ------------------------------------------------------------------------ */
typedef struct {
    int num_steps;
    long capacity; /* <0 none */
    int n_envs_cap;
    long *win;     /* [env][num_steps] */
    int *win_len;  /* [env] */
    /* emitted entries of the most recent call */
    long *emit;    /* [k][num_steps] */
    int *emit_len; /* [k] */
    long n_emit, emit_cap;
} nstep_t;

void *orc_nstep_create(int num_steps, int max_envs) {
    nstep_t *s = (nstep_t *)calloc(1, sizeof(nstep_t));
    s->num_steps = num_steps;
    s->n_envs_cap = max_envs;
    s->win = (long *)calloc((size_t)max_envs * num_steps, sizeof(long));
    s->win_len = (int *)calloc(max_envs, sizeof(int));
    s->emit_cap = num_steps + 1;
    s->emit = (long *)calloc((size_t)s->emit_cap * num_steps, sizeof(long));
    s->emit_len = (int *)calloc(s->emit_cap, sizeof(int));
    return s;
}

void orc_nstep_destroy(void *h) {
    nstep_t *s = (nstep_t *)h;
    free(s->win);
    free(s->win_len);
    free(s->emit);
    free(s->emit_len);
    free(s);
}

static void emit_window(nstep_t *s, int env) {
    int i, n = s->win_len[env];
    for (i = 0; i < n; i++) s->emit[s->n_emit * s->num_steps + i] = s->win[env * s->num_steps + i];
    s->emit_len[s->n_emit] = n;
    s->n_emit++;
}

static void window_del0(nstep_t *s, int env) {
    int i, n = s->win_len[env];
    for (i = 1; i < n; i++) s->win[env * s->num_steps + i - 1] = s->win[env * s->num_steps + i];
    s->win_len[env] = n - 1;
}

/* replay_buffer.py:33-62 append.  Returns number of emitted entries; their
 * contents are read back with orc_nstep_emitted. */
long orc_nstep_append(void *h, int env, long tid, int terminal) {
    nstep_t *s = (nstep_t *)h;
    s->n_emit = 0;
    /* deque(maxlen=num_steps).append */
    if (s->win_len[env] == s->num_steps) window_del0(s, env);
    s->win[env * s->num_steps + s->win_len[env]] = tid;
    s->win_len[env]++;
    if (terminal) {
        while (s->win_len[env] > 0) {
            emit_window(s, env);
            window_del0(s, env);
        }
    } else if (s->win_len[env] == s->num_steps) {
        emit_window(s, env);
    }
    return s->n_emit;
}

/* replay_buffer.py:64-76 stop_current_episode */
long orc_nstep_stop(void *h, int env) {
    nstep_t *s = (nstep_t *)h;
    s->n_emit = 0;
    if (0 < s->win_len[env] && s->win_len[env] < s->num_steps) emit_window(s, env);
    if (0 < s->win_len[env] && s->win_len[env] <= s->num_steps) window_del0(s, env);
    while (s->win_len[env] > 0) {
        emit_window(s, env);
        window_del0(s, env);
    }
    return s->n_emit;
}

void orc_nstep_emitted(void *h, long *tids, int *lens) {
    nstep_t *s = (nstep_t *)h;
    long k;
    int i;
    for (k = 0; k < s->n_emit; k++) {
        lens[k] = s->emit_len[k];
        for (i = 0; i < s->num_steps; i++)
            tids[k * s->num_steps + i] = (i < s->emit_len[k]) ? s->emit[k * s->num_steps + i] : -1;
    }
}

/* ------------------------------------------------------------------------ *
 * batch_experiences: pfrl/replay_buffer.py:157-212.
 *   reward   = sum((gamma**i) * r_i)   Python float (f64) sum starting from
 *              int 0, then torch.as_tensor(..., dtype=float32)
 *   terminal = any(is_state_terminal)
 *   discount = gamma ** len            f64 -> f32
 * gamma_pow[i] = gamma**i is supplied by the caller (computed by Python's own
 * pow so that libm differences cannot enter).
 * entry_tids: [B][n] (-1 padded), rewards/terminals indexed by transition id.
 * ------------------------------------------------------------------------ */
void orc_batch_experiences_scalars(long B, int n, const long *entry_tids, const int *entry_len,
                                   const double *rewards, const uint8_t *terminals,
                                   const double *gamma_pow, float *out_reward, float *out_terminal,
                                   float *out_discount, long *out_first, long *out_last) {
    long b;
    for (b = 0; b < B; b++) {
        int i, any = 0, len = entry_len[b];
        double acc = 0.0;
        for (i = 0; i < len; i++) {
            long tid = entry_tids[b * n + i];
            volatile double term = gamma_pow[i] * rewards[tid];
            acc = acc + term;
            any |= terminals[tid] != 0;
        }
        out_reward[b] = (float)acc;
        out_terminal[b] = any ? 1.0f : 0.0f;
        out_discount[b] = (float)gamma_pow[len];
        out_first[b] = entry_tids[b * n];
        out_last[b] = entry_tids[b * n + len - 1];
    }
}

/* batch_states with phi(x) = np.asarray(x, dtype=np.float32) / 255
 * (examples/atari/train_dqn_batch_ale.py:229-231; batch_states.py:18-36).
 * frames: [n_frames][frame_elems] u8; refs: [M][k] frame ids;
 * out: [M][k][frame_elems] f32. */
void orc_batch_states_u8(long M, int k, long frame_elems, const uint8_t *frames, const int32_t *refs,
                         float scale_div, float *out) {
    long m, e;
    int j;
    for (m = 0; m < M; m++)
        for (j = 0; j < k; j++) {
            const uint8_t *src = frames + (long)refs[m * k + j] * frame_elems;
            float *dst = out + (m * k + j) * frame_elems;
            if (scale_div == 1.0f)
                for (e = 0; e < frame_elems; e++) dst[e] = (float)src[e];
            else
                for (e = 0; e < frame_elems; e++) dst[e] = (float)src[e] / scale_div;
        }
}

/* identity phi on float32 observations (gym/mujoco examples) */
void orc_batch_states_f32(long M, int k, long frame_elems, const float *frames, const int32_t *refs,
                          float *out) {
    long m;
    int j;
    for (m = 0; m < M; m++)
        for (j = 0; j < k; j++)
            memcpy(out + (m * k + j) * frame_elems, frames + (long)refs[m * k + j] * frame_elems,
                   sizeof(float) * frame_elems);
}

/* ------------------------------------------------------------------------ *
 * GAE: pfrl/agents/ppo.py:36-47, one episode fragment, reverse scan.
 * mode 0: reward is a Python float  -> every op in f32 (NEP 50; v_pred and
 *         next_v_pred are np.float32 from .cpu().numpy(), ppo.py:133-142).
 * mode 1: reward is np.float64      -> td_err/adv in f64, but the product
 *         (gamma*nonterminal) * next_v_pred is still rounded to f32 first.
 * mode 2: v_pred / next_v_pred are Python floats (the recurrent dataset stores
 *         float(v), ppo.py:98-107)  -> every op in f64, the product included.
 * ------------------------------------------------------------------------ */
void orc_gae_fragment(long T, const double *reward, const float *v_pred, const float *next_v_pred,
                      const double *nonterminal, double gamma, double lambd, int mode,
                      double *adv_out, double *vt_out) {
    long i;
    if (mode == 0) {
        /* adv starts as Python float 0.0; gamma*lambd is a Python float */
        double gl = gamma * lambd;
        volatile float adv = 0.0f;
        for (i = T - 1; i >= 0; i--) {
            double gn = gamma * nonterminal[i]; /* Python float * float */
            volatile float prod = (float)gn * next_v_pred[i];
            volatile float s1 = (float)reward[i] + prod;
            volatile float td = s1 - v_pred[i];
            volatile float ga = (float)gl * adv;
            adv = td + ga;
            adv_out[i] = (double)adv;
            {
                volatile float vt = adv + v_pred[i];
                vt_out[i] = (double)vt;
            }
        }
    } else {
        double gl = gamma * lambd;
        volatile double adv = 0.0;
        for (i = T - 1; i >= 0; i--) {
            double gn = gamma * nonterminal[i];
            volatile float prodf = (float)gn * next_v_pred[i];
            volatile double prod = mode == 2 ? gn * (double)next_v_pred[i] : (double)prodf;
            volatile double s1 = reward[i] + prod;
            volatile double td = s1 - (double)v_pred[i];
            volatile double ga = gl * adv;
            adv = td + ga;
            adv_out[i] = adv;
            {
                volatile double vt = adv + (double)v_pred[i];
                vt_out[i] = vt;
            }
        }
    }
}

/* A2C._compute_returns, pfrl/agents/a2c.py:150-167; all tensors f32, shape
 * rewards/masks [T][N], value_preds/returns [T+1][N]; value_preds[T] must
 * already hold next_value (use_gae) / returns[T] must hold next_value. */
void orc_a2c_returns(long T, long N, const float *rewards, const float *masks, float *value_preds,
                     float *returns, double gamma_d, double tau_d, int use_gae) {
    long i, e;
    const float gamma = (float)gamma_d;
    const float gt = (float)(gamma_d * tau_d);
    if (use_gae) {
        for (e = 0; e < N; e++) {
            volatile float gae = 0.0f;
            for (i = T - 1; i >= 0; i--) {
                volatile float a = gamma * value_preds[(i + 1) * N + e];
                volatile float b = a * masks[i * N + e];
                volatile float c = rewards[i * N + e] + b;
                volatile float delta = c - value_preds[i * N + e];
                volatile float g2 = gt * masks[i * N + e];
                volatile float g3 = g2 * gae;
                gae = delta + g3;
                returns[i * N + e] = gae + value_preds[i * N + e];
            }
        }
    } else {
        for (e = 0; e < N; e++)
            for (i = T - 1; i >= 0; i--) {
                volatile float a = gamma * returns[(i + 1) * N + e];
                volatile float b = a * masks[i * N + e];
                returns[i * N + e] = rewards[i * N + e] + b;
            }
    }
}

/* ------------------------------------------------------------------------ *
 * PriorityWeightError: pfrl/replay_buffers/prioritized.py:47-66.
 * priority_from_errors for np.float32 errors (the DQN path, dqn.py:449-454),
 * error_min / error_max Python ints, eps and alpha Python floats.
 * Uses this host's libm powf/pow exactly as NumPy's scalar math does.
 * ------------------------------------------------------------------------ */
void orc_priority_from_errors_f32(long n, const float *err, int has_min, double emin, int has_max,
                                  double emax, double eps, double alpha, double *out_v,
                                  int *out_t) {
    long i;
    for (i = 0; i < n; i++) {
        float e = err[i];
        int is_np = 1; /* still np.float32? */
        double cv = 0.0;
        if (has_min && !(e > (float)emin)) { /* max(error_min, error) keeps error_min */
            is_np = 0;
            cv = emin;
        }
        if (has_max) {
            if (is_np) {
                if (!(e < (float)emax)) { /* min(error_max, error) keeps error_max */
                    is_np = 0;
                    cv = emax;
                }
            } else if (!(cv < emax)) {
                cv = emax;
            }
        }
        if (is_np) {
            volatile float s = e + (float)eps;
            out_v[i] = (double)powf(s, (float)alpha);
            out_t[i] = T_F32;
        } else {
            out_v[i] = pow(cv + eps, alpha);
            out_t[i] = T_PY;
        }
    }
}

/* Advantage standardisation statistics: torch.std_mean(all_advs,
 * unbiased=False) (ppo.py:476-478).  Reference value in f64 for tolerance
 * comparison. */
void orc_adv_stats(long n, const float *adv, double *mean, double *std) {
    long i;
    double m = 0.0, s = 0.0;
    for (i = 0; i < n; i++) m += adv[i];
    m /= (double)n;
    for (i = 0; i < n; i++) s += (adv[i] - m) * (adv[i] - m);
    *mean = m;
    *std = sqrt(s / (double)n);
}

/* ------------------------------------------------------------------------
 * Categorical (C51) DQN loss:
 *   pfrl/action_value.py:97-180          q_values = q_dist @ z, greedy = argmax
 *   pfrl/agents/categorical_dqn.py:150-171 (double: categorical_double_dqn.py:25-47)
 *       Tz = r + (1 - terminal) * discount * z
 *   pfrl/agents/categorical_dqn.py:7-57   _apply_categorical_projection: clamp,
 *       bj = (y - v_min) / delta_z clamped to [0, Z-1], l = floor, u = ceil,
 *       z_probs[l] += p * (1 - (bj - l))  for all j, then  z_probs[u] += p * (bj - l)
 *       (the CPU scatter_add_ accumulates in index order, in float32)
 *   pfrl/agents/categorical_dqn.py:373-401 eltwise = -t * log(clamp(y, 1e-10, 1))
 *   pfrl/agents/categorical_dqn.py:60-104  loss accumulation
 * out_grad = d loss / d q_dist (zero except the taken action's row).
 * ------------------------------------------------------------------------ */
void orc_c51_loss(const float *q_dist, const int64_t *action, const float *next_dist,
                  const float *next_select, const float *z, const float *reward,
                  const float *discount, const float *terminal, const float *weights, long B,
                  long A, long Z, int mean, double *out_loss, float *out_grad, float *out_q,
                  float *out_delta, float *out_target) {
    const float v_min = z[0], v_max = z[Z - 1];
    const float delta_z = z[1] - z[0];
    if (next_select == NULL) next_select = next_dist;
    double loss = 0.0;
    memset(out_grad, 0, sizeof(float) * (size_t)(B * A * Z));
    for (long b = 0; b < B; ++b) {
        /* greedy next action: first maximum of the expected values */
        long g = 0;
        double best = 0.0;
        for (long a = 0; a < A; ++a) {
            double qv = 0.0;
            for (long k = 0; k < Z; ++k) qv += (double)next_select[(b * A + a) * Z + k] * z[k];
            if (a == 0 || (float)qv > (float)best) {
                best = qv;
                g = a;
            }
        }
        const float *p = next_dist + (b * A + g) * Z;
        float *t = out_target + b * Z;
        for (long k = 0; k < Z; ++k) t[k] = 0.0f;
        const float scale = (1.0f - terminal[b]) * discount[b];
        /* lower neighbours first, then upper neighbours (two scatter_add_ calls) */
        for (int pass = 0; pass < 2; ++pass) {
            for (long j = 0; j < Z; ++j) {
                float y = reward[b] + scale * z[j];
                y = y < v_min ? v_min : (y > v_max ? v_max : y);
                float bj = (y - v_min) / delta_z;
                bj = bj < 0.0f ? 0.0f : (bj > (float)(Z - 1) ? (float)(Z - 1) : bj);
                const float lo = floorf(bj), up = ceilf(bj);
                const float frac = bj - lo;
                if (pass == 0)
                    t[(long)lo] += p[j] * (1.0f - frac);
                else
                    t[(long)up] += p[j] * frac;
            }
        }
        const long act = action[b];
        const float *y = q_dist + (b * A + act) * Z;
        float coef = weights != NULL ? weights[b] : 1.0f;
        if (mean) coef /= (float)B;
        double d = 0.0, qsa = 0.0;
        for (long k = 0; k < Z; ++k) {
            const float yc = y[k] < 1e-10f ? 1e-10f : (y[k] > 1.0f ? 1.0f : y[k]);
            d += (double)(-t[k] * logf(yc));
            qsa += (double)y[k] * z[k];
            if (y[k] >= 1e-10f && y[k] <= 1.0f)
                out_grad[(b * A + act) * Z + k] = -t[k] / yc * coef;
        }
        out_delta[b] = (float)d;
        out_q[b] = (float)qsa;
        loss += d * (weights != NULL ? (double)weights[b] : 1.0);
    }
    *out_loss = mean ? loss / (double)B : loss;
}

/* ------------------------------------------------------------------------
 * (Double-)DQN TD loss:
 *   pfrl/agents/dqn.py:424-445   y = Q(s)[a]
 *   pfrl/agents/dqn.py:406-422   t = r + discount * (1 - terminal) * max_a Q_target(s')
 *   pfrl/agents/double_dqn.py:15-40  ... Q_target(s')[argmax_a Q_online(s')]
 *   pfrl/agents/dqn.py:44-104    smooth_l1 (delta = 1) or mse / 2; 'sum' | 'mean';
 *                                weighted: sum(loss * w) [/ B]
 * out_grad = d loss / d Q(s)  (zero except the taken action).
 * ------------------------------------------------------------------------ */
void orc_dqn_td_loss(const float *q, const int64_t *action, const float *target_q,
                     const float *next_q_online, const float *reward, const float *discount,
                     const float *terminal, const float *weights, long B, long A, int clip_delta,
                     int mean, double *out_loss, float *out_grad, float *out_y, float *out_t) {
    double loss = 0.0;
    memset(out_grad, 0, sizeof(float) * (size_t)(B * A));
    for (long b = 0; b < B; ++b) {
        const float *sel = next_q_online != NULL ? next_q_online + b * A : target_q + b * A;
        long g = 0;
        for (long a = 1; a < A; ++a)
            if (sel[a] > sel[g]) g = a;          /* first maximum */
        const float nxt = target_q[b * A + g];
        const float t = reward[b] + (discount[b] * (1.0f - terminal[b])) * nxt;
        const float y = q[b * A + action[b]];
        const float d = y - t;
        float l, gl;
        if (clip_delta) {                        /* F.smooth_l1_loss, beta = 1 */
            const float ad = fabsf(d);
            l = ad < 1.0f ? 0.5f * d * d : ad - 0.5f;
            gl = ad < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f);
        } else {                                 /* F.mse_loss / 2 */
            l = 0.5f * d * d;
            gl = d;
        }
        float coef = weights != NULL ? weights[b] : 1.0f;
        if (mean) coef /= (float)B;
        loss += (double)l * (weights != NULL ? (double)weights[b] : 1.0);
        out_grad[b * A + action[b]] = gl * coef;
        out_y[b] = y;
        out_t[b] = t;
    }
    *out_loss = mean ? loss / (double)B : loss;
}
