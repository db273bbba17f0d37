/*
 * pfrl_amd.h -- C ABI of the MI355X (gfx950) replay / rollout data path.
 *
 * This is the drop-in boundary of the hot path.  The
 * reference (pfnet/pfrl) is pure Python and has no FFI of its own; the entry
 * points below are what a ctypes binding for its L2 data path would bind, one
 * per reference function that moved to the device.  Every pointer is a plain
 * device pointer (HBM) unless the name says `host_`; sizes are element counts;
 * `stream` is a hipStream_t passed as void*.  No torch types cross this
 * boundary.  All functions return 0 on success or a hipError_t / negative
 * pfrl error code; pfrl_amd_last_error() gives a message.  Not re-entrant per
 * object: one owner thread per GPU, as the reference's batched path
 * (pfrl/agents/dqn.py:509-549 is single-threaded).
 *
 * Typed scalars.  The reference as executed with NumPy 2 keeps a mixture of
 * Python floats, np.float32 and np.float64 inside its priority trees
 * (SURVEY.md section 7.1).  A tree node here is (double value, uint8 tag):
 */
#ifndef PFRL_AMD_H
#define PFRL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFRL_TAG_ABSENT 0 /* empty list node */
#define PFRL_TAG_PY 1     /* Python float (weak f64) */
#define PFRL_TAG_F32 2    /* np.float32 */
#define PFRL_TAG_F64 3    /* np.float64 */

#define PFRL_MAX_LEVELS 40
#define PFRL_MAX_NSTEP 16
#define PFRL_MAX_STACK 8

#define PFRL_OPT_MAX_TENSORS 24

/* how np.float32 ** alpha of the priority transform is evaluated on the device */
#define PFRL_POW_CORRECTLY_ROUNDED 0 /* (float)pow((double)x, alpha): <= 1 ulp from libm's powf */
#define PFRL_POW_GLIBC 1             /* glibc 2.35 powf restated, separate multiply / add build */
#define PFRL_POW_GLIBC_FMA 2         /* the same, fused multiply-add build (x86-64 -mfma ifunc) */
#define PFRL_ERR_ARG (-2)

int pfrl_amd_version(void);
const char *pfrl_amd_last_error(void);

/* ------------------------------------------------------------------------
 * Observation store: a ring of fixed-size frames in HBM.  An observation is a
 * stack of k frame slots (k = 4 for the Atari frame stack, k = 1 for flat
 * vector observations).  Replaces VectorFrameStack/LazyFrames storage
 * (pfrl/wrappers/vector_frame_stack.py:54-105, atari_wrappers.py:251-272).
 * ------------------------------------------------------------------------ */

/* frames[slots[i]] <- src[i]  (i < n); frame_bytes must be a multiple of 4.
 * New frames of one env step, already on the device: the storage side of
 * VectorFrameStack.step (pfrl/wrappers/vector_frame_stack.py:93-105) and of the
 * `.to(device)` in pfrl/utils/batch_states.py:18-36, done once per frame. */
int pfrl_frames_scatter(void *frames, int64_t frame_bytes, const void *src, const int32_t *slots,
                        int64_t n, void *stream);

/* Synthetic Atari-shaped env (benchmark input only, no
 * counterpart in pfrl/): fills frames[slots[i]]
 * with iid U{0..255} bytes from a counter-based generator keyed by
 * (seed, env_id0 + i, step). */
int pfrl_frames_synth_u8(void *frames, int64_t frame_bytes, const int32_t *slots, int64_t n,
                         uint64_t seed, int64_t env_id0, int64_t step, void *stream);
/* The same for a whole env batch taking consecutive ring positions: frame i goes to slot
 * (seq0 + i) % n_slots, no slot list crosses PCIe. */
int pfrl_frames_synth_u8_ring(void *frames, int64_t frame_bytes, int64_t n_slots, int64_t seq0,
                              int64_t n, uint64_t seed, int64_t env_id0, int64_t step, void *stream);
/* Rewards / dones of that env on the HOST (pure functions of (seed, env, t), same hash as
 * pfrl_amd/envs/synthetic.py reward_done_stream; benchmark input only). */
int pfrl_synth_reward_done(uint64_t seed, int64_t env_id0, int64_t n, int64_t t, double p_done,
                           double *host_reward, uint8_t *host_done);

/* select_action_epsilon_greedily (pfrl/explorers/epsilon_greedy.py:8-12) for a batch, with the
 * greedy actions still on the device (int32 as DiscreteActionValue.greedy_actions yields them,
 * pfrl/action_value.py:59-61, or int64): out[i] = choice[i] >= 0 ? choice[i] : greedy[i], where
 * choice holds the host's draws (pfrl_plan_eps_greedy; -1 = the draw said "greedy").  The
 * actions never have to visit the host (DQN.batch_act, pfrl/agents/dqn.py:490-507). */
int pfrl_select_actions(const void *greedy, int greedy_is_i32, const int32_t *choice, int64_t *out,
                        int64_t n, void *stream);

/* batch_states(states, device, phi) with phi(x) = asarray(x, float32) / divisor
 * (pfrl/utils/batch_states.py:18-36; examples/atari/train_dqn_batch_ale.py:
 * 229-231).  out[m][j][:] = float(frames[refs[m*k+j]][:]) / divisor, IEEE
 * division, bit-exact.  divisor == 1 is the cast-only phi.
 * n_refs = M*k; out is f32 [n_refs][frame_bytes]. */
int pfrl_batch_states_u8(const void *frames, int64_t frame_bytes, const int32_t *refs,
                         int64_t n_refs, float divisor, float *out, void *stream);

/* Same values (pfrl/utils/batch_states.py:18-36 with the Atari phi) for stacks of FOUR
 * frames, emitted channels-last: out is f32
 * [n_obs][frame_bytes][4] = the memory of an NCHW [n_obs][4][H][W] tensor in
 * torch.channels_last format, which a channels_last network reads without the layout
 * conversion PyTorch otherwise performs on every forward and backward pass.
 * refs = int32 [n_obs][4]. */
int pfrl_batch_states_u8_nhwc4(const void *frames, int64_t frame_bytes, const int32_t *refs,
                               int64_t n_obs, float divisor, float *out, void *stream);

/* The same stacks of four u8 frames as u8 NHWC4 pixels, phi NOT applied: out = u8
 * [n_obs][frame_bytes][4] (16-byte aligned), byte c of pixel p = frame refs[obs][c] at p.  For
 * consumers that evaluate phi(x) = float32(x) / divisor in their own operand loader
 * (pfrl_conv2d_u8nhwc4_fwd / _bwd_weight): the minibatch of pfrl/agents/ppo.py:480-487 costs
 * 2 bytes per frame byte here instead of 5. */
int pfrl_batch_states_u8_raw_nhwc4(const void *frames, int64_t frame_bytes, const int32_t *refs,
                                   int64_t n_obs, void *out, void *stream);

/* batch_states with the identity phi on float32 observations
 * (examples/gym/train_dqn_gym.py, mujoco examples): plain gather. */
int pfrl_batch_states_f32(const void *frames, int64_t frame_bytes, const int32_t *refs,
                          int64_t n_refs, float *out, void *stream);

/* ------------------------------------------------------------------------
 * Transition table + n-step entries (pfrl/replay_buffers/replay_buffer.py:
 * 24-76; pfrl/collections/random_access_queue.py).  Structure-of-arrays in
 * HBM, all rings indexed by slot:
 *   t_state_ref, t_next_ref : int32 [R][k]   frame slots of s_t and s_{t+1}
 *   t_action                : int64 [R]  (discrete)  or  float [R][act_dim]
 *   t_reward                : double[R]      (Python float rewards)
 *   t_terminal              : uint8 [R]
 *   e_tids                  : int32 [E][n]   transition slots of an n-step
 *                                            entry, -1 padded
 *   e_len                   : int32 [E]
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t *t_state_ref;
    int32_t *t_next_ref;
    void *t_action;
    double *t_reward;
    uint8_t *t_terminal;
    int32_t *e_tids;
    int32_t *e_len;
    int32_t k;        /* frames per observation */
    int32_t n;        /* num_steps */
    int32_t act_dim;  /* 0: int64 discrete actions, >0: float[act_dim] */
    int32_t reserved;
} pfrl_table_t;

/* ReplayBuffer.append for a batch of transitions already on the device
 * (pfrl/agents/dqn.py:527-544): row i of the staging arrays goes to
 * transition slot t_slots[i]. */
int pfrl_table_append(const pfrl_table_t *tab, int64_t n_rows, const int32_t *t_slots,
                      const int32_t *state_ref, const int32_t *next_ref, const void *action,
                      const double *reward, const uint8_t *terminal, void *stream);

/* Emitted n-step windows (pfrl/replay_buffers/replay_buffer.py:53-62,64-76): entry slot
 * e_slots[i] <- (tids[i][0..n), len[i]). */
int pfrl_entries_append(const pfrl_table_t *tab, int64_t n_rows, const int32_t *e_slots,
                        const int32_t *tids, const int32_t *lens, void *stream);

/* batch_experiences (pfrl/replay_buffer.py:157-212) for B sampled entry slots:
 *   state      f32 [B][k][frame_bytes] = phi(e[0].state)
 *   next_state f32 [B][k][frame_bytes] = phi(e[-1].next_state)
 *   action     = e[0].action
 *   reward     = float32(sum_i gamma_pow[i] * r_i)   (f64 accumulate, :183-190)
 *   terminal   = any(is_state_terminal)              (:194-201)
 *   discount   = float32(gamma_pow[len])             (:202-206)
 * gamma_pow: host array of n+1 doubles (gamma**i computed by the caller's own
 * pow, passed by value).  frame_is_f32 != 0 selects the identity phi on f32
 * frames.  One launch does the scalar collapse and both gathers. */
int pfrl_batch_experiences(const pfrl_table_t *tab, const void *frames, int64_t frame_bytes,
                           int frame_is_f32, float divisor, const int32_t *entry_slots, int64_t B,
                           const double *host_gamma_pow, float *out_state, float *out_next_state,
                           void *out_action, float *out_reward, float *out_terminal,
                           float *out_discount, void *stream);

/* The same launch (pfrl/replay_buffer.py:157-212) with out_state / out_next_state emitted
 * channels-last
 * ([B][frame_bytes][4] f32; see pfrl_batch_states_u8_nhwc4): u8 frames, tab->k == 4. */
int pfrl_batch_experiences_nhwc4(const pfrl_table_t *tab, const void *frames, int64_t frame_bytes,
                                 float divisor, const int32_t *entry_slots, int64_t B,
                                 const double *host_gamma_pow, float *out_state,
                                 float *out_next_state, void *out_action, float *out_reward,
                                 float *out_terminal, float *out_discount, void *stream);

/* ------------------------------------------------------------------------
 * Prioritized buffer: the sliding sum/min trees of
 * pfrl/collections/prioritized.py:135-323 as per-level rings of tagged nodes.
 *   level l (0 = leaves) has M_l = max(smax >> l, 1) slots starting at
 *   level_off[l]; node of absolute leaf coordinate x at level l is
 *   ((x - origin[l]) >> l) & (M_l - 1).
 * The host mirrors the integer bookkeeping of TreeQueue (length, bounds,
 * re-rooting, :207-242) and passes it in this descriptor by value.
 * ------------------------------------------------------------------------ */
typedef struct {
    double *sum_val;
    uint8_t *sum_tag;
    double *min_val;
    uint8_t *min_tag;
    double *maxp_val;   /* device scalar: PrioritizedBuffer.max_priority */
    uint8_t *maxp_tag;
    int64_t level_off[PFRL_MAX_LEVELS];
    int64_t origin[PFRL_MAX_LEVELS];
    int64_t base;       /* absolute coordinate of the frame start (bounds[0]) */
    int64_t head;       /* absolute coordinate of logical index 0 */
    int64_t length;
    int32_t log2_size;  /* frame size = 1 << log2_size */
    int32_t log2_smax;
} pfrl_tree_t;

/* PrioritizedBuffer.append / popleft for a batch (pfrl/collections/prioritized.py:39-54;
 * TreeQueue._write :154-180):
 * leaves x[i] are written in both trees and every ancestor is re-reduced.
 * tag[i] == PFRL_TAG_ABSENT deletes the leaf (popleft);
 * use_maxp[i] != 0 writes the current max_priority (append with
 * priority=None).  x must be unique within one call. n <= 1024. */
int pfrl_tree_write(const pfrl_tree_t *tree, int64_t n, const int64_t *x, const double *val,
                    const uint8_t *tag, const uint8_t *use_maxp, void *stream);

/* pfrl_tree_update_errors_write_f32 and pfrl_tree_sample as ONE launch: the priorities of minibatch k
 * (pfrl/replay_buffers/prioritized.py:117-126 -> collections/prioritized.py:107-116), the appends /
 * pops recorded since (:39-54) and the B dependent draws of minibatch k + 1 (:56-84, 294-312) -- the
 * chain the next forward pass waits for, without the launch boundary and the second dispatch in
 * the middle.  Be + n <= 64; same outputs, bit for bit, as the two calls. */
int pfrl_tree_update_errors_write_sample(
    const pfrl_tree_t *tree, int64_t Be, const int64_t *x, const float *err, int has_min,
    float error_min, double pri_at_min, int has_max, float error_max, double pri_at_max, double eps,
    double alpha, int dedupe, int pow_mode, int64_t n, const int64_t *wx, const double *wval,
    const uint8_t *wtag, const uint8_t *wuse_maxp, int64_t B, const double *u01, int64_t *out_x,
    double *out_pri, uint8_t *out_pri_tag, double *out_prob, float *out_weight, double *out_total,
    uint8_t *out_total_tag, double *out_min_prob, int normalize, double beta, int64_t slot_mod,
    int32_t *out_slot, void *stream);

/* TreeQueue._write on the SUM tree only, for n <= 1024 distinct leaves
 * (pfrl/collections/prioritized.py:278-292 SumTreeQueue.uniform_sample: the leaves sample_n_k
 * picked are zeroed and their previous priorities returned; :289-291 / :308-310: with
 * remove=False the samplers write the priorities back).  val / tag NULL = write Python-float 0.0;
 * old_val / old_tag (NULL or both) receive the leaves' previous contents.  The min tree is
 * untouched, as in the reference. */
int pfrl_tree_write_sum(const pfrl_tree_t *tree, int64_t n, const int64_t *x, const double *val,
                        const uint8_t *tag, double *old_val, uint8_t *old_tag, void *stream);

/* SumTreeQueue.prioritized_sample(n, remove=True) + the probability / weight
 * math of PrioritizedBuffer._sample_indices_and_probabilities (:56-84) and
 * PriorityWeightError.weights_from_probabilities
 * (pfrl/replay_buffers/prioritized.py:57-66).
 *   u01[i]          : the doubles np.random.uniform would consume
 *   out_x[i]        : absolute leaf coordinate (logical index = x - head)
 *   out_pri / _tag  : removed priorities (bit-exact)
 *   out_prob        : f64 probabilities
 *   out_weight      : f32 importance weights
 *   normalize: 0 False, 1 "batch", 2 "memory";  beta: current beta
 *   out_total/_tag, out_min_prob: root sum before sampling, min/total
 *   out_slot (optional)         : int32 x % slot_mod -- the replay buffer's
 *                                 entry-ring slot of each sampled item
 * B <= 1024. */
int pfrl_tree_sample(const pfrl_tree_t *tree, int64_t B, const double *u01, int64_t *out_x,
                     double *out_pri, uint8_t *out_pri_tag, double *out_prob, float *out_weight,
                     double *out_total, uint8_t *out_total_tag, double *out_min_prob,
                     int normalize, double beta, int64_t slot_mod, int32_t *out_slot,
                     void *stream);

/* PrioritizedReplayBuffer.update_errors (pfrl/replay_buffers/prioritized.py:
 * 47-55,125-126) + PrioritizedBuffer.set_last_priority (:107-116) for
 * np.float32 errors resident on the device (|y-t| of DQN, dqn.py:447-454):
 * p = (clip(err, error_min, error_max) + eps) ** alpha with NEP-50 typing;
 * clipped values use the host-supplied constants (Python-float results of
 * (error_min + eps) ** alpha and (error_max + eps) ** alpha).  np.float32 ** alpha is libm's
 * powf in the reference; pow_mode (PFRL_POW_*) selects its device restatement -- bit-exact
 * leaves with the variant pfrl_powf_host_variant() reports for this host.
 * x must be unique unless dedupe != 0 (last occurrence wins). */
int pfrl_tree_update_errors_f32(const pfrl_tree_t *tree, int64_t B, const int64_t *x,
                                const float *err, int has_min, float error_min, double pri_at_min,
                                int has_max, float error_max, double pri_at_max, double eps,
                                double alpha, int dedupe, int pow_mode, void *stream);

/* The same followed by the leaf writes of pfrl_tree_write (n entries: the appends and pops that
 * PrioritizedBuffer.append / popleft, pfrl/collections/prioritized.py:39-54, recorded after this
 * minibatch was sampled) as one launch with one path repair: sequential semantics -- priorities
 * and max_priority first, then the writes (which win on a shared leaf).  Both sets must belong
 * to the frame in `tree`; B + n <= 1024. */
int pfrl_tree_update_errors_write_f32(const pfrl_tree_t *tree, int64_t B, const int64_t *x,
                                      const float *err, int has_min, float error_min,
                                      double pri_at_min, int has_max, float error_max,
                                      double pri_at_max, double eps, double alpha, int dedupe,
                                      int pow_mode, int64_t n, const int64_t *wx, const double *wval,
                                      const uint8_t *wtag, const uint8_t *wuse_maxp, void *stream);

/* HOST helpers of the priority transform (no device work).  pfrl_powf_host evaluates the
 * restated glibc powf (pow_mode PFRL_POW_GLIBC / _FMA) on host arrays; pfrl_powf_host_variant
 * compares both restatements with THIS host's libm powf on n_probe pseudo-random inputs in
 * (0, 2) plus the caller's alpha and returns the PFRL_POW_* mode that reproduces it bit for
 * bit (the FMA build when both do), or -1 when neither does (callers then evaluate the power
 * on the host, as NumPy does). */
int pfrl_powf_host(int pow_mode, const float *host_x, float alpha, float *host_out, int64_t n);
/* the same power as the update kernel evaluates per leaf, on device arrays (any PFRL_POW_*) */
int pfrl_powf_device(int pow_mode, const float *x, float alpha, float *out, int64_t n,
                     void *stream);
int pfrl_powf_host_variant(float alpha, int64_t n_probe);

/* PrioritizedBuffer.set_last_priority (pfrl/collections/prioritized.py:107-116) with
 * explicit typed priorities (host-computed, e.g. from Python-float errors): val/tag
 * are device arrays. */
int pfrl_tree_set_priorities(const pfrl_tree_t *tree, int64_t B, const int64_t *x,
                             const double *val, const uint8_t *tag, int dedupe, void *stream);

/* ------------------------------------------------------------------------
 * On-policy rollouts (pfrl/agents/ppo.py, a2c.py).  Layout [T][N], env minor.
 * ------------------------------------------------------------------------ */

/* _add_advantage_and_value_target_to_episode (pfrl/agents/ppo.py:36-47) for every episode
 * fragment of a T x N rollout.  cut[t][e] != 0 marks the last transition of a
 * fragment (done, reset or rollout end): the scan restarts with adv = 0.
 * mode 0: Python-float rewards (all f32 arithmetic); mode 1: np.float64
 * rewards (f64 accumulate).  adv / v_teacher are written as f32. */
int pfrl_gae_scan(int64_t T, int64_t N, const double *reward, const float *v_pred,
                  const float *next_v_pred, const uint8_t *nonterminal, const uint8_t *cut,
                  double gamma, double lambd, int mode, float *adv, float *v_teacher,
                  void *stream);

/* A2C._compute_returns (pfrl/agents/a2c.py:150-167); value_preds/returns are [T+1][N]
 * with row T already holding next_value as the reference sets it. */
int pfrl_a2c_returns(int64_t T, int64_t N, const float *rewards, const float *masks,
                     const float *value_preds, float *returns, double gamma, double tau,
                     int use_gae, void *stream);

/* torch.std_mean(all_advs, unbiased=False) (pfrl/agents/ppo.py:476-478): out[0] = mean,
 * out[1] = std, f32, accumulated in f64 with wavefront shuffle reductions. */
int pfrl_adv_stats(const float *adv, int64_t n, float *out_mean_std, void *partial_ws,
                   void *stream);

/* PPO minibatch assembly (pfrl/agents/ppo.py:483-511): for dataset positions idx[i]
 *   out_adv   = (adv[idx] - mean) / (std + 1e-8)   if standardize
 *   out_logp  = log_prob[idx], out_v = v_pred[idx], out_vt = v_teacher[idx],
 *   out_action= action[idx] (int64),  out_refs[i][:] = state_refs[idx][:]. */
int pfrl_ppo_minibatch(int64_t M, const int64_t *idx, const float *adv, const float *mean_std,
                       int standardize, const float *log_prob, const float *v_pred,
                       const float *v_teacher, const int64_t *action, const int32_t *state_refs,
                       int32_t k, float *out_adv, float *out_logp, float *out_v, float *out_vt,
                       int64_t *out_action, int32_t *out_refs, void *stream);

/* The two narrow heads of the PPO example network on the ACTING path and what PPO.batch_act does
 * with them (examples/atari/train_ppo_ale.py:257-263, pfrl/agents/ppo.py:759-778,
 * pfrl/policies/softmax_policy.py: Categorical(logits)): logits = h w_policy^T + b_policy,
 * value = h w_value^T + b_value, action ~ Categorical(logits) by inverse CDF on u01[row] in [0, 1),
 * entropy of the distribution, optionally log pi(action).  given_action != NULL: no draw -- the
 * value pass of an update (ppo.py:110-142): log pi(given_action | s) and V(s) (out_action /
 * out_entropy / u01 may be NULL).  h [N][K], w_policy [A][K], A <= 31.  * rows != NULL (device int32[2]): outputs are offset on the device -- out_action by rows[0] * N,
 * out_entropy / out_value by rows[1] * 2 * N -- so that a captured rollout step writes into the
 * rollout's columns with fixed kernel arguments. */
int pfrl_ppo_act_head(const float *h, const float *w_policy, const float *b_policy,
                      const float *w_value, const float *b_value, const float *u01,
                      const int64_t *given_action, int64_t *out_action, float *out_entropy,
                      float *out_value, float *out_log_prob, int32_t N, int32_t K, int32_t A,
                      const int32_t *rows, void *stream);
/* PPO._lossfun (pfrl/agents/ppo.py:634-671) on the logits [M, A] and values [M] of a minibatch:
 * clipped surrogate + (clipped, if clip_eps_vf >= 0) value MSE + entropy bonus, AND its gradient
 * with respect to logits and values (d loss = 1), in one launch + a one-workgroup finish.
 * `adv` is the (standardised) advantage column of the minibatch.  out4 = {loss, loss_policy,
 * loss_value, mean entropy}; partial_ws: 3 * ceil(M / 256) doubles.  A <= 31.  Replaces ~90
 * elementwise / reduction launches of torch.distributions + autograd per minibatch. */
int pfrl_ppo_loss(const float *logits, const float *value, const int64_t *action, const float *adv,
                  const float *log_prob_old, const float *v_pred_old, const float *v_teacher,
                  int32_t M, int32_t A, float clip_eps, float clip_eps_vf, float value_func_coef,
                  float entropy_coef, float *dlogits, float *dvalue, double *partial_ws, float *out4,
                  void *stream);
/* The same loss with the two narrow heads of the example network (examples/atari/train_ppo_ale.py:
 * nn.Linear(512, n_actions) + SoftmaxCategoricalHead, nn.Linear(512, 1) behind one body) and their
 * backward in ONE launch: logits = h Wp^T + bp, v = h Wv^T + bv, the loss of pfrl/agents/ppo.py:634-671
 * and its gradient, dh = dlogits Wp + dv Wv (where backward of the body starts) and the heads'
 * parameter gradients as `blocks` partial slabs dw_part[blocks][(A + 1) * K + pad4(A + 1)] (rows 0..A-1
 * = dWp, row A = dWv, then dbp[A], dbv; the bias block padded to a multiple of 4) for pfrl_splitk_reduce.  h is read once and dh written once.
 * K = 256 or 512, A <= 9; partial_ws: 3 * blocks doubles; out4 as pfrl_ppo_loss. */
int pfrl_ppo_head_loss(const float *h, const float *w_policy, const float *b_policy,
                       const float *w_value, const float *b_value, const int64_t *action,
                       const float *adv, const float *log_prob_old, const float *v_pred_old,
                       const float *v_teacher, int32_t M, int32_t K, int32_t A, float clip_eps,
                       float clip_eps_vf, float value_func_coef, float entropy_coef, float *dh,
                       float *dw_part, int32_t blocks, double *partial_ws, float *out4, void *stream);

/* ------------------------------------------------------------------------
 * Optimizer step of the DQN update (pfrl/agents/dqn.py:360-365 calls
 * optimizer.step(); examples/atari/train_dqn_batch_ale.py:199-206 builds
 * torch.optim.RMSprop(alpha=0.95, eps=1e-2, centered=True)).  One fused
 * multi-tensor launch with torch's RMSprop arithmetic.  The pointer arrays are
 * HOST arrays of device pointers (they are passed to the kernel by value, so
 * the launch can be captured in a HIP graph).  grad_avg may be NULL when
 * centered == 0. */
int pfrl_rmsprop_step(int32_t n_tensors, float *const *host_params,
                      const float *const *host_grads, float *const *host_square_avg,
                      float *const *host_grad_avg, const int64_t *host_numel, float lr,
                      float alpha, float eps, float weight_decay, int centered, void *stream);

/* The optimizer step that finishes the gradients (same reference lines, pfrl/agents/dqn.py:
 * 360-365, at minibatch size).  Each task updates one parameter tensor with RMSprop taking its
 * gradient in the form the backward pass left it:
 *   PLAIN         src = the gradient tensor
 *   SLABS         src = first of n_slabs split-K partial slabs, slab_stride floats apart: summed
 *                 while loading (what pfrl_splitk_reduce would have written)
 *   LOWRANK       the weight of a Linear(K, F) layer whose gradient is dy^T x over a batch of M
 *                 rows: src = dy [M][F] (counted where mask [M][F] > 0: the ReLU of the layer's
 *                 output; mask may be NULL), x [M][K]; formed tile by tile on the matrix cores
 *                 (f32 16x16x4 MFMA, exact f32) and applied from the accumulators -- the F x K
 *                 gradient never exists in memory.  K % 64 == 0, F % 16 == 0, M % 4 == 0, M <= 32
 *   LOWRANK_BIAS  that layer's bias: sum over the M rows of the masked dy column
 *   FOLD          no parameter: out[i] = sum of the slabs (loss terms that rode on the fold)
 * host_tasks is a host array (copied into the kernel arguments: graph-capturable). */
#define PFRL_OPT_MAX_TASKS 16
#define PFRL_OPT_PLAIN 0
#define PFRL_OPT_SLABS 1
#define PFRL_OPT_LOWRANK 2
#define PFRL_OPT_LOWRANK_BIAS 3
#define PFRL_OPT_FOLD 4
typedef struct {
    float *p, *sq, *ga;     /* parameter, square_avg, grad_avg (centered) */
    const float *src;
    float *out;             /* FOLD */
    const float *mask;      /* LOWRANK / LOWRANK_BIAS */
    const float *x;         /* LOWRANK */
    int64_t numel;
    int64_t slab_stride;
    int32_t n_slabs;
    int32_t mode;
    int32_t M, F, K;
    int32_t reserved;
} pfrl_opt_task_t;
int pfrl_rmsprop_fused_step(int32_t n_tasks, const pfrl_opt_task_t *host_tasks, float lr,
                            float alpha, float eps, float weight_decay, int centered, void *stream);

/* DQN / DoubleDQN TD loss and its gradient in one launch
 * (pfrl/agents/dqn.py:388-470 _compute_target_values/_compute_y_and_t/
 * _compute_loss, losses :44-104; pfrl/agents/double_dqn.py:12-40):
 *   y_b    = q[b][action_b]
 *   next_b = target_q[b][argmax_a sel[b][a]],  sel = next_q_online (Double DQN)
 *            or target_q itself (DQN: max_a)
 *   t_b    = reward_b + (discount_b * (1 - terminal_b)) * next_b
 *   loss   = sum_b w_b * L(y_b - t_b)  [/ B if mean],  L = Huber(1) or d^2 / 2
 * Outputs: loss[1], grad_q[B][A] = dloss/dq, y[B], abs_delta[B] = |y - t|
 * (the TD errors handed to PrioritizedReplayBuffer.update_errors).
 * next_q_online and weights may be NULL. */
int pfrl_dqn_td_loss(const float *q, const int64_t *action, const float *target_q,
                     const float *next_q_online, const float *reward, const float *discount,
                     const float *terminal, const float *weights, int64_t B, int32_t A,
                     int clip_delta, int mean, float *out_loss, float *out_grad_q, float *out_y,
                     float *out_abs_delta, void *stream);
/* The narrow head Linear(K, A) (examples/atari/train_dqn_batch_ale.py:35-41), the TD loss above
 * and the head's backward in ONE launch, a wave per row: h [B][K] is the head's input, w [A][K],
 * bias [A]; out_y / out_abs_delta as pfrl_dqn_td_loss, dh [B][K] = dL/dh (the `mean` scaling
 * included).  What couples the rows leaves as one partial slab per four rows,
 * [ceil(B/4)][A*K + 32], for pfrl_splitk_reduce (splits = ceil(B/4), stride = A*K + 32): dL/dw = the fold of [0, A*K), dL/db of
 * [A*K, A*K + A), the loss of [A*K + 16].  A <= 16, K = 256 or 512.  * h_part != NULL: the hidden layer's forward (pfrl_conv2d_nhwc_fwd with splits > 1) left h as
 * h_splits split-K slabs of h_stride floats; each row folds them here (h = relu(sum + h_bias),
 * the order of pfrl_splitk_reduce), uses h and writes it to h_out for the backward pass -- the
 * fold launch between the two disappears.  h is then ignored.
 * dh_masked != NULL (data parallel): also dh with the ReLU mask of h applied (h > 0) and
 * multiplied by dh_scale (1 / world size) -- the hidden layer's batch matrix as the low-rank
 * gradient exchange all-gathers it (pfrl_amd/distributed.py::lowrank_ready). */
int pfrl_dqn_head_td_loss(const float *h, const float *w, const float *bias, const int64_t *action,
                          const float *target_q, const float *next_q_online, const float *reward,
                          const float *discount, const float *terminal, const float *weights,
                          int32_t B, int32_t K, int32_t A, int clip_delta, int mean, float *out_y,
                          float *out_abs_delta, float *dh, float *partials, const float *h_part, int32_t h_splits,
                          int64_t h_stride, const float *h_bias, float *h_out, float *dh_masked,
                          float dh_scale, void *stream);

/* Fused bias + ReLU of the conv trunk (pfrl/nn/atari_cnn.py:40-47: activation(
 * layer(h)) with conv bias) on row-major [rows][C] activations, i.e.
 * channels_last conv outputs flattened over N*H*W.  Forward: y = max(x + b[c], 0)
 * (C % 4 == 0).  Backward: gx = gy * (y > 0) and gb[c] = sum over rows of gx, in
 * ONE launch (workgroup partials published as {value, epoch} granules + fold by
 * the last arriver; C must divide 256).  granule_ws: uint64[blocks * C];
 * counters: uint64[2], zero-initialised once, owned by one (C, blocks) pair and
 * never touched by the host afterwards (they carry the launch epoch, which keeps
 * the kernel valid under HIP-graph replay).  Launches that share a workspace
 * must be ordered on one stream.  The rows are split evenly over `blocks`.
 * blocks == 0 selects the single-workgroup form for few rows of any width (rows <= 1024,
 * e.g. the hidden linear layer at minibatch 32): no workspace, granule_ws / counters may
 * be NULL.
 * planar_hw > 0: y (forward output; y and gy in backward) is PLANAR, [N][C][planar_hw]
 * (plain NCHW) instead of rows of C channels -- for the last convolution of a trunk,
 * whose output is flattened for a linear layer (`h.view(h.size(0), -1)`,
 * pfrl/nn/atari_cnn.py:46): the flatten and its backward are then views instead of
 * layout-copy launches.  x and gx stay channels-last rows. */
int pfrl_bias_relu_fwd(const float *x, const float *bias, float *y, int64_t rows, int32_t C,
                       int64_t planar_hw, void *stream);
int pfrl_bias_relu_bwd(const float *gy, const float *y, float *gx, float *gb, uint64_t *granule_ws,
                       uint64_t *counters, int64_t rows, int32_t C, int32_t blocks,
                       int64_t planar_hw, void *stream);

/* ------------------------------------------------------------------------
 * Categorical (C51) DQN loss in one launch: replaces
 *   pfrl/agents/categorical_dqn.py:7-57   _apply_categorical_projection
 *   pfrl/agents/categorical_dqn.py:60-104 compute_(weighted_)value_loss
 *   pfrl/agents/categorical_dqn.py:150-204 _compute_target_values / _compute_y_and_t / _compute_loss
 *   pfrl/agents/categorical_double_dqn.py:10-52 (next_select = online net at s')
 *   pfrl/action_value.py:97-180 greedy_actions / evaluate_actions(_as_distribution)
 * q_dist, next_dist, next_select: f32 [B][A][Z] probabilities (next_select NULL =
 * next_dist, i.e. plain CategoricalDQN); z_values f32 [Z] evenly spaced, Z <= 64.
 *   g        = first argmax_a sum_z next_select[b][a][z] * z_values[z]
 *   Tz[j]    = reward[b] + ((1 - terminal[b]) * discount[b]) * z_values[j], clamped
 *   t[b][.]  = projection of next_dist[b][g][.] carried by Tz onto z_values
 *   delta[b] = sum_z -t[b][z] * log(clamp(q_dist[b][action[b]][z], 1e-10, 1))
 *   loss     = sum_b w[b] * delta[b]  (w = 1 if weights NULL; / B if mean)
 *   out_grad_q [B][A][Z] = d loss / d q_dist;  out_q[b] = sum_z q_dist[b][a_b][z] z_values[z] */
int pfrl_c51_loss(const float *q_dist, const int64_t *action, const float *next_dist,
                  const float *next_select, const float *z_values, const float *reward,
                  const float *discount, const float *terminal, const float *weights, int32_t B,
                  int32_t A, int32_t Z, int32_t mean, float *out_loss, float *out_grad_q,
                  float *out_q, float *out_delta, void *stream);

/* ------------------------------------------------------------------------
 * Distributional dueling head (pfrl/q_functions/dueling_dqn.py:116-127), forward
 * and backward in one launch each.  ya f32 [B][A][Z] advantage logits, ys f32 [B][Z]
 * state-value logits, Z <= 64:
 *   q[b][a][.] = softmax_z((ya[b][a][z] - sum_a' ya[b][a'][z] / A) + ys[b][z])
 * backward: gq = d loss / d q  ->  g_ya [B][A][Z], g_ys [B][Z]. */
int pfrl_dueling_softmax_fwd(const float *ya, const float *ys, float *q, int64_t B, int32_t A,
                             int32_t Z, void *stream);
int pfrl_dueling_softmax_bwd(const float *gq, const float *q, float *g_ya, float *g_ys, int64_t B,
                             int32_t A, int32_t Z, void *stream);

/* ------------------------------------------------------------------------
 * Factorised NoisyNet weights (pfrl/nn/noisy_linear.py:52-70: `_eps` shaping,
 * `torch.ger`, two `torch.addcmul`), forward and backward in one launch each.
 *   r            f32 [in + out] unit Gaussians: r[0:in] -> eps_x, r[in:] -> eps_y
 *   f(x)         sign(x) * sqrt(|x|)
 *   w_out[o][i]  mu_w + sigma_w * (f(r[in+o]) * f(r[i]));  b_out[o] = mu_b + sigma_b * f(r[in+o])
 *   g_sigma_w    g_w * (f(r[in+o]) * f(r[i]));             g_sigma_b = g_b * f(r[in+o])
 * (the gradients of mu_w / mu_b are g_w / g_b themselves).  mu_b, sigma_b, b_out
 * (and g_b, g_sigma_b) may be NULL for layers without bias.  Row-major, dense. */
int pfrl_noisy_weights_fwd(const float *mu_w, const float *sigma_w, const float *mu_b,
                           const float *sigma_b, const float *r, float *w_out, float *b_out,
                           int64_t out_features, int64_t in_features, void *stream);
int pfrl_noisy_weights_bwd(const float *g_w, const float *g_b, const float *r, float *g_sigma_w,
                           float *g_sigma_b, int64_t out_features, int64_t in_features,
                           void *stream);

/* ------------------------------------------------------------------------
 * Q-network trunk at minibatch sizes (pfrl/nn/atari_cnn.py:17-47 `activation(layer(h))`
 * for the three Nature convolutions and the hidden linear layer, and their autograd
 * backward inside DQN.update, pfrl/agents/dqn.py:316-365): f32 MFMA implicit-GEMM
 * kernels (v_mfma_f32_16x16x4_f32: exact f32 fmaf chains) with bias / ReLU / ReLU-mask
 * epilogues, sized for B = 32 where the library kernels are launch bound.
 * Layouts: activations NHWC (torch.channels_last memory), weights [Cout][R][S][Cin]
 * (a channels_last Conv2d weight); no padding, dilation 1, groups 1; S*C % 32 == 0.
 * A linear layer is the case H = W = R = S = stride = 1, C = in_features.
 *
 * pfrl_conv2d_nhwc_fwd: y = conv(x, w) + bias, ReLU if `relu`; y is NHWC rows
 *   [N*OH*OW][Cout], or plain NCHW if `planar_out` (the convolution in front of a
 *   flatten).  splits > 1: split-K; y receives `splits` raw partial slabs
 *   [splits][N*OH*OW][Cout] (no bias, no ReLU) for pfrl_splitk_reduce.
 * pfrl_conv2d_nhwc_bwd_data: dx = conv_transpose(dy, w) masked by (a_prev > 0) when
 *   a_prev (the ReLU output that is this layer's input; dx's layout) is given; dy is
 *   first masked by (dy_mask > 0) when dy_mask (same layout as dy) is given.  Needs
 *   Cout % 32 == 0, C % 16 == 0, R, S, H, W multiples of stride.  perm_p > 0
 *   (linear layers after a planar flatten, C = perm_c * perm_p): a_prev is read
 *   planar [c][p] and dx is written as NHWC rows [p][c].
 * pfrl_conv2d_nhwc_bwd_weight: dw[co][r][s][ci] and db[co], reduction over the
 *   N*OH*OW rows cut into `splits` ranges; split z writes dw_part + z*dw_stride and
 *   db_part + z*db_stride (splits == 1: the gradients themselves).  db_part may be NULL.
 * pfrl_splitk_reduce: out[e] = act(sum_s part[s*stride + e] + bias[e % ncol]) for up
 *   to 12 tensors in one launch (host arrays of device pointers, passed by value).
 * pfrl_linear_small_fwd / _bwd: a narrow head, y = x w^T + b with out_features <= 16
 *   (Linear(512, n_actions), pfrl/q_functions/state_q_functions.py) and its backward
 *   (dx may be NULL; dw and db may both be NULL for frozen weights).  _bwd alone goes up to
 *   out_features 64, M * out_features <= 10 240: the 2 x action_size policy head of SAC
 *   (Linear(256, 34), train_soft_actor_critic.py:128-141), whose width the tile engine's
 *   32-column gradient tiles do not divide.
 * pfrl_linear_fwd: y = act(x w^T + b), x [M][K], w [N][K], any K and N, no alignment
 *   requirement -- the `nn.Linear` layers of the MLP agents (obs 376 -> 256,
 *   obs + action 393 -> 256: examples/mujoco/reproduction/soft_actor_critic/
 *   train_soft_actor_critic.py:172-243).  splits > 1: partials for pfrl_splitk_reduce. */
int pfrl_conv2d_nhwc_fwd(const float *x, const float *w, const float *bias, float *y, int32_t N,
                         int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t R, int32_t S,
                         int32_t stride, int32_t relu, int32_t planar_out, int32_t splits,
                         void *stream);
int pfrl_conv2d_nhwc_bwd_data(const float *dy, const float *dy_mask, const float *w,
                              const float *a_prev, float *dx, int32_t N, int32_t H, int32_t W,
                              int32_t C, int32_t Cout, int32_t R, int32_t S, int32_t stride,
                              int32_t perm_p, int32_t perm_c, void *stream);
int pfrl_conv2d_nhwc_bwd_weight(const float *dy, const float *dy_mask, const float *x, float *dw_part,
                                float *db_part, int64_t dw_stride, int64_t db_stride, int32_t N,
                                int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t R, int32_t S,
                                int32_t stride, int32_t splits, void *stream);
/* pfrl_conv2d_nhwc_fwd / pfrl_conv2d_nhwc_bwd_weight for a FIRST layer whose input is the u8
 * NHWC4 minibatch of pfrl_batch_states_u8_raw_nhwc4 (C = 4): the feature extractor
 * phi(x) = float32(x) / divisor (pfrl/utils/batch_states.py:18-36 with the example scripts' phi,
 * examples/atari/train_ppo_ale.py:229-231) is evaluated where the operand enters LDS, with the
 * rounding of IEEE division for every byte value (the caller checks the divisor:
 * pfrl_amd.ops.u8_division_exact).  Same tile programs, operands and summation order as the fp32
 * entries on the gathered fp32 minibatch: bit-identical y / dw / db.  Cout % 32 == 0 and
 * Cout % 64 != 0; forward: at least 384 tiles of 32 x 32, no split-K. */
int pfrl_conv2d_u8nhwc4_fwd(const uint8_t *x, float divisor, const float *w, const float *bias,
                            float *y, int32_t N, int32_t H, int32_t W, int32_t Cout, int32_t R,
                            int32_t S, int32_t stride, int32_t relu, int32_t planar_out, void *stream);
int pfrl_conv2d_u8nhwc4_bwd_weight(const float *dy, const float *dy_mask, const uint8_t *x,
                                   float divisor, float *dw_part, float *db_part, int64_t dw_stride,
                                   int64_t db_stride, int32_t N, int32_t H, int32_t W, int32_t Cout,
                                   int32_t R, int32_t S, int32_t stride, int32_t splits, void *stream);
/* pfrl_conv2d_nhwc_bwd_weight with the RMSprop steps of OTHER parameter tensors riding in the
 * same launch (pfrl/agents/dqn.py:360-365 `loss.backward(); optimizer.step()` at minibatch size):
 * the first layer's weight gradient is the last launch of the backward pass, a few hundred
 * latency-bound workgroups; by then the gradients of the layers above are final (ride_grad[i],
 * finished tensors) and their parameters are not read again in this update, so their
 * elementwise steps (the arithmetic of pfrl_rmsprop_step) run as extra workgroups of this launch.
 * 1 <= n_ride <= 8 HOST arrays of device pointers (plus whatever pfrl_ride_set left pending); 16-byte aligned, numel % 4 == 0;
 * ride_grad_avg may be NULL when centered == 0.  The caller must not step these tensors again. */
int pfrl_conv2d_nhwc_bwd_weight_ride(
    const float *dy, const float *dy_mask, const float *x, float *dw_part, float *db_part,
    int64_t dw_stride, int64_t db_stride, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout,
    int32_t R, int32_t S, int32_t stride, int32_t splits, int32_t n_ride, float *const *ride_param,
    const float *const *ride_grad, float *const *ride_square_avg, float *const *ride_grad_avg,
    const int64_t *ride_numel, float lr, float alpha, float eps, float weight_decay, int centered,
    void *stream);
/* Optimizer steps (the arithmetic of pfrl_rmsprop_step) handed to the NEXT backward launch of this
 * host thread -- pfrl_conv2d_nhwc_bwd or pfrl_conv2d_nhwc_bwd_weight_ride -- which runs them as extra
 * workgroups and clears the set (pfrl/agents/dqn.py:360-365 `loss.backward(); optimizer.step()` at
 * minibatch size: by the time a layer's backward launch starts, the gradient of the layer above
 * is complete and its parameters are not read again in the update).  Up to 8 tensors; the
 * gradient of tensor i is grad_src[i] itself (n_slabs[i] == 0) or the sum, in slab order, of
 * n_slabs[i] split-K slabs slab_stride[i] floats apart (what pfrl_conv2d_nhwc_bwd_weight /
 * pfrl_dqn_head_td_loss leave).  n == 0 clears the set.  A pfrl_conv2d_nhwc_bwd whose tile
 * program has no riding form returns an error WITHOUT launching (and clears the set): the caller
 * launches again and steps those tensors elsewhere. */
int pfrl_ride_set(int32_t n, float *const *param, const float *const *grad_src,
                  float *const *square_avg, float *const *grad_avg, const int64_t *numel,
                  const int32_t *n_slabs, const int64_t *slab_stride, float lr, float alpha, float eps,
                  float weight_decay, int centered);
/* Both gradients of one layer in ONE launch (same dy): arguments of _bwd_data and _bwd_weight
 * combined.  Minibatch-sized problems only; returns PFRL_ERR_ARG for larger ones (the caller
 * then issues the two launches). */
int pfrl_conv2d_nhwc_bwd(const float *dy, const float *dy_mask, const float *w, const float *a_prev,
                         const float *x, float *dx, float *dw_part, float *db_part, int64_t dw_stride,
                         int64_t db_stride, int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cout,
                         int32_t R, int32_t S, int32_t stride, int32_t perm_p, int32_t perm_c,
                         int32_t splits, void *stream);
int pfrl_splitk_reduce(int32_t n_tasks, const float *const *host_part, float *const *host_out,
                       const float *const *host_bias, const int64_t *host_stride,
                       const int32_t *host_n, const int32_t *host_splits, const int32_t *host_ncol,
                       const int32_t *host_relu, void *stream);
/* torch.nn.utils.clip_grad_norm_(parameters, max_norm, norm_type=2) as the reference's update
 * calls it between backward and step (pfrl/agents/ppo.py:602-605, dqn.py:362-364): global L2 norm
 * of up to 24 dense f32 gradient tensors, coef = min(max_norm / (norm + 1e-6), 1), every gradient
 * scaled in place -- three launches.  partial_ws: sum over tensors of ceil(numel / 4096) doubles;
 * out_norm_coef[0] = the norm, [1] = the coefficient. */
int pfrl_clip_grad_norm(int32_t n_tensors, float *const *grads, const int64_t *numel, float max_norm,
                        double *partial_ws, float *out_norm_coef, void *stream);
/* First level of a two-level split-K fold: out[g * out_stride + e] = the sum, in slab order, of
 * slabs [g * per, (g + 1) * per) of `part` (per = ceil(splits / groups)), e < n, all groups in one
 * launch; pfrl_splitk_reduce then folds the `groups` partial slabs.  For tensors with thousands
 * of short slabs (the first convolution's weight gradient at rollout size, autograd's
 * `conv2d` backward in PPO._update_once, pfrl/agents/ppo.py:521-532). */
int pfrl_splitk_group(const float *part, int64_t stride, int32_t n, int32_t splits, int32_t groups,
                      float *out, int64_t out_stride, void *stream);
int pfrl_linear_fwd(const float *x, const float *w, const float *bias, float *y, int32_t M, int32_t K,
                    int32_t N, int32_t relu, int32_t splits, void *stream);
/* Weight and bias gradient of such a layer, any in_features (out_features % 16 == 0):
 * partials for pfrl_splitk_reduce laid out as pfrl_conv2d_nhwc_bwd_weight's. */
int pfrl_linear_bwd_weight(const float *dy, const float *dy_mask, const float *x, float *dw_part,
                           float *db_part, int64_t dw_stride, int64_t db_stride, int32_t M, int32_t K,
                           int32_t N, int32_t splits, void *stream);
int pfrl_linear_small_fwd(const float *x, const float *w, const float *bias, float *y, int32_t M,
                          int32_t K, int32_t N, void *stream);
/* The narrow Q head on the ACTING path with what DQN.batch_act does with its output, in one
 * launch (pfrl/agents/dqn.py:490-507: `batch_av.greedy_actions` = argmax over the action values,
 * pfrl/action_value.py:59-61; then per env `select_action_epsilon_greedily`,
 * pfrl/explorers/epsilon_greedy.py:8-12): q = h w^T + b with the arithmetic of
 * pfrl_linear_small_fwd (bit-identical action values; written to `q` unless NULL), greedy = the
 * FIRST maximum, action[m] = choice[m] >= 0 ? choice[m] : greedy[m] (`choice`: the host's
 * epsilon-greedy draws, pfrl_plan_eps_greedy; NULL = greedy).  `greedy` may be NULL.
 * Replaces pfrl_linear_small_fwd + an argmax reduction + a cast + pfrl_select_actions. */
int pfrl_dqn_act_head(const float *h, const float *w, const float *bias, const int32_t *choice,
                      float *q, int64_t *greedy, int64_t *action, int32_t M, int32_t K, int32_t N,
                      void *stream);
/* Forward tile programs (pfrl_conv2d_nhwc_fwd, pfrl_conv2d_u8nhwc4_fwd) planned for a batch of
 * `images` images when a call brings fewer (0: off; per host thread).  No tile program mixes rows,
 * so a row evaluated in a small batch under this plan has bit for bit the value it has inside the
 * large batch: PPO evaluates V(next_state) only for the rows that are not also a state of the
 * rollout and still returns what the reference's second pass over ALL next states
 * (pfrl/agents/ppo.py:119-133) would, pfrl_amd/agents/ppo.py::_next_values_exact. */
int pfrl_qnet_plan_images(int32_t images);
int pfrl_linear_small_bwd(const float *dy, const float *x, const float *w, float *dx, float *dw,
                          float *db, int32_t M, int32_t K, int32_t N, void *stream);

/* ------------------------------------------------------------------------
 * Twin launches (csrc/qnet.hip): the same layer of the two Q-networks of SAC / TD3
 * (pfrl/agents/soft_actor_critic.py:97-110, always evaluated on the same inputs one after
 * the other) as ONE grid.  Every `const float *const *` argument is a HOST array of two
 * device pointers.  Same arithmetic as the single-network entry points.
 * pfrl_linear_fwd_twin: y_t = act(x_t w_t^T + b_t), any in_features, out_features % 32 == 0.
 *   With x2 != NULL the input rows are [x_t (K1 columns) | x2_t (K - K1 columns)]:
 *   ConcatObsAndAction (pfrl/nn/concat_obs_and_action.py) without materialising the
 *   concatenation; likewise for the weight gradient in pfrl_linear_bwd_twin.
 * pfrl_linear_bwd_twin: dx == NULL: weight/bias gradient partials only (any in_features);
 *   dw_part == NULL: input gradients only; both: all four gradients in one launch
 *   (in_features % 32 == 0).
 * pfrl_linear_small_fwd_twin / _bwd_twin: the narrow heads (out_features <= 16).
 * pfrl_twin_input_grad: dx[m][j] = sum_t sum_n (dy_t (.) mask_t)[m][n] w_t[n][col0 + j],
 *   j < ncol <= 32: the gradient w.r.t. the action columns of the twins' first layer, the only
 *   input gradient the policy loss needs (:284-291).
 * pfrl_half_mse_twin_fwd/_bwd (below): both critic losses of one target. */
int pfrl_linear_fwd_twin(const float *const *x, const float *const *x2, int32_t K1,
                         const float *const *w, const float *const *bias, float *const *y, int32_t M,
                         int32_t K, int32_t N, int32_t relu, void *stream);
int pfrl_linear_bwd_twin(const float *const *dy, const float *const *dy_mask, const float *const *w,
                         const float *const *x, const float *const *x2, int32_t K1, float *const *dx,
                         float *const *dw_part,
                         float *const *db_part, int64_t dw_stride, int64_t db_stride, int32_t M,
                         int32_t K, int32_t N, int32_t splits, void *stream);
int pfrl_linear_small_fwd_twin(const float *const *x, const float *const *w, const float *const *bias,
                               float *const *y, int32_t M, int32_t K, int32_t N, void *stream);
int pfrl_linear_small_bwd_twin(const float *const *dy, const float *const *x, const float *const *w,
                               float *const *dx, float *const *dw, float *const *db, int32_t M,
                               int32_t K, int32_t N, void *stream);
int pfrl_twin_input_grad(const float *const *dy, const float *const *dy_mask, const float *const *w,
                         int32_t ldw, int32_t col0, int32_t ncol, float *dx, int32_t M, int32_t N,
                         void *stream);

/* ------------------------------------------------------------------------
 * Actor-critic update helpers (csrc/actor.hip): the elementwise stretches of the
 * SAC / TD3 / DDPG update, pfrl/agents/soft_actor_critic.py:213-330.
 *
 * pfrl_squashed_gaussian_fwd: for the policy head of examples/mujoco/reproduction/
 *   soft_actor_critic/train_soft_actor_critic.py:128-141 -- TransformedDistribution(
 *   Independent(Normal(loc, scale), 1), [TanhTransform(cache_size=1)]) -- the
 *   reparameterised sample action = tanh(loc + eps*scale) [B][A] and its
 *   log-probability logp[B] (what `rsample()` + `log_prob()` compute, agents/
 *   soft_actor_critic.py:228-229, 282-283).  loc / scale rows may be strided (ld in
 *   elements); eps [B][A] is the caller's standard-normal draw; neg_logp (may be NULL)
 *   receives -logp, the entropy estimate recorded at :299-306.
 * pfrl_squashed_gaussian_bwd: gradients w.r.t. loc and scale ([B][A] each) from
 *   dL/daction (may be NULL) and dL/dlogp (may be NULL).
 * pfrl_squashed_head_fwd / _bwd: the same two launches with the example's head function folded
 *   in (train_soft_actor_critic.py:128-141: mean, log_scale = chunk(x, 2); scale =
 *   sqrt(exp(2 clamp(log_scale, lo, hi))), mode 0 -- or exp(clamp(..)), mode 1): x [B][2A] (row
 *   stride ldx) is the last Linear's output, g_x [B][2A] its gradient, clamp mask included.
 * pfrl_soft_update: dst <- (1 - tau) dst + tau src for n tensors in one launch
 *   (pfrl/utils/copy_param.py:10-28; host arrays of device pointers, by value).
 * pfrl_adam_step: torch.optim.Adam's update (no amsgrad; _single_tensor_adam
 *   arithmetic) for n tensors in one launch.  steps[t] is the tensor's device-side f32
 *   step counter: read as t - 1, advanced by one inside the launch.  `ticket` is a
 *   zero-initialised device uint32 owned by the optimizer. */
int pfrl_squashed_gaussian_fwd(const float *loc, int64_t ld_loc, const float *scale, int64_t ld_scale,
                               const float *eps, float *action, float *logp, float *neg_logp, int32_t B,
                               int32_t A, void *stream);
int pfrl_squashed_gaussian_bwd(const float *g_action, const float *g_logp, const float *action,
                               const float *eps, const float *scale, int64_t ld_scale, float *g_loc,
                               float *g_scale, int32_t B, int32_t A, void *stream);
int pfrl_squashed_head_fwd(const float *x, int64_t ldx, float clamp_lo, float clamp_hi, int32_t mode,
                           const float *eps, float *action, float *logp, float *neg_logp, int32_t B,
                           int32_t A, void *stream);
int pfrl_squashed_head_bwd(const float *g_action, const float *g_logp, const float *action,
                           const float *eps, const float *x, int64_t ldx, float clamp_lo, float clamp_hi,
                           int32_t mode, float *g_x, int32_t B, int32_t A, void *stream);
int pfrl_soft_update(int32_t n_tensors, float *const *dst, const float *const *src,
                     const int64_t *numel, double tau, void *stream);
int pfrl_adam_step(int32_t n_tensors, float *const *params, const float *const *grads,
                   float *const *exp_avg, float *const *exp_avg_sq, float *const *steps,
                   const int64_t *numel, double lr, double beta1, double beta2, double eps,
                   double weight_decay, void *ticket, void *stream);
/* pfrl_adam_step with two riders: grads[t] may still be n_slabs[t] split-K partial slabs
 * slab_stride[t] elements apart (summed slab 0 first, as pfrl_splitk_reduce sums them -- that launch
 * then never runs; n_slabs == NULL: all plain), and soft_dst[t] (entry or array may be NULL) takes
 * the soft target update dst <- (1 - tau) dst + tau p from the parameter value just written
 * (pfrl_soft_update's arithmetic; pfrl/agents/soft_actor_critic.py:262-263 then :308). */
int pfrl_adam_step_ex(int32_t n_tensors, float *const *params, const float *const *grads,
                      const int64_t *slab_stride, const int32_t *n_slabs, float *const *exp_avg,
                      float *const *exp_avg_sq, float *const *steps, float *const *soft_dst, double tau,
                      const int64_t *numel, double lr, double beta1, double beta2, double eps,
                      double weight_decay, void *ticket, void *stream);
/* SAC losses on [B] vectors, pfrl/agents/soft_actor_critic.py.  The temperature is
 * exp(*log_temperature) when that device pointer is given (TemperatureHolder, :60-76),
 * else the float argument.
 * pfrl_sac_target_q (:226-240): target_q = reward + discount * (1 - terminal) *
 *   (min(next_q1, next_q2) - T * next_log_prob).
 * pfrl_half_mse_fwd/_bwd (:247-248): loss[0] = 0.5 * mean((target - pred)^2) and its
 *   gradient w.r.t. pred given g_loss[0].
 * pfrl_sac_policy_loss_fwd/_bwd (:284-291): loss[0] = mean(T * log_prob - min(q1, q2)) and
 *   its gradients w.r.t. log_prob, q1, q2 (a tie in the minimum is split evenly).
 * unit_g_* (pfrl_half_mse_twin_fwd, pfrl_sac_policy_loss_fwd; may be NULL): the forward launch
 *   also writes the gradients for an upstream gradient of exactly 1 -- what loss.backward()
 *   passes -- with the _bwd kernels' arithmetic, so that backward needs no launch of its own. */
/* pfrl_sac_temperature_loss (:264-271): loss[0] = -mean(exp(*log_temperature) * (log_prob +
 *   entropy_target)); its derivative w.r.t. log_temperature is the loss itself. */
int pfrl_sac_temperature_loss(const float *log_temperature, const float *log_prob,
                              float entropy_target, float *loss, int32_t B, void *stream);
/* ... and torch.optim.Adam's step on *log_temperature with that loss as its gradient, in the same
 * launch (the optimizer's exp_avg / exp_avg_sq / device-side step counter of that parameter). */
int pfrl_sac_temperature_step(float *log_temperature, const float *log_prob, float entropy_target,
                              float *loss, float *exp_avg, float *exp_avg_sq, float *step, double lr,
                              double beta1, double beta2, double eps, double weight_decay, int32_t B,
                              void *stream);
int pfrl_sac_target_q(const float *reward, const float *discount, const float *terminal,
                      const float *next_q1, const float *next_q2, const float *next_log_prob,
                      const float *log_temperature, float temperature, float *target_q, int32_t B,
                      void *stream);
int pfrl_half_mse_fwd(const float *target, const float *pred, float *loss, int32_t B, void *stream);
int pfrl_half_mse_bwd(const float *g_loss, const float *target, const float *pred, float *g_pred,
                      int32_t B, void *stream);
int pfrl_half_mse_twin_fwd(const float *target, const float *const *pred, float *const *loss,
                           float *const *unit_g_pred, int32_t B, void *stream);
int pfrl_half_mse_twin_bwd(const float *const *g_loss, const float *target, const float *const *pred,
                           float *const *g_pred, int32_t B, void *stream);
int pfrl_sac_policy_loss_fwd(const float *log_prob, const float *q1, const float *q2,
                             const float *log_temperature, float temperature, float *loss,
                             float *unit_g_log_prob, float *unit_g_q1, float *unit_g_q2, int32_t B,
                             void *stream);
int pfrl_sac_policy_loss_bwd(const float *g_loss, const float *q1, const float *q2,
                             const float *log_temperature, float temperature, float *g_log_prob,
                             float *g_q1, float *g_q2, int32_t B, void *stream);

/* Ragged gather of sampled episodes for recurrent updates: batch_recurrent_experiences
 * (pfrl/replay_buffer.py:219-287) over EpisodicReplayBuffer.sample_episodes (pfrl/replay_buffers/
 * episodic.py:48-85, windows cut by random_subseq).  An episode is a run of consecutive
 * one-transition entries of the entry ring; ep_first[e] = entry slot of the first transition of
 * sampled window e (windows sorted by descending length), ep_row0[0..n_eps] = prefix of their
 * lengths, row_start[0..T] = first packed row of time step t (T = longest window).  Emits
 * state / next_state episode-major ([rows][k][frame], u8 -> f32 / divisor as
 * pfrl_batch_experiences) and action / reward / is_state_terminal / discount (= gamma) in
 * time-major packed order (pfrl/utils/recurrent.py flatten_sequences_time_first).  Episode
 * payloads never leave HBM; the host supplies three small int32 arrays per sample. */
int pfrl_batch_episodes(const pfrl_table_t *tab, const void *frames, int64_t frame_bytes,
                        int frames_are_f32, float divisor, const int32_t *ep_first,
                        const int32_t *ep_row0, const int32_t *row_start, int32_t n_eps, int32_t T,
                        int64_t rows, int64_t entry_ring, float gamma, void *out_state,
                        void *out_next_state, void *out_action, float *out_reward,
                        float *out_terminal, float *out_discount, void *stream);

/* hipMemcpyAsync(dst, host_src, nbytes, HostToDevice, stream) for the pinned staging rings
 * (pfrl_amd/staging.py): the H2D of the reference's `.to(device)` in pfrl/utils/batch_states.py:
 * 18-36 and `torch.tensor([...], device=...)` conversions, reduced to index / draw traffic. */
int pfrl_h2d_async(void *dst, const void *host_src, int64_t nbytes, void *stream);

/* ------------------------------------------------------------------------
 * HOST step planner (no device work; every pointer is host memory).  The reference's batched
 * step consumes NumPy's legacy global stream in a fixed order (pfrl/agents/dqn.py:490-549):
 * per env rand() [+ randint(n_actions)] in batch_act (pfrl/explorers/epsilon_greedy.py:8-12),
 * then per env append and, when due, n_times_update x sample_n_k(len, B)
 * (pfrl/replay_buffer.py:329-356, pfrl/utils/random.py:4-28).  These entry points make exactly
 * those draws on NumPy's own generator: `bitgen` is the `bitgen_t *` NumPy publishes
 * (numpy/random/bitgen.h; np.random.mtrand._rand._bit_generator.ctypes.bit_generator), so the
 * stream position before and after is that of the Python loop. */

/* host mirrors and geometry of a device replay store with one-step entries
 * (pfrl_amd/replay_buffers/device_replay.py); arrays are the store's NumPy mirrors */
typedef struct {
    int32_t *h_state_ref;   /* [R][k] */
    int32_t *h_next_ref;    /* [R][k] */
    double *h_reward;       /* [R] */
    uint8_t *h_terminal;    /* [R] */
    int64_t *h_min_fseq;    /* [R] oldest frame sequence number a transition references */
    int64_t *h_e_tids;      /* [E][n] absolute transition ids */
    int32_t *h_e_len;       /* [E] */
    int64_t *h_e_min_fseq;  /* [E] */
    int64_t R, E;           /* ring sizes of the transition / entry tables */
    int64_t maxlen;         /* ReplayBuffer capacity, -1 = unbounded */
    int64_t bound;          /* device allocation of an unbounded buffer */
    int32_t k, n;           /* frames per observation, num_steps (must be 1) */
} pfrl_host_store_t;

#define PFRL_PLAN_DENSE (-10)      /* 3 B >= len: np.random.choice(replace=False) regime; nothing
                                      was drawn or written, take the Python path */
#define PFRL_PLAN_OVERFLOW (-11)   /* unbounded buffer beyond its allocation */
#define PFRL_PLAN_FRAME_RING (-12) /* a sampled entry's frames were overwritten: fatal */

/* sample_n_k(n, k), sparse regime 3 k < n (pfrl/utils/random.py:13-28): k distinct indices. */
int pfrl_plan_sample_n_k(void *bitgen, int64_t n, int32_t k, int64_t *host_out);

/* random.sample(range(n), k=n) of Python's `random` module (the minibatch order of
 * pfrl/agents/ppo.py:247-257, Lib/random.py sample() + _randbelow_with_getrandbits) on the module's
 * own MT19937 state: state625 = the 624 words + index of random.getstate()[1], advanced in place
 * (write it back with random.setstate).  host_out = int64[n]. */
int pfrl_pyrandom_permutation(uint32_t *state625, int64_t n, int64_t *host_out);

/* n_envs x select_action_epsilon_greedily with random_action_func = np.random.randint(n_actions):
 * host_choice[i] = the random action, or -1 where the draw chose the greedy action. */
int pfrl_plan_eps_greedy(void *bitgen, int64_t n_envs, double epsilon, int64_t n_actions,
                         int32_t *host_choice);

/* The per-env loop of DQN._batch_observe_train (pfrl/agents/dqn.py:516-549) -- and the same loop
 * of the vector-observation agents (pfrl/agents/soft_actor_critic.py:354-374, td3.py:283-303,
 * ddpg.py:207-227; k = 1 frame per observation, the float action rows travel behind the planner's
 * part of the block) -- for m envs of one batched step with a uniform one-step ReplayBuffer: m appends (transition + entry rows, host
 * mirrors updated, RandomAccessQueue head advanced) and every index set the loop draws between
 * them.  counters = {n_trans, n_entries, head} in / out; t0 = agent.t before the first env.
 * Everything the device needs goes into ONE pinned block (host_block; offs[0..8] = byte
 * offsets of t_slots i32[m], state_ref i32[m][k], next_ref i32[m][k], reward f64[m], terminal
 * u8[m], e_slots i32[m], e_tids i32[m], e_len i32[m], sampled entry slots i32[U][B];
 * offs[9] = bytes used).  Returns U = number of index sets drawn (>= 0) or a negative
 * PFRL_PLAN_* / PFRL_ERR_ARG code. */
int64_t pfrl_plan_dqn_range(const pfrl_host_store_t *st, void *bitgen, int64_t m,
                            const int32_t *s_refs, const int64_t *s_min_seq, const int32_t *n_refs,
                            const int64_t *n_min_seq, const double *reward, const uint8_t *done,
                            int64_t t0, int64_t replay_start, int64_t update_interval,
                            int32_t n_times_update, int32_t B, int64_t oldest_live_fseq,
                            int64_t *counters, uint8_t *host_block, int64_t block_bytes,
                            int64_t *offs);

/* ------------------------------------------------------------------------
 * Measurement support (bench.py roofline): time every pfrl_batch_experiences
 * (kind 0, units = sampled entries), pfrl_batch_states_u8[_nhwc4] (kind 1, units =
 * frame refs), pfrl_gae_scan (kind 2, units = T * N), pfrl_adv_stats (kind 3) and
 * pfrl_batch_states_u8_raw_nhwc4 (kind 4, units = frame refs) launch with a hipEvent
 * pair attached to the dispatch, on its own stream.  pfrl_profile_collect synchronises, returns durations in
 * microseconds, the unit count and the kind of each timed launch. */
int pfrl_profile_enable(int on);
int64_t pfrl_profile_collect(double *host_out_us, int64_t *host_out_units, int32_t *host_out_kind,
                             int64_t cap);

/* y = act(x W^T + b) of a factorised NoisyNet layer (pfrl/nn/noisy_linear.py:56-70: W = mu.W +
 * sigma.W * outer(f(r_out), f(r_in)), b = mu.b + sigma.b * f(r_out), f(r) = sign(r) sqrt|r|, r = the
 * layer's in + out unit Gaussians, inputs first) without the perturbed weights ever being written:
 * the weight operand is formed chunk by chunk inside the forward kernel.  Bit-identical to
 * pfrl_noisy_weights_fwd followed by pfrl_linear_fwd (same roundings in W, same tile program).
 * Minibatch-sized problems, in_features % 32 == 0, 16-byte aligned operands.  splits > 1: partials
 * [splits][M][N] for pfrl_splitk_reduce_noisy (mu_b / sigma_b may then be NULL). */
int pfrl_linear_noisy_fwd(const float *x, const float *mu_w, const float *sigma_w, const float *mu_b,
                          const float *sigma_b, const float *r, float *y, int32_t M, int32_t K, int32_t N,
                          int32_t relu, int32_t splits, void *stream);

/* Two NoisyNet layers in one launch: the advantage and value streams of the distributional dueling
 * head (pfrl/q_functions/dueling_dqn.py:93-118).  Problem t = 0, 1: y[t] [M, N[t]] = act(rows of x[t]
 * (K floats at a stride of x_row_stride floats: the halves of the hidden activations in place)
 * times the perturbed weights of layer t).  Arrays of two pointers / sizes.  Both layers must be
 * narrow-output minibatch problems (N % 32 != 0); bit-identical to two pfrl_linear_noisy_fwd calls. */
int pfrl_linear_noisy_fwd_pair(const float *const *x, int32_t x_row_stride, const float *const *mu_w,
                               const float *const *sigma_w, const float *const *mu_b,
                               const float *const *sigma_b, const float *const *r, float *const *y,
                               int32_t M, int32_t K, const int32_t *N, int32_t relu, void *stream);

/* pfrl_splitk_reduce whose bias may be a NoisyNet layer's: bias[t] + bias_sigma[t] * f(bias_noise[t])
 * (bias_noise[t] = the out_features Gaussians of the layer's draw, i.e. r + in_features).
 * bias_sigma / bias_noise NULL (or NULL entries) = plain biases. */
int pfrl_splitk_reduce_noisy(int32_t n_tasks, const float *const *part, float *const *out,
                             const float *const *bias, const float *const *bias_sigma,
                             const float *const *bias_noise, const int64_t *stride, const int32_t *n,
                             const int32_t *splits, const int32_t *ncol, const int32_t *relu, void *stream);

/* torch.randn on the device generator, restated (pfrl/nn/noisy_linear.py:52-60 draws one
 * torch.normal(0, 1, size = in + out) per NoisyNet layer and forward pass: nine launches per Rainbow
 * update).  n_calls <= 16 draws in ONE launch: call i writes numel[i] floats at out + out_offsets[i],
 * bit for bit what torch.randn(numel[i]) writes when the CUDA generator holds (seed, base_offset +
 * offsets[i]) -- offsets[] relative to the first call, so that a caller prepares the arrays once;
 * grids[i] = the blocks of 256 threads torch launches for that size (min(CUs * (max threads per CU /
 * 256), ceil(numel / 256))).  variant: 0 / 1 = Box-Muller without / with a * b + c contracted to an
 * fma (what torch's own build does is pinned by tests/test_philox.py).  The caller advances the
 * generator's offset exactly as the torch calls would. */
int pfrl_philox_normal(uint64_t seed, uint64_t base_offset, int32_t n_calls, const uint64_t *offsets,
                       const int64_t *numel, const int64_t *out_offsets, const int32_t *grids, float *out,
                       int32_t variant, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PFRL_AMD_H */
