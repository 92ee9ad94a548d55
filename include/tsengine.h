/*
 * tsengine.h -- C ABI of libtsengine.so, the MI355X (gfx950) update-step engine that drops in
 * behind the Batch -> learn() hot path of thu-ml/tianshou 2.0.1.
 *
 * Boundary rules
 *   - extern "C", plain pointers + int64 sizes + scalar hyper-parameters + a stream handle.
 *     No torch / numpy types.  Every data pointer is a DEVICE pointer unless the parameter
 *     name starts with `h_` (host).  The caller owns every buffer; the library owns only the
 *     opaque ts_workspace it hands out.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  No entry point
 *     synchronises the device; results are ordered on `stream`.
 *   - Return value: TS_OK (0) or a negative TS_ERR_* code; ts_last_error() gives the message
 *     for the calling thread.  The Python wrapper maps TS_ERR_SHAPE to ValueError (mirrors
 *     algorithm_base.py:757-758) and the rest to RuntimeError.
 *   - Re-entrant per stream; no global mutable state except the thread-local error string.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * checkout, thu-ml/tianshou @ 2.0.1).
 */
#ifndef TSENGINE_H
#define TSENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TS_OK 0
#define TS_ERR_INVALID_ARG (-1) /* null pointer, negative size, unsupported flag value      */
#define TS_ERR_SHAPE (-2)       /* size mismatch between arguments -> ValueError            */
#define TS_ERR_HIP (-3)         /* a HIP runtime call / kernel launch failed                 */
#define TS_ERR_UNSUPPORTED (-4) /* valid request the engine does not cover (caller decides)  */
#define TS_ERR_WORKSPACE (-5)   /* workspace too small / missing                             */

typedef void* ts_stream_t; /* hipStream_t */
typedef struct ts_workspace ts_workspace;

const char* ts_version(void);
const char* ts_last_error(void);

/* Opaque scratch (tile aggregates, partial gradients, sum-tree winner table).  Grows on
 * demand up to `max_bytes` (0 = no limit); never shrinks.  One workspace per stream. */
int ts_workspace_create(ts_workspace** out, int device, size_t max_bytes);
int ts_workspace_destroy(ts_workspace* ws);

/* The workspace's two internal side streams (hipStreamNonBlocking, created on first use; which = 0 or 1): the streams on
 * which entry points called with `ws` place work that is independent of the caller's stream (the lagged network's pass of
 * ts_dqn_target_q_fused, the weight gradients of a backward chain).  A caller with independent work of its own (the
 * forward pass that ts_dqn_forward_cache runs ahead of an update) can put it there instead of on one more stream of its
 * own: the runtime multiplexes streams onto four hardware queues, and a fifth stream shares a queue -- and its order --
 * with another one. */
int ts_workspace_side_stream(ts_workspace* ws, int which, ts_stream_t* stream_out);

/* Per-kernel timing for bench.py's roofline figure: between ts_profile_begin and
 * ts_profile_end every kernel launched through `ws` is bracketed by a HIP event pair recorded
 * on the launch stream.  ts_profile_end waits for the events and returns, per kernel kind,
 * the summed duration (ms) and the launch count.  Kinds: */
#define TS_KIND_PPO_STEP 0   /* fused forward/backward (ppo_step_kernel)      */
#define TS_KIND_PPO_REDUCE 1 /* slab reduction                                */
#define TS_KIND_PPO_ADAM 2   /* clip + Adam                                   */
#define TS_KIND_PPO_INFER 3  /* value / log-prob inference                    */
#define TS_KIND_GAE_MAPS 4   /* GAE pass 1 (tile maps)                        */
#define TS_KIND_GAE_APPLY 5  /* GAE pass 2 (carry + outputs)                  */
#define TS_KIND_CONV_FWD 6   /* conv / linear forward (implicit GEMM)           */
#define TS_KIND_CONV_WGRAD 7 /* conv / linear weight gradient                  */
#define TS_KIND_CONV_DGRAD 8 /* conv / linear input gradient                   */
#define TS_N_KINDS 9
int ts_profile_begin(ts_workspace* ws);
int ts_profile_end(ts_workspace* ws, double* h_ms_by_kind, int64_t* h_count_by_kind, int n_kinds);

/* ---------------------------------------------------------------------------------------------
 * Returns / advantages
 * ------------------------------------------------------------------------------------------- */

/* Fused Algorithm.compute_episodic_return (tianshou/algorithm/algorithm_base.py:653-719) with
 * its njit kernel _gae (:1085-1140) and the return-scaling arithmetic around it
 * (tianshou/algorithm/modelfree/a2c.py:134-148):
 *     vs  = v_s  * v_scale;  vs_ = v_s_next * v_scale * !terminated          (:711, a2c.py:135-136)
 *     end = terminated | truncated | (position in cut_pos)                    (:714-715)
 *     adv_i = (rew_i + gamma*vs__i - vs_i) + (1-end_i)*gamma*lambda*adv_{i+1} (reverse scan)
 *     ret_i = adv_i + vs_i ;  returns_out = ret_i / ret_div                   (:717, a2c.py:146)
 * All inputs are the *batch* arrays in batch order (= buffer arrays gathered at `indices`).
 * `cut_pos` lists the batch positions p with indices[p] in buffer.unfinished_index()
 * (any order, may be NULL when n_cut == 0); when `d_n_cut` (device int64, nullable) is given
 * the kernel uses min(*d_n_cut, n_cut) entries, so the count produced by ts_isin_positions
 * never has to visit the host.  Arithmetic is float64 inside (as numba's), the
 * outputs are float32 like the reference's to_torch_as casts (a2c.py:151-152).  Optional
 * float64 outputs (`adv64`, `ret64`, unnormalised returns) serve parity tests.
 * `ret_partials` (nullable, double[2 * ts_gae_num_tiles(n)]) receives per-tile (sum, sum of
 * squares) of the unnormalised returns for RunningMeanStd.update (utils/statistics.py:99-114).
 * rew_dtype: 0 = float32, 1 = float64 (the reference stores rew as float64, buffer_base.py:492). */
int64_t ts_gae_num_tiles(int64_t n);
int ts_gae_scan(ts_workspace* ws, const float* v_s, const float* v_s_next, const void* rew,
                int rew_dtype, const uint8_t* terminated, const uint8_t* truncated,
                const int64_t* cut_pos, int64_t n_cut, const int64_t* d_n_cut, int64_t n,
                double gamma, double gae_lambda, double v_scale, double ret_div, float* adv_out,
                float* returns_out, double* adv64, double* ret64, double* ret_partials,
                ts_stream_t stream);

/* The scan is a single launch: workgroups hand their tile maps to each other through a small
 * persistent area of the workspace (bounded spins, launch-epoch tags).  ts_gae_check reports
 * (after synchronising the stream) whether any spin ever ran out; 0 = all scans valid.
 * Setting TS_GAE_TWO_PASS=1 selects the two-launch variant without cross-workgroup hand-off. */
int ts_gae_check(ts_workspace* ws, int* h_error_out, ts_stream_t stream);

/* Positions of unfinished slots inside a batch: for the general (non-identity) `indices`
 * of compute_episodic_return, algorithm_base.py:715 `np.isin(indices, unfinished_index())`.
 * Writes the matching batch positions to cut_pos_out (capacity >= n_unfinished * dup, see
 * `capacity`) in unspecified order and their count to *n_cut_out (device int64). */
int ts_isin_positions(const int64_t* indices, int64_t n, const int64_t* unfinished,
                      int64_t n_unfinished, int64_t* cut_pos_out, int64_t capacity,
                      int64_t* n_cut_out, ts_stream_t stream);

/* _nstep_return (tianshou/algorithm/algorithm_base.py:1160-1222), same arguments:
 * rew_B float64[B], end_flag_B u8[B], target_q_IA float32[I,A], stacked_indices_NI int64[N,I]
 * -> out float32[I,A] (the reference returns float64 and casts at :811) and optionally the
 * float64 values (`out64`, nullable).  Bit-exact float64 arithmetic (no FMA contraction). */
int ts_nstep_return(const double* rew_B, const uint8_t* end_flag_B, const float* target_q_IA,
                    const int64_t* stacked_indices_NI, int64_t I, int64_t A, int64_t n_step,
                    int64_t B, double gamma, float* out, double* out64, ts_stream_t stream);

/* indices_after_n_steps of Algorithm.compute_nstep_return (algorithm_base.py:772-791):
 * applies ReplayBufferManager.next (n_step - 1) times.  Optionally also writes the whole
 * stack int64[n_step, I] (`stacked_out`, nullable). */
int ts_nstep_indices(const int64_t* indices, int64_t I, int64_t n_step, const int64_t* offset,
                     int64_t E, const uint8_t* done, const int64_t* last_index,
                     const int64_t* lengths, int64_t* after_out, int64_t* stacked_out,
                     ts_stream_t stream);

/* Fused remainder of compute_nstep_return (algorithm_base.py:798-811): walks next() itself,
 * applies value_mask (~terminated[idx_after_n], :798), builds end_flag = done | unfinished
 * (:799-800) on the fly, evaluates _nstep_return.  No [N,I] index matrix, no O(B) copy.
 * n_step <= 32. */
int ts_nstep_return_fused(const int64_t* indices, int64_t I, int64_t n_step,
                          const int64_t* offset, int64_t E, const uint8_t* done,
                          const uint8_t* terminated, const int64_t* last_index,
                          const int64_t* lengths, const double* rew_B, const float* target_q_IA,
                          int64_t A, double gamma, float* out, double* out64,
                          ts_stream_t stream);

/* The part of ts_nstep_return_fused that does not depend on target_q (the index walk, the reward sums, the masks): per
 * index mask = value_mask(idx_after_n) as float32 (:798), gpow = gamma^n_eff and mc = the discounted reward sum of
 * _nstep_return (:1201-1222), both float64, such that returns = float(double(target_q * mask) * gpow + mc).  It can run
 * before (beside) the target network's passes; ts_dqn_target_returns finishes the returns. */
int ts_nstep_coefficients(const int64_t* indices, int64_t I, int64_t n_step, const int64_t* offset, int64_t E,
                          const uint8_t* done_B, const uint8_t* terminated_B, const int64_t* last_index,
                          const int64_t* lengths, const double* rew_B, double gamma, float* mask_out, double* gpow_out,
                          double* mc_out, ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Replay-buffer index math (bit-exact integers)
 * ------------------------------------------------------------------------------------------- */

/* _next_index / _prev_index (tianshou/data/buffer/manager.py:339-363 / :311-336), same
 * arguments.  O(I log E) instead of the reference's O(E * I) masked passes. */
int ts_next_index(const int64_t* index, int64_t I, const int64_t* offset, int64_t E,
                  const uint8_t* done, const int64_t* last_index, const int64_t* lengths,
                  int64_t* out, ts_stream_t stream);
int ts_prev_index(const int64_t* index, int64_t I, const int64_t* offset, int64_t E,
                  const uint8_t* done, const int64_t* last_index, const int64_t* lengths,
                  int64_t* out, ts_stream_t stream);

/* ReplayBufferManager.unfinished_index (manager.py:85-91, buffer_base.py:314-317):
 * ascending list of last_index[e] with lengths[e] > 0 and !done[last_index[e]].
 * out int64[E] (capacity E), *n_out device int64. */
int ts_unfinished_index(const int64_t* offset, int64_t E, const uint8_t* done,
                        const int64_t* last_index, const int64_t* lengths, int64_t* out,
                        int64_t* n_out, ts_stream_t stream);

/* ReplayBufferManager.sample_indices(0) (manager.py:216-234, buffer_base.py:518-525):
 * per sub-buffer [insertion, len) ++ [0, insertion), + offset.  `total` = sum(lengths)
 * (known to the host, which owns the manager state). out int64[total]. */
int ts_sample_indices_all(ts_workspace* ws, const int64_t* offset, int64_t E,
                          const int64_t* lengths, const int64_t* insertion, int64_t total,
                          int64_t* out, ts_stream_t stream);

/* ReplayBufferManager.sample_indices(batch_size > 0) (manager.py:216-234 + ReplayBuffer.sample_indices
 * buffer_base.py:503-517, stack_num == 1): sub-buffer by RandomState.choice(E, bs, p = lengths / sum) ==
 * cdf.searchsorted(u_buffer, side="right"), then `sample_num` uniform slots inside each chosen sub-buffer, output
 * concatenated in sub-buffer order.  The random draws are INPUTS so that a seeded reference run is reproduced
 * bit-exactly: u_buffer f64[bs] = the uniforms `choice` consumes; within_i int64[bs] = the children's randint draws
 * in output order - or within_u f64[bs] uniforms (slot = floor(u * len), device-RNG path); exactly one of the two.
 * *err_flag (device int, caller zeroes it): 1 = empty buffer, 2 = a within_i draw outside its sub-buffer.  E <= 4096. */
int ts_sample_indices_random(const int64_t* offset, int64_t E, const int64_t* lengths, const double* u_buffer,
                             const int64_t* within_i, const double* within_u, int64_t batch_size, int64_t* out,
                             int* err_flag, ts_stream_t stream);
/* The same sampler drawing its own uniforms: Philox-4x32-10 keyed by `seed`, counter (draw index, `counter`) -- the generator
 * of ts_normal_fill; 53-bit doubles in [0, 1).  One launch instead of a generator launch + the sampler; the same
 * (seed, counter) always gives the same indices.  Not the reference's RandomState stream (pass its draws to
 * ts_sample_indices_random for that). */
int ts_sample_indices_seeded(const int64_t* offset, int64_t E, const int64_t* lengths, uint64_t seed, uint64_t counter,
                             int64_t batch_size, int64_t* out, int* err_flag, ts_stream_t stream);

/* Device-side stand-in for the np.random.permutation(len(batch)) that Batch.split draws per repeat
 * (tianshou/data/batch.py:1209): a keyed bijection of [0, n) (4-round Feistel + cycle walking),
 * one thread per slot, ~5 us for 2^20 entries (a sort-based randperm costs ~250 us).  Not the
 * NumPy stream: for bit-for-bit reproduction of a seeded reference run pass the host permutation
 * to ts_ppo_update instead. */
int ts_random_permutation(int64_t* out, int64_t n, uint64_t seed, ts_stream_t stream);

/* eps ~ N(0, 1) for the reparameterised samples of the continuous policies (torch.distributions.Normal.rsample as used by
 * SAC / REDQ `forward`, tianshou/algorithm/modelfree/sac.py:228-236; DDPG / TD3 exploration and target noise): out
 * float32[n] from a counter-based Philox-4x32-10 stream keyed by (seed, offset) + Box-Muller; `offset` = a per-call
 * counter (e.g. the update number) so that successive calls draw fresh numbers.  Not torch's generator stream: to
 * reproduce a seeded reference run pass its noise to the update entry points instead (they all take it as an input). */
int ts_normal_fill(float* out, int64_t n, uint64_t seed, uint64_t offset, ts_stream_t stream);
/* np.random.rand(n) of the engine's own stream (the draws of PrioritizedReplayBuffer.sample_indices, prio.py:65, when the
 * caller does not supply the reference's): out float64[n], out[i] = the 53-bit double in [0, 1) of Philox-4x32-10 keyed by
 * `seed` at counter (i, `counter`) -- the draw ts_sample_indices_seeded makes for its buffer choice. */
int ts_uniform_fill_f64(double* out, int64_t n, uint64_t seed, uint64_t counter, ts_stream_t stream);

/* ReplayBuffer.__getitem__ row gather (buffer_base.py:605-649): out[i,:] = src[index[i],:]
 * for a row of `row_bytes` bytes (any dtype).  16-byte vector path when row_bytes % 16 == 0
 * and both bases are 16-byte aligned. */
int ts_gather_rows(const void* src, int64_t n_rows_src, int64_t row_bytes, const int64_t* index,
                   int64_t I, void* out, ts_stream_t stream);

/* The same for up to 8 keys of a batch at once -- Batch.__getitem__ (data/batch.py:714-738) gathers EVERY key at the same
 * indices; one launch instead of one per key.  h_src / h_out / h_row_bytes: HOST arrays of n_keys device pointers / row sizes
 * (multiples of 4 bytes, 4-byte aligned); every source has n_rows_src rows. */
int ts_gather_rows_multi(int64_t n_keys, const void* const* h_src, const int64_t* h_row_bytes, int64_t n_rows_src,
                         const int64_t* index, int64_t I, void* const* h_out, ts_stream_t stream);

/* Write side: ReplayBufferManager.add (manager.py:131-198) with
 * ReplayBuffer._update_state_pre_add (buffer_base.py:360-418) for K transitions addressed to DISTINCT
 * sub-buffers buffer_ids[k] (NULL = 0..K-1), entirely on the device.  State per sub-buffer, all [E], updated in
 * place: insertion (next write slot, relative to offset[e]), lengths, last_index (global), ep_return (float64),
 * ep_len, ep_start (relative).  Writes rew / terminated / truncated / done at the new slots of the [B] columns,
 * scatters the rows of up to 8 further keys (obs, act, obs_next, ...: h_keys is a HOST array of device
 * pointers), and returns per entry what add() returns (manager.py:193-198): global index, episode return and
 * length (0 unless the episode ended), global episode start index.  Bit-exact, float64 returns included. */
typedef struct ts_scatter_key { void* dst; const void* src; int64_t row_bytes; } ts_scatter_key;
int ts_buffer_add(const int64_t* buffer_ids, int64_t K, const double* rew, const uint8_t* terminated,
                  const uint8_t* truncated, const int64_t* offset, int64_t E, int64_t* insertion, int64_t* lengths,
                  int64_t* last_index, double* ep_return, int64_t* ep_len, int64_t* ep_start, double* rew_B,
                  uint8_t* terminated_B, uint8_t* truncated_B, uint8_t* done_B, const ts_scatter_key* h_keys,
                  int n_keys, int64_t* index_out, double* ep_return_out, int64_t* ep_len_out, int64_t* ep_start_out,
                  ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sum tree / prioritized replay
 * ------------------------------------------------------------------------------------------- */

/* SegmentTree._setitem (tianshou/data/utils/segtree.py:95-101): tree float64[2*bound],
 * index int64[K] (already + bound), value float64[K] (value_dtype 1) or float32[K] (0).
 * Duplicate leaves: the later entry wins, as NumPy fancy assignment does.  */
int ts_segtree_setitem(ts_workspace* ws, double* tree, int64_t bound, const int64_t* index,
                       const void* value, int value_dtype, int64_t K, ts_stream_t stream);

/* SegmentTree._reduce (segtree.py:104-116) -> *out (device double). */
int ts_segtree_reduce(const double* tree, int64_t start, int64_t end, double* out,
                      ts_stream_t stream);

/* SegmentTree._get_prefix_sum_idx (segtree.py:119-134): value float64[K] is mutated in
 * place exactly as the reference does (:131); out int64[K]. */
int ts_segtree_prefix_sum_idx(double* value, int64_t K, int64_t bound, const double* sums,
                              int64_t* out, ts_stream_t stream);

/* PrioritizedReplayBuffer.sample_indices + get_weight + __getitem__ normalisation
 * (tianshou/data/buffer/prio.py:63-79, 104-106) in one launch: u float64[K] are the
 * host-supplied np.random.rand(K) draws (:65).  idx_out int64[K], weight_out float64[K]
 * = (tree[idx+bound]/min_prio)^-beta, divided by its max when weight_norm; min_prio is read
 * from prio_minmax[1] (device double[2] = {max_prio, min_prio}, prio.py:36). */
int ts_per_sample(ts_workspace* ws, const double* tree, int64_t bound, const double* u,
                  int64_t K, const double* prio_minmax, double beta, int weight_norm,
                  int64_t* idx_out, double* weight_out, ts_stream_t stream);

/* PrioritizedReplayBuffer.update_weight (prio.py:81-90): new_weight float32[K] (TD errors);
 * tree[index+bound] = (|w| + eps)^alpha (float32 math like NumPy's), then sum-tree repair;
 * prio_minmax (device double[2] = {max_prio, min_prio}) updated with |w| + eps. */
int ts_per_update_weight(ts_workspace* ws, double* tree, int64_t bound, const int64_t* index,
                         const float* new_weight, int64_t K, double alpha, double* prio_minmax,
                         ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PPO / A2C actor-critic MLP  (obs -> 64 -> 64 -> {mu[act_dim], V}), tanh, fp32 MFMA
 * ------------------------------------------------------------------------------------------- */

/* Flat fp32 parameter vector layout (P = ts_ppo_param_count(obs_dim, act_dim)):
 *   actor : W1[64,obs] b1[64] W2[64,64] b2[64] Wmu[act,64] bmu[act] sigma_param[act]
 *   critic: W1[64,obs] b1[64] W2[64,64] b2[64] Wv[64] bv[1]
 * (row-major, torch nn.Linear convention weight[out,in]).  Reference modules:
 * Net/MLP tianshou/utils/net/common.py:90-178,246-369; ContinuousActorProbabilistic
 * tianshou/utils/net/continuous.py:172-238 (unbounded, state-independent sigma);
 * ContinuousCritic continuous.py:99-169. */
int64_t ts_ppo_param_count(int64_t obs_dim, int64_t act_dim);

/* The collector's inference step: ProbabilisticActorPolicy.forward (reinforce.py:167-192) =
 * mu(obs), act = mu + exp(sigma_param) * noise (dist.sample(); noise NULL = dist.mode, deterministic_eval), and
 * Algorithm.map_action (algorithm_base.py:254-287): bound_method 0 none / 1 "clip" / 2 "tanh", then scaling to
 * [low, high] (device float32[act_dim] each; both NULL = action_scaling False).  act_out float32[n, act_dim] is
 * what goes into the buffer, mapped_out (nullable) what goes to the env. */
int ts_ppo_policy_forward(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim, const float* obs,
                          const float* noise, int64_t n, int bound_method, const float* low, const float* high,
                          float* act_out, float* mapped_out, ts_stream_t stream);
/* ... with the tanh bound of a bounded actor (max_action > 0) and mu_out (nullable) = the distribution's mean. */
int ts_ppo_policy_forward_bounded(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim, double max_action,
                                  const float* obs, const float* noise, int64_t n, int bound_method, const float* low,
                                  const float* high, float* act_out, float* mapped_out, float* mu_out, ts_stream_t stream);
/* dist.sample() + Algorithm.map_action (algorithm_base.py:254-287) for a mean that is already on the device (the per-layer
 * engine's ts_ppo_net_infer mu_out, or a squashed SAC action with noise = NULL): act_out = mu + exp(log_sigma) * noise
 * (noise NULL: act_out = mu), mapped_out (nullable) = clip / tanh bounding (bound_method 1 / 2) and scaling to
 * [low, high] (nullable pair).  log_sigma float32[act_dim] (needed with noise). */
int ts_gauss_sample_map(const float* mu, const float* noise, const float* log_sigma, int64_t n, int64_t act_dim,
                        int bound_method, const float* low, const float* high, float* act_out, float* mapped_out,
                        ts_stream_t stream);

typedef struct ts_ppo_hparams {
    double eps_clip;      /* ppo.py:140 */
    double dual_clip;     /* ppo.py:141; <= 0 means None */
    double vf_coef;       /* a2c.py / ppo.py:211 */
    double ent_coef;      /* ppo.py:211 */
    double max_grad_norm; /* algorithm_base.py:497-499; <= 0 means None */
    double lr;            /* optim.py:89-110 */
    double beta1, beta2;  /* Adam betas */
    double adam_eps;      /* Adam eps */
    int32_t value_clip;   /* ppo.py:199-206 */
    int32_t adv_norm;     /* ppo.py:184-186 */
    int32_t algo;         /* 0 = PPO clipped surrogate (ppo.py:187-196);
                             1 = A2C policy gradient -(logp * adv).mean() + plain MSE value loss
                                 (modelfree/a2c.py:262-273; eps/dual/value clip, adv_norm, logp_old unused) */
    int32_t nets;         /* 0 (or 3): actor and critic.  1: the actor's half of every step only, 2: the critic's -- the other
                             network's gradient and loss sum are zeros (it still takes its Adam step on that zero gradient).
                             For callers whose other network is a stand-in: Reinforce (reinforce.py:371-380 = A2C's actor
                             loss with adv := returns) and the critic iterations of NPG / TRPO (npg.py:142-150 = A2C steps
                             with a zero advantage).  Honoured by ts_ppo_update / ts_ppo_grad (obs_dim <= 31 kernels) */
    /* -- round 6 (appended: a zero-initialised tail is torch.optim.Adam without weight decay on an unbounded actor) -- */
    int32_t optimizer;    /* TS_OPT_ADAM (optim.py:89-110) / TS_OPT_RMSPROP (optim.py:113-140; examples/mujoco/mujoco_a2c.py:117):
                             torch.optim.RMSprop's single-tensor arithmetic, square_avg in the `adam_v` vector, the momentum
                             buffer (rms_momentum > 0) or grad_avg (rms_centered) in `adam_m`; lr, adam_eps are shared */
    int32_t rms_centered; /* RMSprop(centered=True); not together with rms_momentum > 0 (one auxiliary vector) */
    double weight_decay;  /* both optimizers: grad += weight_decay * param after clipping (torch: inside optimizer.step) */
    double rms_alpha;     /* RMSprop smoothing constant */
    double rms_momentum;  /* RMSprop momentum (0: none) */
    double max_action;    /* > 0: ContinuousActorProbabilistic(unbounded=False), the constructor default
                             (utils/net/continuous.py:194, 230-231): mu = max_action * tanh(Linear(h)); 0: unbounded */
} ts_ppo_hparams;
#define TS_OPT_ADAM 0
#define TS_OPT_RMSPROP 1

/* No-grad inference passes of PPO._preprocess_batch / _add_returns_and_advantages
 * (a2c.py:122-129, ppo.py:157-161), whole batch in one launch instead of max_batchsize
 * chunks: v_out[i] = V(obs_i) (nullable), logp_out[i] = log N(act_i; mu(obs_i), sigma)
 * (nullable; needs act).  obs float32[n, obs_dim], act float32[n, act_dim]. */
int ts_ppo_infer(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim, const float* obs,
                 const float* act, int64_t n, float* v_out, float* logp_out,
                 ts_stream_t stream);
/* The same for ContinuousActorProbabilistic(unbounded=False) -- the constructor default (utils/net/continuous.py:194,
 * 230-231): mu = max_action * tanh(Linear(h)) (max_action = 0: unbounded); mu_out (nullable) float32[n, act_dim] receives
 * `logits[0]` of ProbabilisticActorPolicy.forward (reinforce.py:183-190). */
int ts_ppo_infer_bounded(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim, double max_action,
                         const float* obs, const float* act, int64_t n, float* v_out, float* logp_out, float* mu_out,
                         ts_stream_t stream);

/* PPO._update_with_batch (tianshou/algorithm/modelfree/ppo.py:164-224) + Optimizer.step
 * (algorithm_base.py:484-500): all `n_steps` minibatch gradient steps of one update() are
 * enqueued by this one call.  Minibatch s uses rows perm[mb_offset[s] .. mb_offset[s+1])
 * of the batch arrays (host-supplied np.random.permutation, batch.py:1209; mb_offset is a
 * HOST array of n_steps+1 int64).  Per step: fused forward + clipped-surrogate / value /
 * entropy loss + backward (fp32 MFMA), global-norm clip, Adam.  params / adam_m / adam_v
 * (float32[P]) are updated in place; adam_step0 = number of Adam steps already taken.
 * losses_out float32[n_steps,4] = (loss, clip_loss, vf_loss, ent_loss) per step (ppo.py:213-216).
 * grads_out (nullable float32[P]) receives the unclipped gradient of the LAST step (tests). */
int ts_ppo_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v,
                  int64_t adam_step0, int64_t obs_dim, int64_t act_dim, const float* obs,
                  const float* act, const float* adv, const float* returns,
                  const float* logp_old, const float* v_s, int64_t n, const int64_t* perm,
                  const int64_t* h_mb_offset, int64_t n_steps, const ts_ppo_hparams* hp,
                  float* losses_out, float* grads_out, ts_stream_t stream);

/* Packed minibatch records.  The step kernel gathers minibatch rows by permutation; to make every
 * touched 128-byte line fully useful the batch is first packed into one 16-byte-aligned record per
 * sample:  [obs(obs_dim) | act(act_dim) | adv | returns | logp_old | v_s | 0-pad] , W floats with
 * W = ts_ppo_record_width(obs_dim, act_dim) (multiple of 4).  ts_ppo_update packs internally;
 * the data-parallel path packs once per update() and passes the records to ts_ppo_grad. */
int64_t ts_ppo_record_width(int64_t obs_dim, int64_t act_dim);
int ts_ppo_pack_batch(const float* obs, const float* act, const float* adv, const float* returns,
                      const float* logp_old, const float* v_s, int64_t n, int64_t obs_dim,
                      int64_t act_dim, float* rec_out, ts_stream_t stream);

/* Data-parallel variant, split around the gradient all-reduce (RCCL, issued by the caller on
 * the same stream between the two calls; the reference has no equivalent - its only multi-GPU
 * path is nn.DataParallel, tianshou/utils/net/common.py:473-515):
 *   ts_ppo_grad : forward/backward of ONE local minibatch shard (records rec[perm_rows[0..n_rows)],
 *                 or the first n_rows records when perm_rows is NULL) -> grad_out float32[P] = sum
 *                 over the local rows / global_batch, loss_parts_out float32[4] = (loss, clip, vf,
 *                 ent) with clip / vf as local sums / global_batch (all-reduce-sum them too and
 *                 recompute loss = clip + vf_coef*vf - ent_coef*ent).  adv_stats (device float32[2]
 *                 = {mean, std} of the GLOBAL minibatch) is required when hp->adv_norm.
 *                 loss_parts_out[0] is written as 0: the caller composes the loss after the all-reduce.
 *   ts_ppo_apply: clip by global norm + Adam step number `adam_step` (1-based) using the all-reduced gradient;
 *                 `ws` (nullable) = the workspace ts_ppo_grad runs on: its cached weight images are refreshed in
 *                 place (otherwise call ts_ppo_invalidate_image after changing the parameters by other means). */
int ts_ppo_grad(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t act_dim,
                const float* rec, int64_t n, const int64_t* perm_rows, int64_t n_rows,
                int64_t global_batch, const float* adv_stats, const ts_ppo_hparams* hp,
                float* grad_out, float* loss_parts_out, ts_stream_t stream);
int ts_ppo_apply(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                 int64_t act_dim, const float* grad, const ts_ppo_hparams* hp, ts_stream_t stream);
int ts_ppo_invalidate_image(ts_workspace* ws);

/* Which fused step kernel ts_ppo_update / ts_ppo_grad launch for a minibatch of `n_rows` rows and its geometry (reporting
 * only: bench.py's roofline line and strong-scaling projection).  variant_out: 0 = 128-sample workgroups with an LDS weight
 * image (ppo_step2_kernel), 1 / 2 = 32-sample tiles split by features over 4 waves, one network per persistent workgroup, in
 * the 128- / 168-register build (ppo_stepq_kernel / ppo_stepq2_kernel); grid_out = workgroups, slabs_out = gradient slabs
 * the reduction kernel reads.  Any out pointer may be NULL. */
int ts_ppo_step_plan(int64_t obs_dim, int64_t act_dim, int64_t n_rows, int32_t nets, int32_t* variant_out,
                     int32_t* grid_out, int32_t* slabs_out);

/* ---------------------------------------------------------------------------------------------
 * Target networks
 * ------------------------------------------------------------------------------------------- */

/* polyak_parameter_update / full_parameter_update over a flat parameter vector
 * (tianshou/utils/lagged_network.py:8-18, 81-87): tgt = tau * src + (1 - tau) * tgt, same three
 * float32 roundings as torch; tau == 1 is the hard copy DQN uses every target_update_freq steps
 * (modelfree/dqn.py:283-285). */
int ts_polyak_update(float* tgt, const float* src, int64_t n, double tau, ts_stream_t stream);

/* clip_grad_norm_ + Adam over a flat vector: Optimizer.step (algorithm_base.py:484-500) with
 * torch.optim.Adam's single-tensor arithmetic (optim.py:89-110).  `step` is the 1-based Adam step
 * of this call; max_grad_norm <= 0 disables clipping (needs `ws` otherwise). */
int ts_adam_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, const float* grad, int64_t n,
                 int64_t step, double lr, double beta1, double beta2, double eps, double max_grad_norm,
                 ts_stream_t stream);

/* Optimizer.step (algorithm_base.py:484-500) for the other optimizer factories of tianshou/algorithm/optim.py: `kind`
 * TS_OPT_ADAM = torch.optim.Adam incl. weight_decay (optim.py:89-110), TS_OPT_RMSPROP = torch.optim.RMSprop (optim.py:113-140:
 * alpha, eps, weight_decay, momentum, centered -- the optimizer of examples/mujoco/mujoco_a2c.py:117).  Single-tensor
 * arithmetic of torch.optim, applied after clip_grad_norm_(max_grad_norm) (<= 0: none).  state_m / state_v: Adam's exp_avg /
 * exp_avg_sq; RMSprop's momentum buffer (or grad_avg when centered; both together are not supported) / square_avg. */
int ts_optim_step(ts_workspace* ws, int32_t kind, float* params, float* state_m, float* state_v, const float* grad, int64_t n,
                  int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay, double rms_alpha,
                  double rms_momentum, int32_t rms_centered, double max_grad_norm, ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Frame-stacked observations (Atari replay layout)
 * ------------------------------------------------------------------------------------------- */

/* The index part of ReplayBuffer.get(index, "obs") with stack_num > 1 (buffer_base.py:586-596):
 * out int64[I, stack_num], column stack_num-1-j = prev^j(index) (manager.py:311-336). Bit-exact. */
int ts_stack_indices(const int64_t* index, int64_t I, int64_t stack_num, const int64_t* offset, int64_t E,
                     const uint8_t* done, const int64_t* last_index, const int64_t* lengths, int64_t* out,
                     ts_stream_t stream);

/* The gather + np.stack + torch.as_tensor(float32) of buffer_base.py:590-596 / atari_network.py:121
 * in one pass: src u8[n_planes, plane_elems] (one image plane per row), plane_index int64[B, C] ->
 * out float32[B, plane_elems, C] (NHWC).  For a buffer that stores whole [C, H, W] observations
 * plane_index[b, c] = index[b] * C + c. */
int ts_gather_planes_nhwc(const uint8_t* src, int64_t n_planes, int64_t plane_elems, const int64_t* plane_index,
                          int64_t B, int64_t C, float* out, ts_stream_t stream);
/* Same gather with uint8 output [B, plane_elems, C] for the `obs_u8` mode of the network entry points below. */
int ts_gather_planes_nhwc_u8(const uint8_t* src, int64_t n_planes, int64_t plane_elems, const int64_t* plane_index,
                             int64_t B, int64_t C, uint8_t* out, ts_stream_t stream);
/* Both frame-stack gathers of a DQN-family update on a frame buffer that stores single frames (stack_num = 4 through prev(),
 * obs_next read at next(indices_after_n): examples/atari/atari_dqn.py:137-142, buffer_base.py:586-596, 624-626,
 * algorithm_base.py:772-791) in ONE launch: obs_out[b] = stacked observation at index[b], obs_next_out[b] = stacked observation at
 * next(next^(n_step - 1)(index[b])), both uint8 NHWC [B, plane_elems, 4].  Bit-identical to ts_nstep_indices + next() +
 * 2 x (ts_stack_indices + ts_gather_planes_nhwc_u8).  stack_num 4 and plane_elems % 16 == 0 only (TS_ERR_UNSUPPORTED else). */
int ts_dqn_gather_pair(const uint8_t* frames, int64_t n_planes, int64_t plane_elems, const int64_t* index, int64_t B,
                       int64_t n_step, int64_t stack_num, const int64_t* offset, int64_t E, const uint8_t* done,
                       const int64_t* last_index, const int64_t* lengths, uint8_t* obs_out, uint8_t* obs_next_out,
                       ts_stream_t stream);
/* The same pair for vector observations stored as float32 rows [n_rows, row_elems] with frame stacking (DRQN,
 * test/discrete/test_drqn.py:79-101): obs_out[b] = rows at prev^(stack_num-1-t)(index[b]), t = 0 .. stack_num-1 (oldest first,
 * buffer_base.py:586-596); obs_next_out[b] likewise from `rows_next` at indices_after_n = next^(n_step-1)(index[b]) when the
 * buffer stores obs_next, else (rows_next NULL) from `rows` at next(indices_after_n) (buffer_base.py:624-626,
 * algorithm_base.py:772-791); both float32 [B, stack_num, row_elems].  act_col / act_out (both or neither): batch.act =
 * act_col[index] (negative indices count from the end, as the index walks take them).  Bit-identical to ts_nstep_indices + next() + 2 x (ts_stack_indices + ts_gather_rows).
 * 1 <= stack_num <= 16 (TS_ERR_UNSUPPORTED else). */
int ts_stacked_rows_pair(const float* rows, const float* rows_next, int64_t n_rows, int64_t row_elems, const int64_t* index,
                         int64_t B, int64_t n_step, int64_t stack_num, const int64_t* offset, int64_t E, const uint8_t* done,
                         const int64_t* last_index, const int64_t* lengths, const int64_t* act_col, float* obs_out,
                         float* obs_next_out, int64_t* act_out, ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution / linear layers on fp32 MFMA (NHWC activations)
 * ------------------------------------------------------------------------------------------- */

/* h_dims (host int64[8]) = {B, IH, IW, IC, KH, KW, stride, OC}; x float32[B, IH, IW, IC];
 * wb float32[KH*KW*IC + 1, OC] (row (kh, kw, ic) = torch weight[:, ic, kh, kw], last row = bias);
 * y float32[B, OH, OW, OC].  nn.Conv2d / nn.Linear (+ ReLU) forward as used by DQNet
 * (atari_network.py:79-98); a Linear layer is IH = IW = KH = KW = stride = 1.
 * Shape limits: KH*KW*IC % 32 == 0, OC % 32 == 0, 16-byte aligned im2col runs.
 * x_u8 != 0: x is uint8[B, IH, IW, IC] (raw frames, e.g. from ts_gather_planes_nhwc_u8); the kernels convert on load
 * (exact), so the float32 copy of the observations -- 4x the HBM bytes -- is never materialised. */
int ts_conv_forward(ts_workspace* ws, const void* x, int x_u8, const float* wb, float* y, const int64_t* h_dims,
                    int relu, ts_stream_t stream);
/* autograd of the same layer: d_wb float32[KH*KW*IC + 1, OC] = d loss / d wb given dy float32[B, OH, OW, OC];
 * dx (nullable) float32[B, IH, IW, IC] = d loss / d x, multiplied by (mask > 0) when `mask` (the layer
 * input as produced by a ReLU, nullable) is given.  dx needs KH % stride == 0 and IC % 32 == 0. */
int ts_conv_backward(ts_workspace* ws, const void* x, int x_u8, const float* wb, const float* dy, const float* mask,
                     float* d_wb, float* dx, const int64_t* h_dims, ts_stream_t stream);
/* Kernel generation of the layers above and of every network entry point built on them.  Generation 2 (large row
 * counts: minibatch 65,536 of the Atari-shape PPO update) keeps the weight block resident in LDS and feeds the
 * activation operand straight from global memory into the MFMA registers; it sums in the same order as generation 1,
 * so forward results and input gradients are bit-identical.  mode: -1 = generation 1 only, 0 = automatic by row count
 * (default; the environment variable TS_CONV_V2 = 0 / 1 presets -1 / 1), 1 = generation 2 wherever the shape allows.
 * Returns the previous mode.  Process-wide; not meant to be flipped while launches are being enqueued from other threads. */
int ts_conv_set_generation(int mode);

/* ---------------------------------------------------------------------------------------------
 * DQN on NatureCNN (DQNet, tianshou/env/atari/atari_network.py:60-122)
 * ------------------------------------------------------------------------------------------- */

/* Flat fp32 parameter vector: five wb matrices (see ts_conv_forward) back to back:
 *   conv1 [8*8*c + 1, 32] | conv2 [4*4*32 + 1, 64] | conv3 [3*3*64 + 1, 64] | fc1 [F + 1, 512] | fc2 [512 + 1, n_act]
 * with F = 64 * OH3 * OW3 flattened in (h, w, channel) order (the NHWC flatten; torch flattens
 * (channel, h, w), tianshou_amd/dqn.py permutes on import / export).  h_offsets6 receives the five
 * layer offsets and the total; h_geom (nullable, int64[40]) the four ConvGeom rows
 * {B, IH, IW, IC, KH, KW, S, OH, OW, OC}. */
int64_t ts_dqn_param_count(int64_t c, int64_t h, int64_t w, int64_t n_act);
int ts_dqn_layer_offsets(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t* h_offsets6, int64_t* h_geom);

/* DQNet.forward + DiscreteQLearningPolicy.forward (dqn.py:101-143): obs [B, h, w, c] NHWC, raw 0..255 values (no
 * scaling), float32 (obs_u8 == 0) or uint8 (obs_u8 != 0) -> q_out float32[B, n_act], act_out (nullable)
 * int64[B] = argmax_a.  The same obs / obs_u8 convention holds for every network entry point below. */
int ts_dqn_forward(ts_workspace* ws, const float* params, int64_t c, int64_t h, int64_t w, int64_t n_act,
                   const void* obs_nhwc, int obs_u8, int64_t B, float* q_out, int64_t* act_out, ts_stream_t stream);

/* DQN._target_q (dqn.py:365-379) given Q_online(s') and Q_target(s') [B, n_act]:
 * is_double: q_target[b, argmax_a q_online[b, a]], else max_a q_target[b, a] -> out float32[B]. */
int ts_dqn_target_q(const float* q_online, const float* q_target, int64_t B, int64_t n_act, int is_double,
                    float* out, ts_stream_t stream);

/* DQN._target_q end to end (dqn.py:365-379): Q_online(s') and Q_target(s') (params_old; NULL = no lagged net,
 * dqn.py:371-374) evaluated concurrently on two streams, then ts_dqn_target_q.  obs_next float32[B, h, w, c]. */
int ts_dqn_target_q_fused(ts_workspace* ws, const float* params, const float* params_old, int64_t c, int64_t h,
                          int64_t w, int64_t n_act, const void* obs_next_nhwc, int obs_u8, int64_t B, int is_double,
                          float* out, ts_stream_t stream);

/* ts_dqn_target_q_fused followed by the arithmetic of compute_nstep_return (algorithm_base.py:798-811) in the same final
 * kernel: returns_out[b] = float(double(target_q[b] * mask[b]) * gpow[b] + mc[b]) with the coefficients of
 * ts_nstep_coefficients -- bit-identical to ts_dqn_target_q_fused + ts_nstep_return_fused, one launch less on the path
 * between the target passes and the loss. */
int ts_dqn_target_returns(ts_workspace* ws, const float* params, const float* params_old, int64_t c, int64_t h,
                          int64_t w, int64_t n_act, const void* obs_next_nhwc, int obs_u8, int64_t B, int is_double,
                          const float* nstep_mask, const double* nstep_gpow, const double* nstep_mc, float* returns_out,
                          ts_stream_t stream);

typedef struct ts_dqn_hparams {
    double lr;            /* < 0: compute the gradient only (no optimizer step) */
    double beta1, beta2, adam_eps;
    double huber_delta;   /* > 0: Huber loss, mean, PER weights ignored (dqn.py:392-398); else (td^2 * w).mean() */
    double max_grad_norm; /* <= 0: no clipping */
} ts_dqn_hparams;

/* DQN._update_with_batch (dqn.py:381-404) without the periodic target sync (ts_polyak_update, tau = 1):
 * q = Q(obs)[act]; td = returns - q -> td_out float32[B] (the new PER priorities, :401); loss -> loss_out
 * float32[1]; backward through DQNet; clip + Adam (adam_step = 1-based step of this call).
 * weight nullable (= 1.0).  grad_out (nullable) float32[P] receives the unclipped flat gradient. */
int ts_dqn_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t c,
                  int64_t h, int64_t w, int64_t n_act, const void* obs_nhwc, int obs_u8, const int64_t* act,
                  const float* returns, const float* weight, int64_t B, const ts_dqn_hparams* hp, float* td_out,
                  float* loss_out, float* grad_out, ts_stream_t stream);
/* The forward pass on the batch's own observations does not depend on the target computation (DQN._target_q runs on
 * obs_next in _preprocess_batch, dqn.py:257-275, before _update_with_batch forwards batch.obs, dqn.py:381-404, with the same
 * online parameters): ts_dqn_forward_cache runs it ahead of time -- on another stream, beside the two obs_next passes -- and
 * leaves every layer's activations and Q(s) in the caller's `cache` (ts_dqn_cache_bytes bytes, 256-byte aligned);
 * ts_dqn_update_cached is ts_dqn_update without its forward pass, reading them from there.  The caller orders the two calls
 * (stream events) and must not change `params` in between. */
int64_t ts_dqn_cache_bytes(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t B);
int ts_dqn_forward_cache(ts_workspace* ws, const float* params, int64_t c, int64_t h, int64_t w, int64_t n_act,
                         const void* obs_nhwc, int obs_u8, int64_t B, void* cache, int64_t cache_bytes, ts_stream_t stream);
int ts_dqn_update_cached(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t c,
                         int64_t h, int64_t w, int64_t n_act, const void* obs_nhwc, int obs_u8, const int64_t* act,
                         const float* returns, const float* weight, int64_t B, const ts_dqn_hparams* hp, void* cache,
                         float* td_out, float* loss_out, float* grad_out, ts_stream_t stream);

/* Makes `stream` wait until the TD errors (td_out; prio_out of the distributional engines) and the loss of the most recent
 * ts_dqn_update / ts_dqn_update_cached / ts_distq_update / ts_rainbow_update call on `ws` are written -- not for the rest
 * of that update.  PrioritizedReplayBuffer.update_weight (prio.py:89-100,
 * _postprocess_batch algorithm_base.py:562-581) and the sampling / gathering of the next batch need nothing else from the
 * update: issued on another stream behind this wait they run beside its backward pass and optimizer step. */
int ts_dqn_wait_td(ts_workspace* ws, ts_stream_t stream);

/* A device-resident Atari-layout replay buffer: one uint8 frame per slot, the observation of a transition being the stack of
 * the `c` frames ending at it (ReplayBuffer(stack_num = c, save_only_last_obs = True, ignore_obs_next = True),
 * buffer_base.py:60-110, 605-649; examples/atari/atari_dqn.py), held as DeviceReplayBuffer holds the reference's
 * ReplayBufferManager columns; with a sum tree (`tree` != NULL) it is a PrioritizedVectorReplayBuffer (prio.py:25-47). */
typedef struct ts_frame_replay {
    const int64_t* offset;      /* int64[E + 1] sub-buffer offsets                                   */
    int64_t E;
    const int64_t* lengths;     /* int64[E]                                                          */
    const int64_t* last_index;  /* int64[E]                                                          */
    const uint8_t* done;        /* uint8[slots]                                                      */
    const uint8_t* terminated;  /* uint8[slots]                                                      */
    const double* rew;          /* float64[slots]                                                    */
    const uint8_t* frames;      /* uint8[slots, plane_elems] (plane_elems = h * w)                   */
    int64_t plane_elems;
    const int64_t* act_col;     /* int64[slots]                                                      */
    int64_t slots;
    double* tree;               /* float64[2 * bound] sum tree (SegmentTree, segtree.py) or NULL     */
    int64_t bound;
    double* prio_minmax;        /* float64[2] = {max_prio, min_prio} (prio.py:36)                    */
    double alpha, beta;         /* prio.py:32-33                                                     */
    int32_t weight_norm;        /* prio.py:78-79                                                     */
    int32_t reserved;
} ts_frame_replay;

/* OffPolicyAlgorithm.update (algorithm_base.py:583-631: buffer.sample -> _preprocess_batch -> _update_with_batch ->
 * _postprocess_batch) of DQN on DQNet for such a buffer in ONE call.  Update number `counter` draws its batch as
 * PrioritizedReplayBuffer.sample_indices does (prio.py:63-67) from u = ts_uniform_fill_f64(seed, counter) through
 * ts_per_sample (indices + importance weights, prio.py:69-79) -- or, without a tree, as ts_sample_indices_seeded(seed,
 * counter) --, gathers batch.act / obs / obs_next (ts_gather_rows, ts_dqn_gather_pair), computes the n-step returns
 * (ts_nstep_coefficients + ts_dqn_target_returns) and runs ts_dqn_update_cached on a forward pass started beside the target
 * passes (ts_dqn_forward_cache).  On a replay stream behind the loss kernel (ts_dqn_wait_td), beside the backward pass and the
 * optimizer step: the priority update with the TD errors (ts_per_update_weight, prio.py:81-100) and the batch of update
 * counter + 1, left in `scratch` (double-buffered by counter parity).  prepared != 0: the previous call (counter - 1, same
 * scratch, buffer unchanged since) left this update's batch there; 0: it is sampled now.  Every value equals what the
 * separate calls in that order produce.
 * scratch: ts_dqn_learn_scratch_bytes bytes, 256-byte aligned, zeroed once by the caller; ws_aux: a second workspace -- the
 * ahead-of-time forward pass and the replay stream (its first side stream) use it.  sync_target != 0: params_old := params
 * between the returns and the update (dqn.py:283-285).  td_out nullable float32[B], idx_out nullable int64[B]: copies of the
 * update's TD errors / indices on `stream`.  Frame stack 4 and plane sizes that are multiples of 16 bytes (ts_dqn_gather_pair). */
int64_t ts_dqn_learn_scratch_bytes(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t B);
int ts_dqn_learn_step(ts_workspace* ws, ts_workspace* ws_aux, float* params, float* params_old, int sync_target, float* adam_m,
                      float* adam_v, int64_t adam_step, int64_t c, int64_t h, int64_t w, int64_t n_act,
                      const ts_frame_replay* replay, int64_t B, int64_t n_step, double gamma, int is_double,
                      const ts_dqn_hparams* hp, uint64_t seed, uint64_t counter, int prepared, void* scratch,
                      int64_t scratch_bytes, float* td_out, float* loss_out, int64_t* idx_out, ts_stream_t stream);
/* The update of ts_dqn_learn_step for a batch the CALLER drew -- `indices` int64[B] and (nullable) importance weights float32[B]
 * from a host PrioritizedVectorReplayBuffer's sample_indices / get_weight (prio.py:63-79), as HipDQN.update() has them at hook
 * level (OffPolicyAlgorithm.update, algorithm_base.py:586-631; DQN._preprocess_batch / _update_with_batch, dqn.py:257-275, 381-404):
 * batch.act / obs / obs_next gathered by index, n-step returns with the lagged network, the update with a forward pass started
 * beside the target passes -- the same kernels on the same values as the separate calls.  No sampling, no priority update
 * (`replay->tree` is not looked at): the TD errors return in td_out for the caller's _postprocess_batch (prio.py:81-100).
 * returns_out nullable float32[B] (batch.returns).  scratch / ws_aux / sync_target as in ts_dqn_learn_step. */
int ts_dqn_learn_rows(ts_workspace* ws, ts_workspace* ws_aux, float* params, float* params_old, int sync_target, float* adam_m,
                      float* adam_v, int64_t adam_step, int64_t c, int64_t h, int64_t w, int64_t n_act,
                      const ts_frame_replay* replay, const int64_t* indices, const float* weight, int64_t B, int64_t n_step,
                      double gamma, int is_double, const ts_dqn_hparams* hp, void* scratch, int64_t scratch_bytes,
                      float* returns_out, float* td_out, float* loss_out, ts_stream_t stream);
/* With TS_DQN_GRAPH=1 in the environment ts_dqn_learn_step replays its steady state (same arguments as the two calls before,
 * batch prepared ahead, priorities present) from a HIP graph captured from its own stream enqueue, one per (counter parity,
 * sync_target): one graph launch + one scalar kernel (the Philox counter and Adam's step-dependent scalars move to device
 * memory) instead of ~45 kernel launches and ~20 event operations; bit-identical.  Off by default: on ROCm 7.2 hipGraphLaunch
 * costs the host as much as the launches it replaces and the replay is slower than the hand-placed streams
 * (profiles/r06_dqn_learn_step_ab.txt).  A failed capture falls back to the streams for the rest of the process.
 * -> updates replayed from a graph on `ws` so far (-1: capture failed). */
int64_t ts_dqn_learn_graph_launches(ts_workspace* ws);

/* ---------------------------------------------------------------------------------------------
 * DQN on a recurrent Q network (DRQN, test/discrete/test_drqn.py:79-101): Recurrent (tianshou/utils/net/common.py:372-452)
 * = fc1 Linear(obs_dim, H) -> nn.LSTM(H, H, L layers, batch_first) -> fc2 Linear(H, n_act) on the last step.
 * Flat parameter vector (last row of every block = bias; gate order i, f, g, o as in torch):
 *   fc1 [k0 + 1, H] | per layer: W_ih [H + 1, 4H] | W_hh [H + 1, 4H] | fc2 [H + 1, 32]
 * k0 = obs_dim rounded up to 32 (padding rows zero), head columns [0, n_act) = Q (padding columns zero).
 * H a multiple of 32 in [32, 1024], 1 <= L <= 8, n_act <= 32, T <= 4096, B T <= 2^24.
 * h_out int64[4 + 2 L] = {k0, count, off_fc1, off_ih[0], off_hh[0], ..., off_fc2}.
 * ------------------------------------------------------------------------------------------- */
int ts_rnnq_layout(int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act, int64_t* h_out);

/* Recurrent.forward + DiscreteQLearningPolicy.forward (common.py:400-452, dqn.py:101-143): obs float32[B, T, obs_dim]
 * (training: T = stack_num, no state; evaluation: T = 1 with the carried state) -> q_out float32[B, n_act] (nullable),
 * act_out int64[B] = argmax (nullable).  State tensors are LAYER-major float32[L, B, H] (the reference carries
 * [B, L, H], common.py:441-450; the wrapper transposes): h_in / c_in nullable together (= zeros), h_out / c_out nullable. */
int ts_rnnq_forward(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act,
                    const float* obs, int64_t B, int64_t T, const float* h_in, const float* c_in, float* q_out, int64_t* act_out,
                    float* h_out, float* c_out, ts_stream_t stream);

/* DQN._update_with_batch (dqn.py:381-404) on the recurrent network, same contract as ts_dqn_update: TD error -> td_out,
 * loss -> loss_out, backward through the T steps, clip + Adam (hp->lr < 0: gradient only); grad_out nullable float32[count]. */
int ts_rnnq_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                   int64_t hidden, int64_t layers, int64_t n_act, const float* obs, const int64_t* act, const float* returns,
                   const float* weight, int64_t B, int64_t T, const ts_dqn_hparams* hp, float* td_out, float* loss_out,
                   float* grad_out, ts_stream_t stream);

/* DQN._target_q end to end on the recurrent network (dqn.py:365-379), as ts_dqn_target_q_fused: Q_online(s') and
 * Q_target(s') (params_old; NULL = no lagged network) on two streams, then the arg-max / gather -> out float32[B].
 * obs_next float32[B, T, obs_dim]. */
int ts_rnnq_target_q_fused(ts_workspace* ws, const float* params, const float* params_old, int64_t obs_dim, int64_t hidden,
                           int64_t layers, int64_t n_act, const float* obs_next, int64_t B, int64_t T, int is_double, float* out,
                           ts_stream_t stream);
/* ts_rnnq_target_q_fused with the arithmetic half of compute_nstep_return (algorithm_base.py:793-812) folded into its last
 * kernel, on ts_nstep_coefficients' outputs (as ts_dqn_target_returns): returns_out[b] =
 * float(double(target_q[b] * ns_mask[b]) * ns_gpow[b] + ns_mc[b]). */
int ts_rnnq_target_returns(ts_workspace* ws, const float* params, const float* params_old, int64_t obs_dim, int64_t hidden,
                           int64_t layers, int64_t n_act, const float* obs_next, int64_t B, int64_t T, int is_double,
                           const float* ns_mask, const double* ns_gpow, const double* ns_mc, float* returns_out,
                           ts_stream_t stream);

/* A uniform (no priorities) device-resident replay buffer of float32 observation rows, as DeviceReplayBuffer holds the
 * reference's ReplayBufferManager columns (data/buffer/manager.py:25-60, buffer_base.py:60-110). */
typedef struct ts_rows_replay {
    const int64_t* offset;      /* int64[E + 1] sub-buffer offsets                                   */
    int64_t E;
    const int64_t* lengths;     /* int64[E]                                                          */
    const int64_t* last_index;  /* int64[E]                                                          */
    const uint8_t* done;        /* uint8[slots]                                                      */
    const uint8_t* terminated;  /* uint8[slots]                                                      */
    const double* rew;          /* float64[slots]                                                    */
    const float* obs_rows;      /* float32[slots, obs_dim]                                           */
    const float* obs_next_rows; /* float32[slots, obs_dim] or NULL (obs_next read at next(index))    */
    const int64_t* act_col;     /* int64[slots]                                                      */
    int64_t slots;
} ts_rows_replay;

/* OffPolicyAlgorithm.update (algorithm_base.py:583-631: buffer.sample -> _preprocess_batch -> _update_with_batch; no
 * priorities to write back) of DQN on `Recurrent` for such a buffer in ONE call: update number `counter` draws its indices as
 * ts_sample_indices_seeded(seed, counter), gathers batch.obs / obs_next / act (ts_stacked_rows_pair), computes the n-step
 * returns (ts_nstep_coefficients + ts_rnnq_target_returns) and runs ts_rnnq_update_cached on a forward pass started beside
 * the target passes; the batch of update counter + 1 is prepared on a side stream beside it and left in `scratch`
 * (double-buffered by counter parity).  prepared != 0: the previous call (counter - 1, same scratch, buffer unchanged since)
 * left this update's batch there; 0: it is sampled now.  Every value equals what the separate calls produce.
 * scratch: ts_rnnq_learn_scratch_bytes bytes, 256-byte aligned, zeroed once by the caller; ws_aux: a second workspace (the
 * ahead-of-time forward pass runs concurrently with the target passes).  sync_target != 0: params_old := params between the
 * returns and the update, where the reference's periodic hard sync sits (dqn.py:283-285: the first statement of
 * _update_with_batch, after _preprocess_batch used the lagged network); the caller keeps the counter.
 * idx_out nullable int64[B]: the update's indices. */
int64_t ts_rnnq_learn_scratch_bytes(int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act, int64_t B, int64_t T);
int ts_rnnq_learn_step(ts_workspace* ws, ts_workspace* ws_aux, float* params, float* params_old, int sync_target, float* adam_m,
                       float* adam_v, int64_t adam_step, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act,
                       const ts_rows_replay* replay, int64_t B, int64_t T, int64_t n_step, double gamma, int is_double,
                       const ts_dqn_hparams* hp, uint64_t seed, uint64_t counter, int prepared, void* scratch,
                       int64_t scratch_bytes, float* td_out, float* loss_out, int64_t* idx_out, ts_stream_t stream);

/* The forward pass of ts_rnnq_update ahead of time (as ts_dqn_forward_cache / ts_dqn_update_cached for the NatureCNN): it
 * needs nothing from the two obs_next passes of _target_q (same online parameters, dqn.py:257-275 vs 381-404), and a pass at
 * B = 128 occupies eight CUs.  ts_rnnq_forward_cache leaves every activation of the pass in the caller's `cache`
 * (ts_rnnq_cache_bytes bytes, 256-byte aligned); ts_rnnq_update_cached is ts_rnnq_update without its forward pass.  The
 * caller orders the two calls (stream events) and must not change `params` in between. */
int64_t ts_rnnq_cache_bytes(int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act, int64_t B, int64_t T);
int ts_rnnq_forward_cache(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t n_act,
                          const float* obs, int64_t B, int64_t T, void* cache, int64_t cache_bytes, ts_stream_t stream);
int ts_rnnq_update_cached(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                          int64_t hidden, int64_t layers, int64_t n_act, const float* obs, const int64_t* act, const float* returns,
                          const float* weight, int64_t B, int64_t T, const ts_dqn_hparams* hp, void* cache, float* td_out,
                          float* loss_out, float* grad_out, ts_stream_t stream);

/* LSTM trunk + one linear head, generic -- the recurrent actor and critic of the continuous-control nets:
 *   RecurrentActorProb  tianshou/utils/net/continuous.py:241-322: nn.LSTM(obs_dim -> H, L layers, batch_first) on the
 *                       observation itself (no fc1), mu = Linear(H, act)(h_T), bounded as max_action * tanh(mu) unless
 *                       `unbounded` (tanh_scale = max_action, or 0 for none); sigma = exp(sigma_param) is state-
 *                       independent and stays with the caller (conditioned_sigma is not supported);
 *   RecurrentCritic     continuous.py:325-380: the same trunk, fc2 = Linear(H + act_dim, 1) on [h_T | act]
 *                       (extra_dim = act_dim, `extra` = the actions, float32[B, extra_dim]);
 *   has_fc1 = 1         Recurrent (common.py:372-452, the DRQN network of ts_rnnq_*) through the same entry points.
 * Flat layout (ts_lstm_net_layout: h_out int64[5 + 2 L] = {k0, count, off_fc1 or -1, (off_ih, off_hh) per layer, off_head,
 * head_in}): [fc1 [k0 + 1, H] if has_fc1] | layer 0: W_ih [(has_fc1 ? H : k0) + 1, 4H] | W_hh [H + 1, 4H] | layers >= 1:
 * W_ih [H + 1, 4H] | W_hh [H + 1, 4H] | head [head_in + 1, 32], head_in = H (extra_dim = 0) or H + extra_dim rounded up to
 * a multiple of 32; last row of every block = bias (b_ih / b_hh kept separately, as torch does); padding rows / columns
 * are and stay zero.  obs float32[B, T, obs_dim]; h_in / c_in (nullable, together) float32[L, B, H].
 *   forward : out float32[B, out_dim] (after the tanh bound when tanh_scale > 0), h_out / c_out nullable float32[L, B, H].
 *   backward: d_out float32[B, out_dim] = d loss / d out -> grad_out float32[count] = d loss / d params (forward is
 *             re-run inside; `out` nullable receives it).  The optimizer step is ts_adam_step. */
int ts_lstm_net_layout(int64_t obs_dim, int64_t hidden, int64_t layers, int64_t out_dim, int64_t has_fc1, int64_t extra_dim,
                       int64_t* h_out);
int ts_lstm_net_forward(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t out_dim,
                        int64_t has_fc1, int64_t extra_dim, const float* obs, const float* extra, int64_t B, int64_t T,
                        const float* h_in, const float* c_in, double tanh_scale, float* out, float* h_out, float* c_out,
                        ts_stream_t stream);
int ts_lstm_net_backward(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t layers, int64_t out_dim,
                         int64_t has_fc1, int64_t extra_dim, const float* obs, const float* extra, int64_t B, int64_t T,
                         const float* h_in, const float* c_in, double tanh_scale, const float* d_out, float* out, float* grad_out,
                         ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Distributional Q-learning on the Atari networks: QRDQN (tianshou/algorithm/modelfree/qrdqn.py) and
 * C51 (modelfree/c51.py) with QRDQNet / C51Net (tianshou/env/atari/atari_network.py:211-235 / :125-151):
 * DQNet with n_act * n_atoms outputs viewed [B, n_act, n_atoms] (C51: softmax over the atoms).
 * Flat parameter vector: the ts_dqn_param_count layout with the head matrix [513, W], W = n_act * n_atoms rounded up to
 * a multiple of 32; column a * n_atoms + j, the padding columns are (and stay) zero.
 * `aux` (device float32[n_atoms]): QRDQN -> tau_hat (qrdqn.py:87-91), needed by ts_distq_update only;
 * C51 -> the support linspace(v_min, v_max, n_atoms) (c51.py:61-64), needed everywhere.
 * n_act <= 64, 2 <= n_atoms <= 256.
 * ------------------------------------------------------------------------------------------- */
#define TS_DISTQ_QR 0
#define TS_DISTQ_C51 1

int64_t ts_distq_param_count(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t n_atoms);

/* QRDQNet.forward / C51Net.forward + QRDQNPolicy / C51Policy.compute_q_value (qrdqn.py:19-21, c51.py:66-67) +
 * DiscreteQLearningPolicy.forward's argmax (dqn.py:141): dist_out float32[B, n_act, n_atoms] (quantile values or
 * atom probabilities), q_out float32[B, n_act], act_out int64[B]; each nullable. */
int ts_distq_forward(ts_workspace* ws, const float* params, int64_t c, int64_t h, int64_t w, int64_t n_act,
                     int64_t n_atoms, int kind, const float* aux, const void* obs_nhwc, int obs_u8, int64_t B,
                     float* dist_out, float* q_out, int64_t* act_out, ts_stream_t stream);

/* QRDQN._target_q (qrdqn.py:93-104) and the first half of C51._target_dist (c51.py:123-132): greedy action of the
 * online net on obs_next, its distribution under params_old (NULL: no target network, the online net's own)
 * -> out float32[B, n_atoms]. */
int ts_distq_next_dist(ts_workspace* ws, const float* params, const float* params_old, int64_t c, int64_t h,
                       int64_t w, int64_t n_act, int64_t n_atoms, int kind, const float* aux,
                       const void* obs_next_nhwc, int obs_u8, int64_t B, float* out, ts_stream_t stream);

typedef struct ts_distq_hparams {
    double lr;            /* < 0: compute the gradient only (no optimizer step) */
    double beta1, beta2, adam_eps;
    double max_grad_norm; /* <= 0: no clipping */
    double v_min, v_max;  /* C51 support bounds (c51.py:57-60); delta_z = (v_max - v_min) / (n_atoms - 1) */
} ts_distq_hparams;

/* QRDQN._update_with_batch (qrdqn.py:106-131) / C51._update_with_batch (c51.py:143-160) after the periodic sync:
 * forward, loss, backward, clip_grad_norm_ + Adam (algorithm_base.py:484-500).
 * returns float32[B, n_atoms] = batch.returns; next_dist float32[B, n_atoms] = ts_distq_next_dist on batch.obs_next
 * (C51 only, NULL for QRDQN); weight float32[B] nullable (PER importance weights);
 * prio_out float32[B] = the new batch.weight (qrdqn.py:128 / c51.py:157); loss_out float32[1];
 * target_dist_out float32[B, n_atoms] nullable (C51: the projected target distribution, c51.py:136-141);
 * grad_out nullable: the flat gradient. */
int ts_distq_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t c,
                    int64_t h, int64_t w, int64_t n_act, int64_t n_atoms, int kind, const float* aux,
                    const void* obs_nhwc, int obs_u8, const int64_t* act, const float* returns, const float* next_dist,
                    const float* weight, int64_t B, const ts_distq_hparams* hp, float* prio_out, float* loss_out,
                    float* target_dist_out, float* grad_out, ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Rainbow (tianshou/algorithm/modelfree/rainbow.py): C51 on RainbowNet (env/atari/atari_network.py:154-208) -- DQNet
 * feature trunk, NoisyLinear layers (utils/net/discrete.py:317-374) in a Q branch (F -> 512 -> n_act * n_atoms) and a
 * dueling V branch (F -> 512 -> n_atoms), logits = q - mean_a q + v, softmax over the atoms.
 * Flat parameters: conv1 | conv2 | conv3 | Q0.mu [F + 1, 512] | Q0.sigma | Q2.mu [513, ldq] | Q2.sigma | V0.mu | V0.sigma |
 * V2.mu [513, ldv] | V2.sigma  (mu / sigma = the layer's mu_W + mu_bias / sigma_W + sigma_bias in the engine's matrix
 * layout; F in (h, w, c) order; ldq / ldv = n_act * n_atoms / n_atoms rounded up to 32, padding zero).
 * Noise of one network (device float32): Q0.eps_p [F] | Q0.eps_q [512] | Q2.eps_p [512] | Q2.eps_q [ldq] | V0.eps_p [F] |
 * V0.eps_q [512] | V2.eps_p [512] | V2.eps_q [ldv]; NULL selects eval mode (discrete.py:370-372).
 * `support` = C51Policy.support (c51.py:61-64), device float32[n_atoms].
 * h_out20 = {F, ldq, ldv, parameter count, noise count, 3 conv offsets, 4 noisy-layer offsets (mu block; sigma follows),
 * 8 noise offsets}.
 * ------------------------------------------------------------------------------------------- */
int ts_rainbow_layout(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t n_atoms, int64_t* h_out20);

/* RainbowNet.forward + C51Policy.compute_q_value + argmax: dist_out float32[B, n_act, n_atoms] (probabilities),
 * q_out float32[B, n_act], act_out int64[B]; each nullable. */
int ts_rainbow_forward(ts_workspace* ws, const float* params, const float* noise, int64_t c, int64_t h, int64_t w,
                       int64_t n_act, int64_t n_atoms, const float* support, const void* obs_nhwc, int obs_u8, int64_t B,
                       float* dist_out, float* q_out, int64_t* act_out, ts_stream_t stream);

/* First half of C51._target_dist (c51.py:123-132) with the noisy networks: greedy action of (params, noise) on obs_next,
 * its distribution under (params_old, noise_old) (params_old NULL: the online net's own) -> out float32[B, n_atoms]. */
int ts_rainbow_next_dist(ts_workspace* ws, const float* params, const float* noise, const float* params_old,
                         const float* noise_old, int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t n_atoms,
                         const float* support, const void* obs_next_nhwc, int obs_u8, int64_t B, float* out,
                         ts_stream_t stream);

/* RainbowDQN._update_with_batch (rainbow.py:93-101 -> c51.py:143-160) after the noise draws and the periodic sync:
 * forward with `noise`, projection + cross entropy, backward through the dueling heads and the NoisyLinear layers
 * (d mu = d W, d sigma = d W * eps_q x eps_p), clip_grad_norm_ + Adam.  Arguments as ts_distq_update (C51). */
int ts_rainbow_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, const float* noise,
                      int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t n_atoms, const float* support,
                      const void* obs_nhwc, int obs_u8, const int64_t* act, const float* returns, const float* next_dist,
                      const float* weight, int64_t B, const ts_distq_hparams* hp, float* prio_out, float* loss_out,
                      float* target_dist_out, float* grad_out, ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PPO on the Atari actor-critic (examples/atari/atari_ppo.py:106-118): DQNet(features_only=True,
 * output_dim_added_layer=512) shared by DiscreteActor(softmax_output=False) and DiscreteCritic, Categorical policy
 * ------------------------------------------------------------------------------------------- */

/* Flat parameter vector: conv1 | conv2 | conv3 | fc [F + 1, 512] (as ts_dqn_param_count) | head [513, 32] with
 * columns [0, n_act) = DiscreteActor.last (logits), column n_act = DiscreteCritic.last (V), the rest zero.
 * h_offsets7 = the five layer offsets, the total, and the head width (32); h_geom (nullable) int64[50].
 * n_act <= 31. */
int ts_cnn_ac_layer_offsets(int64_t c, int64_t h, int64_t w, int64_t n_act, int64_t* h_offsets7, int64_t* h_geom);

/* No-grad passes of PPO._preprocess_batch (a2c.py:122-129, ppo.py:157-161) on obs float32[B, h, w, c] (NHWC):
 * v_out[b] = V(obs_b) (nullable); logp_out[b] = Categorical(logits(obs_b)).log_prob(act_b) (nullable, needs act);
 * logits_out (nullable) float32[B, n_act].  One trunk pass serves all outputs. */
int ts_cnn_ac_infer(ts_workspace* ws, const float* params, int64_t c, int64_t h, int64_t w, int64_t n_act,
                    const void* obs_nhwc, int obs_u8, const int64_t* act, int64_t B, float* v_out, float* logp_out,
                    float* logits_out, ts_stream_t stream);

/* One minibatch of PPO._update_with_batch (ppo.py:179-216) + Optimizer.step: forward, Categorical log-prob /
 * entropy, clipped surrogate (dual clip optional), (clipped) value loss, backward through both heads and the
 * shared trunk, clip_grad_norm_ + Adam.  adv_stats (device float32[2] = {mean, unbiased std} of this minibatch's
 * advantages) is used when hp->adv_norm.  losses_out4 = {loss, clip, vf, ent}; hp->lr < 0: gradient only;
 * grad_out (nullable) float32[P] receives the unclipped gradient.  hp->algo must be 0. */
int ts_cnn_ppo_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t c,
                    int64_t h, int64_t w, int64_t n_act, const void* obs_nhwc, int obs_u8, const int64_t* act, const float* adv,
                    const float* returns, const float* logp_old, const float* v_old, int64_t B,
                    const float* adv_stats, const ts_ppo_hparams* hp, float* losses_out4, float* grad_out,
                    ts_stream_t stream);

/* The same three operations for an MLP trunk: Net(obs, [hidden, hidden]) ReLU (utils/net/common.py:343-369) shared by
 * DiscreteActor and DiscreteCritic -- BASELINE.json configs[0], test/discrete/test_ppo_discrete.py:88-98 (CartPole
 * shape: obs 4, hidden 64, 2 actions).  The policy is Categorical over the actor's outputs (softmax_output=True with
 * Categorical(probs) and softmax_output=False with Categorical(logits=) are the same distribution).
 * Flat parameter vector: L1 [k0 + 1, hidden] | L2 [hidden + 1, hidden] | head [hidden + 1, 32] (columns [0, n_act) =
 * DiscreteActor.last, column n_act = DiscreteCritic.last; k0 = obs_dim rounded up to 32; last row of a block = bias).
 * hidden a multiple of 32 in [32, 2048], n_act <= 31.  h_out3 = {k0, head width, parameter count};
 * obs float32[B, obs_dim]. */
int ts_mlp_ac_layout(int64_t obs_dim, int64_t hidden, int64_t n_act, int64_t* h_out3);
int ts_mlp_ac_infer(ts_workspace* ws, const float* params, int64_t obs_dim, int64_t hidden, int64_t n_act,
                    const float* obs, const int64_t* act, int64_t B, float* v_out, float* logp_out, float* logits_out,
                    ts_stream_t stream);
int ts_mlp_ppo_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                    int64_t hidden, int64_t n_act, const float* obs, const int64_t* act, const float* adv,
                    const float* returns, const float* logp_old, const float* v_old, int64_t B, const float* adv_stats,
                    const ts_ppo_hparams* hp, float* losses_out4, float* grad_out, ts_stream_t stream);
/* The WHOLE `_update_with_batch` of that network in one launch (PPO._update_with_batch, ppo.py:164-224 without
 * recompute_advantage, + Optimizer.step, algorithm_base.py:484-500) for the small shapes where a gradient step is launch
 * latency and nothing else (BASELINE.json configs[0]: 310 steps of 64 rows): obs_dim <= 32, hidden == 64, n_act <= 31
 * (ts_mlp_ppo_update_supported; TS_ERR_UNSUPPORTED otherwise -- the caller loops over ts_mlp_ppo_step).  One persistent
 * workgroup keeps the weights in LDS and the Adam moments in registers for all n_steps steps.
 *   obs float32[n, obs_dim], act int64[n], adv / returns / logp_old / v_old float32[n]: the whole preprocessed batch;
 *   rows int64[h_mb_offset[n_steps]] (device): the concatenated minibatch row lists (the `repeat` permutations of
 *   Batch.split, batch.py:1205-1215); h_mb_offset (host): step k takes rows[h_mb_offset[k] .. h_mb_offset[k + 1]);
 *   advantage normalisation uses each minibatch's float64 mean / unbiased std (ppo.py:184-186);
 *   adam_step0 = optimizer steps taken before this call; losses_out float32[n_steps, 4] = {loss, clip, vf, ent}. */
int ts_mlp_ppo_update_supported(int64_t obs_dim, int64_t hidden, int64_t n_act);
int ts_mlp_ppo_update(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step0, int64_t obs_dim,
                      int64_t hidden, int64_t n_act, const float* obs, const int64_t* act, const float* adv,
                      const float* returns, const float* logp_old, const float* v_old, int64_t n, const int64_t* rows,
                      const int64_t* h_mb_offset, int64_t n_steps, const ts_ppo_hparams* hp, float* losses_out,
                      ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SAC (tanh-Gaussian actor with state-conditioned sigma, twin critics on concat(obs, act), hidden [256, 256])
 * nets as in examples/mujoco/mujoco_sac.py:82-104
 * ------------------------------------------------------------------------------------------- */

/* Flat parameter vectors (wb matrices as in ts_conv_forward, 1x1 case, back to back):
 *   actor  : L1 [ka + 1, 256] | L2 [257, 256] | head [257, 64]   head columns [0, A) = mu, [32, 32 + A) = the
 *            pre-clamp log-sigma (ContinuousActorProbabilistic.mu / .sigma, continuous.py:205-213)
 *   critic : L1 [kc + 1, 256] | L2 [257, 256] | head [257, 32]   head column 0 = Q (ContinuousCritic.last)
 * with ka = obs_dim and kc = obs_dim + act_dim rounded up to a multiple of 32; padding rows / columns are zero
 * and stay zero.  h_out8 = {ka, kc, actor count, critic count, actor L2 offset, actor head offset,
 * critic L2 offset, critic head offset}.  act_dim <= 32. */
int ts_sac_layout(int64_t obs_dim, int64_t act_dim, int64_t* h_out8);
/* Net(hidden_sizes=[h, h]) with h other than 256 (utils/net/common.py:246-369 takes any): the hidden width is a property
 * of the WORKSPACE -- ts_mlp_set_hidden(ws, h) (h a multiple of 32 in [32, 1024]; 0 = 256) applies to every SAC / TD3 /
 * DDPG / REDQ entry point subsequently called with `ws`; the `_h` layout variants take it explicitly (no workspace there).
 * 256 runs on the fused three-layer kernels, other widths on the per-layer GEMM kernels; in the layouts above every
 * "256" / "257" then reads h / h + 1. */
int ts_mlp_set_hidden(ts_workspace* ws, int64_t hidden);
int ts_sac_layout_h(int64_t obs_dim, int64_t act_dim, int64_t hidden, int64_t* h_out8);
/* Net(hidden_sizes=[h] * depth) for depth other than 2 (utils/net/common.py:246-369 takes any list; round 6): the number of
 * hidden layers is a property of the workspace too -- ts_mlp_set_trunk(ws, h, depth) (depth in [1, TS_MLP_MAX_HIDDEN_LAYERS];
 * 0 = 2) applies to every SAC / TD3 / DDPG / REDQ / DiscreteSAC entry point subsequently called with `ws`;
 * ts_mlp_set_hidden resets the depth to 2.  Flat vectors then hold depth + 1 wb matrices back to back:
 *   L1 [k + 1, h] | L2 .. Ldepth [h + 1, h] each | head [h + 1, head_cols]
 * (k = the input width rounded up to 32; head_cols = 64 for SAC's / REDQ's Gaussian actor, 32 otherwise) -- for depth 2
 * exactly the layouts above.  ts_mlp_layout: h_out[0] = k, h_out[1 + i] = offset of linear layer i (i = 0 .. depth; the
 * head is layer `depth`), h_out[depth + 2] = the element count; h_out has depth + 3 entries.  Depth 2 at the widths the
 * fused three-layer kernels take runs on them; every other trunk runs layer by layer on the GEMM kernels (ReLU fused
 * into the forward GEMM's epilogue and into the input-gradient GEMM's mask). */
#define TS_MLP_MAX_HIDDEN_LAYERS 6
int ts_mlp_set_trunk(ts_workspace* ws, int64_t hidden, int64_t depth);
/* Net(activation=nn.Tanh) in place of the default nn.ReLU (utils/net/common.py:246-369) for the same entry points: TS_NET_ACT_RELU
 * (default; ts_mlp_set_hidden / ts_mlp_set_trunk reset to it) or TS_NET_ACT_TANH after every hidden layer.  Tanh trunks always run
 * layer by layer (the fused three-layer kernels are ReLU). */
int ts_mlp_set_activation(ts_workspace* ws, int activation);
/* ContinuousActorProbabilistic(unbounded=False) -- the class default, utils/net/continuous.py:194, 230-231: mu = max_action *
 * tanh(mu) in front of SAC's / REDQ's Gaussian (the examples pass unbounded=True).  A property of the workspace like the trunk:
 * applies to every ts_sac_* / ts_redq_* entry point subsequently called with `ws` -- forward, target, update (the gradient goes
 * back through max_action * (1 - tanh^2)).  0 = unbounded (the default). */
int ts_sac_set_actor_bound(ts_workspace* ws, double max_action);
int ts_mlp_layout(int64_t in_dim, int64_t hidden, int64_t depth, int64_t head_cols, int64_t* h_out);

/* SACPolicy.forward (sac.py:108-131) with rsample() = loc + noise * scale; noise NULL = dist.mode
 * (deterministic_eval).  act_out (nullable) float32[B, act_dim] = tanh-squashed action, logp_out float32[B]
 * (correct_log_prob_gaussian_tanh, sac.py:25-39); aux_out (nullable) float32[B, 3, act_dim] =
 * {a - mu, sigma, squashed}. */
int ts_sac_policy_forward(ts_workspace* ws, const float* actor, const float* obs, const float* noise, int64_t B,
                          int64_t obs_dim, int64_t act_dim, float* act_out, float* logp_out, float* aux_out,
                          ts_stream_t stream);
/* The same pass as the collector calls it (data/collector.py:735-741): mu_out / sigma_out (nullable) float32[B, act_dim] =
 * `logits` = (loc, scale) of the returned batch (sac.py:114, 125); no backward state is kept. */
int ts_sac_policy_forward_logits(ts_workspace* ws, const float* actor, const float* obs, const float* noise, int64_t B,
                                 int64_t obs_dim, int64_t act_dim, float* act_out, float* logp_out, float* mu_out,
                                 float* sigma_out, ts_stream_t stream);

/* ActorCriticOffPolicyAlgorithm._target_q + SAC._target_q_compute_value (ddpg.py:327-339, td3.py:94-102,
 * sac.py:290-296): a' ~ pi(s') with `noise`, min(Q1_old, Q2_old)(s', a') - alpha * log_prob -> out float32[B].
 * alpha = exp(*log_alpha) when log_alpha (device float32[1]) is given, else fixed_alpha. */
int ts_sac_target_q(ts_workspace* ws, const float* actor, const float* critic1_old, const float* critic2_old,
                    const float* log_alpha, double fixed_alpha, const float* obs_next, const float* noise, int64_t B,
                    int64_t obs_dim, int64_t act_dim, float* out, ts_stream_t stream);

typedef struct ts_sac_state {  /* device pointers, all float32 */
    float *actor, *actor_m, *actor_v;
    float *critic1, *critic1_m, *critic1_v;
    float *critic2, *critic2_m, *critic2_v;
    float *critic1_old, *critic2_old;
    float *log_alpha, *log_alpha_m, *log_alpha_v; /* [1] each; used when auto_alpha */
} ts_sac_state;

typedef struct ts_sac_hparams {
    double actor_lr, critic_lr, alpha_lr; /* a negative lr skips that optimizer step (gradient only) */
    double beta1, beta2, adam_eps;
    double tau;            /* Polyak coefficient (lagged_network.py:17-18); <= 0 skips the target update */
    double alpha;          /* fixed entropy coefficient when !auto_alpha */
    double target_entropy; /* AutoAlpha (sac.py:184-209) */
    int32_t auto_alpha, reserved;
} ts_sac_hparams;

/* SAC._update_with_batch (sac.py:298-336): critic 1 and 2 steps (ddpg.py:279-285), actor step with the updated
 * critics, AutoAlpha.update, Polyak update of both lagged critics.  adam_step = 1-based step of this call (all
 * optimizers advance together).  noise float32[B, act_dim] = the rsample() draws of the actor-loss policy call.
 * stats_out5 float32[5] = {actor_loss, critic1_loss, critic2_loss, alpha (after the update), alpha_loss};
 * weight_out (nullable) float32[B] = (td1 + td2) / 2, the new PER weights (sac.py:306);
 * grads_out (nullable) = the three flat gradients {critic1, critic2, actor}. */
int ts_sac_update(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const float* obs, const float* act,
                  const float* returns, const float* weight, const float* noise, int64_t B, int64_t obs_dim,
                  int64_t act_dim, const ts_sac_hparams* hp, float* stats_out5, float* weight_out, float* grads_out,
                  ts_stream_t stream);
/* The same two operations reading their rows straight from the replay buffer's columns (row b of the minibatch = row rows[b] of
 * obs_buf / act_buf / obs_next_buf, float32 [slots, dim]; rew_buf float64 [slots], terminated_buf uint8 [slots]): the gathers
 * of ReplayBuffer.__getitem__ (buffer_base.py:605-649) happen inside the input-packing kernel instead of in launches of their
 * own.  ts_sac_returns_rows = _target_q (ts_sac_target_q) + the 1-STEP return of compute_nstep_return
 * (algorithm_base.py:785-817 with n_step = 1; the arithmetic of ts_nstep_return_fused: value mask in float32, gamma and
 * reward in float64) in the same launch sequence: returns_out[b] = float(double(tq[b] * !terminated[rows[b]]) * gamma +
 * rew[rows[b]]), bit-identical to ts_sac_target_q + ts_nstep_return_fused.  ts_sac_update_rows = ts_sac_update. */
int ts_sac_returns_rows(ts_workspace* ws, const float* actor, const float* critic1_old, const float* critic2_old,
                        const float* log_alpha, double fixed_alpha, const float* obs_next_buf, const double* rew_buf,
                        const uint8_t* terminated_buf, const int64_t* rows, const float* noise, int64_t B, int64_t obs_dim,
                        int64_t act_dim, double gamma, float* returns_out, ts_stream_t stream);
int ts_sac_update_rows(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const float* obs_buf, const float* act_buf,
                       const int64_t* rows, const float* returns, const float* weight, const float* noise, int64_t B,
                       int64_t obs_dim, int64_t act_dim, const ts_sac_hparams* hp, float* stats_out5, float* weight_out,
                       ts_stream_t stream);
/* One call per SAC update with n_step = 1 (round 6): OffPolicyAlgorithm.update's _preprocess_batch -> _update_with_batch
 * (algorithm_base.py:586-631; sac.py:281-336, ddpg.py:287-301, algorithm_base.py:785-817) = ts_sac_returns_rows(noise2[0]) followed
 * by ts_sac_update_rows(noise2[1]) on the same rows, bit-identical to the two calls (tests/test_gpu_sac.py).  On the one-launch
 * chains (Net[h, h] at the fused kernels' widths) three launches of the 22 go away: both input-packing passes (obs / act and
 * obs_next of the sampled rows) and, with fill_noise != 0, the noise draw -- fill_noise = 1: ts_normal_fill(noise2, 2 * B * act_dim,
 * noise_seed, noise_offset); fill_noise = 2: the two halves as draws of their own, ts_normal_fill(noise2[h], B * act_dim,
 * noise_seed, noise_offset + h) -- share ONE launch, and the 1-step return is formed inside the critic-loss launch from the lagged critics' outputs
 * (no launch of its own).  Other trunks run the two sequences back to back.  noise2: float32[2, B, act_dim] (input, or output
 * when fill_noise); returns_out: float32[B] (always written: what ts_sac_returns_rows returns); weight: nullable PER weights;
 * weight_out: nullable (td1 + td2) / 2. */
typedef struct ts_sac_replay {
    const float* obs;             /* [slots, obs_dim] */
    const float* act;             /* [slots, act_dim] */
    const float* obs_next;        /* [slots, obs_dim] */
    const double* rew;            /* [slots] */
    const uint8_t* terminated;    /* [slots] */
} ts_sac_replay;
int ts_sac_learn_rows(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const ts_sac_replay* replay,
                      const int64_t* rows, const float* weight, float* noise2, int fill_noise, uint64_t noise_seed,
                      uint64_t noise_offset, int64_t B, int64_t obs_dim, int64_t act_dim, const ts_sac_hparams* hp, double gamma,
                      float* returns_out, float* stats_out5, float* weight_out, ts_stream_t stream);

/* One phase of ts_sac_update, for data-parallel replicas (tianshou_amd/distributed.py DataParallelSAC; the reference
 * has no distributed path): phase 1 = forward / loss / backward of both critics on the local batch,
 * 2 = their Adam steps, 4 = actor forward / loss / backward against the updated critics, 8 = actor Adam step,
 * AutoAlpha.update, weight_out, Polyak.  `grads` is the exchange buffer the caller all-reduces between a "grad"
 * phase and its "apply" phase: phases 1/2 float32[2 * critic_params] = {critic1, critic2} gradients of the local
 * mean loss; phases 4/8 float32[actor_params + 1] = {actor gradient, -mean(log_prob)}.  The four calls of one update
 * use the same workspace, batch pointers and B, in the order 1, 2, 4, 8 (the workspace carries the packed inputs,
 * TD errors and policy intermediates from one phase to the next); 1, 2, 4, 8 without an exchange in between is
 * bit-identical to ts_sac_update.  stats_out5 / weight_out as in ts_sac_update (each phase writes its own slots). */
int ts_sac_update_phase(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const float* obs, const float* act,
                        const float* returns, const float* weight, const float* noise, int64_t B, int64_t obs_dim,
                        int64_t act_dim, const ts_sac_hparams* hp, int phase, float* stats_out5, float* weight_out,
                        float* grads, ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * DiscreteSAC (tianshou/algorithm/modelfree/discrete_sac.py): Categorical policy, twin critics that
 * output Q(s, .) for every action; nets of test/discrete/test_discrete_sac.py:88-97: Net(obs, [hidden, hidden]) ReLU
 * under DiscreteActor(softmax_output=False) / DiscreteCritic(last_size=n_act) (utils/net/discrete.py:27-123).
 * All three nets share one flat layout: L1 [ka + 1, hidden] | L2 [hidden + 1, hidden] | head [hidden + 1, hw]
 * (ka, hw = obs_dim, n_act rounded up to multiples of 32; last row of each block = bias; padding zero).
 * n_act in [2, 64]; hidden a multiple of 32 in [32, 2048].
 * ------------------------------------------------------------------------------------------- */

/* h_out3 = {ka, hw, parameter count of one net}. */
int ts_dsac_layout(int64_t obs_dim, int64_t n_act, int64_t hidden, int64_t* h_out3);

/* DiscreteSACPolicy.forward (discrete_sac.py:53-67) up to the Categorical: logits_out float32[B, n_act]. */
int ts_dsac_policy_forward(ts_workspace* ws, const float* actor, const float* obs, int64_t B, int64_t obs_dim,
                           int64_t n_act, int64_t hidden, float* logits_out, ts_stream_t stream);

/* _target_q (ddpg.py:327-339) with _target_q_compute_value (discrete_sac.py:147-155):
 * out[b] = sum_a p(a|s') min(Q1_old, Q2_old)(s', a) + alpha H(p(.|s')); log_alpha (device, nullable) selects
 * alpha = exp(*log_alpha), else fixed_alpha. */
int ts_dsac_target_q(ts_workspace* ws, const float* actor, const float* critic1_old, const float* critic2_old,
                     const float* log_alpha, double fixed_alpha, const float* obs_next, int64_t B, int64_t obs_dim,
                     int64_t n_act, int64_t hidden, float* out, ts_stream_t stream);

/* DiscreteSAC._update_with_batch (discrete_sac.py:157-196): critic 1 and 2 steps on (Q(s)[a] - returns)^2 * weight,
 * actor step on -(alpha H + sum_a p q).mean() with the updated critics, AutoAlpha.update(entropy) (sac.py:203-209),
 * Polyak update of both lagged critics.  State / hyper-parameters: the SAC structs.  act int64[B];
 * stats_out5 = {actor_loss, critic1_loss, critic2_loss, alpha (after the update), alpha_loss};
 * weight_out (nullable) float32[B] = (td1 + td2) / 2 (discrete_sac.py:174);
 * grads_out (nullable) = the three flat gradients {critic1, critic2, actor}. */
int ts_dsac_update(ts_workspace* ws, const ts_sac_state* st, int64_t adam_step, const float* obs, const int64_t* act,
                   const float* returns, const float* weight, int64_t B, int64_t obs_dim, int64_t n_act, int64_t hidden,
                   const ts_sac_hparams* hp, float* stats_out5, float* weight_out, float* grads_out, ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * REDQ (tianshou/algorithm/modelfree/redq.py): SAC's tanh-Gaussian policy + an ensemble of E critics whose
 * Linear layers are EnsembleLinear (utils/net/common.py:518-550), nets of test/continuous/test_redq.py:86-107 with hidden
 * [256, 256].  The ensemble's parameters are E consecutive blocks in ts_sac_layout's critic layout; actor as SAC's.
 * ------------------------------------------------------------------------------------------- */

/* _target_q (ddpg.py:327-339) + _target_q_compute_value (redq.py:248-261): a' ~ pi(s') with the supplied rsample() noise,
 * min (mean_mode 0) or mean (1) over the lagged members h_subset[0..S) (host int32, the np.random.choice draw), minus
 * alpha * log_prob -> out float32[B]. */
int ts_redq_target_q(ts_workspace* ws, const float* actor, const float* critics_old, int64_t E, const int32_t* h_subset,
                     int64_t S, int mean_mode, const float* log_alpha, double fixed_alpha, const float* obs_next,
                     const float* noise, int64_t B, int64_t obs_dim, int64_t act_dim, float* out, ts_stream_t stream);

typedef struct ts_redq_state {  /* device pointers, all float32 */
    float *actor, *actor_m, *actor_v;
    float *critics, *critics_m, *critics_v; /* E blocks each */
    float *critics_old;
    float *log_alpha, *log_alpha_m, *log_alpha_v;
} ts_redq_state;

/* REDQ._update_with_batch (redq.py:263-304): the ensemble loss mean_{e,b}((Q_e - returns)^2 weight) and ONE Adam step over
 * all members (critic_step = 1-based count of these steps); when do_actor (every actor_delay-th update) the actor step on
 * (alpha log_prob - mean_e Q_e).mean() with the updated critics and AutoAlpha.update(-log_prob) (actor_step = 1-based
 * count of actor updates, used by both optimizers); Polyak update of the lagged ensemble.  E <= 64.
 * stats_out4 = {actor_loss, critic_loss, alpha, alpha_loss}: slots 0, 2, 3 are written only when do_actor.
 * weight_out (nullable) float32[B] = mean_e td_e (redq.py:272); grads_out (nullable) = {E critic blocks, actor}. */
int ts_redq_update(ts_workspace* ws, const ts_redq_state* st, int64_t E, int64_t critic_step, int64_t actor_step,
                   int do_actor, const float* obs, const float* act, const float* returns, const float* weight,
                   const float* noise, int64_t B, int64_t obs_dim, int64_t act_dim, const ts_sac_hparams* hp,
                   float* stats_out4, float* weight_out, float* grads_out, ts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * NPG / TRPO (tianshou/algorithm/modelfree/npg.py, trpo.py) on the MuJoCo actor-critic of
 * examples/mujoco/mujoco_npg.py:103-128 (= the PPO nets: Net[h, h] tanh, unbounded Gaussian actor with a state-independent
 * sigma_param, separate critic).
 * Flat layouts: actor  L1 [k0 + 1, h] | L2 [h + 1, h] | head [h + 1, 32] (columns [0, act_dim) = mu) | log_sigma [32];
 *               critic L1 | L2 | head [h + 1, 32] (column 0 = V);  k0 = obs_dim rounded up to 32, last row of a block = bias,
 *               padding zero.  h a multiple of 32 in [32, 1024], act_dim <= 32.
 * ------------------------------------------------------------------------------------------- */

/* h_out3 = {k0, actor parameter count, critic parameter count}. */
int ts_npg_layout(int64_t obs_dim, int64_t hidden, int64_t act_dim, int64_t* h_out3);

/* The no-grad passes of NPG._preprocess_batch (npg.py:129-135, a2c.py:122-129) on obs float32[B, obs_dim]:
 * v_out[b] = V(obs_b) (nullable, needs critic); logp_out[b] = log pi(act_b | obs_b) (nullable, needs actor and act);
 * mu_out float32[B, act_dim] (nullable). */
int ts_npg_infer(ts_workspace* ws, const float* actor, const float* critic, int64_t obs_dim, int64_t hidden, int64_t act_dim,
                 const float* obs, const float* act, int64_t B, float* v_out, float* logp_out, float* mu_out,
                 ts_stream_t stream);

typedef struct ts_npg_hparams {
    double damping;            /* npg.py:118 (_MVP adds damping * v) */
    double trust_region_size;  /* NPG: actor step size (npg.py:174) */
    double max_kl;             /* TRPO (trpo.py:153-160, :179) */
    double backtrack_coeff;    /* TRPO line search (trpo.py:184) */
    double residual_tol;       /* conjugate gradients (npg.py:206, 1e-10) */
    int32_t algo;              /* 0: NPG, 1: TRPO */
    int32_t cg_iters;          /* npg.py:167 / trpo.py:150: 10 */
    int32_t max_backtracks;    /* TRPO, <= 32 */
    int32_t reserved;
} ts_npg_hparams;

/* The actor half of one minibatch (npg.py:149-177 / trpo.py:132-191): vanilla gradient of the surrogate, natural gradient
 * by conjugate gradients on Fisher-vector products, parameter step (NPG) or step size + backtracking line search (TRPO);
 * `actor` is updated in place.  logp_old: TRPO only.  stats_out3 = {actor_loss, kl(old || new), step_size (TRPO, else 0)}.
 * dbg_out (nullable) float32[3 P] = {flat gradient, F^-1 g (= -search_direction), F g + damping g}, P = actor count.
 * No host synchronisation: the early exit of conjugate gradients and the line search are decided on the device. */
int ts_npg_actor_step(ts_workspace* ws, float* actor, int64_t obs_dim, int64_t hidden, int64_t act_dim, const float* obs,
                      const float* act, const float* adv, const float* logp_old, int64_t B, const ts_npg_hparams* hp,
                      float* stats_out3, float* dbg_out, ts_stream_t stream);

/* The vanilla policy gradient alone: loss = -mean(log pi(act | obs) * weight) and its flat gradient (actor layout) --
 * Reinforce._update_with_batch (modelfree/reinforce.py:363-382, weight = the discounted returns) before Optimizer.step
 * (ts_adam_step on the flat actor vector).  loss_out float32[1], grad_out float32[actor count]. */
int ts_npg_actor_grad(ts_workspace* ws, const float* actor, int64_t obs_dim, int64_t hidden, int64_t act_dim, const float* obs,
                      const float* act, const float* weight, int64_t B, float* loss_out, float* grad_out, ts_stream_t stream);

/* One critic iteration (npg.py:180-183): vf_loss = mse_loss(returns, V(obs)), clip_grad_norm_ + Adam on the critic
 * (algorithm_base.py:484-500).  lr < 0: gradient only.  loss_out float32[1]; grad_out nullable. */
int ts_npg_critic_step(ts_workspace* ws, float* critic, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                       int64_t hidden, const float* obs, const float* returns, int64_t B, double lr, double beta1, double beta2,
                       double adam_eps, double max_grad_norm, float* loss_out, float* grad_out, ts_stream_t stream);

/* The optim_critic_iters iterations of a minibatch (npg.py:179-183: the same obs / returns every iteration) in one call:
 * iteration k takes Adam step number adam_step + k.  loss_out float32[1] / grad_out (nullable) hold the LAST iteration's
 * loss and gradient.  Hidden 64 and obs_dim <= 32: one kernel + one small sum per gradient (csrc/ts_npg_q.h). */
int ts_npg_critic_steps(ts_workspace* ws, float* critic, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                        int64_t hidden, const float* obs, const float* returns, int64_t B, int64_t iters, double lr, double beta1,
                        double beta2, double adam_eps, double max_grad_norm, float* loss_out, float* grad_out, ts_stream_t stream);

/* One minibatch of PPO._update_with_batch / A2C._update_with_batch (ppo.py:179-216, a2c.py:262-283) for ANY Net[h, h]
 * tanh actor-critic (obs_dim >= 1, hidden a multiple of 32 up to 1024, act_dim <= 32): the shapes the fused kernels behind
 * ts_ppo_update do not cover (they are specialised to hidden 64, obs <= 31, act <= 8) run on the implicit-GEMM layers
 * of ts_conv_forward / ts_conv_backward.  `params` = [actor | critic] in the ts_npg_layout order (one contiguous
 * vector, so that clip_grad_norm_ over ActorCritic (a2c.py:103-107) and Adam are one pass); adam_m / adam_v likewise.
 * obs .. v_old: the minibatch rows, already gathered ([B, obs_dim], [B, act_dim], [B]...).  global_batch >= B scales
 * the loss (data-parallel use).  hp->lr < 0: gradient only (grad_out, float32[actor count + critic count]).
 * losses_out4 = {loss, clip / actor loss, vf loss, entropy}. */
int ts_ppo_wide_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step, int64_t obs_dim,
                     int64_t hidden, int64_t act_dim, const float* obs, const float* act, const float* adv, const float* returns,
                     const float* logp_old, const float* v_old, int64_t B, int64_t global_batch, const float* adv_stats,
                     const ts_ppo_hparams* hp, float* losses_out4, float* grad_out, ts_stream_t stream);

/* The same minibatch step and the inference passes for actor-critics with ANY trunk the reference's Net / MLP builds from
 * `hidden_sizes` and one activation (utils/net/common.py:90-178, 246-369): 1 .. TS_NET_MAX_HIDDEN hidden layers of any
 * widths (stored padded to multiples of 32: a padding unit has zero weights on both sides and keeps them), activation tanh /
 * ReLU / none, actor and critic described separately (they are separate Net instances in examples/mujoco/mujoco_ppo.py:
 * 108-114).  Per layer one block [K_pad + 1, N_pad] (last row = bias) as in ts_npg_layout; actor = blocks | head
 * [h_last + 1, 32] (columns [0, act) = mu) | log_sigma[32]; critic = blocks | head [h_last + 1, 32] (column 0 = V).
 * ts_net_layout: h_out3 = {k0 (obs padded to 32), floats of an actor with this trunk, floats of a critic with this trunk}. */
#define TS_NET_MAX_HIDDEN 7
#define TS_NET_ACT_TANH 0
#define TS_NET_ACT_RELU 1
#define TS_NET_ACT_NONE 2
typedef struct ts_net_desc {
    int64_t obs_dim;
    int32_t n_hidden;                  /* 1 .. TS_NET_MAX_HIDDEN */
    int32_t activation;                /* TS_NET_ACT_* after every hidden layer */
    int64_t hidden[TS_NET_MAX_HIDDEN]; /* widths as configured (1 .. 1024 each) */
    int64_t flags;                     /* actor only: TS_NET_CONDITIONED_SIGMA -- sigma = exp(clamp(Linear(h), -20, 2)) per sample
                                          (ContinuousActorProbabilistic(conditioned_sigma=True), utils/net/continuous.py:212-234):
                                          the head block's columns [16, 16 + act) are that Linear layer (act_dim <= 16), the
                                          trailing log_sigma[32] block is unused and stays zero */
    double max_action;                 /* actor only (round 6): > 0 = ContinuousActorProbabilistic(unbounded=False), the constructor
                                          default (utils/net/continuous.py:194, 230-231): mu = max_action * tanh(Linear(h)), and
                                          the gradient goes back through it; 0 = unbounded */
    double ln_eps;                     /* TS_NET_LAYERNORM: the LayerNorm modules' eps (0 = torch's default 1e-5) */
} ts_net_desc;
#define TS_NET_CONDITIONED_SIGMA 1
/* flags | TS_NET_LAYERNORM (round 6): MLP(norm_layer=nn.LayerNorm) -- every hidden layer is Linear -> LayerNorm(width) ->
 * activation (utils/net/common.py:25-39, 99-137; elementwise affine, biased variance over the layer's configured width).  The
 * flat vector then holds, behind every hidden layer's wb block, gamma[N_pad] | beta[N_pad] (padding entries zero), and the
 * gradients follow the same layout.  PPO / A2C entry points (ts_ppo_net_infer, ts_ppo_net_step); not the NPG / TRPO ones. */
#define TS_NET_LAYERNORM 2
int ts_net_layout(const ts_net_desc* net, int64_t act_dim, int64_t* h_out3);
int ts_ppo_net_infer(ts_workspace* ws, const float* actor, const float* critic, const ts_net_desc* actor_net,
                     const ts_net_desc* critic_net, int64_t act_dim, const float* obs, const float* act, int64_t B,
                     float* v_out, float* logp_out, float* mu_out, ts_stream_t stream);
int ts_ppo_net_step(ts_workspace* ws, float* params, float* adam_m, float* adam_v, int64_t adam_step,
                    const ts_net_desc* actor_net, const ts_net_desc* critic_net, int64_t act_dim, const float* obs,
                    const float* act, const float* adv, const float* returns, const float* logp_old, const float* v_old,
                    int64_t B, int64_t global_batch, const float* adv_stats, const ts_ppo_hparams* hp, float* losses_out4,
                    float* grad_out, ts_stream_t stream);
/* NPG / TRPO (modelfree/npg.py:123-224, trpo.py:123-214) on the same generic trunks (round 6): ts_npg_actor_step /
 * ts_npg_critic_steps with a ts_net_desc in place of the single hidden width -- any 1 .. TS_NET_MAX_HIDDEN hidden layers of any
 * widths, tanh / ReLU / no activation (the actor: unbounded, state-independent sigma, i.e. flags = 0 and max_action = 0).  Flat
 * layouts as ts_net_layout (critic: its first h_out3[2] entries -- trunk + head -- without the log_sigma block).  The
 * Fisher-vector product is the Gauss-Newton form J^T diag(1 / sigma^2) J v / B + 2 v_sigma with one forward-mode pass (layer i:
 * d_i = (d_{i-1} W_i + h_{i-1} V_i + bv_i) act'(h_i)) and one reverse pass per product; inference: ts_ppo_net_infer. */
int ts_npg_net_actor_step(ts_workspace* ws, float* actor, const ts_net_desc* net, int64_t act_dim, const float* obs, const float* act,
                          const float* adv, const float* logp_old, int64_t B, const ts_npg_hparams* hp, float* stats_out3,
                          float* dbg_out, ts_stream_t stream);
int ts_npg_net_critic_steps(ts_workspace* ws, float* critic, float* adam_m, float* adam_v, int64_t adam_step, const ts_net_desc* net,
                            const float* obs, const float* returns, int64_t B, int64_t iters, double lr, double beta1, double beta2,
                            double adam_eps, double max_grad_norm, float* loss_out, float* grad_out, ts_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * TD3 / DDPG: ContinuousActorDeterministic (utils/net/continuous.py:26-85) + the SAC critics,
 * nets of examples/mujoco/mujoco_td3.py:85-103 / mujoco_ddpg.py
 * ------------------------------------------------------------------------------------------- */

/* Flat vectors: actor L1 [ka + 1, 256] | L2 [257, 256] | head [257, 32] (columns [0, act_dim) = last);
 * critics as in ts_sac_layout.  h_out4 = {ka, kc, actor count, critic count}. */
int ts_td3_layout(int64_t obs_dim, int64_t act_dim, int64_t* h_out4);
int ts_td3_layout_h(int64_t obs_dim, int64_t act_dim, int64_t hidden, int64_t* h_out4);

/* ContinuousDeterministicPolicy.forward (ddpg.py:162-180): act = max_action * tanh(actor(obs)). */
int ts_td3_policy_forward(ts_workspace* ws, const float* actor, const float* obs, int64_t B, int64_t obs_dim,
                          int64_t act_dim, double max_action, float* act_out, ts_stream_t stream);

/* _target_q (ddpg.py:327-339) with the lagged actor: DDPG (critic2_old NULL, noise NULL; ddpg.py:397-399) or
 * TD3 (td3.py:94-102, 190-202): a' = actor_old(s') + clamp(noise * policy_noise, +-noise_clip) (noise_clip <= 0:
 * no clamp), min of the two lagged critics.  noise float32[B, act_dim] = the torch.randn draws. */
int ts_td3_target_q(ts_workspace* ws, const float* actor_old, const float* critic1_old, const float* critic2_old,
                    const float* obs_next, const float* noise, int64_t B, int64_t obs_dim, int64_t act_dim,
                    double max_action, double policy_noise, double noise_clip, float* out, ts_stream_t stream);

typedef struct ts_td3_state {  /* device pointers, float32; critic2* NULL = DDPG */
    float *actor, *actor_m, *actor_v;
    float *critic1, *critic1_m, *critic1_v;
    float *critic2, *critic2_m, *critic2_v;
    float *actor_old, *critic1_old, *critic2_old;
} ts_td3_state;

typedef struct ts_td3_hparams {
    double actor_lr, critic_lr; /* negative: gradient only */
    double beta1, beta2, adam_eps;
    double tau, max_action;
    int32_t update_actor; /* this call updates the actor and the lagged networks (td3.py:215, _cnt % freq == 0) */
    int32_t reserved;
} ts_td3_hparams;

/* TD3._update_with_batch (td3.py:204-226) / DDPG._update_with_batch (ddpg.py:401-411): critic step(s); when
 * hp->update_actor, actor step on -Q1(s, pi(s)).mean() with the updated critic and the Polyak updates.
 * critic_step / actor_step = 1-based Adam steps of this call (the actor's counts only its own updates).
 * stats_out3 = {actor_loss (written only when the actor is updated), critic1_loss, critic2_loss};
 * weight_out (nullable) = (td1 + td2) / 2 or td1; grads_out (nullable) = {critic1, critic2, actor} gradients. */
int ts_td3_update(ts_workspace* ws, const ts_td3_state* st, int64_t critic_step, int64_t actor_step, const float* obs,
                  const float* act, const float* returns, const float* weight, int64_t B, int64_t obs_dim,
                  int64_t act_dim, const ts_td3_hparams* hp, float* stats_out3, float* weight_out, float* grads_out,
                  ts_stream_t stream);

/* ---- data-parallel exchange ------------------------------------------------------------------
 * One process per GPU; the reference has no distributed path (its only multi-GPU mechanism is single-process
 * nn.DataParallel, tianshou/utils/net/common.py:473-515).  In-place sum all-reduce of a flat fp32 buffer over RCCL
 * (xGMI inside a node), issued on the caller's stream: ordered after the kernel that wrote the buffer
 * (ts_ppo_grad / ts_sac_update_phase ...) and before the one that consumes it, without a host synchronisation.
 * RCCL is resolved at run time; TS_ERR_UNSUPPORTED when librccl.so.1 cannot be loaded.
 *   rank 0: ts_allreduce_unique_id(id) -> the host ships the 128 bytes to every rank (any rendezvous it has)
 *   all   : ts_allreduce_init(id, rank, world, device, &comm)           (collective: every rank must call it)
 *   all   : ts_allreduce(comm, buf, n, stream) per exchange;  ts_allreduce_destroy(comm) at the end.
 * Hosts inside PyTorch can use torch.distributed (backend "nccl" = RCCL) instead: same collective, same result. */
typedef struct ts_comm ts_comm;
int ts_allreduce_unique_id(uint8_t* h_id128);
int ts_allreduce_init(const uint8_t* h_id128, int64_t rank, int64_t world, int device, ts_comm** out);
int ts_allreduce(ts_comm* comm, float* buf, int64_t n, ts_stream_t stream);
int ts_allreduce_destroy(ts_comm* comm);
/* *world = the rank count the communicator was created with; *rccl_ranks = what RCCL itself reports for it (ncclCommCount;
 * 0 for a communicator without an RCCL side, see ts_allreduce_from_small). */
int ts_allreduce_ranks(const ts_comm* comm, int64_t* world, int64_t* rccl_ranks);

/* One-shot all-reduce for payloads of at most 16,384 floats (64 KB) -- the [gradient | loss parts] vector of a PPO
 * minibatch step is 44 KB and sits on the critical path between two ~55 us kernels, where a ring collective's
 * latency is what matters.  Every rank owns one device buffer which the others map through HIP IPC (xGMI peer
 * access inside a node; also valid between processes sharing one GPU); a call is ONE single-workgroup kernel on
 * `stream`: publish (system-scope write-through stores + flag), poll the peers' flags, sum the payloads in rank order
 * 0 .. W-1 -- every replica obtains bit-identical sums (for W = 2 identical to ts_allreduce as well).
 *   every rank: ts_allreduce_small_create(device, max_floats, &comm, handle64)   -> 64-byte IPC handle of its buffer
 *   host      : all-gather the handles (any rendezvous) into h_handles[world * 64], rank order
 *   every rank: ts_allreduce_small_connect(comm, h_handles, rank, world)          (at most 8 ranks)
 *   per step  : ts_allreduce_small(comm, buf, n, stream)      in place; every rank must call it the same number of times
 *   optional  : ts_allreduce_small_status(comm, stream)       synchronises; error if a peer never arrived (bounded spin)
 * Replaces the per-step ncclAllReduce of the data-parallel PPO update (reference: single-process nn.DataParallel,
 * tianshou/utils/net/common.py:473-515). */
typedef struct ts_small_comm ts_small_comm;
int ts_allreduce_small_create(int device, int64_t max_floats, ts_small_comm** out, uint8_t* h_handle64);
int ts_allreduce_small_connect(ts_small_comm* comm, const uint8_t* h_handles, int64_t rank, int64_t world);
int ts_allreduce_small(ts_small_comm* comm, float* buf, int64_t n, ts_stream_t stream);
int ts_allreduce_small_status(ts_small_comm* comm, ts_stream_t stream);
int ts_allreduce_small_destroy(ts_small_comm* comm);
int64_t ts_allreduce_small_capacity(const ts_small_comm* comm);
/* ts_allreduce (and therefore ts_ppo_dp_step) takes the one-shot path for payloads within `small`'s capacity once it is
 * attached (NULL detaches; not owned: destroy it separately, after the communicator).  ts_allreduce_from_small makes a
 * communicator that has ONLY the one-shot path (no RCCL; larger payloads are refused) -- e.g. ranks sharing one GPU. */
int ts_allreduce_attach_small(ts_comm* comm, ts_small_comm* small);
int ts_allreduce_from_small(ts_small_comm* small, ts_comm** out);

/* One data-parallel PPO / A2C gradient step in one call: ts_ppo_grad on the local minibatch shard ->
 * ts_allreduce of step_buf[0 .. P + 4) (gradient sums / global_batch and the four loss parts; skipped when comm is NULL =
 * one rank) -> ts_ppo_apply (clip by the global norm + Adam step number `adam_step`), all on `stream`, no host
 * synchronisation in between.  step_buf: device float32[>= P + 4], 16-byte aligned, one row per step if the caller
 * wants to keep the loss parts (step_buf[P .. P + 4) = (0, clip, vf, ent) summed over the ranks: divide ent by the world
 * size and compose loss = clip + vf_coef * vf - ent_coef * ent).  Arguments as for ts_ppo_grad / ts_ppo_apply. */
int ts_ppo_dp_step(ts_workspace* ws, ts_comm* comm, float* params, float* adam_m, float* adam_v, int64_t adam_step,
                   int64_t obs_dim, int64_t act_dim, const float* rec, int64_t n, const int64_t* perm_rows, int64_t n_rows,
                   int64_t global_batch, const float* adv_stats, const ts_ppo_hparams* hp, float* step_buf,
                   ts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TSENGINE_H */
