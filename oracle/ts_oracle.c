/*
 * ts_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle + reported CPU baseline).
 *
 * Plain-C, single-threaded restatement of the compiled (numba @njit) leaf functions on
 * the Batch -> learn() hot path of thu-ml/tianshou 2.0.1, plus the thin NumPy glue that
 * surrounds them.  Every function cites the reference file:line it follows (paths are
 * relative to the reference checkout).  Nothing under tianshou_amd/ may link, load or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Pinned (tests/test_oracle_golden.py) against
 *   - the literal known-answer vectors of the reference's own tests
 *     (test/base/test_returns.py, test/base/test_buffer.py) and
 *   - outputs of the unmodified reference executed in the authoring container
 *     (oracle/gen_golden.py -> the .npz files in tests/golden).
 *
 * Arithmetic notes
 *   - numba types float32-array (op) float64-scalar as float64, so all value math is
 *     done in double here, exactly one rounding per source-level operation.  Build with
 *     -ffp-contract=off so that "a + b * c" is mul-then-add as in numba (no FMA).
 *   - index math follows Python floor-modulo semantics (pymod below).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int64_t pymod(int64_t a, int64_t m) {
    int64_t r = a % m;
    return (r < 0) ? r + m : r;
}

/* ------------------------------------------------------------------------------------
 * _gae                      tianshou/algorithm/algorithm_base.py:1085-1140
 *   returns = zeros; delta = rew + v_s_ * gamma - v_s;
 *   discount = (1.0 - end_flag) * (gamma * gae_lambda);
 *   for i reversed: gae = delta[i] + discount[i] * gae; returns[i] = gae
 * ---------------------------------------------------------------------------------- */
void oracle_gae(const double* v_s, const double* v_s_, const double* rew,
                const uint8_t* end_flag, int64_t n, double gamma, double gae_lambda,
                double* adv_out) {
    const double gl = gamma * gae_lambda;
    double gae = 0.0;
    for (int64_t i = n - 1; i >= 0; --i) {
        const double delta = rew[i] + v_s_[i] * gamma - v_s[i];
        const double discount = (1.0 - (double)(end_flag[i] != 0)) * gl;
        gae = delta + discount * gae;
        adv_out[i] = gae;
    }
}

/* ------------------------------------------------------------------------------------
 * compute_episodic_return    tianshou/algorithm/algorithm_base.py:653-719
 *   v_s_ = v_s_ * value_mask(buffer, indices)      (:711, value_mask = ~terminated :651)
 *   end_flag = terminated | truncated               (:714)
 *   end_flag[isin(indices, unfinished_index())] = True   (:715)
 *   advantage = _gae(...); returns = advantage + v_s     (:716-717)
 * `terminated`/`truncated` are the *batch* arrays (already gathered at `indices`);
 * value_mask gathers buffer.terminated[indices], which is the same array.
 * `unfinished` holds buffer slots; position p is cut iff indices[p] is among them.
 * ---------------------------------------------------------------------------------- */
void oracle_compute_episodic_return(const double* v_s, const double* v_s_,
                                    const double* rew, const uint8_t* terminated,
                                    const uint8_t* truncated, const int64_t* indices,
                                    int64_t n, const int64_t* unfinished,
                                    int64_t n_unfinished, double gamma, double gae_lambda,
                                    double* returns_out, double* adv_out) {
    double* vnext = (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    uint8_t* end = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
    /* np.isin(indices, unfinished) (:715): NumPy sorts / hashes, it does not compare every pair.  Same result
     * here with a sorted copy of `unfinished` and a binary search (O((n + E) log E) instead of O(n E): the timed
     * CPU baseline of bench.py must not be penalised by the restatement). */
    int64_t* unf = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_unfinished > 0 ? n_unfinished : 1));
    for (int64_t k = 0; k < n_unfinished; ++k) {            /* insertion sort: `unfinished` arrives ascending */
        int64_t v = unfinished[k], j = k;
        while (j > 0 && unf[j - 1] > v) { unf[j] = unf[j - 1]; --j; }
        unf[j] = v;
    }
    for (int64_t i = 0; i < n; ++i) {
        vnext[i] = v_s_[i] * (double)(terminated[i] == 0);
        uint8_t e = (uint8_t)((terminated[i] != 0) | (truncated[i] != 0));
        if (!e && n_unfinished > 0) {
            int64_t lo = 0, hi = n_unfinished - 1;
            const int64_t key = indices[i];
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (unf[mid] < key) lo = mid + 1; else hi = mid; }
            e = (uint8_t)(unf[lo] == key);
        }
        end[i] = e;
    }
    free(unf);
    oracle_gae(v_s, vnext, rew, end, n, gamma, gae_lambda, adv_out);
    for (int64_t i = 0; i < n; ++i) returns_out[i] = adv_out[i] + v_s[i];
    free(vnext);
    free(end);
}

/* ------------------------------------------------------------------------------------
 * episode_mc_return_to_go   tianshou/algorithm/algorithm_base.py:1143-1157
 * ---------------------------------------------------------------------------------- */
void oracle_episode_mc_return_to_go(const double* rewards, int64_t n, double gamma,
                                    double* out) {
    if (n <= 0) return;
    out[n - 1] = rewards[n - 1];
    for (int64_t j = n - 2; j >= 0; --j) out[j] = rewards[j] + gamma * out[j + 1];
}

/* ------------------------------------------------------------------------------------
 * _nstep_return             tianshou/algorithm/algorithm_base.py:1160-1222
 *   gamma_buffer[i] = gamma_buffer[i-1] * gamma                        (:1203-1205)
 *   for n = N-1..0: now = idx[n]; where end_flag[now]: gammas = n+1, mc = 0;
 *                   mc = rew[now] + gamma * mc                          (:1214-1218)
 *   out = target_q * gamma_buffer[gammas] + mc                          (:1220-1222)
 * target_q_IA is float32 in the reference (to_numpy of a torch f32 tensor, :796).
 * ---------------------------------------------------------------------------------- */
void oracle_nstep_return(const double* rew_B, const uint8_t* end_flag_B,
                         const float* target_q_IA, const int64_t* stacked_indices_NI,
                         int64_t I, int64_t A, int64_t N, double gamma, double* out_IA) {
    double* gamma_buffer = (double*)malloc(sizeof(double) * (size_t)(N + 1));
    gamma_buffer[0] = 1.0;
    for (int64_t i = 1; i <= N; ++i) gamma_buffer[i] = gamma_buffer[i - 1] * gamma;
    for (int64_t i = 0; i < I; ++i) {
        int64_t gammas = N;
        for (int64_t a = 0; a < A; ++a) out_IA[i * A + a] = 0.0;
        for (int64_t n = N - 1; n >= 0; --n) {
            const int64_t now = stacked_indices_NI[n * I + i];
            if (end_flag_B[now]) {
                gammas = n + 1;
                for (int64_t a = 0; a < A; ++a) out_IA[i * A + a] = 0.0;
            }
            for (int64_t a = 0; a < A; ++a) {
                const double t = gamma * out_IA[i * A + a];
                out_IA[i * A + a] = rew_B[now] + t;
            }
        }
        for (int64_t a = 0; a < A; ++a) {
            const double q = (double)target_q_IA[i * A + a] * gamma_buffer[gammas];
            out_IA[i * A + a] = q + out_IA[i * A + a];
        }
    }
    free(gamma_buffer);
}

/* ------------------------------------------------------------------------------------
 * _prev_index               tianshou/data/buffer/manager.py:311-336
 *   index %= offset[-1]; per sub-buffer [start,end): cur_len = max(1, len);
 *   subind = (index - start - 1) % cur_len
 *   end_flag = done[subind + start] | (subind + start == last)
 *   prev = (subind + end_flag) % cur_len + start
 * The reference loops over sub-buffers with masks (O(E*I)); per element this is the
 * same as locating the element's sub-buffer.  Indices matched by no sub-buffer keep the
 * zeros_like() value 0, as in the reference.
 * ---------------------------------------------------------------------------------- */
static int64_t find_sub(const int64_t* offset, int64_t E, int64_t idx) {
    for (int64_t e = 0; e < E; ++e)
        if (offset[e] <= idx && idx < offset[e + 1]) return e;
    return -1;
}

void oracle_prev_index(const int64_t* index, int64_t I, const int64_t* offset, int64_t E,
                       const uint8_t* done, const int64_t* last_index,
                       const int64_t* lengths, int64_t* out) {
    for (int64_t i = 0; i < I; ++i) {
        const int64_t idx = pymod(index[i], offset[E]);
        const int64_t e = find_sub(offset, E, idx);
        if (e < 0) { out[i] = 0; continue; }
        const int64_t start = offset[e];
        const int64_t cur_len = lengths[e] > 1 ? lengths[e] : 1;
        const int64_t subind = pymod(idx - start - 1, cur_len);
        const int64_t end_flag = (done[subind + start] != 0) | (subind + start == last_index[e]);
        out[i] = pymod(subind + end_flag, cur_len) + start;
    }
}

/* _next_index               tianshou/data/buffer/manager.py:339-363
 *   end_flag = done[subind] | (subind == last)
 *   next = (subind - start + 1 - end_flag) % cur_len + start                          */
void oracle_next_index(const int64_t* index, int64_t I, const int64_t* offset, int64_t E,
                       const uint8_t* done, const int64_t* last_index,
                       const int64_t* lengths, int64_t* out) {
    for (int64_t i = 0; i < I; ++i) {
        const int64_t idx = pymod(index[i], offset[E]);
        const int64_t e = find_sub(offset, E, idx);
        if (e < 0) { out[i] = 0; continue; }
        const int64_t start = offset[e];
        const int64_t cur_len = lengths[e] > 1 ? lengths[e] : 1;
        const int64_t end_flag = (done[idx] != 0) | (idx == last_index[e]);
        out[i] = pymod(idx - start + 1 - end_flag, cur_len) + start;
    }
}

/* ------------------------------------------------------------------------------------
 * unfinished_index          tianshou/data/buffer/buffer_base.py:314-317 (per sub-buffer)
 *                           tianshou/data/buffer/manager.py:85-91 (concatenate + offset)
 *   last = (insertion_idx - 1) % size if size else 0;  [last] if size and not done[last]
 * `last_index[e]` (manager.py:176) is exactly offset[e] + last whenever size > 0.
 * Returns the number of entries written.
 * ---------------------------------------------------------------------------------- */
int64_t oracle_unfinished_index(const int64_t* offset, int64_t E, const uint8_t* done,
                                const int64_t* last_index, const int64_t* lengths,
                                int64_t* out) {
    int64_t k = 0;
    (void)offset;
    for (int64_t e = 0; e < E; ++e)
        if (lengths[e] > 0 && !done[last_index[e]]) out[k++] = last_index[e];
    return k;
}

/* ------------------------------------------------------------------------------------
 * sample_indices(0)         tianshou/data/buffer/manager.py:216-234 with
 *                           tianshou/data/buffer/buffer_base.py:518-525
 *   per sub-buffer: [insertion_idx, size) ++ [0, insertion_idx), + offset, concatenated.
 *   insertion_idx[e] is the child's _insertion_idx (next write slot, modulo maxsize).
 * Returns the number of entries written (= sum(lengths)).
 * ---------------------------------------------------------------------------------- */
int64_t oracle_sample_indices_all(const int64_t* offset, int64_t E, const int64_t* lengths,
                                  const int64_t* insertion_idx, int64_t* out) {
    int64_t k = 0;
    for (int64_t e = 0; e < E; ++e) {
        for (int64_t j = insertion_idx[e]; j < lengths[e]; ++j) out[k++] = offset[e] + j;
        for (int64_t j = 0; j < insertion_idx[e] && j < lengths[e]; ++j) out[k++] = offset[e] + j;
    }
    return k;
}

/* ------------------------------------------------------------------------------------
 * SegmentTree._setitem      tianshou/data/utils/segtree.py:95-101
 *   tree[index] = value; while index[0] > 1: index //= 2;
 *                                          tree[index] = tree[2*index] + tree[2*index+1]
 * `index` already includes +bound (segtree.py:52).  Duplicate leaves: later wins (NumPy
 * fancy assignment order).  All leaves sit at the same depth, so index[0] > 1 is the
 * common level counter.  `index` is mutated in place by the reference (index //= 2);
 * here a private copy is used because callers never read it back.
 * ---------------------------------------------------------------------------------- */
void oracle_segtree_setitem(double* tree, const int64_t* index, const double* value,
                            int64_t K) {
    if (K <= 0) return;
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
    for (int64_t k = 0; k < K; ++k) { idx[k] = index[k]; tree[idx[k]] = value[k]; }
    while (idx[0] > 1) {
        for (int64_t k = 0; k < K; ++k) idx[k] /= 2;
        for (int64_t k = 0; k < K; ++k) tree[idx[k]] = tree[idx[k] * 2] + tree[idx[k] * 2 + 1];
    }
    free(idx);
}

/* SegmentTree._reduce       tianshou/data/utils/segtree.py:104-116 */
double oracle_segtree_reduce(const double* tree, int64_t start, int64_t end) {
    double result = 0.0;
    while (end - start > 1) {
        if (pymod(start, 2) == 0) result += tree[start + 1];
        start = (start >= 0) ? start / 2 : -((-start + 1) / 2);
        if (pymod(end, 2) == 1) result += tree[end - 1];
        end = end / 2;
    }
    return result;
}

/* SegmentTree._get_prefix_sum_idx   tianshou/data/utils/segtree.py:119-134
 *   index = 1; while index[0] < bound: index *= 2; lsons = sums[index];
 *   direct = lsons < value; value -= lsons * direct; index += direct;  index -= bound
 * The reference mutates `value` in place (segtree.py:131); so does this port.          */
void oracle_segtree_prefix_sum_idx(double* value, int64_t K, int64_t bound,
                                   const double* sums, int64_t* out) {
    for (int64_t k = 0; k < K; ++k) {
        int64_t index = 1;
        while (index < bound) {
            index *= 2;
            const double lsons = sums[index];
            const int direct = lsons < value[k];
            value[k] -= lsons * (double)direct;
            index += direct;
        }
        out[k] = index - bound;
    }
}

/* ------------------------------------------------------------------------------------
 * PrioritizedReplayBuffer.get_weight + __getitem__ normalisation
 *                           tianshou/data/buffer/prio.py:69-79, 104-106
 *   w = (tree[index + bound] / min_prio) ** (-beta);  w /= max(w) if weight_norm
 * ---------------------------------------------------------------------------------- */
void oracle_per_get_weight(const double* tree, int64_t bound, const int64_t* index,
                           int64_t K, double min_prio, double beta, int weight_norm,
                           double* out) {
    double mx = -INFINITY;
    for (int64_t k = 0; k < K; ++k) {
        out[k] = pow(tree[index[k] + bound] / min_prio, -beta);
        if (out[k] > mx) mx = out[k];
    }
    if (weight_norm)
        for (int64_t k = 0; k < K; ++k) out[k] = out[k] / mx;
}

/* PrioritizedReplayBuffer.update_weight   tianshou/data/buffer/prio.py:81-90
 *   weight = |new_weight| + eps; tree[index] = weight ** alpha;
 *   max_prio = max(max_prio, weight.max()); min_prio = min(min_prio, weight.min())
 * new_weight is the TD error (float32 in DQN, dqn.py:401); eps = float32 machine eps.
 * np.abs(f32) + python-float eps stays float32 under NumPy-2 promotion, and
 * float32 ** python-float stays float32 as well; the f32 result is then written into
 * the float64 tree.  `f32_math` selects that behaviour (1) or pure double math (0, what
 * a float64 new_weight would give).                                                    */
void oracle_per_update_weight(double* tree, int64_t bound, const int64_t* index,
                              const double* new_weight, int64_t K, double alpha,
                              double eps, int f32_math, double* max_prio, double* min_prio) {
    double* w = (double*)malloc(sizeof(double) * (size_t)(K > 0 ? K : 1));
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(K > 0 ? K : 1));
    double* val = (double*)malloc(sizeof(double) * (size_t)(K > 0 ? K : 1));
    for (int64_t k = 0; k < K; ++k) {
        if (f32_math) {
            const float a = fabsf((float)new_weight[k]) + (float)eps;
            w[k] = (double)a;
            val[k] = (double)powf(a, (float)alpha);
        } else {
            w[k] = fabs(new_weight[k]) + eps;
            val[k] = pow(w[k], alpha);
        }
        idx[k] = index[k] + bound;
        if (w[k] > *max_prio) *max_prio = w[k];
        if (w[k] < *min_prio) *min_prio = w[k];
    }
    oracle_segtree_setitem(tree, idx, val, K);
    free(w);
    free(idx);
    free(val);
}

/* ------------------------------------------------------------------------------------
 * RunningMeanStd.update     tianshou/utils/statistics.py:99-114
 *   batch_mean = mean(x); batch_var = var(x) (population); parallel-variance merge.
 * NumPy's mean/var use pairwise summation; a plain Kahan-free double loop differs by
 * O(1e-16) relative, far below the 1e-5 parity tolerance.
 * state = {mean, var, count}
 * ---------------------------------------------------------------------------------- */
void oracle_rms_update(double* state, const double* x, int64_t n) {
    if (n <= 0) return;
    long double s = 0.0L;
    for (int64_t i = 0; i < n; ++i) s += x[i];
    const double batch_mean = (double)(s / (long double)n);
    long double q = 0.0L;
    for (int64_t i = 0; i < n; ++i) {
        const long double d = (long double)x[i] - (long double)batch_mean;
        q += d * d;
    }
    const double batch_var = (double)(q / (long double)n);
    const double batch_count = (double)n;
    const double delta = batch_mean - state[0];
    const double total_count = state[2] + batch_count;
    const double new_mean = state[0] + delta * batch_count / total_count;
    const double m_a = state[1] * state[2];
    const double m_b = batch_var * batch_count;
    const double m_2 = m_a + m_b + delta * delta * state[2] * batch_count / total_count;
    state[0] = new_mean;
    state[1] = m_2 / total_count;
    state[2] = total_count;
}

/* ReplayBufferManager.add index / episode bookkeeping (tianshou/data/buffer/manager.py:131-198) with
 * ReplayBuffer._update_state_pre_add (data/buffer/buffer_base.py:360-418) for every entry, in order.
 * State per sub-buffer e: insertion[e] (next write slot, relative), lengths[e], last_index[e] (global),
 * ep_return[e], ep_len[e], ep_start[e] (relative).  Outputs per entry: global insertion index, episode
 * return / length (0 unless done), global episode start index.  Also scatters rew / flags / done. */
void oracle_buffer_add(const int64_t* buffer_ids, int64_t K, const double* rew, const uint8_t* terminated,
                       const uint8_t* truncated, const int64_t* offset, int64_t* insertion, int64_t* lengths,
                       int64_t* last_index, double* ep_return, int64_t* ep_len, int64_t* ep_start,
                       double* rew_B, uint8_t* terminated_B, uint8_t* truncated_B, uint8_t* done_B,
                       int64_t* index_out, double* ep_return_out, int64_t* ep_len_out, int64_t* ep_start_out) {
    for (int64_t k = 0; k < K; ++k) {
        const int64_t e = buffer_ids ? buffer_ids[k] : k;
        const int64_t maxsize = offset[e + 1] - offset[e];
        const int done = terminated[k] || truncated[k];                 /* manager.py:150 */
        const int64_t cur = insertion[e];                               /* buffer_base.py:381 */
        lengths[e] = lengths[e] + 1 < maxsize ? lengths[e] + 1 : maxsize;   /* :382 */
        insertion[e] = (cur + 1) % maxsize;                             /* :383 */
        ep_return[e] += rew[k];                                         /* :385 */
        ep_len[e] += 1;                                                 /* :386 */
        index_out[k] = cur + offset[e];                                 /* manager.py:170 */
        ep_return_out[k] = done ? ep_return[e] : 0.0;                   /* buffer_base.py:397-410 */
        ep_len_out[k] = done ? ep_len[e] : 0;
        ep_start_out[k] = ep_start[e] + offset[e];                      /* manager.py:171 */
        if (done) { ep_return[e] = 0.0; ep_len[e] = 0; ep_start[e] = insertion[e]; }   /* :414-418 */
        last_index[e] = cur + offset[e];                                /* manager.py:176 */
        rew_B[index_out[k]] = rew[k];                                   /* manager.py:180: _meta[idx] = batch */
        terminated_B[index_out[k]] = terminated[k] != 0;
        truncated_B[index_out[k]] = truncated[k] != 0;
        done_B[index_out[k]] = (uint8_t)done;
    }
}
