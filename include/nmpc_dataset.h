/* nmpc_dataset.h -- C-ABI of the device-resident training database ("DAgger
 * aggregation ... and normalisation"), the data side of the learning update (libnmpc_hip.so).
 *
 * Replaces the arithmetic of the reference's `Database` for rows that already live in HBM (rollout
 * states from nmpc_rollout_batch, expert actions from nmpc_solve_batch):
 *   Database.append               DAgger/utils/database.py:105-154   ring buffer of `limit` rows
 *   Database.calc_input_mean_std  DAgger/utils/database.py:208-255   per-column mean / population std
 *   Database.__getitem__          DAgger/utils/database.py:54-84     x = [state_norm, goal], y = action
 *   `.float()` of the batch       DAgger/utils/train_locosafedagger.py:95
 * The ring's bookkeeping (start, length) is two integers and stays with the caller; these entry
 * points move and reduce the rows.  Row storage is fp32 row-major [row][column]; statistics are
 * float64 like the reference's (numpy) ones.  Calls are stream-ordered, never allocate and never
 * synchronise.  Return values: NMPC_OK / NMPC_E_* of nmpc.h; nmpc_dataset_last_error() explains the
 * last failure on the calling thread. */
#ifndef NMPC_DATASET_H
#define NMPC_DATASET_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *nmpc_dataset_last_error(void);

/* Database.append (database.py:123-141) for one field: row i of src goes to ring slot
 * (first_slot + i) % limit, first_slot = (start + length) % limit BEFORE the call.  When n > limit the
 * reference's loop overwrites the early rows; only the surviving last `limit` rows are written. */
int nmpc_ring_append(const float *src, int row_len, long long n, float *ring, long long limit,
                     long long first_slot, void *stream);

/* np.mean(rows, axis=0), np.std(rows, axis=0) (database.py:220-221): two passes in float64 over
 * data[rows][cols] (cols <= 64), summed in a fixed order (results are reproducible run to run).
 * mean, std: cols doubles each.  scratch: nmpc_column_stats_scratch(cols) doubles. */
size_t nmpc_column_stats_scratch(int cols);
int nmpc_column_stats(const float *data, long long rows, int cols, double *mean, double *std,
                      double *scratch, void *stream);

/* Batch assembly = Database.__getitem__ over idx[0..n_idx), cast to fp32:
 *   x[i] = [ s[0..s_first), (s[s_first..) - s_mean) / s_std,  (g - g_mean) / g_std ]   in float64, then fp32
 *   y[i] = action
 * with s = states[idx[i]], g = goals[idx[i]].  s_mean/s_std NULL: states raw (norm_input False);
 * g_mean/g_std NULL: goals raw (the 'vc' goal type: mean 0, std 1, database.py:240-243).  The reference
 * leaves the phase column unnormalised: s_first = 1 (database.py:227-230).  A zero std divides by zero
 * exactly as numpy does (inf / nan).  x: [n_idx][n_state + n_goal], y: [n_idx][n_action]; actions and y
 * may be NULL together.  n_rows: rows of the tables; an idx outside [0, n_rows) is not read, its
 * output row is NaN. */
int nmpc_assemble_batch(const float *states, int n_state, const double *s_mean, const double *s_std,
                        int s_first, const float *goals, int n_goal, const double *g_mean,
                        const double *g_std, const float *actions, int n_action, long long n_rows,
                        const int *idx, int n_idx, float *x, float *y, void *stream);

#ifdef __cplusplus
}
#endif
#endif
