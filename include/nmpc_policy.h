/* nmpc_policy.h -- C-ABI of the learning update: the reference's policy network
 * and its behaviour-cloning training step as hand-written gfx950 kernels (libnmpc_hip.so).
 *
 * Replaces, for batches that already live on the device (rollout states from nmpc_rollout_batch,
 * expert actions from nmpc_solve_batch):
 *   GoalConditionedPolicyNet.forward              DAgger/utils/network.py:72-81
 *       in -> [Linear, BatchNorm1d, ReLU] x L -> Linear -> out
 *   one iteration of BehavioralCloning.train_network   DAgger/utils/train_locosafedagger.py:93-102
 *       optimizer.zero_grad(); loss = L1Loss(network(x), y); loss.backward(); optimizer.step()   (Adam)
 * All tensors are fp32 device pointers, row-major [batch][feature]; calls are stream-ordered, never
 * allocate and never synchronise.  Return values: NMPC_OK / NMPC_E_* of nmpc.h.
 *
 * Parameter vector theta (the order of torch's net.parameters()): for each hidden layer
 * W[hidden][fan_in], b[hidden], then gamma[hidden], beta[hidden] if batch_norm; finally
 * W[n_out][hidden], b[n_out].  BatchNorm buffers: running_mean[L][hidden], running_var[L][hidden]. */
#ifndef NMPC_POLICY_H
#define NMPC_POLICY_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int n_in;        /* policy input: state + goal (47 in cfgs/iter_locosafedagger.yaml)      */
    int n_out;       /* action dimension (12)                                                 */
    int n_hidden;    /* hidden layers L >= 1 (3)                                              */
    int hidden;      /* nodes per hidden layer (512)                                          */
    int batch_norm;  /* BatchNorm1d between Linear and ReLU (1)                               */
    int batch_max;   /* largest batch of a forward / training call                            */
} nmpc_policy_dims;

int nmpc_policy_create(const nmpc_policy_dims *dims, int device_id, void **handle);
void nmpc_policy_destroy(void *handle);
const char *nmpc_policy_last_error(void *handle);

/* length of theta */
size_t nmpc_policy_param_count(void *handle);

/* Copy parameters and BatchNorm buffers in / out (device pointers; running_* may be NULL without
 * batch_norm).  set_params also resets the optimiser state (Adam moments, step count). */
int nmpc_policy_set_params(void *handle, const float *theta, const float *running_mean,
                           const float *running_var, void *stream);
int nmpc_policy_get_params(void *handle, float *theta, float *running_mean, float *running_var,
                           void *stream);

/* network.eval(); Y = network(X)      X[B][n_in] -> Y[B][n_out] */
int nmpc_policy_forward(void *handle, int B, const float *X, float *Y, void *stream);

/* One Adam step on the L1 loss of a batch (train mode: batch statistics, running statistics updated
 * with momentum 0.1).  loss: device scalar, the mean absolute error BEFORE the step (may be NULL);
 * pred: the train-mode prediction [B][n_out] (may be NULL).  B >= 2 with batch_norm. */
int nmpc_policy_train_step(void *handle, int B, const float *X, const float *Y, float lr, float *loss,
                           float *pred, void *stream);

/* torch.utils.data.WeightedRandomSampler(weights, num_samples, replacement=True)
 * (Behavior_Cloning/examples/test_train_policy.py:128-134) on device weights -- e.g. the OOD weights
 * nmpc_tracking_error wrote:  idx[i] ~ weights / sum(weights),  0 <= i < num_samples.  Sample i is the
 * inverse-CDF lookup of a uniform number made by the counter-based Philox-4x32-10 generator from
 * (seed, i): reproducible for a seed whatever the launch shape, and restated bit for bit by the oracle.
 * scratch: n + n/2048 + 2 doubles of device memory.  Stateless (no handle). */
int nmpc_weighted_sample(const float *weights, long long n, int num_samples, unsigned long long seed,
                         double *scratch, int *idx, void *stream);

/* Batch assembly behind the sampler: dst[i][0..row_len) = src[idx[i]][0..row_len) for a table of n_rows
 * rows; an idx outside [0, n_rows) is not read, its output row is NaN. */
int nmpc_gather_rows(const float *src, long long n_rows, int row_len, const int *idx, int n_idx, float *dst,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif
