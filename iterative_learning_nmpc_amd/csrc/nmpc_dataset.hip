// nmpc_dataset.hip -- the device-resident training database on gfx950 (C-ABI: include/nmpc_dataset.h).
//
// Ring-buffer append, per-column mean / standard deviation and normalised batch assembly of the
// reference's Database (DAgger/utils/database.py:105-154, 208-255, 54-84).  All of it is HBM-bound
// byte moving and fp64 summation: no LDS tiling games, coalesced dword streams and a fixed summation
// order.  At the reference's database_size (1e7 rows x 44 columns, cfgs/iter_locosafedagger.yaml:61)
// the state table is 1.76 GB: it stays in HBM next to the rollouts that fill it.
#include <hip/hip_runtime.h>

#include "nmpc_device_guard.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>

#include "../../include/nmpc.h"
#include "../../include/nmpc_dataset.h"

namespace nmpc_dataset {

constexpr int STAT_BLOCKS_MAX = 1024;   // partial sums per column
constexpr int STAT_SEGS = 16;           // segments of the final reduction

__global__ void ring_append_kernel(const float* __restrict__ src, int row_len, long long n, float* __restrict__ ring,
                                   long long limit, long long first_slot, long long skip) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)(n - skip) * row_len) return;
    const long long i = skip + (long long)(e / row_len);
    const int j = (int)(e % row_len);
    const long long slot = (first_slot + i) % limit;
    ring[(size_t)slot * row_len + j] = src[(size_t)i * row_len + j];
}

// Column sums of data[rows][cols] without a division or a transposition: a wave walks "super rows" of
// 64 rows (64 * cols consecutive floats); lane l reads the elements l + 64 j, j < cols, of each, and since a
// super row is a whole number of rows, element (l, j) belongs to the same column (l + 64 j) % cols in
// every super row: one fp64 accumulator per j, in registers, and every load is a fully coalesced 256 B.
//   PASS 0: sum of x                 -> mean
//   PASS 1: sum of (x - mean)^2      -> population variance (np.std: ddof = 0)
// COLS > 0: cols is the compile-time COLS (branch-free unrolled body); COLS == 0: any cols <= 64.
// The block's 4 waves are folded in wave order through the LDS and each column is summed over its 64
// (lane, j) slots in index order, so the partial sums -- and the statistics -- are the same on every run.
template <int COLS, int PASS>
__global__ __launch_bounds__(256) void colstat_partial_kernel(const float* __restrict__ data, long long rows, int cols_rt,
                                                              const double* __restrict__ mean, double* __restrict__ part) {
    constexpr int NACC = COLS > 0 ? COLS : 64;
    const int cols = COLS > 0 ? COLS : cols_rt;
    __shared__ double flat[64 * NACC];
    __shared__ double mean_s[NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (PASS == 1) {
        if ((int)threadIdx.x < cols) mean_s[threadIdx.x] = mean[threadIdx.x];
        __syncthreads();
    }
    const long long super = 64LL * cols;
    const long long n_super = rows / 64;                          // full super rows; the < 64 rows left go to block 0
    const int step = 64 % cols;

    double acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.0;

    for (long long sr = (long long)blockIdx.x * 4 + wave; sr < n_super; sr += (long long)gridDim.x * 4) {
        const float* __restrict__ p = data + sr * super;          // wave-uniform base, 32-bit lane offsets
        float v[NACC];
#pragma unroll
        for (int j = 0; j < NACC; ++j)
            if (j < cols) v[j] = p[lane + 64 * j];
        int c = lane % cols;
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (j < cols) {
                if (PASS == 0) {
                    acc[j] += (double)v[j];
                } else {
                    const double d = (double)v[j] - mean_s[c];
                    acc[j] += d * d;
                    c += step;
                    if (c >= cols) c -= cols;
                }
            }
        }
    }
    // waves in order, then the 64 slots of a column in order, then (block 0) the rows beyond the last super row
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NACC; ++j)
                if (j < cols) flat[64 * j + lane] = (w == 0 ? 0.0 : flat[64 * j + lane]) + acc[j];
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < cols) {
        double s = 0.0;
        for (int r = 0; r < 64; ++r) s += flat[threadIdx.x + cols * r];
        if (blockIdx.x == 0) {
            for (long long r = n_super * 64; r < rows; ++r) {
                const double x = (double)data[r * cols + threadIdx.x];
                s += PASS == 0 ? x : (x - mean_s[threadIdx.x]) * (x - mean_s[threadIdx.x]);
            }
        }
        part[(size_t)blockIdx.x * cols + threadIdx.x] = s;
    }
}

// mean = sum / rows   or   std = sqrt(sum / rows): thread (c, seg) sums the partials of its segment of
// blocks, the segments are added in order.
template <int PASS>
__global__ __launch_bounds__(64 * STAT_SEGS) void colstat_final_kernel(const double* __restrict__ part, int n_blocks, int cols,
                                                                        long long rows, double* __restrict__ out) {
    __shared__ double seg_sum[STAT_SEGS][64];
    const int c = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int per = (n_blocks + STAT_SEGS - 1) / STAT_SEGS;
    double s = 0.0;
    if (c < cols) {
        const int b0 = seg * per, b1 = min(n_blocks, b0 + per);
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += part[(size_t)b * cols + c];
    }
    seg_sum[seg][c] = s;
    __syncthreads();
    if (seg == 0 && c < cols) {
        double t = 0.0;
        for (int k = 0; k < STAT_SEGS; ++k) t += seg_sum[k][c];
        out[c] = PASS == 0 ? t / (double)rows : sqrt(t / (double)rows);
    }
}

__global__ void assemble_batch_kernel(const float* __restrict__ states, int n_state, const double* __restrict__ s_mean,
                                      const double* __restrict__ s_std, int s_first, const float* __restrict__ goals, int n_goal,
                                      const double* __restrict__ g_mean, const double* __restrict__ g_std,
                                      const float* __restrict__ actions, int n_action, long long n_rows,
                                      const int* __restrict__ idx, int n_idx, float* __restrict__ x, float* __restrict__ y) {
    const int n_x = n_state + n_goal, width = n_x + n_action;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n_idx * width) return;
    const int i = (int)(e / width), j = (int)(e % width);
    const long long r = idx[i];
    if (r < 0 || r >= n_rows) {   // never read outside the tables: an index out of range shows up as NaN
        if (j < n_x) x[(size_t)i * n_x + j] = NAN; else y[(size_t)i * n_action + j - n_x] = NAN;
        return;
    }
    const size_t row = (size_t)r;
    if (j < n_state) {
        const float s = states[row * n_state + j];
        x[(size_t)i * n_x + j] = (s_mean && j >= s_first) ? (float)(((double)s - s_mean[j]) / s_std[j]) : s;
    } else if (j < n_x) {
        const int k = j - n_state;
        const float g = goals[row * n_goal + k];
        x[(size_t)i * n_x + j] = g_mean ? (float)(((double)g - g_mean[k]) / g_std[k]) : g;
    } else {
        const int k = j - n_x;
        y[(size_t)i * n_action + k] = actions[row * n_action + k];
    }
}

}  // namespace nmpc_dataset

// ================================================================================================
namespace {

using namespace nmpc_dataset;

thread_local std::string g_dataset_error;

int dfail(int code, const std::string& msg) {
    g_dataset_error = msg;
    return code;
}
int launched() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NMPC_OK : dfail(NMPC_E_HIP, hipGetErrorString(e));
}

// One resident round: as many blocks as the device holds at once (a grid-stride kernel launched wider
// than that runs a second, mostly idle round), at most STAT_BLOCKS_MAX (the scratch size).
template <int COLS, int PASS>
int resident_blocks() {
    static int cache[16] = {0};                 // per device: the occupancy query is the device's, not the process's
    int dev_now = 0;
    if (hipGetDevice(&dev_now) != hipSuccess || dev_now < 0 || dev_now >= 16) return STAT_BLOCKS_MAX;
    int& cached = cache[dev_now];
    if (cached == 0) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, colstat_partial_kernel<COLS, PASS>, 256, 0) != hipSuccess ||
            cus < 1 || per_cu < 1)
            return STAT_BLOCKS_MAX;
        cached = std::min(STAT_BLOCKS_MAX, cus * per_cu);
    }
    return cached;
}

template <int COLS>
void column_stats(hipStream_t st, const float* data, long long rows, int cols, double* mean, double* std_out, double* part) {
    const long long need = std::max<long long>(1, (rows / 64 + 3) / 4);      // blocks that have a super row to read
    const int b0 = (int)std::min<long long>(resident_blocks<COLS, 0>(), need);
    const int b1 = (int)std::min<long long>(resident_blocks<COLS, 1>(), need);
    hipLaunchKernelGGL((colstat_partial_kernel<COLS, 0>), dim3(b0), dim3(256), 0, st, data, rows, cols, nullptr, part);
    hipLaunchKernelGGL((colstat_final_kernel<0>), dim3(1), dim3(64 * STAT_SEGS), 0, st, part, b0, cols, rows, mean);
    hipLaunchKernelGGL((colstat_partial_kernel<COLS, 1>), dim3(b1), dim3(256), 0, st, data, rows, cols, mean, part);
    hipLaunchKernelGGL((colstat_final_kernel<1>), dim3(1), dim3(64 * STAT_SEGS), 0, st, part, b1, cols, rows, std_out);
}

}  // namespace

extern "C" {

const char* nmpc_dataset_last_error(void) { return g_dataset_error.c_str(); }

int nmpc_ring_append(const float* src, int row_len, long long n, float* ring, long long limit, long long first_slot,
                     void* stream) {
    if (n == 0) return NMPC_OK;
    if (!src || !ring) return dfail(NMPC_E_ARG, "null argument");
    if (row_len < 1 || n < 0 || limit < 1 || first_slot < 0 || first_slot >= limit)
        return dfail(NMPC_E_ARG, "need row_len >= 1, n >= 0, limit >= 1, 0 <= first_slot < limit");
    nmpc::DeviceGuard guard(nmpc::device_of(ring));
    const long long skip = n > limit ? n - limit : 0;
    const size_t elems = (size_t)(n - skip) * row_len;
    if ((elems + 255) / 256 > 0x7fffffffULL) return dfail(NMPC_E_ARG, "append too large for one launch");
    hipLaunchKernelGGL(ring_append_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, row_len, n, ring, limit, first_slot, skip);
    return launched();
}

size_t nmpc_column_stats_scratch(int cols) { return cols < 1 ? 0 : (size_t)STAT_BLOCKS_MAX * cols; }

int nmpc_column_stats(const float* data, long long rows, int cols, double* mean, double* std_out, double* scratch,
                      void* stream) {
    if (!data || !mean || !std_out || !scratch) return dfail(NMPC_E_ARG, "null argument");
    if (rows < 1 || cols < 1 || cols > 64) return dfail(NMPC_E_ARG, "need rows >= 1, 1 <= cols <= 64");
    nmpc::DeviceGuard guard(nmpc::device_of(data));
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (cols) {   // the reference's row widths: state 44, action 12, contact goal 8, velocity goal 3
        case 44: column_stats<44>(st, data, rows, cols, mean, std_out, scratch); break;
        case 12: column_stats<12>(st, data, rows, cols, mean, std_out, scratch); break;
        case 8: column_stats<8>(st, data, rows, cols, mean, std_out, scratch); break;
        case 3: column_stats<3>(st, data, rows, cols, mean, std_out, scratch); break;
        default: column_stats<0>(st, data, rows, cols, mean, std_out, scratch); break;
    }
    return launched();
}

int nmpc_assemble_batch(const float* states, int n_state, const double* s_mean, const double* s_std, int s_first,
                        const float* goals, int n_goal, const double* g_mean, const double* g_std, const float* actions,
                        int n_action, long long n_rows, const int* idx, int n_idx, float* x, float* y, void* stream) {
    if (n_idx == 0) return NMPC_OK;
    if (!states || !idx || !x) return dfail(NMPC_E_ARG, "null argument");
    if (n_state < 1 || n_goal < 0 || n_action < 0 || n_idx < 0 || s_first < 0 || n_rows < 1)
        return dfail(NMPC_E_ARG, "need n_state, n_rows >= 1, n_goal, n_action, n_idx, s_first >= 0");
    if ((n_goal > 0 && !goals) || (n_action > 0 && (!actions || !y)))
        return dfail(NMPC_E_ARG, "goals / actions / y missing for a non-zero width");
    if ((s_mean == nullptr) != (s_std == nullptr) || (g_mean == nullptr) != (g_std == nullptr))
        return dfail(NMPC_E_ARG, "mean and std come in pairs");
    nmpc::DeviceGuard guard(nmpc::device_of(states));
    const size_t elems = (size_t)n_idx * (n_state + n_goal + n_action);
    hipLaunchKernelGGL(assemble_batch_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), states, n_state, s_mean, s_std, s_first, goals, n_goal, g_mean, g_std,
                       actions, n_action, n_rows, idx, n_idx, x, y);
    return launched();
}

}  // extern "C"
