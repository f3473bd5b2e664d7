// Rectangular linear sum assignment, bit-exact with scipy.optimize.linear_sum_assignment
// (scipy 1.15.3 `_lsap`, the modified Jonker-Volgenant / Crouse shortest augmenting path
// solver), which the reference calls at
// mmdet/core/bbox/assigners/gfl_hungarian_assigner.py:143-151 after a device->host copy.
//
// Behaviour reproduced:
//  * float32 costs are widened to float64 exactly; all dual arithmetic is IEEE double with
//    the same operation order  r = ((minVal + c) - u[i]) - v[j]  (no contraction possible:
//    there is no multiply);
//  * nr > nc  ->  the transposed problem is solved and the pairs are re-sorted by row;
//  * NaN or -inf anywhere  ->  "invalid numeric entries"; an unreachable sink -> "infeasible";
//  * tie rule of the sequential scan over the swap-removed `remaining` list: among the
//    minimal columns the LAST scanned unassigned one wins, else the FIRST scanned.  The
//    device version keeps every column's position in that list and reduces the key
//    (value, unassigned, position) across the wave, so it picks the same column.
//
// Device mapping: one wavefront per problem (problems are tiny and serial in the
// augmentation index; 6 layers x B images of them run side by side on different CUs in ONE
// launch, instead of 6*B device->host->device round trips).  All solver state lives in the
// workgroup's LDS; lanes stride over columns.
#include "common.h"
#include <math.h>
#include <vector>
#include <algorithm>
#include <numeric>

namespace dskd {
namespace {

constexpr int kMaxDim = 1024;

// ------------------------------------------------------------------ host solver
int lsap_host_impl(const float* cost_in, int nr_in, int nc_in, int64_t* row, int64_t* col) {
  if (nr_in == 0 || nc_in == 0) return DSKD_OK;
  const bool tr = nc_in < nr_in;
  const int nr = tr ? nc_in : nr_in, nc = tr ? nr_in : nc_in;
  std::vector<double> cost((size_t)nr * nc);
  for (int i = 0; i < nr_in; ++i)
    for (int j = 0; j < nc_in; ++j) {
      const double c = (double)cost_in[(size_t)i * nc_in + j];
      if (c != c || c == -INFINITY) return DSKD_ERR_INVALID_COST;
      if (tr) cost[(size_t)j * nc + i] = c; else cost[(size_t)i * nc + j] = c;
    }
  std::vector<double> u(nr, 0.0), v(nc, 0.0), spc(nc);
  std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
  std::vector<char> SR(nr), SC(nc);
  for (int cur = 0; cur < nr; ++cur) {
    double minVal = 0.0;
    int num_remaining = nc;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(spc.begin(), spc.end(), INFINITY);
    int sink = -1, i = cur;
    while (sink == -1) {
      int index = -1;
      double lowest = INFINITY;
      SR[i] = 1;
      for (int it = 0; it < num_remaining; ++it) {
        const int j = remaining[it];
        const double r = minVal + cost[(size_t)i * nc + j] - u[i] - v[j];
        if (r < spc[j]) { path[j] = i; spc[j] = r; }
        if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
          lowest = spc[j];
          index = it;
        }
      }
      minVal = lowest;
      if (minVal == INFINITY) return DSKD_ERR_INFEASIBLE;
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--num_remaining];
    }
    u[cur] += minVal;
    for (int k = 0; k < nr; ++k)
      if (SR[k] && k != cur) u[k] += minVal - spc[col4row[k]];
    for (int j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= minVal - spc[j];
    int j = sink;
    while (true) {
      const int k = path[j];
      row4col[j] = k;
      std::swap(col4row[k], j);
      if (k == cur) break;
    }
  }
  if (tr) {
    std::vector<int> order(nr);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return col4row[a] < col4row[b]; });
    for (int k = 0; k < nr; ++k) { row[k] = col4row[order[k]]; col[k] = order[k]; }
  } else {
    for (int k = 0; k < nr; ++k) { row[k] = k; col[k] = col4row[k]; }
  }
  return DSKD_OK;
}

// ------------------------------------------------------------------ device solver
struct ProbDesc {
  int nr, nc;
  long long cost_off, out_off;
};

struct Cand {
  double val;
  int score;  // unassigned ? 4096 + position : 2047 - position ; larger wins on ties
};

__device__ __forceinline__ bool better(const Cand& a, const Cand& b) {
  return a.val < b.val || (a.val == b.val && a.score > b.score);
}

__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Cand t;
    t.val = __shfl_xor(c.val, o);
    t.score = __shfl_xor(c.score, o);
    if (better(t, c)) c = t;
  }
  return c;
}

constexpr int kPack = 64;  // problems per launch: descriptors ride in the kernel arguments
struct DescPack {
  ProbDesc d[kPack];
};

__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ cost_all,
                                                  DescPack descs,
                                                  int64_t* __restrict__ row_out,
                                                  int64_t* __restrict__ col_out,
                                                  int* __restrict__ status) {
  __shared__ double s_spc[kMaxDim];
  __shared__ double s_v[kMaxDim];
  __shared__ double s_u[kMaxDim];
  __shared__ int s_path[kMaxDim];
  __shared__ int s_row4col[kMaxDim];
  __shared__ int s_col4row[kMaxDim];
  __shared__ int s_remaining[kMaxDim];
  __shared__ int s_pos[kMaxDim];
  __shared__ unsigned char s_SC[kMaxDim];
  __shared__ unsigned char s_SR[kMaxDim];

  const int lane = threadIdx.x;
  const ProbDesc d = descs.d[blockIdx.x];
  const float* cost = cost_all + d.cost_off;
  const int nr_in = d.nr, nc_in = d.nc;
  if (nr_in == 0 || nc_in == 0) {
    if (lane == 0) status[blockIdx.x] = 0;
    return;
  }
  const bool tr = nc_in < nr_in;
  const int nr = tr ? nc_in : nr_in, nc = tr ? nr_in : nc_in;
  // element (i, j) of the work matrix
  const int si = tr ? 1 : nc_in, sj = tr ? nc_in : 1;

  // validation (scipy: NaN or -inf anywhere -> invalid)
  int bad = 0;
  for (int e = lane; e < nr_in * nc_in; e += 64) {
    const float c = cost[e];
    bad |= (c != c) || (c == -INFINITY);
  }
  // On an error the status word carries the scipy error class and the outputs are filled with
  // an in-range identity pairing, so that a caller that indexes with them before looking at
  // the status (asynchronous pipelines do) can never go out of bounds.
  int64_t* ro = row_out + d.out_off;
  int64_t* co = col_out + d.out_off;
  if (__any(bad)) {
    for (int k = lane; k < nr; k += 64) { ro[k] = k; co[k] = k; }
    if (lane == 0) status[blockIdx.x] = DSKD_ERR_INVALID_COST;
    return;
  }

  for (int j = lane; j < nc; j += 64) { s_v[j] = 0.0; s_row4col[j] = -1; s_path[j] = -1; }
  for (int i = lane; i < nr; i += 64) { s_u[i] = 0.0; s_col4row[i] = -1; }
  wave_lds_sync();

  for (int cur = 0; cur < nr; ++cur) {
    double minVal = 0.0;
    int num_remaining = nc;
    for (int j = lane; j < nc; j += 64) {
      s_remaining[nc - 1 - j] = j;  // remaining[it] = nc - it - 1
      s_pos[j] = nc - 1 - j;
      s_SC[j] = 0;
      s_spc[j] = INFINITY;
    }
    for (int i = lane; i < nr; i += 64) s_SR[i] = 0;
    wave_lds_sync();

    int sink = -1, i = cur;
    while (sink == -1) {
      if (lane == 0) s_SR[i] = 1;
      const double ui = s_u[i];
      const float* crow = cost + (size_t)i * si;
      Cand best;
      best.val = INFINITY;
      best.score = -1;
      for (int j = lane; j < nc; j += 64) {
        if (s_SC[j]) continue;
        const double r = minVal + (double)crow[(size_t)j * sj] - ui - s_v[j];
        double sp = s_spc[j];
        if (r < sp) { s_path[j] = i; s_spc[j] = r; sp = r; }
        Cand c;
        c.val = sp;
        c.score = (s_row4col[j] == -1) ? 4096 + s_pos[j] : 2047 - s_pos[j];
        if (better(c, best)) best = c;
      }
      best = wave_best(best);
      minVal = best.val;
      if (minVal == INFINITY) {  // wave-uniform
        for (int k = lane; k < nr; k += 64) { ro[k] = k; co[k] = k; }
        if (lane == 0) status[blockIdx.x] = DSKD_ERR_INFEASIBLE;
        return;
      }
      const int index = best.score >= 4096 ? best.score - 4096 : 2047 - best.score;
      const int j = s_remaining[index];
      const int r4c = s_row4col[j];
      if (r4c == -1) sink = j; else i = r4c;
      --num_remaining;
      wave_lds_sync();  // all lanes have read remaining[index] before it is replaced
      if (lane == 0) {
        s_SC[j] = 1;
        const int jl = s_remaining[num_remaining];
        s_remaining[index] = jl;
        s_pos[jl] = index;
      }
      wave_lds_sync();
    }

    // dual update, then augmentation (same order as the sequential solver)
    for (int k = lane; k < nr; k += 64)
      if (s_SR[k] && k != cur) s_u[k] += minVal - s_spc[s_col4row[k]];
    for (int j = lane; j < nc; j += 64)
      if (s_SC[j]) s_v[j] -= minVal - s_spc[j];
    if (lane == 0) s_u[cur] += minVal;
    wave_lds_sync();
    if (lane == 0) {
      int j = sink;
      while (true) {
        const int k = s_path[j];
        s_row4col[j] = k;
        const int t = s_col4row[k];
        s_col4row[k] = j;
        j = t;
        if (k == cur) break;
      }
    }
    wave_lds_sync();
  }

  if (tr) {
    // pairs sorted by original row = col4row value (all distinct): rank by counting
    for (int k = lane; k < nr; k += 64) {
      const int mine = s_col4row[k];
      int rank = 0;
      for (int m = 0; m < nr; ++m) rank += s_col4row[m] < mine;
      ro[rank] = mine;
      co[rank] = k;
    }
  } else {
    for (int k = lane; k < nr; k += 64) { ro[k] = k; co[k] = s_col4row[k]; }
  }
  if (lane == 0) status[blockIdx.x] = 0;
}

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int dskd_lsap_host(const float* cost, int nr, int nc, int64_t* row, int64_t* col) {
  if (nr < 0 || nc < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_lsap_host: negative size");
  if ((nr && nc) && (!cost || !row || !col)) return fail(DSKD_ERR_INVALID_ARG, "dskd_lsap_host: null pointer");
  const int rc = lsap_host_impl(cost, nr, nc, row, col);
  if (rc == DSKD_ERR_INVALID_COST) return fail(rc, "matrix contains invalid numeric entries");
  if (rc == DSKD_ERR_INFEASIBLE) return fail(rc, "cost matrix is infeasible");
  return rc;
}

extern "C" int dskd_lsap_batched(const float* cost, const int32_t* nr, const int32_t* nc,
                                 const int64_t* offsets, int nprob, int64_t* row,
                                 int64_t* col, const int64_t* out_offsets, int32_t* status,
                                 void* stream) {
  if (nprob < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_lsap_batched: nprob < 0");
  if (nprob == 0) return DSKD_OK;
  if (!cost || !nr || !nc || !offsets || !row || !col || !out_offsets || !status)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_lsap_batched: null pointer");
  for (int p = 0; p < nprob; ++p)
    if (nr[p] < 0 || nc[p] < 0 || nr[p] > kMaxDim || nc[p] > kMaxDim)
      return fail(DSKD_ERR_INVALID_ARG, "dskd_lsap_batched: problem %d is %dx%d, limit %d", p,
                  nr[p], nc[p], kMaxDim);
  hipStream_t st = (hipStream_t)stream;
  // No allocation, no copy, no sync: graph-capturable.
  for (int p0 = 0; p0 < nprob; p0 += kPack) {
    const int n = std::min(kPack, nprob - p0);
    DescPack pack;
    for (int k = 0; k < kPack; ++k) {
      const int p = p0 + (k < n ? k : 0);
      pack.d[k] = ProbDesc{nr[p], nc[p], (long long)offsets[p], (long long)out_offsets[p]};
    }
    hipLaunchKernelGGL(lsap_kernel, dim3(n), dim3(64), 0, st, cost, pack, row, col, status + p0);
    if (int rc = check_launch("dskd_lsap_batched")) return rc;
  }
  return DSKD_OK;
}
