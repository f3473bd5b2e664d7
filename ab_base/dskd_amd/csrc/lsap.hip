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

// ------------------------------------------------------------------------------------------------------------------
// Several waves per problem (r4): ONE COLUMN PER THREAD.  The one-wave kernel above strides its 64 lanes over the columns
// (300 queries: 5 rounds of five LDS reads + one strided global read each, then 18 ds_bpermute steps of reduction: ~2 us per
// scan, 477 us for 300 x 110).  Here a column's state -- v, shortest path cost, predecessor, assignment, position in the
// `remaining` list, scanned flag -- lives in the REGISTERS of its thread, the work matrix is staged TRANSPOSED in LDS when it
// fits (row i contiguous over the columns: one conflict-free read per scan), a wave reduces (value, tie score) with DPP row
// operations + four v_readlane, and the waves meet in ONE barrier per scan through a double-buffered table of per-wave
// winners.  Same arithmetic (IEEE double, same operation order), same tie rule -- the key (value, unassigned?, position) is
// reduced lexicographically, first inside a wave, then over the waves -- so the result is the sequential solver's, bit for bit.
struct MwCand {      // (carrying the next row's dual along as well was measured: 254 vs 241 us at 300 x 110 -- the larger entry costs more)
  double val;
  int score, j, r4c, pad;
};

template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double dmin(double a, double b) { return b < a ? b : a; }
__device__ __forceinline__ double readlane_d(double v, int l) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_min_d(double v) {      // no NaN among the inputs (validated)
  v = dmin(v, dpp_d<0xB1>(v));       // quad_perm [1, 0, 3, 2]
  v = dmin(v, dpp_d<0x4E>(v));       // quad_perm [2, 3, 0, 1]
  v = dmin(v, dpp_d<0x141>(v));      // row_half_mirror
  v = dmin(v, dpp_d<0x140>(v));      // row_mirror: every lane of a 16-lane row holds the row's minimum
  return dmin(dmin(readlane_d(v, 0), readlane_d(v, 16)), dmin(readlane_d(v, 32), readlane_d(v, 48)));
}
__device__ __forceinline__ float wave_min_f32(float v) {
  v = fminf(v, dpp_quad_xor1(v));
  v = fminf(v, dpp_quad_xor2(v));
  v = fminf(v, dpp_half_mirror(v));
  v = fminf(v, dpp_row_mirror(v));
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fminf(fminf(r0, r1), fminf(r2, r3));
}
__device__ __forceinline__ int wave_max_i(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// dynamic LDS: doubles u[nr_max], spc[nc_max] | MwCand red[2][16] | ints path[nc_max], row4col[nc_max], remaining[nc_max],
// col4row[nr_max] | bytes SR[nr_max] (padded to 16) | floats costT[nr * nc] (LDS_COST)
struct MwDims {
  int nr_max, nc_max;
};
__host__ __device__ inline size_t mw_state_bytes(int nr_max, int nc_max) {
  size_t b = (size_t)(nr_max + nc_max) * 8 + 2 * 16 * sizeof(MwCand) + (size_t)(3 * nc_max + nr_max) * 4;
  b += ((size_t)nr_max + 15) / 16 * 16;
  return (b + 15) / 16 * 16;
}

template <bool LDS_COST, int CPT>      // CPT: columns per thread (column j = t + k NT); up to 1024 columns in every form
__global__ __launch_bounds__(CPT == 1 ? 1024 : 512) void lsap_mw_kernel(const float* __restrict__ cost_all, DescPack descs, MwDims dims,
                                                       int64_t* __restrict__ row_out, int64_t* __restrict__ col_out,
                                                       int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char mw_smem[];
  double* s_u = reinterpret_cast<double*>(mw_smem);
  double* s_spc = s_u + dims.nr_max;
  MwCand* s_red = reinterpret_cast<MwCand*>(s_spc + dims.nc_max);
  int* s_path = reinterpret_cast<int*>(s_red + 2 * 16);
  int* s_row4col = s_path + dims.nc_max;
  int* s_remaining = s_row4col + dims.nc_max;
  int* s_col4row = s_remaining + dims.nc_max;
  unsigned char* s_SR = reinterpret_cast<unsigned char*>(s_col4row + dims.nr_max);
  float* s_cost = reinterpret_cast<float*>(mw_smem + mw_state_bytes(dims.nr_max, dims.nc_max));

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int NT = blockDim.x, NW = NT >> 6;
  const ProbDesc d = descs.d[blockIdx.x];
  const float* cost = cost_all + d.cost_off;
  const int nr_in = d.nr, nc_in = d.nc;
  if (nr_in == 0 || nc_in == 0) {
    if (t == 0) status[blockIdx.x] = 0;
    return;
  }
  const bool tr = nc_in < nr_in;
  const int nr = tr ? nc_in : nr_in, nc = tr ? nr_in : nc_in;
  const int si = tr ? 1 : nc_in, sj = tr ? nc_in : 1;          // element (i, j) of the work matrix = cost[i si + j sj]
  int64_t* ro = row_out + d.out_off;
  int64_t* co = col_out + d.out_off;

  // validation (scipy: NaN or -inf anywhere -> invalid) + the transposed copy of the work matrix
  int bad = 0;
  for (int e = t; e < nr_in * nc_in; e += NT) {
    const float c = cost[e];
    bad |= (c != c) || (c == -INFINITY);
    if constexpr (LDS_COST) {
      const int a = e / nc_in, b2 = e - a * nc_in;             // source (row a, column b2)
      const int i = tr ? b2 : a, j = tr ? a : b2;
      s_cost[i * nc + j] = c;
    }
  }
  if (__syncthreads_or(bad)) {
    for (int k = t; k < nr; k += NT) { ro[k] = k; co[k] = k; }
    if (t == 0) status[blockIdx.x] = DSKD_ERR_INVALID_COST;
    return;
  }
  double v[CPT], spc[CPT];
  int path[CPT], r4c[CPT], pos[CPT];
  bool SC[CPT], mine[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int j = t + k * NT;
    mine[k] = j < nc;
    v[k] = 0.0; spc[k] = INFINITY; path[k] = -1; r4c[k] = -1; pos[k] = 0; SC[k] = false;
    if (mine[k]) s_row4col[j] = -1;
  }
  for (int i = t; i < nr; i += NT) { s_u[i] = 0.0; s_col4row[i] = -1; }

  for (int cur = 0; cur < nr; ++cur) {
    double minVal = 0.0;
    int num_remaining = nc;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int j = t + k * NT;
      if (mine[k]) s_remaining[nc - 1 - j] = j;      // remaining[it] = nc - it - 1
      pos[k] = nc - 1 - j;
      SC[k] = false;
      spc[k] = INFINITY;
    }
    for (int i = t; i < nr; i += NT) s_SR[i] = 0;
    __syncthreads();

    int sink = -1, i = cur, par = 0;
    while (sink == -1) {
      if (t == 0) s_SR[i] = 1;
      const double ui = s_u[i];
      double cval = INFINITY;
      int cscore = -1, cj = 0, cr4c = -1;
      double cst[CPT];
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const int j = t + k * NT;
        const int jj = mine[k] ? j : 0;
        cst[k] = LDS_COST ? (double)s_cost[i * nc + jj] : (double)cost[(size_t)i * si + (size_t)jj * sj];
      }
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        if (mine[k] && !SC[k]) {
          const double r = minVal + cst[k] - ui - v[k];
          if (r < spc[k]) { path[k] = i; spc[k] = r; }
          const int sc = (r4c[k] == -1) ? 4096 + pos[k] : 2047 - pos[k];
          if (spc[k] < cval || (spc[k] == cval && sc > cscore)) { cval = spc[k]; cscore = sc; cj = t + k * NT; cr4c = r4c[k]; }
        }
      }
      // Wave minimum.  Rounding to f32 is monotone, so the double minimum sits among the lanes that hold the f32 minimum:
      // four v_min_f32 with DPP operands find those; usually it is ONE lane and its double is the minimum.  Otherwise
      // (values closer than an f32 ulp, or real ties) the exact reduction runs over those lanes only.
      const float c32 = (float)cval;
      const float m32 = wave_min_f32(c32);
      unsigned long long tie = __builtin_amdgcn_ballot_w64(c32 == m32);
      double m;
      if (__builtin_popcountll(tie) == 1) {               // wave-uniform
        m = readlane_d(cval, __builtin_amdgcn_readfirstlane(__builtin_ctzll(tie)));
      } else {
        const double cv2 = c32 == m32 ? cval : INFINITY;
        m = wave_min_d(cv2);
        tie = __builtin_amdgcn_ballot_w64(cv2 == m);
        if (__builtin_popcountll(tie) > 1) {             // several columns at the minimum: the tie score decides
          const int bs = wave_max_i(cv2 == m ? cscore : -2);
          tie = __builtin_amdgcn_ballot_w64(cv2 == m && cscore == bs);
        }
      }
      const int wl = __builtin_amdgcn_readfirstlane(__builtin_ctzll(tie));
      const int w_score = __builtin_amdgcn_readlane(cscore, wl), w_r4c = __builtin_amdgcn_readlane(cr4c, wl);
      const int w_j = __builtin_amdgcn_readlane(cj, wl);
      MwCand best = MwCand{m, w_score, w_j, w_r4c, 0};
      if (NW > 1) {
        if (lane == 0) s_red[par * 16 + wave] = best;
        __syncthreads();
        best = s_red[par * 16];
        for (int w = 1; w < NW; ++w) {
          const MwCand o = s_red[par * 16 + w];
          if (o.val < best.val || (o.val == best.val && o.score > best.score)) best = o;
        }
        par ^= 1;
      }
      minVal = best.val;
      if (minVal == INFINITY) {                           // uniform over the workgroup
        for (int k = t; k < nr; k += NT) { ro[k] = k; co[k] = k; }
        if (t == 0) status[blockIdx.x] = DSKD_ERR_INFEASIBLE;
        return;
      }
      const int index = best.score >= 4096 ? best.score - 4096 : 2047 - best.score;
      const int j = best.j;
      if (best.r4c == -1) sink = j; else i = best.r4c;
      --num_remaining;
      // swap-remove j from `remaining` (the slot written here is read again only behind the next scan's barrier)
      if (NW == 1) wave_lds_sync();
      const int jl = s_remaining[num_remaining];
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        if (t + k * NT == j) SC[k] = true;
        if (t + k * NT == jl) pos[k] = index;
      }
      if (NW == 1) wave_lds_sync();                       // every lane has read remaining[num_remaining] before lane 0 writes
      if (t == 0) s_remaining[index] = jl;
    }

    // dual update, then augmentation (same order as the sequential solver)
#pragma unroll
    for (int k = 0; k < CPT; ++k)
      if (mine[k]) { s_spc[t + k * NT] = spc[k]; s_path[t + k * NT] = path[k]; }
    __syncthreads();
    for (int k = t; k < nr; k += NT)
      if (s_SR[k] && k != cur) s_u[k] += minVal - s_spc[s_col4row[k]];
#pragma unroll
    for (int k = 0; k < CPT; ++k)
      if (mine[k] && SC[k]) v[k] -= minVal - spc[k];
    __syncthreads();
    if (t == 0) {
      s_u[cur] += minVal;
      int j = sink;
      while (true) {
        const int k = s_path[j];
        s_row4col[j] = k;
        const int tmp = s_col4row[k];
        s_col4row[k] = j;
        j = tmp;
        if (k == cur) break;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CPT; ++k)
      if (mine[k]) r4c[k] = s_row4col[t + k * NT];
  }

  if (tr) {
    // pairs sorted by original row = col4row value (all distinct): rank by counting
    for (int k = t; k < nr; k += NT) {
      const int mine_c = s_col4row[k];
      int rank = 0;
      for (int m2 = 0; m2 < nr; ++m2) rank += s_col4row[m2] < mine_c;
      ro[rank] = mine_c;
      co[rank] = k;
    }
  } else {
    for (int k = t; k < nr; k += NT) { ro[k] = k; co[k] = s_col4row[k]; }
  }
  if (t == 0) status[blockIdx.x] = 0;
}

template <bool LDS_COST, int CPT>
static int launch_mw(int n, int threads, size_t lds, hipStream_t st, const float* cost, const DescPack& pack, MwDims dims,
                     int64_t* row, int64_t* col, int32_t* status) {
  static bool done[64] = {};
  if (!reserve_lds((const void*)lsap_mw_kernel<LDS_COST, CPT>, 156 * 1024, done))
    return fail(DSKD_ERR_LAUNCH, "dskd_lsap_batched: cannot reserve LDS");
  hipLaunchKernelGGL((lsap_mw_kernel<LDS_COST, CPT>), dim3(n), dim3(threads), lds, st, cost, pack, dims, row, col, status);
  return DSKD_OK;
}

}  // namespace
}  // namespace dskd

using namespace dskd;

static int g_lsap_mode = 0;
/* test / A-B hook: 0 = automatic (default), 1 = the one-wave-per-problem kernel of round 1 for every size, 2 / 3 = the
 * register-resident kernel with 1 / 2 columns per thread */
extern "C" int dskd_lsap_tune(int mode) {
  if (mode < 0 || mode > 3)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_lsap_tune: mode %d", mode);
  g_lsap_mode = mode;
  return DSKD_OK;
}

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
    // the work matrix of a problem has max(nr, nc) columns: more than 64 -> one column per thread, several waves per problem
    int nc_max = 1, nr_max = 1;
    long long elems = 0;
    for (int k = 0; k < n; ++k) {
      const int a = nr[p0 + k], b = nc[p0 + k];
      nc_max = std::max(nc_max, std::max(a, b));
      nr_max = std::max(nr_max, std::min(a, b));
      elems = std::max(elems, (long long)a * b);
    }
    if (nc_max > 64 && g_lsap_mode != 1) {
      // columns per thread: 1 (automatic) | 2 (dskd_lsap_tune(3)).  Measured at 300 x 110 (profiles/r04_lsap_several_waves.txt):
      // 1: 236 us, 2: 233, 5: 339, 16 (one wave, no barrier at all): 632 -- the scan is bound by the dependent double-precision
      // chain of ONE column plus the reduction, so more columns per thread only lengthen it; the 5 / 16 forms were removed
      const int cpt = g_lsap_mode == 3 ? 2 : 1;
      const int threads = ((nc_max + cpt - 1) / cpt + 63) / 64 * 64;
      const MwDims dims{nr_max, nc_max};
      const size_t state = mw_state_bytes(nr_max, nc_max), with_cost = state + (size_t)elems * sizeof(float);
      const bool lc = with_cost <= 156 * 1024;
      const size_t lds = lc ? with_cost : state;
      int rc = DSKD_OK;
#define DSKD_MW(C) (lc ? launch_mw<true, C>(n, threads, lds, st, cost, pack, dims, row, col, status + p0) \
                       : launch_mw<false, C>(n, threads, lds, st, cost, pack, dims, row, col, status + p0))
      rc = cpt == 1 ? DSKD_MW(1) : DSKD_MW(2);
#undef DSKD_MW
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(lsap_kernel, dim3(n), dim3(64), 0, st, cost, pack, row, col, status + p0);
    }
    if (int rc = check_launch("dskd_lsap_batched")) return rc;
  }
  return DSKD_OK;
}
