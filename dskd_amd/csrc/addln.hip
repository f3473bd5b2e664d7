// Residual add + dropout + LayerNorm (+ positional add) of a transformer sub-layer, one pass
// each way.
//
// Every sub-layer of ext-mmcv's BaseTransformerLayer ends in
//     x = LayerNorm(identity + dropout(sublayer_out))            ('self_attn'/'ffn' then 'norm')
// and the next deformable-attention layer starts with  query = x + query_pos.  In PyTorch under
// autocast that is dropout, a mixed-dtype add, an fp32 LayerNorm, a cast back to bf16 for the
// next GEMM and the positional add: five launches and ~0.7 GB of HBM traffic per sub-layer at
// 88 892 tokens x 256, and twice that backward.  Here:
//     forward   reads h, residual (and pos) once, writes y (and q = y + pos) once, plus the
//               pre-norm sum z and (mean, rstd) for backward when training
//     backward  reads dy (and dq), z once; writes d(residual) and d(h) once; column sums for
//               d(gamma), d(beta) accumulate per wave in registers -> LDS -> one atomic per
//               column and workgroup
// The dropout mask is never stored: Philox4x32-10 keyed by (seed, offset) and counted by
// (row, lane) is regenerated in backward.
//
// HBM-bound streaming kernels.  D == 256 only (the DSKD transformer width): one 64-lane wave
// owns a row, a lane owns 4 consecutive columns (8-byte bf16 / 16-byte f32 accesses), the row
// statistics are two wave reductions.
#include "common.h"

namespace dskd {
namespace {

constexpr int kD = 256;
constexpr int kRowsPerBlock = 4;       // waves per workgroup, one row each per step

__device__ __forceinline__ float as_float(unsigned u) { return __builtin_bit_cast(float, u); }

template <typename T>
__device__ __forceinline__ void load4(const T* __restrict__ p, float* f) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    const unsigned lo = v.x, hi = v.y;
    f[0] = as_float(lo << 16); f[1] = as_float(lo & 0xFFFF0000u);
    f[2] = as_float(hi << 16); f[3] = as_float(hi & 0xFFFF0000u);
  }
}

// round to T and return the rounded values in f (what a later reader of the store will see)
template <typename T>
__device__ __forceinline__ void store4(T* __restrict__ p, float* f) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
  } else {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 v = {(__bf16)f[0], (__bf16)f[1], (__bf16)f[2], (__bf16)f[3]};
    *reinterpret_cast<bf16x4*>(p) = v;
    f[0] = (float)v.x; f[1] = (float)v.y; f[2] = (float)v.z; f[3] = (float)v.w;
  }
}

// Philox4x32-10 (Salmon et al. 2011): 4 x 32 random bits per (row, lane).
__device__ __forceinline__ u32x4 philox(unsigned long long row, unsigned lane, unsigned long long seed,
                                        unsigned long long offset) {
  unsigned c0 = (unsigned)row, c1 = (unsigned)(row >> 32), c2 = lane, c3 = (unsigned)offset;
  unsigned k0 = (unsigned)seed ^ (unsigned)(offset >> 32), k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

struct Drop {
  unsigned thresh;          // drop when the random word < thresh  (thresh = p * 2^32)
  float scale;              // 1 / (1 - p)
  unsigned long long seed, offset;
};

template <typename T>
__global__ __launch_bounds__(kRowsPerBlock * 64) void add_ln_fwd_kernel(
    const T* __restrict__ h, const T* __restrict__ res, const float* __restrict__ pos, long long pos_rows,
    const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, T* __restrict__ q,
    T* __restrict__ z, float* __restrict__ stats, long long rows, float eps, Drop dr) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane * 4;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
  const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
  for (long long row = (long long)blockIdx.x * kRowsPerBlock + wave; row < rows;
       row += (long long)gridDim.x * kRowsPerBlock) {
    float hv[4], zv[4];
    load4(h + row * kD + c, hv);
    load4(res + row * kD + c, zv);
    if (dr.thresh) {
      const u32x4 rnd = philox((unsigned long long)row, (unsigned)lane, dr.seed, dr.offset);
      const unsigned r4[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) hv[i] = r4[i] < dr.thresh ? 0.f : hv[i] * dr.scale;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) zv[i] += hv[i];
    if (z) {
      store4(z + row * kD + c, zv);            // statistics of the ROUNDED sum: backward sees the same z
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) zv[i] = (float)(__bf16)zv[i];
    }
    const float mean = wave_sum(zv[0] + zv[1] + zv[2] + zv[3]) * (1.0f / kD);
    float d[4], ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { d[i] = zv[i] - mean; ss += d[i] * d[i]; }
    const float rstd = rsqrtf(wave_sum(ss) * (1.0f / kD) + eps);
    if (stats && lane == 0) *reinterpret_cast<f32x2*>(stats + row * 2) = f32x2{mean, rstd};
    float yv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) yv[i] = d[i] * rstd * g4[i] + b4[i];
    store4(y + row * kD + c, yv);
    if (q) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pos + (row % pos_rows) * kD + c);
      float qv[4] = {yv[0] + pv.x, yv[1] + pv.y, yv[2] + pv.z, yv[3] + pv.w};
      store4(q + row * kD + c, qv);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kRowsPerBlock * 64) void add_ln_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ dq, const T* __restrict__ z,
    const float* __restrict__ stats, const float* __restrict__ gamma, T* __restrict__ dres,
    T* __restrict__ dh, float* __restrict__ dgamma, float* __restrict__ dbeta, long long rows, Drop dr) {
  __shared__ float s_part[2][kRowsPerBlock][kD];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane * 4;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
  const float g4[4] = {gm.x, gm.y, gm.z, gm.w};
  float accg[4] = {0.f, 0.f, 0.f, 0.f}, accb[4] = {0.f, 0.f, 0.f, 0.f};
  for (long long row = (long long)blockIdx.x * kRowsPerBlock + wave; row < rows;
       row += (long long)gridDim.x * kRowsPerBlock) {
    float g[4], zv[4];
    load4(dy + row * kD + c, g);
    if (dq) {
      float t[4];
      load4(dq + row * kD + c, t);
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] += t[i];
    }
    load4(z + row * kD + c, zv);
    const f32x2 st = *reinterpret_cast<const f32x2*>(stats + row * 2);
    float xh[4], gg[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xh[i] = (zv[i] - st.x) * st.y;
      gg[i] = g[i] * g4[i];
      s1 += gg[i];
      s2 += gg[i] * xh[i];
      accg[i] += g[i] * xh[i];
      accb[i] += g[i];
    }
    s1 = wave_sum(s1) * (1.0f / kD);
    s2 = wave_sum(s2) * (1.0f / kD);
    float dz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dz[i] = st.y * (gg[i] - s1 - xh[i] * s2);
    if (dh) {
      float dv[4];
      const u32x4 rnd = philox((unsigned long long)row, (unsigned)lane, dr.seed, dr.offset);
      const unsigned r4[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) dv[i] = r4[i] < dr.thresh ? 0.f : dz[i] * dr.scale;
      store4(dh + row * kD + c, dv);
    }
    store4(dres + row * kD + c, dz);
  }
  *reinterpret_cast<f32x4*>(&s_part[0][wave][c]) = f32x4{accg[0], accg[1], accg[2], accg[3]};
  *reinterpret_cast<f32x4*>(&s_part[1][wave][c]) = f32x4{accb[0], accb[1], accb[2], accb[3]};
  __syncthreads();
  const int col = threadIdx.x;     // 256 threads, 256 columns
  float sg = 0.f, sb = 0.f;
#pragma unroll
  for (int w = 0; w < kRowsPerBlock; ++w) { sg += s_part[0][w][col]; sb += s_part[1][w][col]; }
  atomicAdd(dgamma + col, sg);
  atomicAdd(dbeta + col, sb);
}

inline int grid_for(long long rows) {
  const long long blocks = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
  return (int)(blocks < 2048 ? blocks : 2048);     // 8 workgroups per CU, grid-stride over rows
}

inline bool make_drop(float p, unsigned long long seed, unsigned long long offset, Drop* d) {
  if (!(p >= 0.f) || p >= 1.f) return false;
  const double t = (double)p * 4294967296.0;
  d->thresh = p > 0.f ? (unsigned)(t < 1.0 ? 1.0 : t) : 0u;
  d->scale = 1.0f / (1.0f - p);
  d->seed = seed;
  d->offset = offset;
  return true;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int dskd_add_ln_fwd(const void* h, const void* res, const float* pos, int64_t pos_rows,
                               const float* gamma, const float* beta, void* y, void* q, void* z,
                               float* stats, int64_t rows, int D, float eps, float drop_p,
                               uint64_t seed, uint64_t offset, int dtype, void* stream) {
  if (D != kD) return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_fwd: only D=256 supported (got %d)", D);
  if (dtype != DSKD_DTYPE_F32 && dtype != DSKD_DTYPE_BF16)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_fwd: unknown dtype %d", dtype);
  if (rows < 0 || !h || !res || !gamma || !beta || !y)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_fwd: null pointer or negative row count");
  if (q && (!pos || pos_rows <= 0))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_fwd: q requested without pos");
  if ((z == nullptr) != (stats == nullptr))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_fwd: z and stats are saved together");
  if (!aligned16(h) || !aligned16(res) || !aligned16(y) || !aligned16(q) || !aligned16(z) || !aligned16(pos) ||
      !aligned16(gamma) || !aligned16(beta) || !aligned16(stats))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_fwd: pointers must be 16-byte aligned");
  Drop dr;
  if (!make_drop(drop_p, seed, offset, &dr)) return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_fwd: drop_p=%f", drop_p);
  if (rows == 0) return DSKD_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for(rows)), block(kRowsPerBlock * 64);
  if (dtype == DSKD_DTYPE_F32)
    hipLaunchKernelGGL(add_ln_fwd_kernel<float>, grid, block, 0, st, (const float*)h, (const float*)res, pos,
                       (long long)pos_rows, gamma, beta, (float*)y, (float*)q, (float*)z, stats, (long long)rows,
                       eps, dr);
  else
    hipLaunchKernelGGL(add_ln_fwd_kernel<__bf16>, grid, block, 0, st, (const __bf16*)h, (const __bf16*)res, pos,
                       (long long)pos_rows, gamma, beta, (__bf16*)y, (__bf16*)q, (__bf16*)z, stats,
                       (long long)rows, eps, dr);
  return check_launch("dskd_add_ln_fwd");
}

extern "C" int dskd_add_ln_bwd(const void* dy, const void* dq, const void* z, const float* stats,
                               const float* gamma, void* dres, void* dh, float* dgamma, float* dbeta,
                               int64_t rows, int D, float drop_p, uint64_t seed, uint64_t offset,
                               int dtype, void* stream) {
  if (D != kD) return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: only D=256 supported (got %d)", D);
  if (dtype != DSKD_DTYPE_F32 && dtype != DSKD_DTYPE_BF16)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: unknown dtype %d", dtype);
  if (rows < 0 || !dy || !z || !stats || !gamma || !dres || !dgamma || !dbeta)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: null pointer or negative row count");
  if (!aligned16(dy) || !aligned16(dq) || !aligned16(z) || !aligned16(stats) || !aligned16(gamma) ||
      !aligned16(dres) || !aligned16(dh))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: pointers must be 16-byte aligned");
  Drop dr;
  if (!make_drop(drop_p, seed, offset, &dr)) return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: drop_p=%f", drop_p);
  if ((dr.thresh != 0) != (dh != nullptr))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: dh is written exactly when drop_p > 0 (else d(h) == d(res))");
  if (rows == 0) return DSKD_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for(rows)), block(kRowsPerBlock * 64);
  if (dtype == DSKD_DTYPE_F32)
    hipLaunchKernelGGL(add_ln_bwd_kernel<float>, grid, block, 0, st, (const float*)dy, (const float*)dq,
                       (const float*)z, stats, gamma, (float*)dres, (float*)dh, dgamma, dbeta, (long long)rows, dr);
  else
    hipLaunchKernelGGL(add_ln_bwd_kernel<__bf16>, grid, block, 0, st, (const __bf16*)dy, (const __bf16*)dq,
                       (const __bf16*)z, stats, gamma, (__bf16*)dres, (__bf16*)dh, dgamma, dbeta,
                       (long long)rows, dr);
  return check_launch("dskd_add_ln_bwd");
}
