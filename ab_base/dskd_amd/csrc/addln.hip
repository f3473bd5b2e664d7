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
// HBM-bound streaming kernels.  D == 256 only (the DSKD transformer width): 16 bytes per lane
// (4 f32 / 8 bf16 columns), so a 64-lane wave owns one f32 row or two bf16 rows per step; the
// row statistics are two shuffle reductions over the row's lanes.
#include "common.h"

namespace dskd {
namespace {

constexpr int kD = 256;
constexpr int kRowsPerBlock = 4;       // waves per workgroup

__device__ __forceinline__ float as_float(unsigned u) { return __builtin_bit_cast(float, u); }

// 16 bytes per lane: 4 f32 or 8 bf16 elements; a 256-wide row takes 64 or 32 lanes, so a wave
// owns one f32 row or two bf16 rows per step.
template <typename T> struct Lay;
template <> struct Lay<float> { static constexpr int EPL = 4, LPR = 64, RPW = 1; };
template <> struct Lay<__bf16> { static constexpr int EPL = 8, LPR = 32, RPW = 2; };

template <typename T>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, float* f) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p);
    const unsigned a = v.x, b = v.y, c = v.z, d = v.w;      // scalars first (bit_cast of a vector element reads [0])
    f[0] = as_float(a << 16); f[1] = as_float(a & 0xFFFF0000u);
    f[2] = as_float(b << 16); f[3] = as_float(b & 0xFFFF0000u);
    f[4] = as_float(c << 16); f[5] = as_float(c & 0xFFFF0000u);
    f[6] = as_float(d << 16); f[7] = as_float(d & 0xFFFF0000u);
  }
}

// round to T and return the rounded values in f (what a later reader of the store will see)
template <typename T>
__device__ __forceinline__ void store_vec(T* __restrict__ p, float* f) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
  } else {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const bf16x8 v = {(__bf16)f[0], (__bf16)f[1], (__bf16)f[2], (__bf16)f[3],
                      (__bf16)f[4], (__bf16)f[5], (__bf16)f[6], (__bf16)f[7]};
    *reinterpret_cast<bf16x8*>(p) = v;
    f[0] = (float)v[0]; f[1] = (float)v[1]; f[2] = (float)v[2]; f[3] = (float)v[3];
    f[4] = (float)v[4]; f[5] = (float)v[5]; f[6] = (float)v[6]; f[7] = (float)v[7];
  }
}

// sum over the LPR lanes of a row (LPR = 32: the two halves of the wave reduce independently)
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Philox4x32-10 (Salmon et al. 2011): 4 x 32 random bits per (row, lane) = 8 x 16-bit fields.
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
  unsigned thresh;          // drop when the element's 16-bit random field < thresh  (thresh = p * 2^16)
  float scale;              // 1 / (1 - p)
  unsigned long long seed, offset;
  const unsigned long long* epoch;   // device word added to `offset` (or NULL): the per-step part of the key, so that a
                                     // launch captured into a hipGraph draws a new mask on every replay
};
__device__ __forceinline__ unsigned long long drop_offset(const Drop& dr) {
  return dr.offset + (dr.epoch ? *dr.epoch : 0ull);
}

// element i (< 8) of a lane is dropped when its 16-bit field is below the threshold
__device__ __forceinline__ bool dropped(const u32x4& rnd, int i, unsigned thresh) {
  const unsigned w[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
  return ((w[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu) < thresh;
}

template <typename T>
__global__ __launch_bounds__(kRowsPerBlock * 64) void add_ln_fwd_kernel(
    const T* __restrict__ h, const T* __restrict__ res, const float* __restrict__ pos, long long pos_rows,
    const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, T* __restrict__ q,
    T* __restrict__ z, float* __restrict__ stats, long long rows, float eps, Drop dr) {
  using L = Lay<T>;
  constexpr int E = L::EPL;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rl = lane % L::LPR, rsub = lane / L::LPR;       // lane inside its row, row of the wave
  const int c = rl * E;
  float g4[E], b4[E];
#pragma unroll
  for (int i = 0; i < E; i += 4) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c + i);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c + i);
    g4[i] = gm.x; g4[i + 1] = gm.y; g4[i + 2] = gm.z; g4[i + 3] = gm.w;
    b4[i] = bt.x; b4[i + 1] = bt.y; b4[i + 2] = bt.z; b4[i + 3] = bt.w;
  }
  const long long stride = (long long)gridDim.x * kRowsPerBlock * L::RPW;
  for (long long row0 = ((long long)blockIdx.x * kRowsPerBlock + wave) * L::RPW; row0 < rows; row0 += stride) {
    const long long row = row0 + rsub;
    const bool ok = row < rows;
    const long long rr = ok ? row : rows - 1;               // the idle half of a wave recomputes the last row
    float hv[E], zv[E];
    load_vec(h + rr * kD + c, hv);
    load_vec(res + rr * kD + c, zv);
    if (dr.thresh) {
      const u32x4 rnd = philox((unsigned long long)rr, (unsigned)rl, dr.seed, drop_offset(dr));
#pragma unroll
      for (int i = 0; i < E; ++i) hv[i] = dropped(rnd, i, dr.thresh) ? 0.f : hv[i] * dr.scale;
    }
#pragma unroll
    for (int i = 0; i < E; ++i) zv[i] += hv[i];
    if (z) {
      if (ok) store_vec(z + rr * kD + c, zv);              // statistics of the ROUNDED sum: backward sees the same z
      else if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < E; ++i) zv[i] = (float)(__bf16)zv[i];
      }
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int i = 0; i < E; ++i) zv[i] = (float)(__bf16)zv[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) s += zv[i];
    const float mean = row_sum<L::LPR>(s) * (1.0f / kD);
    float d[E], ss = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) { d[i] = zv[i] - mean; ss += d[i] * d[i]; }
    const float rstd = rsqrtf(row_sum<L::LPR>(ss) * (1.0f / kD) + eps);
    if (!ok) continue;
    if (stats && rl == 0) *reinterpret_cast<f32x2*>(stats + row * 2) = f32x2{mean, rstd};
    float yv[E];
#pragma unroll
    for (int i = 0; i < E; ++i) yv[i] = d[i] * rstd * g4[i] + b4[i];
    store_vec(y + row * kD + c, yv);
    if (q) {
      float qv[E];
#pragma unroll
      for (int i = 0; i < E; i += 4) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(pos + (row % pos_rows) * kD + c + i);
        qv[i] = yv[i] + pv.x; qv[i + 1] = yv[i + 1] + pv.y; qv[i + 2] = yv[i + 2] + pv.z; qv[i + 3] = yv[i + 3] + pv.w;
      }
      store_vec(q + row * kD + c, qv);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kRowsPerBlock * 64) void add_ln_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ dq, const T* __restrict__ z,
    const float* __restrict__ stats, const float* __restrict__ gamma, T* __restrict__ dres,
    T* __restrict__ dh, float* __restrict__ dgamma, float* __restrict__ dbeta, int copies, long long rows,
    Drop dr) {
  using L = Lay<T>;
  constexpr int E = L::EPL;
  __shared__ float s_part[2][kRowsPerBlock * L::RPW][kD];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rl = lane % L::LPR, rsub = lane / L::LPR;
  const int c = rl * E;
  float g4[E];
#pragma unroll
  for (int i = 0; i < E; i += 4) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c + i);
    g4[i] = gm.x; g4[i + 1] = gm.y; g4[i + 2] = gm.z; g4[i + 3] = gm.w;
  }
  float accg[E], accb[E];
#pragma unroll
  for (int i = 0; i < E; ++i) { accg[i] = 0.f; accb[i] = 0.f; }
  const long long stride = (long long)gridDim.x * kRowsPerBlock * L::RPW;
  for (long long row0 = ((long long)blockIdx.x * kRowsPerBlock + wave) * L::RPW; row0 < rows; row0 += stride) {
    const long long row = row0 + rsub;
    const bool ok = row < rows;
    const long long rr = ok ? row : rows - 1;
    float g[E], zv[E];
    load_vec(dy + rr * kD + c, g);
    if (dy2) {          // r4: y had two consumers -- their gradients arrive as two tensors and are summed here, not by a launch
      float t[E];
      load_vec(dy2 + rr * kD + c, t);
#pragma unroll
      for (int i = 0; i < E; ++i) g[i] += t[i];
    }
    if (dq) {
      float t[E];
      load_vec(dq + rr * kD + c, t);
#pragma unroll
      for (int i = 0; i < E; ++i) g[i] += t[i];
    }
    load_vec(z + rr * kD + c, zv);
    const f32x2 st = *reinterpret_cast<const f32x2*>(stats + rr * 2);
    float xh[E], gg[E], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) {
      xh[i] = (zv[i] - st.x) * st.y;
      gg[i] = g[i] * g4[i];
      s1 += gg[i];
      s2 += gg[i] * xh[i];
    }
    s1 = row_sum<L::LPR>(s1) * (1.0f / kD);
    s2 = row_sum<L::LPR>(s2) * (1.0f / kD);
    if (!ok) continue;
    float dz[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      accg[i] += g[i] * xh[i];
      accb[i] += g[i];
      dz[i] = st.y * (gg[i] - s1 - xh[i] * s2);
    }
    if (dh) {
      float dv[E];
      const u32x4 rnd = philox((unsigned long long)row, (unsigned)rl, dr.seed, drop_offset(dr));
#pragma unroll
      for (int i = 0; i < E; ++i) dv[i] = dropped(rnd, i, dr.thresh) ? 0.f : dz[i] * dr.scale;
      store_vec(dh + row * kD + c, dv);
    }
    store_vec(dres + row * kD + c, dz);
  }
  const int slot = wave * L::RPW + rsub;
#pragma unroll
  for (int i = 0; i < E; ++i) { s_part[0][slot][c + i] = accg[i]; s_part[1][slot][c + i] = accb[i]; }
  __syncthreads();
  const int col = threadIdx.x;     // 256 threads, 256 columns
  float sg = 0.f, sb = 0.f;
#pragma unroll
  for (int w = 0; w < kRowsPerBlock * L::RPW; ++w) { sg += s_part[0][w][col]; sb += s_part[1][w][col]; }
  const int copy = blockIdx.x % copies;      // several copies: fewer workgroups contend for one address
  atomicAdd(dgamma + copy * kD + col, sg);
  atomicAdd(dbeta + copy * kD + col, sb);
}

inline int grid_for(long long rows, int rows_per_wave) {
  const long long per_block = (long long)kRowsPerBlock * rows_per_wave;
  const long long blocks = (rows + per_block - 1) / per_block;
  return (int)(blocks < 2048 ? blocks : 2048);     // 8 workgroups per CU, grid-stride over rows
}

inline bool make_drop(float p, unsigned long long seed, unsigned long long offset, const uint64_t* epoch, Drop* d) {
  if (!(p >= 0.f) || p >= 1.f) return false;
  const double t = (double)p * 65536.0 + 0.5;
  d->thresh = p > 0.f ? (unsigned)(t < 1.0 ? 1.0 : t) : 0u;
  d->scale = 1.0f / (1.0f - p);
  d->seed = seed;
  d->offset = offset;
  d->epoch = reinterpret_cast<const unsigned long long*>(epoch);
  return true;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace dskd

using namespace dskd;

// q = T(x + pos[r % pos_rows]) for a [rows, D] bf16 token tensor and an f32 positional table: the first encoder layer's
// ``query + query_pos`` (every later one comes out of add_ln_fwd).  ATen runs the mixed-dtype add as a generic kernel
// (120 us at 88 892 x 256) followed by a cast (30 us); this is one streaming pass (45 + 91 MB in, 45 MB out).
__global__ __launch_bounds__(256) void add_pos_kernel(const __bf16* __restrict__ x, const float* __restrict__ pos,
                                                      __bf16* __restrict__ q, long long nvec, long long pos_nvec) {
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
    const long long pi = (i % pos_nvec) * 8;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(pos + pi), p1 = *reinterpret_cast<const f32x4*>(pos + pi + 4);
    const unsigned a = v.x, b = v.y, c = v.z, d = v.w;
    const bf16x8_t o = {(__bf16)(__builtin_bit_cast(float, a << 16) + p0.x), (__bf16)(__builtin_bit_cast(float, a & 0xFFFF0000u) + p0.y),
                        (__bf16)(__builtin_bit_cast(float, b << 16) + p0.z), (__bf16)(__builtin_bit_cast(float, b & 0xFFFF0000u) + p0.w),
                        (__bf16)(__builtin_bit_cast(float, c << 16) + p1.x), (__bf16)(__builtin_bit_cast(float, c & 0xFFFF0000u) + p1.y),
                        (__bf16)(__builtin_bit_cast(float, d << 16) + p1.z), (__bf16)(__builtin_bit_cast(float, d & 0xFFFF0000u) + p1.w)};
    *reinterpret_cast<bf16x8_t*>(q + i * 8) = o;
  }
}

extern "C" int dskd_add_pos(const void* x, const float* pos, void* q, int64_t rows, int64_t pos_rows, int D, int dtype,
                            void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_add_pos: bf16 only");
  if (!x || !pos || !q || rows < 0 || pos_rows <= 0 || D <= 0 || D % 8 || rows % pos_rows)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_pos: null pointer, D %% 8 != 0 or rows not a multiple of pos_rows");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(q)) & 15)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_pos: pointers must be 16-byte aligned");
  if (rows == 0) return DSKD_OK;
  const long long nvec = (long long)rows * D / 8, want = (nvec + 255) / 256;
  hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)x, pos, (__bf16*)q, nvec, (long long)pos_rows * D / 8);
  return check_launch("dskd_add_pos");
}

extern "C" int dskd_add_ln_fwd(const void* h, const void* res, const float* pos, int64_t pos_rows,
                               const float* gamma, const float* beta, void* y, void* q, void* z,
                               float* stats, int64_t rows, int D, float eps, float drop_p,
                               uint64_t seed, uint64_t offset, const uint64_t* epoch, int dtype, void* stream) {
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
  if (!make_drop(drop_p, seed, offset, epoch, &dr)) return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_fwd: drop_p=%f", drop_p);
  if (rows == 0) return DSKD_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for(rows, dtype == DSKD_DTYPE_F32 ? 1 : 2)), block(kRowsPerBlock * 64);
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

extern "C" int dskd_add_ln_bwd2(const void* dy, const void* dy2, const void* dq, const void* z, const float* stats,
                                const float* gamma, void* dres, void* dh, float* dgamma, float* dbeta,
                                int copies, int64_t rows, int D, float drop_p, uint64_t seed,
                                uint64_t offset, const uint64_t* epoch, int dtype, void* stream);

extern "C" int dskd_add_ln_bwd(const void* dy, const void* dq, const void* z, const float* stats,
                               const float* gamma, void* dres, void* dh, float* dgamma, float* dbeta,
                               int copies, int64_t rows, int D, float drop_p, uint64_t seed,
                               uint64_t offset, const uint64_t* epoch, int dtype, void* stream) {
  return dskd_add_ln_bwd2(dy, nullptr, dq, z, stats, gamma, dres, dh, dgamma, dbeta, copies, rows, D, drop_p, seed, offset,
                          epoch, dtype, stream);
}

extern "C" int dskd_add_ln_bwd2(const void* dy, const void* dy2, const void* dq, const void* z, const float* stats,
                                const float* gamma, void* dres, void* dh, float* dgamma, float* dbeta,
                                int copies, int64_t rows, int D, float drop_p, uint64_t seed,
                                uint64_t offset, const uint64_t* epoch, int dtype, void* stream) {
  if (D != kD) return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: only D=256 supported (got %d)", D);
  if (dtype != DSKD_DTYPE_F32 && dtype != DSKD_DTYPE_BF16)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: unknown dtype %d", dtype);
  if (rows < 0 || copies < 1 || !dy || !z || !stats || !gamma || !dres || !dgamma || !dbeta)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: null pointer, negative row count or copies < 1");
  if (!aligned16(dy) || !aligned16(dy2) || !aligned16(dq) || !aligned16(z) || !aligned16(stats) || !aligned16(gamma) ||
      !aligned16(dres) || !aligned16(dh))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: pointers must be 16-byte aligned");
  Drop dr;
  if (!make_drop(drop_p, seed, offset, epoch, &dr)) return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: drop_p=%f", drop_p);
  if ((dr.thresh != 0) != (dh != nullptr))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_add_ln_bwd: dh is written exactly when drop_p > 0 (else d(h) == d(res))");
  if (rows == 0) return DSKD_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for(rows, dtype == DSKD_DTYPE_F32 ? 1 : 2)), block(kRowsPerBlock * 64);
  if (dtype == DSKD_DTYPE_F32)
    hipLaunchKernelGGL(add_ln_bwd_kernel<float>, grid, block, 0, st, (const float*)dy, (const float*)dy2, (const float*)dq,
                       (const float*)z, stats, gamma, (float*)dres, (float*)dh, dgamma, dbeta, copies, (long long)rows, dr);
  else
    hipLaunchKernelGGL(add_ln_bwd_kernel<__bf16>, grid, block, 0, st, (const __bf16*)dy, (const __bf16*)dy2, (const __bf16*)dq,
                       (const __bf16*)z, stats, gamma, (__bf16*)dres, (__bf16*)dh, dgamma, dbeta, copies,
                       (long long)rows, dr);
  return check_launch("dskd_add_ln_bwd");
}
