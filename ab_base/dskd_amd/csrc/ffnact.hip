// Dropout after the FFN's Linear + ReLU, and the matching backward, as streaming passes.
//
// ext-mmcv FFN = Sequential(Linear, ReLU, Dropout), Linear, Dropout (mmcv/cnn/bricks/
// transformer.py; the feed-forward of every transformer layer run by mmdet/models/utils/
// transformer.py:454-483).  Its hidden activation is [88 892, 1024] at B=4 (182 MB in bf16), and
// PyTorch walks it five times around the two GEMMs: dropout forward (plus a 91 MB mask),
// dropout backward, ReLU backward, and the bias-gradient column sum.  Here:
//   forward   in-place dropout on the GEMM's bias+ReLU output, mask from Philox4x32-10, not stored
//   backward  g' = g * (y_dropped != 0) / (1 - p)  -- `y_dropped != 0` IS (ReLU active AND kept) --
//             and the column sums of g' (= the Linear's bias gradient) from the same read
// HBM-bound: 364 MB forward, 546 MB backward per layer instead of 455 + 1 183 MB.
#include "common.h"

namespace dskd {
namespace {

__device__ __forceinline__ u32x4 philox_ctr(unsigned long long idx, unsigned long long seed,
                                            unsigned long long offset) {
  unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
  const unsigned a = v.x, b = v.y, c = v.z, d = v.w;     // scalars first: see msda.hip on bit_cast
  f[0] = __builtin_bit_cast(float, a << 16); f[1] = __builtin_bit_cast(float, a & 0xFFFF0000u);
  f[2] = __builtin_bit_cast(float, b << 16); f[3] = __builtin_bit_cast(float, b & 0xFFFF0000u);
  f[4] = __builtin_bit_cast(float, c << 16); f[5] = __builtin_bit_cast(float, c & 0xFFFF0000u);
  f[6] = __builtin_bit_cast(float, d << 16); f[7] = __builtin_bit_cast(float, d & 0xFFFF0000u);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// y: [n] bf16, n % 8 == 0.  One 16-bit random field per element: drop when field < thresh16.
__global__ __launch_bounds__(256) void dropout_fwd_kernel(__bf16* __restrict__ y, long long nvec, unsigned thresh16,
                                                          float scale, unsigned long long seed,
                                                          unsigned long long offset0,
                                                          const unsigned long long* __restrict__ epoch) {
  const unsigned long long offset = offset0 + (epoch ? *epoch : 0ull);   // per-step part of the key: graph-replay safe
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(y + i * 8), f);
    const u32x4 r = philox_ctr((unsigned long long)i, seed, offset);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned field = (w[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
      f[k] = field < thresh16 ? 0.f : f[k] * scale;
    }
    const bf16x8 o = {(__bf16)f[0], (__bf16)f[1], (__bf16)f[2], (__bf16)f[3],
                      (__bf16)f[4], (__bf16)f[5], (__bf16)f[6], (__bf16)f[7]};
    *reinterpret_cast<bf16x8*>(y + i * 8) = o;
  }
}

// g, yd, out: [rows, C] bf16.  TPR = C / 8 threads cover a row; a thread keeps its 8 columns for
// the whole grid-stride loop, so the column sums accumulate in registers.
template <int TPR>
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const __bf16* __restrict__ g,
                                                               const __bf16* __restrict__ yd,
                                                               __bf16* __restrict__ out, float* __restrict__ colsum,
                                                               int copies, long long rows, float scale) {
  constexpr int RPB = 256 / TPR;                  // rows per workgroup and step
  constexpr int C = TPR * 8;
  __shared__ float s_part[RPB][C];
  const int t = threadIdx.x % TPR, rsub = threadIdx.x / TPR;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long row = (long long)blockIdx.x * RPB + rsub; row < rows; row += (long long)gridDim.x * RPB) {
    const long long e = row * C + t * 8;
    float gv[8], yv[8];
    unpack8(*reinterpret_cast<const u32x4*>(g + e), gv);
    unpack8(*reinterpret_cast<const u32x4*>(yd + e), yv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gv[k] = yv[k] != 0.f ? gv[k] * scale : 0.f;
      acc[k] += gv[k];
    }
    const bf16x8 o = {(__bf16)gv[0], (__bf16)gv[1], (__bf16)gv[2], (__bf16)gv[3],
                      (__bf16)gv[4], (__bf16)gv[5], (__bf16)gv[6], (__bf16)gv[7]};
    *reinterpret_cast<bf16x8*>(out + e) = o;
  }
  if (colsum) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s_part[rsub][t * 8 + k] = acc[k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < RPB; ++r) s += s_part[r][c];
      atomicAdd(colsum + (blockIdx.x % copies) * C + c, s);
    }
  }
}

// Column sums of a tall [rows, C] bf16 matrix (the bias gradient of a Linear: sum over tokens of
// the output gradient).  ATen's generic reduction takes ~40 us for 88 892 x 256 (45 MB); this is
// the same streaming structure as above: a thread keeps 8 columns, partial sums per workgroup in
// LDS, one atomic per column and workgroup.
template <int TPR>
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16* __restrict__ x, float* __restrict__ colsum,
                                                     int copies, long long rows) {
  constexpr int RPB = 256 / TPR;
  constexpr int C = TPR * 8;
  __shared__ float s_part[RPB][C];
  const int t = threadIdx.x % TPR, rsub = threadIdx.x / TPR;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rsub < RPB) {
    const long long step = (long long)gridDim.x * RPB;
    long long row = (long long)blockIdx.x * RPB + rsub;
    for (; row + 3 * step < rows; row += 4 * step) {       // four independent 16-byte loads in flight
      u32x4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const u32x4*>(x + (row + u * step) * C + t * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        unpack8(r[u], v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
      }
    }
    for (; row < rows; row += step) {
      float v[8];
      unpack8(*reinterpret_cast<const u32x4*>(x + row * C + t * 8), v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_part[rsub][t * 8 + k] = acc[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RPB; ++r) s += s_part[r][c];
    atomicAdd(colsum + (blockIdx.x % copies) * C + c, s);
  }
}

// Column sums of a SHORT [rows, C] bf16 matrix (the decoder's 1 200 query rows, the head branches' 7 200) written as bf16 in
// ONE launch: a workgroup owns 64 columns (8 lanes x 16 bytes) and all rows, 128 row lanes with independent 16-byte loads,
// one LDS tree.  No atomics, no zero fill, no cast pass (the tall form above needs all three); the ones-row GEMM it
// replaces costs 10-19 us through the library at these sizes.
__global__ __launch_bounds__(1024) void colsum_short_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ out,
                                                            int rows, int C) {
  __shared__ float s_part[128][65];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + cg * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < C) {
    for (int r = rl; r < rows; r += 128) {
      float v[8];
      unpack8(*reinterpret_cast<const u32x4*>(x + (size_t)r * C + c0), v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) s_part[rl][cg * 8 + k] = acc[k];
  __syncthreads();
  __shared__ float s_half[8][65];
  if (threadIdx.x < 512) {                       // two steps instead of a seven-barrier tree: 16 rows per thread, then 8
    const int g = threadIdx.x >> 6, c = threadIdx.x & 63;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += s_part[g * 16 + r][c];
    s_half[g][c] = t;
  }
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += s_half[g][threadIdx.x];
    out[blockIdx.x * 64 + threadIdx.x] = (__bf16)t;
  }
}

// out[p][c] = sum over the copies of acc[p][k][c]; acc = 0.  The column-sum accumulators of dskd_colsum / dskd_add_ln_bwd /
// dskd_ffn_bwd / dskd_relu_dropout_bwd are PERSISTENT buffers kept zeroed by this hand-over (as dskd_cvt_clear does for the
// weight gradients): one launch where a zero fill before, a reduction over the copies and a cast after made three.
template <typename TO>
__global__ __launch_bounds__(256) void sum_clear_kernel(float* __restrict__ acc, int planes, int copies, int C,
                                                        TO* __restrict__ out) {
  // 32 outputs per workgroup, 8 lanes of copies each (copy k, k + 8, ..: independent loads), one LDS step over the 8
  __shared__ float s_part[8][33];
  const int cl = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (i < planes * C) {
    const int p = i / C, c = i - p * C;
    float* a = acc + (size_t)p * copies * C + c;
    for (int k = kg; k < copies; k += 8) {
      s += a[(size_t)k * C];
      a[(size_t)k * C] = 0.f;
    }
  }
  s_part[kg][cl] = s;
  __syncthreads();
  if (kg == 0 && i < planes * C) {
    float t = s_part[0][cl];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += s_part[k][cl];
    out[i] = (TO)t;
  }
}

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int dskd_colsum(const void* x, float* colsum, int copies, int64_t rows, int C, int dtype,
                           void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_colsum: bf16 only");
  if (rows == 0) return DSKD_OK;
  if (!x || !colsum || rows < 0 || copies < 1 || (reinterpret_cast<uintptr_t>(x) & 15))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_colsum: null / unaligned pointer, negative row count or copies < 1");
  hipStream_t st = (hipStream_t)stream;
  const __bf16* xp = (const __bf16*)x;
#define DSKD_LAUNCH_COLSUM(TPR)                                                                          \
  {                                                                                                      \
    const long long want = (rows + (256 / TPR) - 1) / (256 / TPR);                                       \
    hipLaunchKernelGGL(colsum_kernel<TPR>, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(256), 0,   \
                       st, xp, colsum, copies, (long long)rows);                                         \
  }
  switch (C) {
    case 256: DSKD_LAUNCH_COLSUM(32) break;
    case 384: DSKD_LAUNCH_COLSUM(48) break;
    case 512: DSKD_LAUNCH_COLSUM(64) break;
    case 1024: DSKD_LAUNCH_COLSUM(128) break;
    case 2048: DSKD_LAUNCH_COLSUM(256) break;
    default: return fail(DSKD_ERR_INVALID_ARG, "dskd_colsum: C must be 256, 384, 512, 1024 or 2048 (got %d)", C);
  }
#undef DSKD_LAUNCH_COLSUM
  return check_launch("dskd_colsum");
}

extern "C" int dskd_sum_clear(float* acc, int planes, int copies, int C, void* out, int out_dtype, void* stream) {
  if (!acc || !out || planes < 1 || copies < 1 || C < 1)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_sum_clear: null pointer or planes / copies / C < 1");
  const unsigned blocks = (unsigned)(((long long)planes * C + 31) / 32);
  if (out_dtype == DSKD_DTYPE_BF16)
    hipLaunchKernelGGL(sum_clear_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, planes, copies, C,
                       (__bf16*)out);
  else if (out_dtype == DSKD_DTYPE_F32)
    hipLaunchKernelGGL(sum_clear_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, planes, copies, C,
                       (float*)out);
  else
    return fail(DSKD_ERR_INVALID_ARG, "dskd_sum_clear: out_dtype must be f32 or bf16");
  return check_launch("dskd_sum_clear");
}

extern "C" int dskd_colsum_short(const void* x, void* out, int64_t rows, int C, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_colsum_short: bf16 only");
  if (!x || !out || rows < 0 || rows > 65536 || C <= 0 || (C & 7) || (reinterpret_cast<uintptr_t>(x) & 15))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_colsum_short: null / unaligned pointer, rows outside [0, 65536] or C %% 8 != 0");
  hipLaunchKernelGGL(colsum_short_kernel, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, (hipStream_t)stream,
                     (const __bf16*)x, (__bf16*)out, (int)rows, C);
  return check_launch("dskd_colsum_short");
}

extern "C" int dskd_dropout_fwd(void* y, int64_t n, float p, uint64_t seed, uint64_t offset,
                                const uint64_t* epoch, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_dropout_fwd: bf16 only");
  if (!y || n < 0 || n % 8 != 0 || (reinterpret_cast<uintptr_t>(y) & 15))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_dropout_fwd: need a 16-byte aligned buffer of n %% 8 == 0 elements");
  if (!(p >= 0.f) || p >= 1.f) return fail(DSKD_ERR_INVALID_ARG, "dskd_dropout_fwd: p=%f", p);
  if (n == 0 || p == 0.f) return DSKD_OK;
  const unsigned t = (unsigned)((double)p * 65536.0 + 0.5);
  const long long nvec = n / 8, want = (nvec + 255) / 256;
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0,
                     (hipStream_t)stream, (__bf16*)y, nvec, t < 1 ? 1u : t, 1.0f / (1.0f - p),
                     (unsigned long long)seed, (unsigned long long)offset,
                     reinterpret_cast<const unsigned long long*>(epoch));
  return check_launch("dskd_dropout_fwd");
}

extern "C" int dskd_relu_dropout_bwd(const void* g, const void* y_dropped, void* out, float* colsum,
                                     int copies, int64_t rows, int C, float p, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_relu_dropout_bwd: bf16 only");
  if (!g || !y_dropped || !out || rows < 0 || copies < 1)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_relu_dropout_bwd: null pointer, negative row count or copies < 1");
  if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y_dropped) | reinterpret_cast<uintptr_t>(out)) & 15)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_relu_dropout_bwd: pointers must be 16-byte aligned");
  if (!(p >= 0.f) || p >= 1.f) return fail(DSKD_ERR_INVALID_ARG, "dskd_relu_dropout_bwd: p=%f", p);
  if (rows == 0) return DSKD_OK;
  const float scale = 1.0f / (1.0f - p);
  hipStream_t st = (hipStream_t)stream;
  const __bf16* gp = (const __bf16*)g;
  const __bf16* yp = (const __bf16*)y_dropped;
  __bf16* op = (__bf16*)out;
#define DSKD_LAUNCH_TPR(TPR)                                                                          \
  {                                                                                                   \
    const long long want = (rows + (256 / TPR) - 1) / (256 / TPR);                                    \
    hipLaunchKernelGGL(relu_dropout_bwd_kernel<TPR>, dim3((unsigned)(want < 2048 ? want : 2048)),    \
                       dim3(256), 0, st, gp, yp, op, colsum, copies, (long long)rows, scale);         \
  }
  switch (C) {
    case 256: DSKD_LAUNCH_TPR(32) break;
    case 512: DSKD_LAUNCH_TPR(64) break;
    case 1024: DSKD_LAUNCH_TPR(128) break;
    case 2048: DSKD_LAUNCH_TPR(256) break;
    default: return fail(DSKD_ERR_INVALID_ARG, "dskd_relu_dropout_bwd: C must be 256, 512, 1024 or 2048 (got %d)", C);
  }
#undef DSKD_LAUNCH_TPR
  return check_launch("dskd_relu_dropout_bwd");
}
