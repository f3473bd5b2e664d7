// GroupNorm(32, 256) of the neck on channels_last activations, two streaming passes each way.
//
// ChannelMapper = conv -> GroupNorm per level (mmdet/models/necks/channel_mapper.py:10-100, ext-mmcv ConvModule with
// norm_cfg=dict(type='GN', num_groups=32)); teacher and student each run it on four levels (100x167 ... 13x21 at
// 800x1333).  Through ATen under autocast the level-0 map (4 x 256 x 16 700) takes a bf16->f32 cast, a channels_last ->
// NCHW copy, RowwiseMoments, the normalisation pass and, later, the transposing copy into the encoder's token tensor:
// ~350 us per model and step.  In channels_last memory a pixel is one 512-byte row holding all 32 groups of 8 channels,
// so with lane = group one pass over the rows yields every group's moments and a second pass normalises:
//   forward    gn_stats (sum, sum of squares per (image, group): f32 per thread over <= 128 values, f64 per workgroup,
//              written as partials) -> gn_finalize (partials summed in a FIXED order: the forward is deterministic --
//              an untrained detector sits on assignment near-ties, and last-bit noise in the features flips them)
//              gn_apply (y = (x - mean) rstd gamma + beta, output dtype = input dtype, same layout = token-major rows)
//   backward   gn_bwd_stats (S1 = sum gy, S2 = sum gy xhat per (image, group); dgamma, dbeta per channel)
//              gn_bwd_apply (dx = rstd (gy - S1/N - xhat S2/N), gy = dy gamma)
// HBM-bound: 2 reads + 1 write of the map forward, 4 reads + 1 write backward.
#include "common.h"

namespace dskd {
namespace {

constexpr int kC = 256, kG = 32, kCpg = 8;
constexpr int kRows = 8;                 // pixel rows per workgroup pass (256 threads = 8 rows x 32 groups)

template <typename T>
__device__ __forceinline__ void load8(const T* p, float* f);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float* f) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<__bf16>(const __bf16* p, float* f) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(p);
  const unsigned a = v.x, b = v.y, c = v.z, d = v.w;
  f[0] = __builtin_bit_cast(float, a << 16); f[1] = __builtin_bit_cast(float, a & 0xFFFF0000u);
  f[2] = __builtin_bit_cast(float, b << 16); f[3] = __builtin_bit_cast(float, b & 0xFFFF0000u);
  f[4] = __builtin_bit_cast(float, c << 16); f[5] = __builtin_bit_cast(float, c & 0xFFFF0000u);
  f[6] = __builtin_bit_cast(float, d << 16); f[7] = __builtin_bit_cast(float, d & 0xFFFF0000u);
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float* f);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float* f) {
  *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <>
__device__ __forceinline__ void store8<__bf16>(__bf16* p, const float* f) {
  const bf16x8 o = {(__bf16)f[0], (__bf16)f[1], (__bf16)f[2], (__bf16)f[3],
                    (__bf16)f[4], (__bf16)f[5], (__bf16)f[6], (__bf16)f[7]};
  *reinterpret_cast<bf16x8*>(p) = o;
}

// part[b][chunk][g] = {sum, sum of squares} in f64.  grid (chunks, B); a workgroup walks `span` pixels.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, double* __restrict__ part, long long HW,
                                                       long long x_bs, int span) {
  __shared__ float s_part[kRows][kG][2];
  const int g = threadIdx.x & 31, row = threadIdx.x >> 5, b = blockIdx.y;
  const long long p0 = (long long)blockIdx.x * span, p1 = p0 + span < HW ? p0 + span : HW;
  const T* xb = x + (size_t)b * x_bs + g * kCpg;
  float s = 0.f, ss = 0.f;
  for (long long p = p0 + row; p < p1; p += kRows) {
    float v[8];
    load8<T>(xb + p * kC, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += v[k]; ss += v[k] * v[k]; }
  }
  s_part[row][g][0] = s; s_part[row][g][1] = ss;
  __syncthreads();
  if (threadIdx.x < 2 * kG) {
    const int gg = threadIdx.x >> 1, k = threadIdx.x & 1;
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < kRows; ++r) t += (double)s_part[r][gg][k];
    part[(((size_t)b * gridDim.x + blockIdx.x) * kG + gg) * 2 + k] = t;
  }
}

// stats[b][g] = {mean, rstd} (f32) from the partial sums in a FIXED order: thread (j, g) adds chunks j, j + 8, ... and
// the eight partial results are added in order of j.  grid B, 256 threads (a single thread per group walking all
// chunks is a chain of ~130 dependent L2 loads: 16 us).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, float* __restrict__ stats,
                                                         int chunks, long long HW, float eps) {
  __shared__ double s_acc[kRows][kG][2];
  const int g = threadIdx.x & 31, j = threadIdx.x >> 5, b = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int c = j; c < chunks; c += kRows) {
    s += part[(((size_t)b * chunks + c) * kG + g) * 2];
    ss += part[(((size_t)b * chunks + c) * kG + g) * 2 + 1];
  }
  s_acc[j][g][0] = s; s_acc[j][g][1] = ss;
  __syncthreads();
  if (j == 0) {
    s = 0.0; ss = 0.0;
#pragma unroll
    for (int r = 0; r < kRows; ++r) { s += s_acc[r][g][0]; ss += s_acc[r][g][1]; }
    const double n = (double)HW * kCpg, mean = s / n;
    double var = ss / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    stats[((size_t)b * kG + g) * 2] = (float)mean;
    stats[((size_t)b * kG + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// y in x's dtype and layout.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, long long HW, long long x_bs, long long y_bs,
                                                       int span, int relu) {
  const int g = threadIdx.x & 31, row = threadIdx.x >> 5, b = blockIdx.y;
  const float mean = stats[((size_t)b * kG + g) * 2], rstd = stats[((size_t)b * kG + g) * 2 + 1];
  float ga[8], be[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ga[k] = gamma[g * kCpg + k] * rstd;
    be[k] = beta[g * kCpg + k] - mean * ga[k];
  }
  const long long p0 = (long long)blockIdx.x * span, p1 = p0 + span < HW ? p0 + span : HW;
  const T* xb = x + (size_t)b * x_bs + g * kCpg;
  T* yb = y + (size_t)b * y_bs + g * kCpg;
  for (long long p = p0 + row; p < p1; p += kRows) {
    float v[8];
    load8<T>(xb + p * kC, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = v[k] * ga[k] + be[k];
      if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    store8<T>(yb + p * kC, v);
  }
}

// S[b][g] = {sum gy, sum gy xhat} (f64); dgb[copy][0][c] += sum dy xhat, dgb[copy][1][c] += sum dy.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           double* __restrict__ S, float* __restrict__ dgb, int copies,
                                                           long long HW, long long x_bs, long long dy_bs, int span,
                                                           int relu) {
  __shared__ float s_part[kRows][kC][2];
  const int g = threadIdx.x & 31, row = threadIdx.x >> 5, b = blockIdx.y;
  const float mean = stats[((size_t)b * kG + g) * 2], rstd = stats[((size_t)b * kG + g) * 2 + 1];
  float ga[8], fa[8], fb[8];                            // fa, fb: the forward's y = x fa + fb (the ReLU mask is its sign)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ga[k] = gamma[g * kCpg + k];
    fa[k] = ga[k] * rstd;
    fb[k] = (relu ? beta[g * kCpg + k] : 0.f) - mean * fa[k];
  }
  const long long p0 = (long long)blockIdx.x * span, p1 = p0 + span < HW ? p0 + span : HW;
  const T* xb = x + (size_t)b * x_bs + g * kCpg;
  const T* db = dy + (size_t)b * dy_bs + g * kCpg;
  float dg[8], dbt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) dg[k] = dbt[k] = 0.f;
  for (long long p = p0 + row; p < p1; p += kRows) {
    float v[8], d[8];
    load8<T>(xb + p * kC, v);
    load8<T>(db + p * kC, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (v[k] - mean) * rstd;
      if (relu && !(v[k] * fa[k] + fb[k] > 0.f)) d[k] = 0.f;
      dg[k] += d[k] * xh;
      dbt[k] += d[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    s_part[row][g * kCpg + k][0] = dg[k];
    s_part[row][g * kCpg + k][1] = dbt[k];
  }
  __syncthreads();
  {
    const int c = threadIdx.x;                         // 256 threads = 256 channels
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int r = 0; r < kRows; ++r) { t0 += s_part[r][c][0]; t1 += s_part[r][c][1]; }
    float* dst = dgb + (size_t)((blockIdx.x + blockIdx.y * gridDim.x) % copies) * 2 * kC;
    atomicAdd(dst + c, t0);
    atomicAdd(dst + kC + c, t1);
    // group sums of gy = dy gamma and gy xhat from the channel sums: S1 = sum_c gamma_c dbeta_c, S2 = sum_c gamma_c dgamma_c
    s_part[0][c][0] = t1 * gamma[c];
    s_part[0][c][1] = t0 * gamma[c];
  }
  __syncthreads();
  if (threadIdx.x < 2 * kG) {
    const int gg = threadIdx.x >> 1, k = threadIdx.x & 1;
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < kCpg; ++j) t += (double)s_part[0][gg * kCpg + j][k];
    atomicAdd(S + ((size_t)b * kG + gg) * 2 + k, t);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const double* __restrict__ S, T* __restrict__ dx, long long HW,
                                                           long long x_bs, long long dy_bs, long long dx_bs, int span,
                                                           int relu) {
  const int g = threadIdx.x & 31, row = threadIdx.x >> 5, b = blockIdx.y;
  const float mean = stats[((size_t)b * kG + g) * 2], rstd = stats[((size_t)b * kG + g) * 2 + 1];
  const double n = (double)HW * kCpg;
  const float m1 = (float)(S[((size_t)b * kG + g) * 2] / n), m2 = (float)(S[((size_t)b * kG + g) * 2 + 1] / n);
  float ga[8], fa[8], fb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ga[k] = gamma[g * kCpg + k];
    fa[k] = ga[k] * rstd;
    fb[k] = (relu ? beta[g * kCpg + k] : 0.f) - mean * fa[k];
  }
  const long long p0 = (long long)blockIdx.x * span, p1 = p0 + span < HW ? p0 + span : HW;
  const T* xb = x + (size_t)b * x_bs + g * kCpg;
  const T* db = dy + (size_t)b * dy_bs + g * kCpg;
  T* ob = dx + (size_t)b * dx_bs + g * kCpg;
  for (long long p = p0 + row; p < p1; p += kRows) {
    float v[8], d[8];
    load8<T>(xb + p * kC, v);
    load8<T>(db + p * kC, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (v[k] - mean) * rstd;
      if (relu && !(v[k] * fa[k] + fb[k] > 0.f)) d[k] = 0.f;
      v[k] = rstd * (d[k] * ga[k] - m1 - xh * m2);
    }
    store8<T>(ob + p * kC, v);
  }
}

// out[b][c][p] (f32, NCHW planes) = x[b][p][c] (channels_last rows, f32 | bf16): the layout + dtype the feature-map
// distillation kernel (fgkd.hip) reads.  ATen does this as a strided copy plus a cast (82 + 40 us for the level-0 map);
// here 32 pixels x 256 channels go through LDS: 512-byte rows in, 128-byte runs out.  grid (ceil(HW / 32), B).
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const T* __restrict__ x, float* __restrict__ out,
                                                               long long HW, long long x_bs) {
  __shared__ float tile[kC][33];
  const int g = threadIdx.x & 31, row = threadIdx.x >> 5, b = blockIdx.y;
  const long long p0 = (long long)blockIdx.x * 32;
  const T* xb = x + (size_t)b * x_bs + g * kCpg;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = row + 8 * i;
    if (p0 + pl < HW) {
      float v[8];
      load8<T>(xb + (p0 + pl) * kC, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) tile[g * kCpg + k][pl] = v[k];
    }
  }
  __syncthreads();
  float* ob = out + (size_t)b * kC * HW + p0;
  if (p0 + g < HW) {
#pragma unroll 8
    for (int c = row; c < kC; c += 8) ob[(size_t)c * HW + g] = tile[c][g];
  }
}

int gn_span(long long HW) { return HW >= 8192 ? 128 : 64; }

bool bad_ptr(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int dskd_nhwc_to_nchw_f32(const void* x, float* out, int B, int64_t HW, int C, int64_t x_batch_stride,
                                     int dtype, void* stream) {
  if (C != kC) return fail(DSKD_ERR_INVALID_ARG, "dskd_nhwc_to_nchw_f32: 256 channels only (got %d)", C);
  if (!x || !out || B < 0 || HW < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_nhwc_to_nchw_f32: null pointer or negative size");
  if (bad_ptr(x) || x_batch_stride % 8) return fail(DSKD_ERR_INVALID_ARG, "dskd_nhwc_to_nchw_f32: x must be 16-byte aligned rows");
  if (dtype != DSKD_DTYPE_F32 && dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_nhwc_to_nchw_f32: dtype");
  if (B == 0 || HW == 0) return DSKD_OK;
  const dim3 grid((unsigned)((HW + 31) / 32), (unsigned)B);
  if (dtype == DSKD_DTYPE_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, out,
                       (long long)HW, (long long)x_batch_stride);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, out,
                       (long long)HW, (long long)x_batch_stride);
  return check_launch("dskd_nhwc_to_nchw_f32");
}

extern "C" int64_t dskd_gn_workspace(int B, int64_t HW) {
  if (B < 0 || HW < 0) return -1;
  const int span = gn_span(HW);
  return (int64_t)B * ((HW + span - 1) / span) * kG * 2 * (int64_t)sizeof(double);
}

extern "C" int dskd_gn_fwd(const void* x, const float* gamma, const float* beta, void* y, double* sums, float* stats,
                           int B, int64_t HW, int C, int groups, int64_t x_batch_stride, int64_t y_batch_stride,
                           float eps, int relu, int dtype, void* stream) {
  if (C != kC || groups != kG)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gn_fwd: 256 channels in 32 groups only (got %d / %d)", C, groups);
  if (!x || !gamma || !beta || !y || !sums || !stats || B < 0 || HW < 0)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gn_fwd: null pointer or negative size");
  if (bad_ptr(x) || bad_ptr(y) || x_batch_stride % 8 || y_batch_stride % 8)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gn_fwd: x / y must be 16-byte aligned with batch strides that are multiples of 8");
  if (dtype != DSKD_DTYPE_F32 && dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_gn_fwd: dtype");
  if (B == 0 || HW == 0) return DSKD_OK;
  const int span = gn_span(HW);
  const dim3 grid((unsigned)((HW + span - 1) / span), (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DSKD_DTYPE_BF16) {
    hipLaunchKernelGGL(gn_stats_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)x, sums, (long long)HW, (long long)x_batch_stride, span);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)B), dim3(256), 0, st, sums, stats, (int)grid.x, (long long)HW, eps);
    hipLaunchKernelGGL(gn_apply_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)x, stats, gamma, beta, (__bf16*)y,
                       (long long)HW, (long long)x_batch_stride, (long long)y_batch_stride, span, relu);
  } else {
    hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)x, sums, (long long)HW, (long long)x_batch_stride, span);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)B), dim3(256), 0, st, sums, stats, (int)grid.x, (long long)HW, eps);
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)x, stats, gamma, beta, (float*)y,
                       (long long)HW, (long long)x_batch_stride, (long long)y_batch_stride, span, relu);
  }
  return check_launch("dskd_gn_fwd");
}

extern "C" int dskd_gn_bwd(const void* x, const void* grad_y, const float* stats, const float* gamma, const float* beta,
                           void* grad_x, double* sums, float* grad_gamma_beta, int copies, int B, int64_t HW, int C,
                           int groups, int64_t x_batch_stride, int64_t gy_batch_stride, int64_t gx_batch_stride, int relu,
                           int dtype, void* stream) {
  if (relu && !beta) return fail(DSKD_ERR_INVALID_ARG, "dskd_gn_bwd: relu needs beta (the mask is the sign of the forward's output)");
  if (C != kC || groups != kG)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gn_bwd: 256 channels in 32 groups only (got %d / %d)", C, groups);
  if (!x || !grad_y || !stats || !gamma || !grad_x || !sums || !grad_gamma_beta || B < 0 || HW < 0 || copies < 1)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gn_bwd: null pointer, negative size or copies < 1");
  if (bad_ptr(x) || bad_ptr(grad_y) || bad_ptr(grad_x) || x_batch_stride % 8 || gy_batch_stride % 8 || gx_batch_stride % 8)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_gn_bwd: tensors must be 16-byte aligned with batch strides that are multiples of 8");
  if (dtype != DSKD_DTYPE_F32 && dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_gn_bwd: dtype");
  if (B == 0 || HW == 0) return DSKD_OK;
  const int span = gn_span(HW);
  const dim3 grid((unsigned)((HW + span - 1) / span), (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DSKD_DTYPE_BF16) {
    hipLaunchKernelGGL(gn_bwd_stats_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)x, (const __bf16*)grad_y, stats, gamma, beta, sums,
                       grad_gamma_beta, copies, (long long)HW, (long long)x_batch_stride, (long long)gy_batch_stride, span, relu);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)x, (const __bf16*)grad_y, stats, gamma, beta, sums,
                       (__bf16*)grad_x, (long long)HW, (long long)x_batch_stride, (long long)gy_batch_stride,
                       (long long)gx_batch_stride, span, relu);
  } else {
    hipLaunchKernelGGL(gn_bwd_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)grad_y, stats, gamma, beta, sums,
                       grad_gamma_beta, copies, (long long)HW, (long long)x_batch_stride, (long long)gy_batch_stride, span, relu);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)grad_y, stats, gamma, beta, sums,
                       (float*)grad_x, (long long)HW, (long long)x_batch_stride, (long long)gy_batch_stride,
                       (long long)gx_batch_stride, span, relu);
  }
  return check_launch("dskd_gn_bwd");
}
