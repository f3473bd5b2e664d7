// Global-norm gradient clipping + AdamW over ALL parameter tensors of the detector in two launches.
//
// Replaces the optimizer hook of the reference's runner (config `optimizer_config = dict(grad_clip=dict(max_norm=0.1,
// norm_type=2))` + AdamW, configs/deformable_detr/*_il.py:213-224; ext-mmcv OptimizerHook: clip_grad_norm_ then
// optimizer.step()), which PyTorch runs as ~40 multi-tensor launches (per-tensor norms, norm of norms, scale, fused AdamW
// per parameter group: 1.0 ms per step at ~40 M parameters, 13 % of HBM peak).  Here:
//   launch 1: one workgroup per 32 K-element chunk of a gradient -> sum of squares -> partials[chunk]
//   launch 2: every workgroup adds the partials (a few KB, L2-resident) in a FIXED order -> total norm -> clip coefficient
//             min(1, max_norm / (norm + 1e-6)) -> AdamW update of its chunk with the coefficient folded into the gradient
// The gradients are read twice (second time mostly from the Infinity Cache) and never rewritten: the clipped gradient
// exists only in registers.  Tensors are addressed through device tables of pointers (parameters / moments: fixed;
// gradients: refreshed by the host when an address changes), so nothing has to live in one flat buffer.
// Arithmetic = torch.optim.AdamW (decoupled decay, lerp form of the first moment, bias corrections from the host).
#include "common.h"

namespace dskd {
namespace {

constexpr int kChunk = 32768;
constexpr int kMaxGroups = 8;

struct OptArgs {
  const long long* ptrs;     // [4][n_tensors]: param, grad, exp_avg, exp_avg_sq (device addresses)
  const int* meta;           // [n_tensors][2]: numel, group
  const int* chunks;         // [n_chunks][2]: tensor, first element
  float* partials;           // [n_chunks]
  float* norm_out;           // [2]: total norm, clip coefficient
  int n_tensors, n_chunks;
  float lr[kMaxGroups], wd[kMaxGroups];
  float beta1, beta2, eps, bc1, bc2_sqrt;     // bias corrections 1 - beta^t (bc2 as its square root)
  float max_norm;            // <= 0: no clipping
};

__global__ __launch_bounds__(256) void grad_sq_kernel(const OptArgs a) {
  const int c = blockIdx.x;
  const int t = a.chunks[2 * c], start = a.chunks[2 * c + 1];
  const int n = min(a.meta[2 * t] - start, kChunk);
  const float* g = reinterpret_cast<const float*>(a.ptrs[a.n_tensors + t]) + start;
  float s = 0.f;
  const int n4 = n >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) s = fmaf(g[i], g[i], s);
  s = wave_sum(s);
  __shared__ float sw[4];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) a.partials[c] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

__global__ __launch_bounds__(256) void clip_adamw_kernel(const OptArgs a) {
  __shared__ float s_coef;
  {
    // the same fixed-order sum in every workgroup: all of them get the same coefficient, bit for bit
    float s = 0.f;
    for (int i = threadIdx.x; i < a.n_chunks; i += 256) s += a.partials[i];
    s = wave_sum(s);
    __shared__ float sw[4];
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float norm = sqrtf((sw[0] + sw[1]) + (sw[2] + sw[3]));
      float coef = 1.f;
      if (a.max_norm > 0.f) coef = fminf(a.max_norm / (norm + 1e-6f), 1.f);     // torch.nn.utils.clip_grad_norm_
      if (!(norm == norm)) coef = norm;                                         // NaN gradients stay visible
      s_coef = coef;
      if (blockIdx.x == 0) { a.norm_out[0] = norm; a.norm_out[1] = coef; }
    }
    __syncthreads();
  }
  const float coef = s_coef;
  const int c = blockIdx.x;
  const int t = a.chunks[2 * c], start = a.chunks[2 * c + 1];
  const int n = min(a.meta[2 * t] - start, kChunk);
  const int grp = a.meta[2 * t + 1];
  const float lr = a.lr[grp], decay = 1.f - lr * a.wd[grp];
  const float step_size = lr / a.bc1, one_b1 = 1.f - a.beta1, one_b2 = 1.f - a.beta2;
  float* p = reinterpret_cast<float*>(a.ptrs[t]) + start;
  const float* g = reinterpret_cast<const float*>(a.ptrs[a.n_tensors + t]) + start;
  float* m = reinterpret_cast<float*>(a.ptrs[2 * a.n_tensors + t]) + start;
  float* v = reinterpret_cast<float*>(a.ptrs[3 * a.n_tensors + t]) + start;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= coef;
    pp *= decay;
    mm = mm + (gg - mm) * one_b1;
    vv = a.beta2 * vv + one_b2 * gg * gg;
    const float denom = sqrtf(vv) / a.bc2_sqrt + a.eps;
    pp -= step_size * (mm / denom);
  };
  const int n4 = n >> 2;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) upd(pa[k], ga[k], ma[k], va[k]);
    reinterpret_cast<f32x4*>(p)[i] = f32x4{pa[0], pa[1], pa[2], pa[3]};
    reinterpret_cast<f32x4*>(m)[i] = f32x4{ma[0], ma[1], ma[2], ma[3]};
    reinterpret_cast<f32x4*>(v)[i] = f32x4{va[0], va[1], va[2], va[3]};
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
}

}  // namespace
}  // namespace dskd

using namespace dskd;

// ------------------------------------------------------------------------------------------------------------------
// Multi-tensor cast with an optional per-row scale (r4): the step's low-precision parameter copies and their gradients.
//   direction 0:  dst (bf16) = src (f32) * scale[row]     the bf16 copies of the Linear / attention parameters
//                                                         (transformer._CastParams) and the BN-folded convolution weights
//                                                         w * gamma / sqrt(var + eps) of the trainable ResNet stages
//                                                         (backbones._FoldTrainable; row = output channel)
//   direction 1:  dst (f32) = src (bf16) * scale[row]     their gradients on the way back to the f32 masters
// ONE launch over a device table of (src, dst, scale, numel, inner) rows instead of ATen's multi-tensor passes (fold: mul
// with the scale EXPANDED to the weight's shape, then a copy: 414 MB of traffic each way for 92 MB of weights, at ~2 TB/s
// in 18 multi_tensor_apply launches: 0.43 ms per step).  row = element / inner in MEMORY order, which is the output
// channel for a dense [N, C, kh, kw] weight in either memory format.
constexpr int kCastChunk = 8192;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void cast_scale_many_kernel(const long long* __restrict__ table, const int* __restrict__ first,
                                                              int n, int dir) {
  const int b = blockIdx.x;
  int lo = 0, hi = n - 1;                      // first[t] <= b < first[t + 1]
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const long long* row = table + 5ll * lo;
  const long long numel = row[3], inner = row[4];
  const float* scale = reinterpret_cast<const float*>(row[2]);
  const long long base = (long long)(b - first[lo]) * kCastChunk;
  const long long end = base + kCastChunk < numel ? base + kCastChunk : numel;
  // 8 consecutive elements share a row, and both pointers take 16-byte accesses (a slot of a flat gradient buffer may not)
  const bool vec = ((inner & 7) == 0 || !scale) && (((unsigned long long)row[0] | (unsigned long long)row[1]) & 15ull) == 0ull;
  if (dir == 0) {
    const float* src = reinterpret_cast<const float*>(row[0]);
    __bf16* dst = reinterpret_cast<__bf16*>(row[1]);
    long long i = base + threadIdx.x * 8;
    for (; vec && i + 8 <= end; i += 256 * 8) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + i), a1 = *reinterpret_cast<const f32x4*>(src + i + 4);
      const float sc = scale ? scale[i / inner] : 1.f;
      bf16x8_t o;
      o[0] = (__bf16)(a0.x * sc); o[1] = (__bf16)(a0.y * sc); o[2] = (__bf16)(a0.z * sc); o[3] = (__bf16)(a0.w * sc);
      o[4] = (__bf16)(a1.x * sc); o[5] = (__bf16)(a1.y * sc); o[6] = (__bf16)(a1.z * sc); o[7] = (__bf16)(a1.w * sc);
      *reinterpret_cast<bf16x8_t*>(dst + i) = o;
    }
    // tail of the tensor (numel % 8) or the scalar path: one element per thread and pass
    for (long long j = (vec ? end - ((end - base) & 7) : base) + threadIdx.x; j < end; j += 256)
      dst[j] = (__bf16)(src[j] * (scale ? scale[j / inner] : 1.f));
  } else {
    const __bf16* src = reinterpret_cast<const __bf16*>(row[0]);
    float* dst = reinterpret_cast<float*>(row[1]);
    long long i = base + threadIdx.x * 8;
    for (; vec && i + 8 <= end; i += 256 * 8) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(src + i);
      const float sc = scale ? scale[i / inner] : 1.f;
      *reinterpret_cast<f32x4*>(dst + i) = f32x4{(float)a[0] * sc, (float)a[1] * sc, (float)a[2] * sc, (float)a[3] * sc};
      *reinterpret_cast<f32x4*>(dst + i + 4) = f32x4{(float)a[4] * sc, (float)a[5] * sc, (float)a[6] * sc, (float)a[7] * sc};
    }
    for (long long j = (vec ? end - ((end - base) & 7) : base) + threadIdx.x; j < end; j += 256)
      dst[j] = (float)src[j] * (scale ? scale[j / inner] : 1.f);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The weights the INPUT-GRADIENT launches of a trainable ResNet stage read, for all of its convolutions in one launch:
// dst[k][taps - 1 - t][n] = src[n][t][k]  (bf16; src = a [N, K, 1, 1] or channels_last [N, K, 3, 3] weight as it lies in memory,
// taps = 1 | 9) -- the plain transpose of a 1x1 weight (dX = dY W) and the tap-flipped, channel-swapped 3x3 weight
// (dX = conv3x3(dY, W')).  Before r4 every Bottleneck's backward made its three or four copies itself (flip + strided copy:
// ~75 launches of 4-5 us per step on the backward's launch chain).  Table rows {src, dst, N, K, taps}; block b of tensor i
// (first[i] <= b < first[i + 1]) turns one 64 x 64 tile of one tap through LDS (128-byte rows in, 128-byte rows out).
__global__ __launch_bounds__(256) void weight_t_many_kernel(const long long* __restrict__ table, const int* __restrict__ first,
                                                            int n) {
  __shared__ __bf16 tile[64][64 + 2];
  const int b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const long long* row = table + 5ll * lo;
  const __bf16* src = reinterpret_cast<const __bf16*>(row[0]);
  __bf16* dst = reinterpret_cast<__bf16*>(row[1]);
  const int N = (int)row[2], K = (int)row[3], taps = (int)row[4];
  int lb = b - first[lo];
  const int tk = lb % (K >> 6); lb /= (K >> 6);
  const int tn = lb % (N >> 6);
  const int t = lb / (N >> 6);
  const int r = threadIdx.x >> 2, part = threadIdx.x & 3;
  {
    const __bf16* sp = src + ((long long)(tn * 64 + r) * taps + t) * K + tk * 64 + part * 16;
    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(sp), c = *reinterpret_cast<const bf16x8_t*>(sp + 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { tile[r][part * 16 + j] = a[j]; tile[r][part * 16 + 8 + j] = c[j]; }
  }
  __syncthreads();
  {
    bf16x8_t a, c;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = tile[part * 16 + j][r]; c[j] = tile[part * 16 + 8 + j][r]; }
    __bf16* dp = dst + ((long long)(tk * 64 + r) * taps + (taps - 1 - t)) * N + tn * 64 + part * 16;
    *reinterpret_cast<bf16x8_t*>(dp) = a;
    *reinterpret_cast<bf16x8_t*>(dp + 8) = c;
  }
}

extern "C" int dskd_clip_adamw_chunk(void) { return kChunk; }

extern "C" int dskd_clip_adamw(const int64_t* ptrs, const int32_t* meta, const int32_t* chunks, float* partials,
                               float* norm_out, int n_tensors, int n_chunks, const float* lr, const float* weight_decay,
                               int n_groups, float beta1, float beta2, float eps, int64_t step, float max_norm,
                               void* stream) {
  if (!ptrs || !meta || !chunks || !partials || !norm_out || !lr || !weight_decay)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_clip_adamw: null pointer");
  if (n_tensors < 0 || n_chunks < 0 || n_groups < 1 || n_groups > kMaxGroups || step < 1)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_clip_adamw: bad sizes (tensors=%d chunks=%d groups=%d step=%lld; at most %d groups)",
                n_tensors, n_chunks, n_groups, (long long)step, kMaxGroups);
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_clip_adamw: bad hyper-parameters (beta1=%g beta2=%g eps=%g)", beta1, beta2, eps);
  if (n_chunks == 0) return DSKD_OK;
  OptArgs a;
  a.ptrs = reinterpret_cast<const long long*>(ptrs); a.meta = meta; a.chunks = chunks; a.partials = partials;
  a.norm_out = norm_out; a.n_tensors = n_tensors; a.n_chunks = n_chunks;
  for (int i = 0; i < kMaxGroups; ++i) { a.lr[i] = i < n_groups ? lr[i] : 0.f; a.wd[i] = i < n_groups ? weight_decay[i] : 0.f; }
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.max_norm = max_norm;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(grad_sq_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(clip_adamw_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, a);
  return check_launch("dskd_clip_adamw");
}

extern "C" int dskd_cast_scale_chunk(void) { return kCastChunk; }

extern "C" int dskd_weight_t_many(const int64_t* table, const int32_t* first, int n, int total_blocks, int dtype, void* stream) {
  if (dtype != DSKD_DTYPE_BF16) return fail(DSKD_ERR_INVALID_ARG, "dskd_weight_t_many: bf16 only");
  if (n < 0 || total_blocks < 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_weight_t_many: bad argument (n=%d blocks=%d)", n, total_blocks);
  if (n == 0 || total_blocks == 0) return DSKD_OK;
  if (!table || !first || (reinterpret_cast<uintptr_t>(table) & 7) || (reinterpret_cast<uintptr_t>(first) & 3))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_weight_t_many: null / misaligned table");
  hipLaunchKernelGGL(weight_t_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(table), first, n);
  return check_launch("dskd_weight_t_many");
}

extern "C" int dskd_cast_scale_many(const int64_t* table, const int32_t* first, int n, int total_chunks, int direction,
                                    void* stream) {
  if (n < 0 || total_chunks < 0 || (direction != 0 && direction != 1))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_cast_scale_many: bad argument (n=%d chunks=%d direction=%d)", n, total_chunks, direction);
  if (n == 0 || total_chunks == 0) return DSKD_OK;
  if (!table || !first || (reinterpret_cast<uintptr_t>(table) & 7) || (reinterpret_cast<uintptr_t>(first) & 3))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_cast_scale_many: null / misaligned table");
  hipLaunchKernelGGL(cast_scale_many_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(table), first, n, direction);
  return check_launch("dskd_cast_scale_many");
}
