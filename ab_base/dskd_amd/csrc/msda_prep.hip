// Prologue / epilogue of the MultiScaleDeformableAttention MODULE around the sampling kernel
// (ext-mmcv MultiScaleDeformableAttention.forward, restated in dskd_amd/transformer.py):
//   attention_weights = softmax_{levels*points}(logits)
//   sampling_locations = reference_points[:, :, None, :, None, :] + offsets / (W_l, H_l)
// and the matching backward.  In PyTorch this is a chain of ~6 elementwise launches forward
// (two dtype casts, softmax, div, add, views) and as many backward over [B, 22k, 8, 16(,2)]
// tensors; here it is ONE pass each way: read the projection output once (bf16 or f32, offsets
// and logits side by side as produced by the fused GEMM), write loc/attn in f32 for the
// sampling kernel.  HBM-bound elementwise work: 64 lanes = 64 consecutive points = four
// (query, head) groups of 16, softmax reductions with DPP inside the 16-lane rows.
#include "common.h"

namespace dskd {
namespace {

constexpr int kMaxLevels = 4;

struct PrepGeom {
  float W[kMaxLevels], H[kMaxLevels];
};

__device__ __forceinline__ float sel4(const float* a, int i) {
  // arithmetic select (no branches): levels <= 4
  const float m1 = i >= 1 ? 1.f : 0.f, m2 = i >= 2 ? 1.f : 0.f, m3 = i >= 3 ? 1.f : 0.f;
  return a[0] + m1 * (a[1] - a[0]) + m2 * (a[2] - a[1]) + m3 * (a[3] - a[2]);
}

// both: [NQ, heads*16*3]  (offsets [heads,16,2] then logits [heads,16]);  ref: [NQ, levels, 2]
template <typename T>
__global__ __launch_bounds__(256) void prep_fwd_kernel(const T* __restrict__ both,
                                                       const float* __restrict__ ref,
                                                       float* __restrict__ loc,
                                                       float* __restrict__ attn, PrepGeom g,
                                                       long long npoints, int heads, int levels,
                                                       int points) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // global point index
  const bool ok = p < npoints;
  const int per_q = heads * 16;
  const long long q = (ok ? p : npoints - 1) / per_q;
  const int r = (int)((ok ? p : npoints - 1) - q * per_q);
  const int lvl = (r & 15) / points;
  const T* row = both + q * (per_q * 3);
  const float ox = (float)row[r * 2], oy = (float)row[r * 2 + 1];
  const float lg = (float)row[per_q * 2 + r];
  const float mx = row16_max(lg);
  const float e = __expf(lg - mx);
  const float a = e / row16_sum(e);
  const f32x2 rf = *reinterpret_cast<const f32x2*>(ref + (q * levels + lvl) * 2);
  if (ok) {
    *reinterpret_cast<f32x2*>(loc + p * 2) = f32x2{rf.x + ox / sel4(g.W, lvl), rf.y + oy / sel4(g.H, lvl)};
    attn[p] = a;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void prep_bwd_kernel(const float* __restrict__ grad_loc,
                                                       const float* __restrict__ grad_attn,
                                                       const float* __restrict__ attn,
                                                       T* __restrict__ grad_both, PrepGeom g,
                                                       long long npoints, int heads, int points) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = p < npoints;
  const long long pc = ok ? p : npoints - 1;
  const int per_q = heads * 16;
  const long long q = pc / per_q;
  const int r = (int)(pc - q * per_q);
  const int lvl = (r & 15) / points;
  const float a = attn[pc], ga = grad_attn[pc];
  const float dot = row16_sum(a * ga);          // softmax backward: a * (ga - sum_t a_t ga_t)
  const f32x2 gl = *reinterpret_cast<const f32x2*>(grad_loc + pc * 2);
  if (ok) {
    T* row = grad_both + q * (per_q * 3);
    row[r * 2] = (T)(gl.x / sel4(g.W, lvl));
    row[r * 2 + 1] = (T)(gl.y / sel4(g.H, lvl));
    row[per_q * 2 + r] = (T)(a * (ga - dot));
  }
}

int fill(const int64_t* spatial_shapes, int levels, int points, PrepGeom* g) {
  if (levels < 1 || levels > kMaxLevels || levels * points != 16)
    return fail(DSKD_ERR_INVALID_ARG, "msda_prep: needs levels*points == 16 and levels <= 4 (got %d x %d)", levels, points);
  for (int l = 0; l < kMaxLevels; ++l) {
    g->H[l] = l < levels ? (float)spatial_shapes[2 * l] : 1.f;
    g->W[l] = l < levels ? (float)spatial_shapes[2 * l + 1] : 1.f;
  }
  return DSKD_OK;
}

// d(reference points)[q, l, :] = sum over heads and points of d(loc)[q, h, l, p, :]: loc = ref[q, l] + offset / (W, H).
// One thread per (query, level, x|y), heads * points strided loads.  (ATen's reduction over the [nq, H, L, P, 2] view takes
// 14 us for the decoder's 1 200 queries; this sits on the decoder's launch chain six times per backward.)
__global__ __launch_bounds__(256) void grad_ref_kernel(const float* __restrict__ gl, float* __restrict__ out, long long n,
                                                       int heads, int levels, int points) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (q, l, xy)
  if (i >= n) return;
  const int xy = (int)(i & 1);
  const int l = (int)((i >> 1) % levels);
  const long long q = (i >> 1) / levels;
  const float* base = gl + (q * heads * levels + l) * (long long)points * 2 + xy;
  float s = 0.f;
  for (int h = 0; h < heads; ++h)
    for (int p = 0; p < points; ++p) s += base[((long long)h * levels * points + p) * 2];
  out[i] = s;
}

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int dskd_msda_grad_ref(const float* grad_loc, float* grad_ref, int64_t n_query, int heads, int levels, int points,
                                  void* stream) {
  if (n_query < 0 || heads < 1 || levels < 1 || points < 1) return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_grad_ref: bad sizes");
  if (n_query == 0) return DSKD_OK;
  if (!grad_loc || !grad_ref) return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_grad_ref: null pointer");
  const long long n = (long long)n_query * levels * 2;
  hipLaunchKernelGGL(grad_ref_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_loc, grad_ref,
                     n, heads, levels, points);
  return check_launch("dskd_msda_grad_ref");
}

extern "C" int dskd_msda_prep_fwd(const void* both, const float* ref, const int64_t* spatial_shapes,
                                  float* loc, float* attn, int64_t n_query, int heads, int levels,
                                  int points, int dtype, void* stream) {
  PrepGeom g;
  if (int rc = fill(spatial_shapes, levels, points, &g)) return rc;
  if (n_query <= 0) return DSKD_OK;
  if (!both || !ref || !loc || !attn) return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_prep_fwd: null pointer");
  const long long np = (long long)n_query * heads * 16;
  const dim3 grid((unsigned)((np + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DSKD_DTYPE_F32)
    hipLaunchKernelGGL(prep_fwd_kernel<float>, grid, block, 0, st, (const float*)both, ref, loc, attn, g, np, heads, levels, points);
  else if (dtype == DSKD_DTYPE_BF16)
    hipLaunchKernelGGL(prep_fwd_kernel<__bf16>, grid, block, 0, st, (const __bf16*)both, ref, loc, attn, g, np, heads, levels, points);
  else
    return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_prep_fwd: unknown dtype %d", dtype);
  return check_launch("dskd_msda_prep_fwd");
}

extern "C" int dskd_msda_prep_bwd(const float* grad_loc, const float* grad_attn, const float* attn,
                                  const int64_t* spatial_shapes, void* grad_both, int64_t n_query,
                                  int heads, int levels, int points, int dtype, void* stream) {
  PrepGeom g;
  if (int rc = fill(spatial_shapes, levels, points, &g)) return rc;
  if (n_query <= 0) return DSKD_OK;
  if (!grad_loc || !grad_attn || !attn || !grad_both) return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_prep_bwd: null pointer");
  const long long np = (long long)n_query * heads * 16;
  const dim3 grid((unsigned)((np + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DSKD_DTYPE_F32)
    hipLaunchKernelGGL(prep_bwd_kernel<float>, grid, block, 0, st, grad_loc, grad_attn, attn, (float*)grad_both, g, np, heads, points);
  else if (dtype == DSKD_DTYPE_BF16)
    hipLaunchKernelGGL(prep_bwd_kernel<__bf16>, grid, block, 0, st, grad_loc, grad_attn, attn, (__bf16*)grad_both, g, np, heads, points);
  else
    return fail(DSKD_ERR_INVALID_ARG, "dskd_msda_prep_bwd: unknown dtype %d", dtype);
  return check_launch("dskd_msda_prep_bwd");
}
