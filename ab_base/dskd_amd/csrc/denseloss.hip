// The dense detection losses of ALL decoder layers x images in two launches (forward) + one (backward).
//
// Replaces the per-layer `loss_single_split` arithmetic of the reference head
// (mmdet/models/dense_heads/gfl_deformable_detr_head_il.py:1453-1529) on precomputed dense targets, i.e.
//   QualityFocalLoss(beta = 2)   mmdet/models/losses/gfocal_loss.py:12-53     (target score = IoU(pred, target) of the
//                                positives, WITH its gradient into the boxes, as the reference's index_put keeps it)
//   L1Loss                       mmdet/models/losses/smooth_l1_loss.py
//   GIoULoss(eps = 1e-6)         mmdet/models/losses/iou_loss.py (bbox_overlaps mode 'giou', boxes scaled to pixels)
//   DistributionFocalLoss        mmdet/models/losses/gfocal_loss.py:103-125
// each reduced per layer as sum / (avg_factor + eps_f32) (`weight_reduce_loss`, losses/utils.py).
// In PyTorch this is 368 elementwise / reduction launches of 4-6 us on [6, 1200, ..] tensors per step (2 ms of GPU time
// even when replayed as a hipGraph).  Here: one wave per (layer, query) row computes the row's four loss terms AND the
// unit-upstream gradients w.r.t. its logits / box / distribution inputs (nothing couples rows except the per-layer sums
// and the scalar avg_factor); a one-workgroup kernel sums the rows of each (term, layer) in a fixed order (deterministic);
// the backward scales the stored gradients by the upstream gradient of their (term, layer).
// Tie / clamp conventions follow PyTorch autograd (maximum / minimum split the gradient at ties, clamp(min) passes it at
// the bound, abs' gradient is sign()).
#include "common.h"

namespace dskd {
namespace {

constexpr float kEpsF32 = 1.1920929e-07f;      // torch.finfo(torch.float32).eps

// value and gradient (w.r.t. the first box, xyxy) of bbox_overlaps(b, t, mode, is_aligned=True, eps)
template <bool GIOU>
__device__ __forceinline__ float overlap_grad(const float* b, const float* t, float eps, float* g) {
  const float aw = b[2] - b[0], ah = b[3] - b[1];
  const float a1 = aw * ah, a2 = (t[2] - t[0]) * (t[3] - t[1]);
  const float wxr = fminf(b[2], t[2]) - fmaxf(b[0], t[0]);
  const float wyr = fminf(b[3], t[3]) - fmaxf(b[1], t[1]);
  const float wx = fmaxf(wxr, 0.f), wy = fmaxf(wyr, 0.f);
  const float ov = wx * wy;
  const float un_raw = a1 + a2 - ov;
  const float un = fmaxf(un_raw, eps);
  const float iou = ov / un;
  // max(b, t): gradient to b where b > t (half at ties); min(b, t): where b < t
  auto gmax = [](float x, float y) { return x > y ? 1.f : (x == y ? 0.5f : 0.f); };
  auto gmin = [](float x, float y) { return x < y ? 1.f : (x == y ? 0.5f : 0.f); };
  const float cx = wxr >= 0.f ? 1.f : 0.f, cy = wyr >= 0.f ? 1.f : 0.f;
  // d ov / d b
  const float dov[4] = {-cx * gmax(b[0], t[0]) * wy, -cy * gmax(b[1], t[1]) * wx, cx * gmin(b[2], t[2]) * wy,
                        cy * gmin(b[3], t[3]) * wx};
  const float da1[4] = {-ah, -aw, ah, aw};
  const float pass_un = un_raw >= eps ? 1.f : 0.f;
  float dun[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    dun[k] = pass_un * (da1[k] - dov[k]);
    g[k] = (dov[k] * un - ov * dun[k]) / (un * un);
  }
  if constexpr (!GIOU) return iou;
  const float ewr = fmaxf(b[2], t[2]) - fminf(b[0], t[0]);
  const float ehr = fmaxf(b[3], t[3]) - fminf(b[1], t[1]);
  const float ew = fmaxf(ewr, 0.f), eh = fmaxf(ehr, 0.f);
  const float enc_raw = ew * eh;
  const float enc = fmaxf(enc_raw, eps);
  const float ex = ewr >= 0.f ? 1.f : 0.f, ey = ehr >= 0.f ? 1.f : 0.f;
  const float pass_enc = enc_raw >= eps ? 1.f : 0.f;
  const float denc[4] = {-pass_enc * ex * gmin(b[0], t[0]) * eh, -pass_enc * ey * gmin(b[1], t[1]) * ew,
                         pass_enc * ex * gmax(b[2], t[2]) * eh, pass_enc * ey * gmax(b[3], t[3]) * ew};
  // giou = iou - (enc - un) / enc
#pragma unroll
  for (int k = 0; k < 4; ++k) g[k] -= ((denc[k] - dun[k]) * enc - (enc - un) * denc[k]) / (enc * enc);
  return iou - (enc - un) / enc;
}

// gradient w.r.t. (cx, cy, w, h) from the gradient w.r.t. (x1, y1, x2, y2) of bbox_cxcywh_to_xyxy
__device__ __forceinline__ void xyxy_grad_to_cxcywh(const float* g, float* o) {
  o[0] = g[0] + g[2];
  o[1] = g[1] + g[3];
  o[2] = 0.5f * (g[2] - g[0]);
  o[3] = 0.5f * (g[3] - g[1]);
}

__device__ __forceinline__ float softplus_neg_abs(float x) { return log1pf(expf(-fabsf(x))); }

struct DenseArgs {
  const float* cls;        // [R, C] logits
  const float* box;        // [R, 4] cxcywh, normalised
  const float* lrtb;       // [R, 4 * R1] distribution logits
  const long long* labels; // [R] (background = C)
  const float* tgt;        // [R, 4] cxcywh targets (0 for negatives)
  const unsigned char* pos;  // [R] bool
  const float* factors;    // [N, 4] (w, h, w, h) of the query's image
  float* row_loss;         // [4][R]: qfl, l1, giou, dfl of the row (weighted by pos where the reference does)
  float* d_cls;            // [R, C]
  float* d_box;            // [3][R, 4]: through the QFL score, L1, GIoU
  float* d_lrtb;           // [R, 4 * R1]
  int R, N, C, R1;
};

__global__ __launch_bounds__(256) void dense_loss_rows_kernel(const DenseArgs a) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.R) return;
  const int n = r % a.N;
  const bool is_pos = a.pos[r] != 0;
  const float posf = is_pos ? 1.f : 0.f;
  float box[4], tg[4], fac[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { box[k] = a.box[(size_t)r * 4 + k]; tg[k] = a.tgt[(size_t)r * 4 + k]; fac[k] = a.factors[(size_t)n * 4 + k]; }
  const float b[4] = {box[0] - 0.5f * box[2], box[1] - 0.5f * box[3], box[0] + 0.5f * box[2], box[1] + 0.5f * box[3]};
  const float t[4] = {tg[0] - 0.5f * tg[2], tg[1] - 0.5f * tg[3], tg[0] + 0.5f * tg[2], tg[1] + 0.5f * tg[3]};

  // ---- IoU quality of the positives: the QFL target score, with its gradient into the box
  float giou_b[4], diou[4];
  const float iou = overlap_grad<false>(b, t, 1e-6f, giou_b);
  xyxy_grad_to_cxcywh(giou_b, diou);
  const float score = is_pos ? iou : 0.f;

  // ---- quality focal loss over the classes (beta = 2); lane = class (two passes cover C <= 128)
  const long long label = a.labels[r];
  const bool lab_ok = label >= 0 && label < a.C;
  float qsum = 0.f, dscore = 0.f;
  for (int c = lane; c < a.C; c += 64) {
    const float x = a.cls[(size_t)r * a.C + c];
    const float sg = 1.f / (1.f + expf(-x));
    const float sp = softplus_neg_abs(x);
    float term, dx;
    if (lab_ok && c == (int)label) {
      const float bce = fmaxf(x, 0.f) - x * score + sp;
      const float d = score - sg;
      term = bce * d * d;
      dx = (sg - score) * d * d - 2.f * bce * d * sg * (1.f - sg);
      dscore = -x * d * d + 2.f * bce * d;
    } else {
      const float bce = fmaxf(x, 0.f) + sp;                 // target 0
      term = bce * sg * sg;
      dx = sg * sg * sg + 2.f * bce * sg * sg * (1.f - sg);
    }
    qsum += term;
    a.d_cls[(size_t)r * a.C + c] = dx;
  }
  qsum = wave_sum(qsum);
  dscore = wave_sum(dscore);                                // one lane holds it

  // ---- distribution focal loss: lanes 0..3 = the four sides (targets w/2, w/2, h/2, h/2 as in the reference)
  float dfl = 0.f;
  if (lane < 4) {
    const float lab = (lane < 2 ? tg[2] : tg[3]) * 0.5f;
    int left = (int)lab;
    left = left < 0 ? 0 : (left > a.R1 - 2 ? a.R1 - 2 : left);
    const float wl = (float)(left + 1) - lab, wr = lab - (float)left;
    const float* p = a.lrtb + (size_t)r * 4 * a.R1 + lane * a.R1;
    float m = -3.0e38f;
    for (int j = 0; j < a.R1; ++j) m = fmaxf(m, p[j]);
    float s = 0.f;
    for (int j = 0; j < a.R1; ++j) s += expf(p[j] - m);
    const float lse = m + logf(s);
    dfl = ((lse - p[left]) * wl + (lse - p[left + 1]) * wr) * posf;
    float* d = a.d_lrtb + (size_t)r * 4 * a.R1 + lane * a.R1;
    for (int j = 0; j < a.R1; ++j) {
      float gj = (wl + wr) * expf(p[j] - lse);
      if (j == left) gj -= wl;
      if (j == left + 1) gj -= wr;
      d[j] = gj * posf;
    }
  }
  dfl = wave_sum(dfl);

  if (lane == 0) {
    // ---- L1 on the normalised cxcywh
    float l1 = 0.f;
    float* d1 = a.d_box + ((size_t)a.R + r) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = box[k] - tg[k];
      l1 += fabsf(d);
      d1[k] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * posf;
    }
    // ---- GIoU on the pixel boxes
    float bp[4], tp[4], gg[4], gc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { bp[k] = b[k] * fac[k]; tp[k] = t[k] * fac[k]; }
    const float giou = overlap_grad<true>(bp, tp, 1e-6f, gg);
#pragma unroll
    for (int k = 0; k < 4; ++k) gg[k] *= fac[k];
    xyxy_grad_to_cxcywh(gg, gc);
    float* d2 = a.d_box + ((size_t)2 * a.R + r) * 4;
    float* d0 = a.d_box + (size_t)r * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      d2[k] = -gc[k] * posf;
      d0[k] = is_pos ? dscore * diou[k] : 0.f;
    }
    a.row_loss[r] = qsum;
    a.row_loss[(size_t)a.R + r] = l1 * posf;
    a.row_loss[(size_t)2 * a.R + r] = (1.f - giou) * posf;
    a.row_loss[(size_t)3 * a.R + r] = dfl;
  }
}

// losses[k][l] = w_k * sum_n row_loss[k][l * N + n] / (avg_k + eps): 32 lanes per (term, layer), fixed summation order
__global__ __launch_bounds__(1024) void dense_loss_reduce_kernel(const float* __restrict__ row_loss, const float* __restrict__ avg_pos,
                                                                 float* __restrict__ losses, int nl, int N, float w0, float w1,
                                                                 float w2, float w3) {
  const int pair = threadIdx.x >> 5, sub = threadIdx.x & 31;
  for (int p = pair; p < 4 * nl; p += blockDim.x >> 5) {
    const int k = p / nl, l = p - k * nl;
    const float* src = row_loss + ((size_t)k * nl + l) * N;
    float s = 0.f;
    for (int i = sub; i < N; i += 32) s += src[i];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
    if (sub == 0) {
      const float avg = avg_pos[0] * (k == 3 ? 4.f : 1.f);
      const float w = k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : w3));
      losses[p] = w * s / (avg + kEpsF32);
    }
  }
}

struct DenseBwdArgs {
  const float* g;          // [4][nl] upstream gradients of (cls, bbox, iou, dfl) per layer
  const float* avg_pos;
  const float* d_cls; const float* d_box; const float* d_lrtb;
  float* g_cls; float* g_box; float* g_lrtb;
  int nl, N, C, R1;
  float w[4];
};

__global__ __launch_bounds__(256) void dense_loss_bwd_kernel(const DenseBwdArgs a) {
  const int R = a.nl * a.N;
  const int per_row = a.C + 4 * a.R1 + 4;
  const long long total = (long long)R * per_row;
  const float avg = a.avg_pos[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / per_row), e = (int)(i - (long long)r * per_row);
    const int l = r / a.N;
    const float s_cls = a.g[l] * a.w[0] / (avg + kEpsF32);
    if (e < a.C) {
      a.g_cls[(size_t)r * a.C + e] = a.d_cls[(size_t)r * a.C + e] * s_cls;
    } else if (e < a.C + 4 * a.R1) {
      const int j = e - a.C;
      const float s_dfl = a.g[3 * a.nl + l] * a.w[3] / (avg * 4.f + kEpsF32);
      a.g_lrtb[(size_t)r * 4 * a.R1 + j] = a.d_lrtb[(size_t)r * 4 * a.R1 + j] * s_dfl;
    } else {
      const int k = e - a.C - 4 * a.R1;
      const float s_l1 = a.g[a.nl + l] * a.w[1] / (avg + kEpsF32);
      const float s_iou = a.g[2 * a.nl + l] * a.w[2] / (avg + kEpsF32);
      a.g_box[(size_t)r * 4 + k] = a.d_box[(size_t)r * 4 + k] * s_cls + a.d_box[((size_t)R + r) * 4 + k] * s_l1 +
                                   a.d_box[((size_t)2 * R + r) * 4 + k] * s_iou;
    }
  }
}

}  // namespace
}  // namespace dskd

extern "C" int dskd_dense_loss_fwd(const float* cls, const float* box, const float* lrtb, const int64_t* labels,
                                   const float* tgt, const unsigned char* pos, const float* factors, const float* avg_pos,
                                   float* losses, float* row_loss, float* d_cls, float* d_box, float* d_lrtb, int nl, int N,
                                   int C, int R1, float w_cls, float w_bbox, float w_iou, float w_dfl, void* stream) {
  using namespace dskd;
  if (nl < 0 || N < 0 || C < 1 || C > 128 || R1 < 2 || R1 > 64)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_dense_loss_fwd: need 1 <= C <= 128 and 2 <= reg_max + 1 <= 64 (got %d, %d)", C, R1);
  if (nl == 0) return DSKD_OK;
  if (!cls || !box || !lrtb || !labels || !tgt || !pos || !factors || !avg_pos || !losses || !row_loss || !d_cls || !d_box || !d_lrtb)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_dense_loss_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int R = nl * N;
  if (R > 0) {
    DenseArgs a{cls, box, lrtb, (const long long*)labels, tgt, pos, factors, row_loss, d_cls, d_box, d_lrtb, R, N, C, R1};
    hipLaunchKernelGGL(dense_loss_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(dense_loss_reduce_kernel, dim3(1), dim3(1024), 0, st, row_loss, avg_pos, losses, nl, N, w_cls, w_bbox, w_iou,
                     w_dfl);
  return check_launch("dskd_dense_loss_fwd");
}

extern "C" int dskd_dense_loss_bwd(const float* grad_losses, const float* avg_pos, const float* d_cls, const float* d_box,
                                   const float* d_lrtb, float* grad_cls, float* grad_box, float* grad_lrtb, int nl, int N, int C,
                                   int R1, float w_cls, float w_bbox, float w_iou, float w_dfl, void* stream) {
  using namespace dskd;
  if (nl <= 0 || N <= 0) return DSKD_OK;
  if (!grad_losses || !avg_pos || !d_cls || !d_box || !d_lrtb || !grad_cls || !grad_box || !grad_lrtb)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_dense_loss_bwd: null pointer");
  DenseBwdArgs a{grad_losses, avg_pos, d_cls, d_box, d_lrtb, grad_cls, grad_box, grad_lrtb, nl, N, C, R1, {w_cls, w_bbox, w_iou, w_dfl}};
  const long long total = (long long)nl * N * (C + 4 * R1 + 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(dense_loss_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("dskd_dense_loss_bwd");
}
