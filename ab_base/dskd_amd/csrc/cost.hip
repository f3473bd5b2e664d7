// Fused matching cost of GFLHungarianAssigner for all (decoder layer, image) problems of a
// step in one launch.  Replaces the ~40 small launches per problem of the reference's cost
// build (mmdet/core/bbox/assigners/gfl_hungarian_assigner.py:120-140):
//   cost = QualityFocalLossCost (match_cost.py:193-230, weight w_cls)
//        + BBoxL1Cost 'xywh'    (match_cost.py:34-51,  weight w_reg)
//        + IoUCost 'giou'       (match_cost.py:460-476, weight w_iou)
// with bbox_overlaps from mmdet/core/bbox/iou_calculators/iou2d_calculator.py:190-261
// (eps = 1e-6 clamps on union and enclosing area) and the box conversions of
// mmdet/core/bbox/transforms.py:245-270.  Pure elementwise work over [Q, G]: one thread per
// entry, HBM-trivial; the win is the launch count and staying on the device.
#include "common.h"
#include <algorithm>

namespace dskd {
namespace {

constexpr int kPack = 64;

struct CostProb {
  long long gt_start;
  int G;
  float img_w, img_h;
};
struct CostPack {
  CostProb p[kPack];
};

__device__ __forceinline__ float iou_like(float ax1, float ay1, float ax2, float ay2,
                                          float bx1, float by1, float bx2, float by2,
                                          bool giou) {
  const float eps = 1e-6f;
  const float area1 = (ax2 - ax1) * (ay2 - ay1);
  const float area2 = (bx2 - bx1) * (by2 - by1);
  const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f);
  const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float overlap = w * h;
  const float uni = fmaxf(area1 + area2 - overlap, eps);
  const float iou = overlap / uni;
  if (!giou) return iou;
  const float ew = fmaxf(fmaxf(ax2, bx2) - fminf(ax1, bx1), 0.f);
  const float eh = fmaxf(fmaxf(ay2, by2) - fminf(ay1, by1), 0.f);
  const float earea = fmaxf(ew * eh, eps);
  return iou - (earea - uni) / earea;
}

__global__ __launch_bounds__(256) void match_cost_kernel(
    const float* __restrict__ bbox_pred, const float* __restrict__ cls_pred,
    const float* __restrict__ gt_bboxes, const int64_t* __restrict__ gt_labels,
    float* __restrict__ cost, CostPack pack, int p0, int Q, int C, float w_cls, float w_reg,
    float w_iou) {
#pragma clang fp contract(off)
  const CostProb pr = pack.p[blockIdx.y];
  const int p = p0 + blockIdx.y;
  const int G = pr.G;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Q * G) return;
  const int q = e / G, gi = e - q * G;

  const f32x4 pb = *reinterpret_cast<const f32x4*>(bbox_pred + ((size_t)p * Q + q) * 4);
  const f32x4 gb = *reinterpret_cast<const f32x4*>(gt_bboxes + (size_t)(pr.gt_start + gi) * 4);
  long long label = gt_labels[pr.gt_start + gi];
  label = label < 0 ? 0 : (label >= C ? C - 1 : label);
  const float x = cls_pred[((size_t)p * Q + q) * C + label];

  // normalised boxes
  const float px1 = pb.x - 0.5f * pb.z, py1 = pb.y - 0.5f * pb.w;
  const float px2 = pb.x + 0.5f * pb.z, py2 = pb.y + 0.5f * pb.w;
  const float gx1 = gb.x / pr.img_w, gy1 = gb.y / pr.img_h;
  const float gx2 = gb.z / pr.img_w, gy2 = gb.w / pr.img_h;

  // L1 on (cx, cy, w, h)
  const float gcx = (gx1 + gx2) / 2.f, gcy = (gy1 + gy2) / 2.f;
  const float gw = gx2 - gx1, gh = gy2 - gy1;
  const float reg = (fabsf(pb.x - gcx) + fabsf(pb.y - gcy) + fabsf(pb.z - gw) + fabsf(pb.w - gh)) * w_reg;

  // -GIoU on pixel boxes
  const float giou = iou_like(px1 * pr.img_w, py1 * pr.img_h, px2 * pr.img_w, py2 * pr.img_h,
                              gb.x, gb.y, gb.z, gb.w, true);
  const float iouc = -giou * w_iou;

  // quality focal cost: BCE-with-logits(x, IoU) * |IoU - sigmoid(x)|^2
  const float score = iou_like(px1, py1, px2, py2, gx1, gy1, gx2, gy2, false);
  const float sig = 1.f / (1.f + expf(-x));
  const float logsig = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
  const float bce = (1.f - score) * x - logsig;
  const float sf = fabsf(score - sig);
  const float cls = bce * (sf * sf) * w_cls;

  cost[(size_t)Q * pr.gt_start + e] = cls + reg + iouc;
}

}  // namespace
}  // namespace dskd

using namespace dskd;

extern "C" int dskd_match_cost(const float* bbox_pred, const float* cls_pred,
                               const float* gt_bboxes, const int64_t* gt_labels,
                               const int64_t* gt_start, const float* img_wh, float* cost,
                               int nprob, int Q, int C, float w_cls, float w_reg,
                               float w_iou, void* stream) {
  if (nprob < 0 || Q < 0 || C <= 0) return fail(DSKD_ERR_INVALID_ARG, "dskd_match_cost: bad sizes");
  if (nprob == 0 || Q == 0) return DSKD_OK;
  if (!bbox_pred || !cls_pred || !gt_start || !img_wh)
    return fail(DSKD_ERR_INVALID_ARG, "dskd_match_cost: null pointer");
  if (gt_start[nprob] > gt_start[0] && (!gt_bboxes || !gt_labels || !cost))
    return fail(DSKD_ERR_INVALID_ARG, "dskd_match_cost: null pointer");
  hipStream_t st = (hipStream_t)stream;
  for (int p0 = 0; p0 < nprob; p0 += kPack) {
    const int n = std::min(kPack, nprob - p0);
    CostPack pack;
    int maxG = 0;
    for (int k = 0; k < kPack; ++k) {
      const int p = p0 + (k < n ? k : 0);
      const long long G = gt_start[p + 1] - gt_start[p];
      if (G < 0 || G > (1 << 20)) return fail(DSKD_ERR_INVALID_ARG, "dskd_match_cost: bad gt_start at %d", p);
      pack.p[k] = CostProb{(long long)gt_start[p], (int)G, img_wh[2 * p], img_wh[2 * p + 1]};
      if (k < n) maxG = std::max(maxG, (int)G);
    }
    if (maxG == 0) continue;
    const dim3 grid((unsigned)((Q * maxG + 255) / 256), (unsigned)n), block(256);
    hipLaunchKernelGGL(match_cost_kernel, grid, block, 0, st, bbox_pred, cls_pred, gt_bboxes,
                       gt_labels, cost, pack, p0, Q, C, w_cls, w_reg, w_iou);
    if (int rc = check_launch("dskd_match_cost")) return rc;
  }
  return DSKD_OK;
}
