/*
 * dskd_hip.h -- C-ABI of the MI355X (gfx950) hot-path library for DSKD.
 *
 * This is the drop-in boundary: plain pointers and sizes,
 * no torch types.  Every pointer marked "device" is an HBM address owned by the
 * caller (torch allocations in our host code); the library allocates nothing
 * persistent, launches asynchronously on the passed stream (a hipStream_t passed
 * as void*), and never throws.  Return value: 0 on success, a negative code on
 * failure; dskd_last_error() then holds a human readable reason (thread local).
 *
 * Reference interfaces each entry point replaces (paths relative to the
 * reference checkout, smilekitty7/DSKD):
 *
 *   dskd_msda_fwd / dskd_msda_bwd
 *       ext-mmcv `MultiScaleDeformableAttnFunction.forward/backward`
 *       (mmcv-full>=1.3.17,<=1.6.2, mmcv/ops/multi_scale_deform_attn.py),
 *       imported at mmdet/models/utils/transformer.py:22-29 and reached through
 *       the encoder/decoder calls at transformer.py:985-995 and :1032-1043.
 *   dskd_lsap_host / dskd_lsap_batched
 *       `scipy.optimize.linear_sum_assignment(cost)` at
 *       mmdet/core/bbox/assigners/gfl_hungarian_assigner.py:143-151.
 *   dskd_match_cost
 *       the cost build of `GFLHungarianAssigner.assign`
 *       (gfl_hungarian_assigner.py:120-140) = BBoxL1Cost + IoUCost +
 *       QualityFocalLossCost (mmdet/core/bbox/match_costs/match_cost.py:34-51,
 *       :193-230, :460-476).
 *   dskd_dense_loss_fwd / dskd_dense_loss_bwd
 *       `loss_single_split` for all decoder layers: QFL / L1 / GIoU / DFL and their gradients
 *       (gfl_deformable_detr_head_il.py:1453-1529, mmdet/models/losses/gfocal_loss.py,
 *       iou_loss.py, smooth_l1_loss.py).
 *   dskd_proto_corr_fwd
 *       prototype accumulation + `correlation_mat` + MSELoss
 *       (mmdet/models/dense_heads/gfl_deformable_detr_head_il.py:525-555,
 *       :1197-1222).
 *   dskd_fgkd_fwd
 *       the `decode_v1` feature distillation loop + KL loss
 *       (gfl_deformable_detr_head_il.py:664-718, mmdet/models/losses/kd_loss.py:10-43).
 *   dskd_add_ln_fwd / dskd_add_ln_bwd
 *       the tail of every transformer sub-layer of ext-mmcv `BaseTransformerLayer.forward`
 *       (mmcv/cnn/bricks/transformer.py, imported at mmdet/models/utils/transformer.py:13-15
 *       and run by DetrTransformerEncoder :454-483 / DeformableDetrTransformerDecoder
 *       :625-710): `identity + dropout(out)` of MultiScaleDeformableAttention / FFN, the
 *       following 'norm' (nn.LayerNorm) and the next layer's `query + query_pos`.
 *   dskd_dropout_fwd / dskd_relu_dropout_bwd
 *       `Sequential(Linear, ReLU, Dropout)` of ext-mmcv FFN (mmcv/cnn/bricks/transformer.py),
 *       the feed-forward of every transformer layer (mmdet/models/utils/transformer.py:454-483):
 *       the Dropout forward, and Dropout + ReLU backward + the Linear's bias gradient.
 *   dskd_bias_act
 *       the elementwise tail of a ResNet conv->BN->ReLU group with the frozen BN folded into
 *       the convolution: `self.relu(norm(conv(x)))` and `out += identity; out = self.relu(out)`
 *       (mmdet/models/backbones/resnet.py:271-303).
 */
#ifndef DSKD_HIP_H
#define DSKD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* value/out element type of the MSDA entry points */
#define DSKD_DTYPE_F32 0
#define DSKD_DTYPE_BF16 1

#define DSKD_OK 0
#define DSKD_ERR_INVALID_ARG (-1)   /* shape / pointer the kernels do not support   */
#define DSKD_ERR_LAUNCH (-2)        /* HIP runtime refused the launch               */
#define DSKD_ERR_INVALID_COST (-3)  /* scipy: "matrix contains invalid numeric entries" */
#define DSKD_ERR_INFEASIBLE (-4)    /* scipy: "cost matrix is infeasible"           */

/* ABI version of this header; bumped on any signature change. */
int dskd_abi_version(void);
/* Last error message of the calling thread ("" when none). */
const char* dskd_last_error(void);
/* Zero `bytes` bytes at `p` (device, 16-byte aligned, bytes % 16 == 0) with a KERNEL on `stream`.  For the buffers the
 * entry points below want zeroed by the caller: hipMemsetAsync (what torch.zeros issues) captured into a hipGraph
 * replays with a garbage fill value on this ROCm runtime (csrc/common.h). */
int dskd_zero_fill(void* p, int64_t bytes, void* stream);
/* Number of HIP devices visible to the library (0 when there is no GPU). */
int dskd_device_count(void);

/* ---------------------------------------------------------------------------
 * Multi-scale deformable attention, sampling + aggregation.
 *
 *   out[b,q,h,:] = sum_{l,p} attn[b,q,h,l,p] * bilinear(value_l[b,:,h,:], loc[b,q,h,l,p])
 *
 * value          device, [B, Nv, heads, ch]   f32 or bf16 (dtype)
 * spatial_shapes host,   [levels, 2] int64    (H_l, W_l)
 * level_start    host,   [levels]   int64     row offset of level l inside Nv
 * loc            device, [B, Nq, heads, levels, points, 2] f32, (x, y) in [0,1]
 * attn           device, [B, Nq, heads, levels, points]    f32
 * out            device, [B, Nq, heads*ch]    same dtype as value
 * Supported: heads == 8, ch == 32, levels <= 4, levels*points <= 16.
 * Nq == Nv with 4 levels x 4 points (the encoder) in bf16 takes the windowed kernel (coarse levels of one head in
 * LDS), bit-identical to the plain one.
 * ------------------------------------------------------------------------- */
int dskd_msda_fwd(const void* value, const int64_t* spatial_shapes,
                  const int64_t* level_start, const float* loc, const float* attn,
                  void* out, int B, int Nv, int Nq, int heads, int ch, int levels,
                  int points, int dtype, void* stream);

/* The same with the module's prologue folded in (no-gradient forward: frozen teacher, inference):
 * `both` (device, [B*Nq, heads*16*3] same dtype as value: offsets [heads,16,2] then logits
 * [heads,16], as for dskd_msda_prep_fwd) and `ref` (device, [B*Nq, levels, 2] f32) replace loc /
 * attn, which are never materialised.  Bit-identical to dskd_msda_prep_fwd + dskd_msda_fwd.
 * Requires levels*points == 16. */
int dskd_msda_fwd_fused(const void* value, const int64_t* spatial_shapes,
                        const int64_t* level_start, const void* both, const float* ref, void* out,
                        int B, int Nv, int Nq, int heads, int ch, int levels, int points, int dtype,
                        void* stream);

/* Backward of the above.
 * grad_out    device, [B, Nq, heads*ch]  same dtype as value
 * grad_value  device, [B, Nv, heads, ch] f32, MUST be zeroed by the caller
 *             (contributions are accumulated with float atomics)
 * grad_loc    device, [B, Nq, heads, levels, points, 2] f32 (overwritten)
 * grad_attn   device, [B, Nq, heads, levels, points]    f32 (overwritten)
 */
int dskd_msda_bwd(const void* value, const int64_t* spatial_shapes,
                  const int64_t* level_start, const float* loc, const float* attn,
                  const void* grad_out, float* grad_value, float* grad_loc,
                  float* grad_attn, int B, int Nv, int Nq, int heads, int ch,
                  int levels, int points, int dtype, void* stream);

/* The same with a caller-owned workspace, which unlocks the fastest encoder-shape path (Nq == Nv, 4 levels x 4
 * points): grad_value of the finest level is produced by a tiled gather ("pull") kernel with plain stores -- f32
 * accumulation in registers, no atomics, no fixed point -- and only samples that stray further than a few cells
 * from their query go through a list in the workspace and are added afterwards; in bf16, grad_value of levels 1-3 is
 * formed on the matrix cores (csrc/msda_mm.hip: S^T G per window tile, scaled by the gather kernel's statistics,
 * which also live in the workspace).
 * grad_value   need NOT be zeroed: it is overwritten.
 * workspace    device, 16-byte aligned, at least dskd_msda_bwd_workspace(...) bytes (its header is zeroed at the start
 *              of every call); one workspace per stream in flight.
 * Environment (A/B measurements only): DSKD_MSDA_PULL_LEVELS=<digits> levels handled by the pull kernel (default
 * "0"; "none" = none); DSKD_MSDA_MM=<digits> levels on the matrix-core kernel (default "123"; "0" = none: the
 * windowed fixed-point kernels). */
int64_t dskd_msda_bwd_workspace(int B, int Nv, int Nq, int heads, int levels, int points);
int dskd_msda_bwd_ws(const void* value, const int64_t* spatial_shapes,
                     const int64_t* level_start, const float* loc, const float* attn,
                     const void* grad_out, float* grad_value, float* grad_loc,
                     float* grad_attn, int B, int Nv, int Nq, int heads, int ch,
                     int levels, int points, int dtype, void* workspace,
                     int64_t workspace_bytes, void* stream);

/* Prologue / epilogue of the MultiScaleDeformableAttention module around the sampling op
 * (what ext-mmcv's module does in PyTorch between its Linear layers and the CUDA op):
 *   attn = softmax over levels*points of the logits;  loc = ref + offsets / (W_l, H_l).
 * both    device, [n_query, heads*16*3] f32|bf16: offsets [heads,16,2] then logits [heads,16]
 * ref     device, [n_query, levels, 2] f32;   loc/attn device f32 outputs (layouts as above)
 * Requires levels*points == 16, levels <= 4.  n_query = B*Nq. */
int dskd_msda_prep_fwd(const void* both, const float* ref, const int64_t* spatial_shapes,
                       float* loc, float* attn, int64_t n_query, int heads, int levels,
                       int points, int dtype, void* stream);
/* Backward of the above: grad_both (same dtype/layout as both) from grad_loc, grad_attn and
 * the saved attn.  reference points get no gradient from this entry point. */
int dskd_msda_prep_bwd(const float* grad_loc, const float* grad_attn, const float* attn,
                       const int64_t* spatial_shapes, void* grad_both, int64_t n_query,
                       int heads, int levels, int points, int dtype, void* stream);
/* grad_ref[q, l, 0:2] (f32) = sum over heads and points of grad_loc[q, h, l, p, 0:2]: the gradient of the reference points
 * (the decoder's come from a trainable Linear on the query embedding: mmdet/models/utils/transformer.py:1016-1017). */
int dskd_msda_grad_ref(const float* grad_loc, float* grad_ref, int64_t n_query, int heads, int levels, int points,
                       void* stream);

/* ---------------------------------------------------------------------------
 * Rectangular linear sum assignment, bit-exact with scipy 1.15.3
 * `linear_sum_assignment` (shortest augmenting path, Crouse 2016), including
 * its tie breaking, the transpose when nr > nc, and the row-sorted output.
 *
 * Host version: cost is a host pointer, [nr, nc] row-major f32 (cast to f64
 * exactly like scipy does with a float32 array). row/col hold min(nr,nc) pairs.
 * ------------------------------------------------------------------------- */
int dskd_lsap_host(const float* cost, int nr, int nc, int64_t* row, int64_t* col);

/* Batched device version: nprob independent problems in ONE launch.
 * cost     device, problem p is [nr[p], nc[p]] row-major f32 at cost + offsets[p]
 * nr, nc   host,   [nprob] int32
 * offsets  host,   [nprob] int64 (element offsets)
 * row,col  device, int64; problem p writes min(nr,nc) pairs at out_offsets[p]
 * out_offsets host, [nprob] int64
 * status   device, [nprob] int32: 0 ok, DSKD_ERR_INVALID_COST / _INFEASIBLE
 * Limits: min(nr,nc) <= 1024 and max(nr,nc) <= 1024.
 */
int dskd_lsap_batched(const float* cost, const int32_t* nr, const int32_t* nc,
                      const int64_t* offsets, int nprob, int64_t* row, int64_t* col,
                      const int64_t* out_offsets, int32_t* status, void* stream);
/* Problems whose larger side exceeds 64 (300 queries x G boxes always does) run on the register-resident kernel: every
 * column's state in the registers of its thread, the work matrix transposed in LDS when it fits (r4: 300 x 110 in half of
 * the one-wave kernel's 477 us); same arithmetic and tie rule, bit-identical results.  dskd_lsap_tune(mode): 0 automatic
 * (default), 1 the one-wave kernel of round 1 for every size, 2 / 3 = 1 / 2 columns per thread (tests and A/B runs).
 * Process-global. */
int dskd_lsap_tune(int mode);

/* ---------------------------------------------------------------------------
 * Fused matching cost of GFLHungarianAssigner for nprob (layer, image) problems.
 *
 * bbox_pred  device, [nprob, Q, 4] f32 normalised (cx, cy, w, h)
 * cls_pred   device, [nprob, Q, C] f32 logits
 * gt_bboxes  device, [sum G, 4]    f32 pixel (x1, y1, x2, y2), problems concatenated
 * gt_labels  device, [sum G]       int64
 * gt_start   host,   [nprob + 1]   int64 prefix offsets into gt_bboxes/gt_labels
 * img_wh     host,   [nprob, 2]    f32 (img_w, img_h) of the un-padded image
 * cost       device, problem p is [Q, G_p] row-major f32 at cost + Q*gt_start[p]
 * weights: w_cls (QFL cost), w_reg (L1 on cxcywh), w_iou (-GIoU)
 * ------------------------------------------------------------------------- */
int dskd_match_cost(const float* bbox_pred, const float* cls_pred,
                    const float* gt_bboxes, const int64_t* gt_labels,
                    const int64_t* gt_start, const float* img_wh, float* cost,
                    int nprob, int Q, int C, float w_cls, float w_reg, float w_iou,
                    void* stream);

/* ---------------------------------------------------------------------------
 * The dense detection losses of all decoder layers x images at once: the per-layer arithmetic of
 * `GFLDeformableDETRHead_il.loss_single_split` (gfl_deformable_detr_head_il.py:1453-1529) on
 * precomputed dense targets -- QualityFocalLoss beta = 2 with the IoU of the positives as
 * target score AND its gradient into the boxes (mmdet/models/losses/gfocal_loss.py:12-53),
 * L1Loss on normalised cxcywh, GIoULoss eps = 1e-6 on pixel boxes (losses/iou_loss.py,
 * core/bbox/iou_calculators/iou2d_calculator.py:190-261), DistributionFocalLoss
 * (gfocal_loss.py:103-125) with the reference's targets (w/2, w/2, h/2, h/2) -- each reduced
 * per layer as loss_weight * sum / (avg_factor + eps_f32) (losses/utils.py weight_reduce_loss;
 * avg_factor = avg_pos, 4 * avg_pos for DFL).  Rows r = layer * N + query, R = nl * N.
 *
 * cls      device [R, C] f32 logits                 box   device [R, 4] f32 cxcywh in [0, 1]
 * lrtb     device [R, 4 * R1] f32 (R1 = reg_max+1)   labels device [R] int64 (background = C)
 * tgt      device [R, 4] f32 cxcywh targets          pos   device [R] bool (1 byte)
 * factors  device [N, 4] f32 (w, h, w, h) of the query's image
 * avg_pos  device [1] f32 = clamp(mean over ranks of num_total_pos, 1)
 * losses   device [4, nl] f32 OUT: loss_cls, loss_bbox, loss_iou, loss_dfl per layer
 * row_loss device [4, R] f32 workspace; d_cls [R, C], d_box [3, R, 4], d_lrtb [R, 4 * R1] f32 OUT:
 *          unit-upstream gradients, consumed by dskd_dense_loss_bwd
 * dskd_dense_loss_bwd: grad_losses device [4, nl] f32 (upstream gradients of `losses`) ->
 *          grad_cls [R, C], grad_box [R, 4], grad_lrtb [R, 4 * R1].
 * Deterministic (no atomics).  C <= 128, 2 <= R1 <= 64; tie / clamp conventions of PyTorch autograd.
 * ------------------------------------------------------------------------- */
int dskd_dense_loss_fwd(const float* cls, const float* box, const float* lrtb, const int64_t* labels,
                        const float* tgt, const unsigned char* pos, const float* factors, const float* avg_pos,
                        float* losses, float* row_loss, float* d_cls, float* d_box, float* d_lrtb, int nl, int N,
                        int C, int R1, float w_cls, float w_bbox, float w_iou, float w_dfl, void* stream);
int dskd_dense_loss_bwd(const float* grad_losses, const float* avg_pos, const float* d_cls, const float* d_box,
                        const float* d_lrtb, float* grad_cls, float* grad_box, float* grad_lrtb, int nl, int N, int C,
                        int R1, float w_cls, float w_bbox, float w_iou, float w_dfl, void* stream);

/* ---------------------------------------------------------------------------
 * DSKD loss 1: between-class distance-matrix distillation.
 *
 * hs_s        device, [N, D] f32   student last-layer query embeddings (N = B*Q)
 * labels_s    device, [N] int64    assigned labels of the last decoder layer
 * prev_mask   device, [C] uint8    1 where class id is a previous-task label
 * hs_t        device, [N, D] f32   teacher last-layer query embeddings
 * keepid_t    device, [M] int64    flattened teacher query index per detection
 * labels_t    device, [M] int64    teacher label per detection
 * L           number of previous classes (rows kept, reference `[:prev_length]`)
 * loss        device, [1] f32      MSE(D_t, D_s).mean() / L * loss_weight
 * grad_hs_s   device, [N, D] f32   d loss / d hs_s (overwritten, dense)
 * workspace   device, >= dskd_proto_corr_workspace(L, D) bytes
 * ------------------------------------------------------------------------- */
int64_t dskd_proto_corr_workspace(int L, int D);
int dskd_proto_corr_fwd(const float* hs_s, const int64_t* labels_s,
                        const uint8_t* prev_mask, const float* hs_t,
                        const int64_t* keepid_t, const int64_t* labels_t, int N,
                        int D, int C, int M, int L, float loss_weight, float* loss,
                        float* grad_hs_s, void* workspace, void* stream);

/* ---------------------------------------------------------------------------
 * DSKD loss 2 (`decode_v1`): semantic-guided feature-map distillation.
 *
 * feat_s[l], feat_t[l]  device, [B, C, H_l, W_l] f32 student / teacher neck maps
 *                       (host arrays of `levels` device pointers)
 * shapes      host,   [levels, 2] int32 (H_l, W_l)
 * boxes       device, [M, 4] f32 teacher boxes, pixel xyxy, images concatenated
 * box_start   host,   [B + 1] int32 prefix offsets of boxes per image
 * img_hw      host,   [B, 2]  f32 (img_h, img_w) un-padded
 * hs_t        device, [N, D] f32; keepid_t device [M] int64 (teacher row per box)
 * hs_s        device, [N, D] f32; labels_s device [N] int64; prev_mask device [C]
 *             the k-th student row (ascending) whose label is a previous-task
 *             label is paired with box k (reference `id_pred`)
 * T, loss_weight  KL temperature and weight
 * loss        device, [1] f32   sum over levels and images / B
 * grad_hs_s   device, [N, D] f32 d loss / d hs_s (overwritten, dense)
 * workspace   device, >= dskd_fgkd_workspace(...) bytes
 * status      device, [1] int32: 0 ok, 1 when fewer paired student rows than boxes
 * Requires C == D (the reference multiplies a D-vector into C channels).
 * ------------------------------------------------------------------------- */
int64_t dskd_fgkd_workspace(int B, int C, int levels, const int32_t* shapes, int M, int N);
int dskd_fgkd_fwd(const float* const* feat_s, const float* const* feat_t,
                  const int32_t* shapes, int levels, int B, int C,
                  const float* boxes, const int32_t* box_start, const float* img_hw,
                  const float* hs_t, const int64_t* keepid_t, const float* hs_s,
                  const int64_t* labels_s, const uint8_t* prev_mask, int N, int D,
                  int NC, int M, float T, float loss_weight, float* loss,
                  float* grad_hs_s, void* workspace, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------
 * Fused sub-layer tail:   z = res + dropout(h);  y = LayerNorm(z) * gamma + beta;  q = y + pos
 *
 * h, res      device, [rows, D] f32|bf16 (dtype): sub-layer output and residual (identity)
 * pos         device, [pos_rows, D] f32 or NULL; row r uses pos[r % pos_rows] (positional
 *             encoding shared by the images of a batch-first batch)
 * gamma, beta device, [D] f32
 * y           device, [rows, D] dtype (overwritten)
 * q           device, [rows, D] dtype or NULL (requires pos)
 * z, stats    device, [rows, D] dtype and [rows, 2] f32 (mean, rstd), saved for backward; both
 *             NULL in inference.  With bf16 the statistics are those of the ROUNDED z.
 * drop_p      dropout probability in [0, 1); the mask is Philox4x32-10(seed, offset + *epoch) counted by
 *             (row, lane) and is regenerated by the backward call from the same (seed, offset, epoch).
 * epoch       device, one uint64 or NULL (= 0): the part of the key that changes from step to step.  (seed,
 *             offset) are launch arguments and are frozen into a captured hipGraph; the caller bumps *epoch
 *             between replays (never between a forward and its backward), so every replay draws new masks.
 * All pointers 16-byte aligned.  Supported: D == 256.
 * ------------------------------------------------------------------------- */
int dskd_add_ln_fwd(const void* h, const void* res, const float* pos, int64_t pos_rows,
                    const float* gamma, const float* beta, void* y, void* q, void* z,
                    float* stats, int64_t rows, int D, float eps, float drop_p, uint64_t seed,
                    uint64_t offset, const uint64_t* epoch, int dtype, void* stream);
/* Backward of the above.
 * dy, dq      device, [rows, D] dtype: gradients of y and (or NULL) of q
 * dres        device, [rows, D] dtype: d loss / d res (overwritten)
 * dh          device, [rows, D] dtype: d loss / d h; pass NULL exactly when drop_p == 0
 *             (then d loss / d h == dres)
 * dgamma, dbeta device, [copies, D] f32, MUST be zeroed by the caller: workgroup w accumulates into
 *             copy w % copies with atomics (thousands of workgroups on ONE 1 KB row serialise in
 *             the L2 atomic units); the gradient is the sum over the copies.  copies >= 1.
 */
int dskd_add_ln_bwd(const void* dy, const void* dq, const void* z, const float* stats,
                    const float* gamma, void* dres, void* dh, float* dgamma, float* dbeta,
                    int copies, int64_t rows, int D, float drop_p, uint64_t seed, uint64_t offset,
                    const uint64_t* epoch, int dtype, void* stream);
/* The same with a SECOND gradient of y (dy2, same shape, may be NULL): when y feeds two consumers (the next sub-layer and the
 * next residual add) their gradients arrive as two tensors and are summed in this launch instead of by an add launch (r4). */
int dskd_add_ln_bwd2(const void* dy, const void* dy2, const void* dq, const void* z, const float* stats,
                     const float* gamma, void* dres, void* dh, float* dgamma, float* dbeta,
                     int copies, int64_t rows, int D, float drop_p, uint64_t seed, uint64_t offset,
                     const uint64_t* epoch, int dtype, void* stream);
/* q = bf16(x + pos[r % pos_rows]): x, q device [rows, D] bf16, pos device f32 [pos_rows, D]; D % 8 == 0 -- the first
 * encoder layer's `query + query_pos` (ext-mmcv MultiScaleDeformableAttention.forward), later ones come from
 * dskd_add_ln_fwd(want q). */
int dskd_add_pos(const void* x, const float* pos, void* q, int64_t rows, int64_t pos_rows, int D, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * In-place epilogue of a folded convolution:  x = act(x + bias[c] (+ identity))
 * x, identity  device, channels_last activation [N, H, W, C] (C innermost), n = N*H*W*C elements,
 *              f32 | bf16 (dtype); identity may be NULL
 * bias         device, [C] same dtype;  relu != 0 applies max(., 0)
 * Requires C % 8 == 0 (bf16) / C % 4 == 0 (f32), 16-byte aligned pointers.
 * ------------------------------------------------------------------------- */
int dskd_bias_act(void* x, const void* bias, const void* identity, int64_t n, int C, int relu,
                  int dtype, void* stream);
/* The stem's tail without gradients: y = maxpool_3x3_s2_p1(relu(x + bias[c])) in ONE pass
 * (`x = self.relu(x); x = self.maxpool(x)` behind the folded norm1, mmdet/models/backbones/resnet.py:633-640).
 * x [B, H, W, C] channels_last (the convolution's output WITHOUT bias), y [B, (H-1)/2+1, (W-1)/2+1, C]; bit-identical
 * with dskd_bias_act(relu) followed by the pooling (rounding and ReLU are monotonic, the bias constant over a window). */
int dskd_bias_relu_maxpool(const void* x, const void* bias, void* y, int B, int H, int W, int C, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * FFN hidden activation (bf16 only).
 * dskd_dropout_fwd: y (device, n elements, n % 8 == 0) is overwritten with dropout_p(y); the mask
 *   is Philox4x32-10(seed, offset + *epoch) (epoch as in dskd_add_ln_fwd) and is NOT stored.
 * dskd_relu_dropout_bwd: for y_dropped = dropout_p(relu(.)),
 *       out = g * (y_dropped != 0) / (1 - p)        (ReLU active AND kept <=> y_dropped != 0)
 *       colsum[c] += sum over rows of out[:, c]     (the bias gradient of the Linear before the
 *                                                    ReLU; NULL to skip, else zeroed by the caller)
 *   g, y_dropped, out device [rows, C]; C in {256, 512, 1024, 2048}.
 * ------------------------------------------------------------------------- */
int dskd_dropout_fwd(void* y, int64_t n, float p, uint64_t seed, uint64_t offset,
                     const uint64_t* epoch, int dtype, void* stream);
int dskd_relu_dropout_bwd(const void* g, const void* y_dropped, void* out, float* colsum,
                          int copies, int64_t rows, int C, float p, int dtype, void* stream);
/* colsum[c] += sum over rows of x[:, c]  -- the bias gradient of a Linear (`grad.sum(0)` in
 * AddmmBackward of every nn.Linear of the transformer).  x device [rows, C] bf16, C in
 * {256, 384, 512, 1024, 2048}; colsum device [copies, C] f32, zeroed by the caller (copies as in
 * dskd_add_ln_bwd; same for dskd_relu_dropout_bwd). */
int dskd_colsum(const void* x, float* colsum, int copies, int64_t rows, int C, int dtype,
                void* stream);
/* out[c] = sum over rows of x[:, c] as bf16, ONE launch without atomics / zero fill -- the same bias gradient for SHORT
 * inputs (the 1 200 query rows of the decoder's nn.Linear layers, 7 200 of the head branches); rows <= 65 536, C % 8 == 0. */
int dskd_colsum_short(const void* x, void* out, int64_t rows, int C, int dtype, void* stream);
/* out[p][c] = sum_k acc[p][k][c] (f32 or bf16), then acc = 0: hands the [planes][copies][C] f32 accumulators of dskd_colsum,
 * dskd_add_ln_bwd (planes = 2: d gamma, d beta), dskd_ffn_bwd and dskd_relu_dropout_bwd over in the parameter's dtype and
 * leaves them zeroed for their next use -- the caller keeps them as persistent buffers instead of zero-filling a fresh
 * one per call (same idea as dskd_cvt_clear for the weight gradients). */
int dskd_sum_clear(float* acc, int planes, int copies, int C, void* out, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------
 * The encoder FFN as one MFMA kernel per direction (bf16; d_model 256, hidden 1024 -- other sizes are refused and the
 * caller keeps the GEMM chain).  Replaces, for the tall encoder activation, the reference's
 *   Linear -> ReLU -> Dropout -> Linear        (ext-mmcv FFN.layers, run by mmdet/models/utils/transformer.py:454-483)
 * and its autograd backward; the trailing Dropout + residual + LayerNorm stay in dskd_add_ln_fwd.
 *
 * dskd_ffn_pack     w1 [hidden, d_model], w2 [d_model, hidden] (nn.Linear layout, device) -> the two weight images
 *                   in MFMA fragment order, dskd_ffn_packed_bytes() each; packed_bwd may be NULL (inference).
 *                   Call again whenever the weights change (once per optimiser step).
 * dskd_ffn_fwd      y = (dropout_p(relu(x w1^T + b1))) w2^T + b2       x, y [tokens, d_model]; b1, b2 bf16
 *                   h_out [tokens, hidden] receives H = dropout_p(relu(.)) for the backward; NULL (p must be 0) skips it.
 *                   Dropout mask = that of dskd_dropout_fwd on H (Philox4x32-10(seed, offset + *epoch), not stored).
 * dskd_ffn_bwd      grad_h = (grad_y w2) * [h != 0] / (1 - p)          [tokens, hidden]  (input of the w1 / b1 gradients)
 *                   grad_x = grad_h w1 (+ grad_x_add)                  [tokens, d_model]; grad_x_add [tokens, d_model] bf16 or
 *                                             NULL: another gradient of the same x (the residual branch of the following
 *                                             LayerNorm), added in the epilogue instead of by a separate pass
 *                   grad_b1[k % copies][:] += column sums of grad_h over workgroup k's tokens  (f32 [copies, hidden],
 *                                             zeroed by the caller, summed over copies by the caller; NULL to skip)
 *                   The weight gradients are plain GEMMs over the tokens: grad_w2 = grad_y^T h, grad_w1 = grad_h^T x;
 *                   grad_b2 = dskd_colsum(grad_y).
 * All pointers 16-byte aligned.
 * ------------------------------------------------------------------------- */
int64_t dskd_ffn_packed_bytes(int d_model, int hidden);
int dskd_ffn_pack(const void* w1, const void* w2, void* packed_fwd, void* packed_bwd, int d_model, int hidden,
                  int dtype, void* stream);
int dskd_ffn_fwd(const void* x, const void* packed_fwd, const void* b1, const void* b2, void* h_out, void* y,
                 int64_t tokens, int d_model, int hidden, float p, uint64_t seed, uint64_t offset,
                 const uint64_t* epoch, int dtype, void* stream);
int dskd_ffn_bwd(const void* grad_y, const void* h, const void* packed_bwd, void* grad_h, void* grad_x,
                 const void* grad_x_add, float* grad_b1, int copies, int64_t tokens, int d_model, int hidden, float p,
                 int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * y[tokens, N] = act(x[tokens, 256] w^T + bias) for a very tall x (bf16; K = 256, N a multiple of 32 up to 512): the
 * encoder's 256 -> 256 / 384 nn.Linear layers (value_proj, output_proj, sampling_offsets | attention_weights of ext-mmcv
 * MultiScaleDeformableAttention) and, with the weight packed transposed, their input-gradient GEMMs, as one hand-written
 * MFMA kernel (csrc/ffn_mfma.hip: GEMM-1 of the FFN loop).  Memory-bound; hipBLASLt takes 43 us for 88 892 tokens.
 *   dskd_lin256_pack   w [N, 256] (nn.Linear layout) -> fragment-order image of dskd_lin256_packed_bytes(N) bytes;
 *                      transposed != 0: w is [256, N] and the image is that of w^T  (dX = grad_y w for a [256, 256] w)
 *   dskd_lin256_fwd    bias bf16 [N] or NULL; relu != 0 applies max(., 0)
 * ------------------------------------------------------------------------- */
int64_t dskd_lin256_packed_bytes(int N);
int dskd_lin256_pack(const void* w, void* packed, int N, int K, int transposed, int dtype, void* stream);
/* The same for many weights in one launch: table = device int64 [n, 4] rows {w pointer, packed pointer, N, transposed}, every
 * row as dskd_lin256_pack would take it (the caller validates: N a multiple of 32 in [32, 512], K = 256, 16-byte alignment). */
int dskd_lin256_pack_many(const int64_t* table, int n, int dtype, void* stream);
int dskd_lin256_fwd(const void* x, const void* packed, const void* bias, void* y, int64_t tokens, int N, int K, int relu,
                    int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * y[M, N] = act(x[M, K] w[N, K]^T + bias[N] (+ res[M, N]))  (bf16 in / out, f32 accumulation; N, K multiples of 64):
 * a 1x1 convolution on a channels_last activation with the folded-BN shift, the residual and the ReLU in the epilogue --
 * conv1 / conv3 / downsample of every Bottleneck (mmdet/models/backbones/resnet.py:271-303, the conv -> norm -> relu
 * chains of :271-296 and `out += identity; out = self.relu(out)` of :298-303) and ChannelMapper's lateral convolutions
 * (mmdet/models/necks/channel_mapper.py:90-100), which the reference runs as cuDNN convolution + BatchNorm + add + ReLU
 * launches.  One hand-written MFMA kernel (csrc/gemm_nt.hip); with w = W^T it is the input-gradient GEMM dX = dY W.
 *   x        activation rows, row stride K elements; stride == 0: row m of x; stride s > 0 (a strided 1x1 convolution):
 *            output row m = (img, ho, wo) of an [.., Ho, Wo] map reads row ((img * Hi + s * ho) * Wi + s * wo)
 *   bias     bf16 [N] or NULL;  res bf16 [M, N] or NULL (added before the activation);  relu != 0: max(., 0)
 * ------------------------------------------------------------------------- */
int dskd_gemm_nt(const void* x, const void* w, const void* bias, const void* res, void* y, int64_t M, int N, int K,
                 int relu, int stride, int Ho, int Wo, int Hi, int Wi, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * y = act(conv3x3(x, w, stride, padding 1) + bias (+ res)) on a channels_last bf16 activation x [B, Hi, Wi, C] with the
 * channels_last weight w [N][3][3][C] -- conv2 of every Bottleneck (mmdet/models/backbones/resnet.py:283-288: conv2 ->
 * bn2 -> relu, BN folded) as an implicit GEMM on the kernel of dskd_gemm_nt: K = 9 C, stage = 64 channels of one tap, the
 * activation rows of a tap addressed in place (a page of zeros outside the image), same fused epilogue.  With the taps
 * flipped and the weight transposed it is the stride-1 input-gradient convolution.  C = 64 * 2^k, N a multiple of 64.
 *   y [B, Ho, Wo, N], Ho = (Hi - 1) / stride + 1 (same for Wo);  bias bf16 [N] or NULL;  res bf16 [B, Ho, Wo, N] or NULL
 * ------------------------------------------------------------------------- */
int dskd_conv3x3(const void* x, const void* w, const void* bias, const void* res, void* y, int B, int Hi, int Wi, int C, int N,
                 int stride, int relu, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Input gradients of the two convolutions above with the neighbouring elementwise steps of a Bottleneck's backward
 * (mmdet/models/backbones/resnet.py:271-303 run backwards by autograd: relu -> threshold_backward, `out += identity`
 * -> a gradient add) folded into the epilogue:
 *   dskd_gemm_nt_dx   y[M, N] = (gate > 0) ? g[M, K] wt[N, K]^T + res : 0      (1x1 convolution / Linear, stride 1)
 *   dskd_conv3x3_dx   y       = (gate > 0) ? conv3x3(g, wt, stride 1, padding 1) : 0   (wt: taps flipped, roles swapped)
 * gate = the ReLU OUTPUT that fed the convolution (its own input x, shape of y) or NULL; res = the gradient arriving over
 * the identity path (shape of y) or NULL.  bf16, same alignment / size rules as the forward entry points.
 * ------------------------------------------------------------------------- */
int dskd_gemm_nt_dx(const void* g, const void* wt, const void* res, const void* gate, void* y, int64_t M, int N, int K,
                    int dtype, void* stream);
int dskd_conv3x3_dx(const void* g, const void* wt, const void* gate, void* y, int B, int Hi, int Wi, int C, int N, int dtype,
                    void* stream);

/* ---------------------------------------------------------------------------
 * The same four products with a caller-owned scratch buffer (r4): the general forms
 *   dskd_gemm_nt_ws   y[M, N] = gate_relu(x[M, K] w[N, K]^T + bias + res)   (= dskd_gemm_nt / dskd_gemm_nt_dx)
 *   dskd_conv3x3_ws   y       = gate_relu(conv3x3(x, w, stride) + bias + res)   (= dskd_conv3x3 / dskd_conv3x3_dx)
 * with gate_relu(v) = relu ? max(v, 0) : v, zeroed where gate <= 0 (gate NULL: no mask).  For K >= 256 the library picks a
 * big output tile (128 x 128 .. 256 x 128, csrc/gemm_nt.hip gemm_big_kernel) per layer shape; the tiles of the grid's
 * last, partial round are then split along K over the idle CUs, which write f32 partial tiles into `scratch`
 * (dskd_gemm_nt_scratch_bytes() bytes cover every shape; NULL / too small: no split) and a second launch sums them in a
 * fixed order and applies the epilogue -- deterministic, nothing persistent inside the library.  One scratch per stream in
 * flight.  Reference: the convolutions of mmdet/models/backbones/resnet.py:271-303 and necks/channel_mapper.py:90-100.
 * Which tile runs where is a measured table (profiles/r04_gemm_big_microbench.txt): the big tiles serve the 3x3 convolutions
 * of ResNet stage 4, the 64 x 128 kernel everything else (with its epilogue through LDS -- whole 128-byte lines -- for M >= 8192).
 * dskd_gemm_nt_tune(cfg, splits): tuning hook of the microbenchmarks / tests (cfg < 0: automatic (default), 0: the small-tile
 * kernel, 1..6: a fixed big tile, 7 / 8 / 9: the small tile with the register / LDS epilogue / LDS epilogue without the early
 * residual + gate reads; splits 0: automatic, 1: never, > 1: forced).  Process-global; not for use while launches of another thread are in flight.
 * ------------------------------------------------------------------------- */
int64_t dskd_gemm_nt_scratch_bytes(void);
int dskd_gemm_nt_ws(const void* x, const void* w, const void* bias, const void* res, const void* gate, void* y, int64_t M,
                    int N, int K, int relu, int stride, int Ho, int Wo, int Hi, int Wi, int dtype, void* scratch,
                    int64_t scratch_bytes, void* stream);
int dskd_conv3x3_ws(const void* x, const void* w, const void* bias, const void* res, const void* gate, void* y, int B, int Hi,
                    int Wi, int C, int N, int stride, int relu, int dtype, void* scratch, int64_t scratch_bytes, void* stream);
int dskd_gemm_nt_tune(int cfg, int splits);

/* ---------------------------------------------------------------------------
 * c[N, K] += g[M, N]^T x[M, K]  (bf16 in, f32 out; N, K multiples of 128): the weight gradient dW = dY^T X of an
 * nn.Linear / 1x1 convolution over M tokens -- what autograd's mm / convolution_backward compute for the transformer's
 * Linear layers (ext-mmcv FFN, MultiScaleDeformableAttention projections) and the Bottleneck's 1x1 convolutions
 * (mmdet/models/backbones/resnet.py:271-303).  Split over the tokens; every workgroup adds its f32 tile with atomics:
 * c must be zero-filled (or hold a value to accumulate onto).  ldg / ldx: row strides of g / x in elements.
 * ------------------------------------------------------------------------- */
int dskd_gemm_tn(const void* g, const void* x, float* c, int64_t M, int N, int K, int ldg, int ldx, int dtype, void* stream);
/* out[N, K] (bf16) = g[M, N]^T x[M, K]: the same product with the split-K partial products written as plain stores into
 * `scratch` ([splits, N, K] f32, dskd_gemm_tn_scratch_bytes(M, N, K) bytes) and summed + cast by a second launch in a fixed
 * order -- no float atomics (their rate, ~1.3 TB/s chip-wide, made the 16 MB flush 12 us of every launch), no accumulator
 * to keep zeroed, deterministic.  out is overwritten.  M > 0. */
int64_t dskd_gemm_tn_scratch_bytes(int64_t M, int N, int K);
int dskd_gemm_tn_bf16(const void* g, const void* x, void* out, void* scratch, int64_t scratch_bytes, int64_t M, int N, int K,
                      int ldg, int ldx, int dtype, void* stream);
/* The same launch pair with the BIAS gradient as a by-product: db_out [N] (bf16) = column sums of g -- one more product per g
 * fragment (g^T x ones) on the waves of k-tile 0, its planes behind the product planes of the scratch, summed by the same
 * reduction launch: the separate column-sum + hand-over launches of a Linear layer's backward are gone.  Same scratch size. */
int dskd_gemm_tn_bias_bf16(const void* g, const void* x, void* out, void* db_out, void* scratch, int64_t scratch_bytes, int64_t M,
                           int N, int K, int ldg, int ldx, int dtype, void* stream);
/* Weight gradient of a 3x3 convolution (padding 1, stride 1 | 2): dw[n][ky][kx][c] (bf16 = a [N, C, 3, 3] channels_last
 * weight) = sum over the output pixels of g[pixel][n] * x[pixel shifted by the tap][c]; g [B, Ho, Wo, N] and x [B, Hi, Wi, C]
 * channels_last bf16, Ho = (Hi - 1) / stride + 1.  The split-K kernel of dskd_gemm_tn_bf16 over a virtual [pixels, 9 C]
 * operand (only its producers' source addresses differ): deterministic, scratch from ..._scratch_bytes (-1: bad shape).
 * C and N multiples of 128 (ResNet stages 2-4); other shapes stay with the library.  Replaces the weight half of
 * aten::convolution_backward for conv2 of a Bottleneck (mmdet/models/backbones/resnet.py:283-288). */
int64_t dskd_conv3x3_wgrad_scratch_bytes(int B, int Hi, int Wi, int C, int N, int stride);
int dskd_conv3x3_wgrad(const void* g, const void* x, void* dw, void* scratch, int64_t scratch_bytes, int B, int Hi, int Wi, int C,
                       int N, int stride, int dtype, void* stream);
/* ... with db_out [N] (bf16) = the sums of g over all output pixels (the gradient of the folded-BN bias) as a by-product, as in
 * dskd_gemm_tn_bias_bf16. */
int dskd_conv3x3_wgrad_bias(const void* g, const void* x, void* dw, void* db_out, void* scratch, int64_t scratch_bytes, int B,
                            int Hi, int Wi, int C, int N, int stride, int dtype, void* stream);
/* dst (bf16, n elements) = src (f32); src = 0 -- the accumulator of dskd_gemm_tn handed over in the parameter's dtype and
 * left zeroed for its next use (n a multiple of 4). */
int dskd_cvt_clear(float* src, void* dst, int64_t n, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Window attention of the Swin backbone (BASELINE configs[3]): softmax(q k^T * scale + bias (+ shift mask)) v for
 * windows of 49 tokens and head dimension 32 -- WindowMSA.forward of the reference between its qkv Linear and its output
 * projection (mmdet/models/backbones/swin.py:81-126; the mask of ShiftWindowMSA :180-286) -- and its backward, one wave
 * per (window, head) on the matrix cores (csrc/winattn.hip).  bf16.
 *   qkv    [windows, 49, 3, heads, 32]   the projection's output as it stands (q, k, v interleaved per token)
 *   table  [types][heads][64][64] f32    additive term [key][query]: relative position bias + mask of that type; -30000
 *                                        on the padded keys (>= 49)
 *   wtype  [nW] int32 or NULL            mask type of the nW windows of one image (window w uses wtype[w % nW])
 *   out    [windows, 49, heads * 32]     what the output projection reads;  dout: its gradient, same layout
 *   dqkv   [windows, 49, 3, heads, 32];  dtable [heads][64][64] f32: += the bias gradient [key][query] (zero it first)
 * ------------------------------------------------------------------------- */
int dskd_winattn_fwd(const void* qkv, const float* table, const int32_t* wtype, void* out, int windows, int heads, int nW,
                     int tokens, int head_dim, float scale, int dtype, void* stream);
int dskd_winattn_bwd(const void* qkv, const float* table, const int32_t* wtype, const void* dout, void* dqkv, float* dtable,
                     int windows, int heads, int nW, int tokens, int head_dim, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Self-attention of the decoder's object queries: dropout(softmax(q k^T * scale)) v per (image, head), 8 heads of 32
 * channels, up to 320 tokens -- the core of ext-mmcv MultiheadAttention (nn.MultiheadAttention between its input and output
 * projections; configs/deformable_detr/..._il.py:82-87, first sub-layer of each decoder layer, mmdet/models/utils/
 * transformer.py:639-709) -- and its backward, one wave per 32 queries / 32 keys of an (image, head) on the matrix cores
 * (csrc/attn.hip).  bf16.  q, k, v, out and their gradients are [B, L, heads * 32] in ANY (batch, row) strides, so q | k
 * are read in place from the joint projection's [.., 2 E] rows and both token layouts ([B, L, E], [L, B, E]) need no copy:
 *   strides [8] int64, elements          batch, row stride of q; of k; of v; of out.  dq / dk / dv / dout use the strides
 *                                        of q / k / v / out.  Multiples of 8; pointers 16-byte aligned.
 *   stats   [B, heads, L, 2] f32         row maximum of the scaled scores, 1 / row sum: forward writes (NULL: inference),
 *                                        backward reads
 *   delta   [B, heads, L] f32            backward scratch (sum_d dout * out per query), written by its pre-pass
 *   drop_p, seed, offset, epoch          attention dropout: element (b, h, query, key) is dropped when a counter hash of
 *                                        its index keyed by (seed, offset + *epoch) falls below drop_p; the backward must
 *                                        get the forward's values.  epoch: device word (NULL = 0), see dskd_dropout_fwd.
 * ------------------------------------------------------------------------- */
int dskd_attn_fwd(const void* q, const void* k, const void* v, void* out, float* stats, int B, int heads, int L, int head_dim,
                  const int64_t* strides, float scale, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* epoch,
                  int dtype, void* stream);
int dskd_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* stats,
                  float* delta, void* dq, void* dk, void* dv, int B, int heads, int L, int head_dim, const int64_t* strides,
                  float scale, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* epoch, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Global-norm gradient clipping + AdamW for every trainable tensor in two launches -- the reference's optimizer hook
 * (configs/deformable_detr/..._il.py:213-224: AdamW, grad_clip max_norm 0.1; ext-mmcv OptimizerHook = clip_grad_norm_ then
 * optimizer.step()), f32 parameters / gradients / moments.  Device tables (int64 addresses, filled by the caller):
 *   ptrs   [4][n_tensors]  parameter, gradient, exp_avg, exp_avg_sq
 *   meta   [n_tensors][2]  number of elements, parameter group
 *   chunks [n_chunks][2]   tensor, first element -- one entry per dskd_clip_adamw_chunk() elements of every tensor
 *   partials [n_chunks] f32 scratch;  norm_out [2] f32: total gradient norm, clip coefficient
 * lr / weight_decay: HOST arrays, one entry per group (<= 8); step = 1, 2, ... (bias corrections); max_norm <= 0: no
 * clipping.  The gradients are not modified (the clipped gradient exists only inside the update).
 * ------------------------------------------------------------------------- */
int dskd_clip_adamw_chunk(void);
int dskd_clip_adamw(const int64_t* ptrs, const int32_t* meta, const int32_t* chunks, float* partials, float* norm_out,
                    int n_tensors, int n_chunks, const float* lr, const float* weight_decay, int n_groups, float beta1,
                    float beta2, float eps, int64_t step, float max_norm, void* stream);

/* ---------------------------------------------------------------------------
 * Multi-tensor cast with an optional per-row scale in ONE launch (r4): the step's low-precision parameter copies and the
 * way back of their gradients -- what ext-mmcv's fp16 hooks / torch.autocast do tensor by tensor, and the fold of the frozen
 * BatchNorm into the trainable convolution weights, w * gamma / sqrt(var + eps) (mmdet/models/backbones/resnet.py:271-303
 * with norm_eval=True, configs/deformable_detr/.._il.py:30-37).
 *   table  device int64 [n][5] = {src, dst, scale (0: none), numel, inner}: dst[i] = src[i] * scale[i / inner], i in memory
 *          order (inner = elements per output channel of a dense conv weight in either memory format)
 *   first  device int32 [n + 1]: prefix sums of ceil(numel / dskd_cast_scale_chunk()) (workgroup -> row by binary search)
 *   direction 0: src f32 -> dst bf16;  1: src bf16 -> dst f32.  16-byte accesses where both pointers allow, else scalar.
 * ------------------------------------------------------------------------- */
int dskd_cast_scale_chunk(void);
int dskd_cast_scale_many(const int64_t* table, const int32_t* first, int n, int total_chunks, int direction, void* stream);
/* dst_i[k][taps - 1 - t][n] = src_i[n][t][k] for a list of bf16 convolution weights in ONE launch: the operands of the
 * input-gradient launches of a trainable ResNet stage (dskd_gemm_nt_dx: the transposed 1x1 weight; dskd_conv3x3_dx: the
 * tap-flipped, channel-swapped 3x3 weight -- what `w.flip(2, 3).transpose(0, 1)` of a channels_last [N, K, 3, 3] tensor holds;
 * mmdet/models/backbones/resnet.py:271-303 backward).  table: device rows {src, dst, N, K, taps} (int64; N, K multiples of 64,
 * taps 1 | 9, pointers 16-byte aligned), first[i] = number of 64 x 64 tiles (taps * N / 64 * K / 64 each) of the tensors
 * before i, first[n] = total_blocks. */
int dskd_weight_t_many(const int64_t* table, const int32_t* first, int n, int total_blocks, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * GroupNorm(32 groups, 256 channels) on a channels_last activation -- the norm of every ChannelMapper level
 * (mmdet/models/necks/channel_mapper.py:10-100: ext-mmcv ConvModule(conv, GN)); replaces F.group_norm and, under
 * autocast, the f32 casts and layout copies around it.  Other channel / group counts are refused.
 *   x, y, grad_y, grad_x   device, [B, HW, 256] rows (= a [B, 256, H, W] tensor in channels_last memory), f32 | bf16,
 *                          each with its own batch stride in elements (a level's slice of a concatenated token tensor
 *                          is a valid y / grad_y); 16-byte aligned
 *   gamma, beta            device f32 [256]
 *   sums                   forward: device scratch of dskd_gn_workspace(B, HW) bytes (partial moments, summed in a
 *                          fixed order: the forward is deterministic); backward: device f64 [B, 32, 2], ZEROED by the caller
 *   stats                  device f32 [B, 32, 2] = {mean, rstd}: written by the forward, read by the backward
 *   relu                   != 0: y = max(GroupNorm(x), 0) (ext-mmcv ConvModule with act_cfg=ReLU: the GFL head's towers); the
 *                          backward then masks grad_y by the sign of the forward's output, recomputed from x (needs beta)
 *   grad_gamma_beta        device f32 [copies, 2, 256], zeroed by the caller: [k][0] += partial grad_gamma, [k][1] +=
 *                          partial grad_beta (copies as in dskd_add_ln_bwd)
 * ------------------------------------------------------------------------- */
int64_t dskd_gn_workspace(int B, int64_t HW);
/* out [B, 256, HW] f32 (NCHW planes) = x [B, HW, 256] (channels_last rows, f32 | bf16): layout + dtype of the feature maps
 * dskd_fgkd_fwd reads, from the layout the neck produces. */
int dskd_nhwc_to_nchw_f32(const void* x, float* out, int B, int64_t HW, int C, int64_t x_batch_stride, int dtype,
                          void* stream);
int dskd_gn_fwd(const void* x, const float* gamma, const float* beta, void* y, double* sums, float* stats, int B,
                int64_t HW, int C, int groups, int64_t x_batch_stride, int64_t y_batch_stride, float eps, int relu,
                int dtype, void* stream);
int dskd_gn_bwd(const void* x, const void* grad_y, const float* stats, const float* gamma, const float* beta,
                void* grad_x, double* sums, float* grad_gamma_beta, int copies, int B, int64_t HW, int C, int groups,
                int64_t x_batch_stride, int64_t gy_batch_stride, int64_t gx_batch_stride, int relu, int dtype,
                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSKD_HIP_H */
