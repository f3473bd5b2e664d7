"""CPU restatement of the two DSKD losses (TEST INFRASTRUCTURE), written as the same
sequence of tensor operations the reference performs, so that autograd yields the
reference's gradients.

  loss_corr      mmdet/models/dense_heads/gfl_deformable_detr_head_il.py:525-555 (prototype
                 accumulation) and :1197-1222 (correlation_mat) + MSELoss
                 (mmdet/models/losses/mse_loss.py:10-13, :16-55, reduction 'mean')
  loss_fg (decode_v1)  gfl_deformable_detr_head_il.py:664-718 +
                 KnowledgeDistillationKLDivLoss (mmdet/models/losses/kd_loss.py:10-43, 'sum')
Pinned by tests/golden/loss_b1_l40.npz / loss_b2_l70.npz, produced by running the reference's own
``GFLDeformableDETRHead_il.loss`` (tests/golden/gen_golden.py).
"""
import torch
import torch.nn.functional as F


def proto_corr_loss(hs_s, labels_s, prev_mask, hs_t, keepid_t, labels_t, L, loss_weight=1.0):
    C = prev_mask.numel()
    D = hs_s.shape[1]
    corr_s = hs_s.new_zeros((C, D + 1))
    sel = torch.zeros_like(labels_s)
    for lab in torch.nonzero(prev_mask).flatten().tolist():     # :534-535
        sel[labels_s == lab] = 1
    for idx in torch.nonzero(sel).flatten().tolist():           # :536-539, ascending
        corr_s[labels_s[idx]][:-1] += hs_s[idx]
        corr_s[labels_s[idx]][-1] += 1
    corr_t = hs_s.new_zeros((C, D + 1))
    for i in range(len(labels_t)):                              # :549-551
        corr_t[labels_t[i]][:-1] += hs_t[keepid_t[i]]
        corr_t[labels_t[i]][-1] += 1
    # correlation_mat :1197-1222
    c_t, num_t = corr_t[:L, :-1], corr_t[:L, -1]
    idx_t = torch.nonzero(num_t).squeeze(1)
    c_t[idx_t] = c_t[idx_t] / num_t[idx_t].unsqueeze(1).repeat(1, D)
    c_s, num_s = corr_s[:L, :-1], corr_s[:L, -1]
    idx_s = torch.nonzero(num_t).squeeze(1)                     # :1205 (teacher counts, sic)
    c_s[idx_s] = c_s[idx_s] / num_s[idx_s].unsqueeze(1).repeat(1, D)
    mat_t = c_t.new_zeros((L, L))
    mat_s = c_t.new_zeros((L, L))
    for i in range(L):
        for j in range(L):
            mat_t[i][j] = torch.dist(c_t[i], c_t[j], p=2)
            mat_s[i][j] = torch.dist(c_s[i], c_s[j], p=2)
    return loss_weight * F.mse_loss(mat_t, mat_s, reduction="none").mean() / L


def kd_kl_sum(pred, soft, T, loss_weight):
    """kd_loss.py:10-43 with reduction='sum' on [C,H,W] inputs (softmax over dim=1 = H)."""
    target = F.softmax(soft / T, dim=1).detach()
    kd = F.kl_div(F.log_softmax(pred / T, dim=1), target, reduction="none").mean(1) * (T * T)
    return loss_weight * kd.sum()


def fgkd_loss(feats_s, feats_t, boxes, img_shapes, hs_t, keepid_t, hs_s, labels_s, prev_mask,
              T=2.0, loss_weight=1.0):
    sel = torch.zeros_like(labels_s)
    for lab in torch.nonzero(prev_mask).flatten().tolist():
        sel[labels_s == lab] = 1
    id_pred = torch.nonzero(sel).squeeze(1)                     # :672
    total = 0
    for sp in range(len(feats_s)):                              # :677
        fp, fsoft = feats_s[sp], feats_t[sp]
        N, C, H, W = fp.shape
        mask = torch.zeros((N, C, H, W), dtype=fp.dtype)
        idx = 0
        for i in range(N):
            bx = boxes[i]
            nb = torch.ones_like(bx)
            nb[:, 0] = bx[:, 0] / img_shapes[i][1] * W          # :688-691 (un-padded img_shape)
            nb[:, 2] = bx[:, 2] / img_shapes[i][1] * W
            nb[:, 1] = bx[:, 1] / img_shapes[i][0] * H
            nb[:, 3] = bx[:, 3] / img_shapes[i][0] * H
            wmin, wmax = torch.floor(nb[:, 0]).int(), torch.ceil(nb[:, 2]).int()
            hmin, hmax = torch.floor(nb[:, 1]).int(), torch.ceil(nb[:, 3]).int()
            for j in range(len(bx)):                            # :701-707, later boxes overwrite
                out_mask = hs_t[keepid_t[idx]] - hs_s[id_pred[idx]]
                mask[i, :, hmin[j]:hmax[j], wmin[j]:wmax[j]] = \
                    out_mask.abs().softmax(dim=0).unsqueeze(1).unsqueeze(2).repeat(
                        1, int(hmax[j] - hmin[j]), int(wmax[j] - wmin[j]))
                idx += 1
            fea_student = fp[i] * mask[i]                       # :709 (named fg_fea_t there)
            fea_teacher = fsoft[i] * mask[i]                    # :710
            total = total + kd_kl_sum(fea_teacher, fea_student, T, loss_weight)   # :715 pred=teacher side
    return total / len(boxes)                                   # :716-717
