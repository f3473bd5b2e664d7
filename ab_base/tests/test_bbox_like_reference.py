"""The reference's behavioural test of ``bbox_overlaps`` (/root/reference/tests/test_metrics/test_box_overlap.py:11-106),
applied to ``dskd_amd.bbox.bbox_overlaps`` (restating mmdet/core/bbox/iou_calculators/iou2d_calculator.py:190-261):
aligned / pairwise, batch dimensions, empty inputs, the known GIoU answers, 'iof'."""
import numpy as np
import pytest
import torch

from dskd_amd.bbox import bbox_overlaps


def _construct_bbox(g, num_bbox=None):
    img_h, img_w = int(torch.randint(3, 1000, (1,), generator=g)), int(torch.randint(3, 1000, (1,), generator=g))
    if num_bbox is None:
        num_bbox = int(torch.randint(1, 10, (1,), generator=g))
    x1y1 = torch.rand((num_bbox, 2), generator=g)
    x2y2 = torch.max(torch.rand((num_bbox, 2), generator=g), x1y1)
    bboxes = torch.cat((x1y1, x2y2), -1)
    bboxes[:, 0::2] *= img_w
    bboxes[:, 1::2] *= img_h
    return bboxes, num_bbox


def test_bbox_overlaps_2d(eps=1e-7):
    g = torch.Generator().manual_seed(0)
    # aligned
    b1, n = _construct_bbox(g)
    b2, _ = _construct_bbox(g, n)
    gious = bbox_overlaps(b1, b2, "giou", True)
    assert gious.size() == (n,) and torch.all(gious >= -1) and torch.all(gious <= 1)
    # aligned, empty
    gious = bbox_overlaps(torch.empty((0, 4)), torch.empty((0, 4)), "giou", True)
    assert gious.size() == (0,)
    # aligned, batch dimensions (and the assertion when they differ)
    b1, n = _construct_bbox(g)
    b2, _ = _construct_bbox(g, n)
    b1 = b1.unsqueeze(0).repeat(2, 1, 1)
    with pytest.raises(AssertionError):
        bbox_overlaps(b1, b2.unsqueeze(0).repeat(3, 1, 1), "giou", True)
    b2 = b2.unsqueeze(0).repeat(2, 1, 1)
    gious = bbox_overlaps(b1, b2, "giou", True)
    assert gious.size() == (2, n) and torch.all(gious >= -1) and torch.all(gious <= 1)
    gious = bbox_overlaps(b1.unsqueeze(0).repeat(2, 1, 1, 1), b2.unsqueeze(0).repeat(2, 1, 1, 1), "giou", True)
    assert gious.size() == (2, 2, n)
    # pairwise
    b1, n1 = _construct_bbox(g)
    b2, n2 = _construct_bbox(g)
    gious = bbox_overlaps(b1, b2, "giou")
    assert gious.size() == (n1, n2) and torch.all(gious >= -1) and torch.all(gious <= 1)
    b1, b2 = b1.unsqueeze(0).repeat(2, 1, 1), b2.unsqueeze(0).repeat(2, 1, 1)
    assert bbox_overlaps(b1, b2, "giou").size() == (2, n1, n2)
    assert bbox_overlaps(b1.unsqueeze(0), b2.unsqueeze(0), "giou").size() == (1, 2, n1, n2)
    # pairwise, empty first set
    gious = bbox_overlaps(torch.empty(1, 2, 0, 4), b2.unsqueeze(0), "giou")
    assert gious.size() == (1, 2, 0, n2)
    # the known answers of the official implementation (four decimals)
    b1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    b2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    gious = bbox_overlaps(b1, b2, "giou", is_aligned=True, eps=eps).numpy().round(4)
    assert np.allclose(gious, np.array([0.5000, -0.0500, -0.8214]), rtol=0, atol=eps)
    # 'iof'
    ious = bbox_overlaps(b1, b2, "iof", is_aligned=True, eps=eps)
    assert ious.size() == (3,) and torch.all(ious >= -1) and torch.all(ious <= 1)
    ious = bbox_overlaps(b1, b2, "iof", eps=eps)
    assert ious.size() == (3, 3) and torch.all(ious >= -1) and torch.all(ious <= 1)
