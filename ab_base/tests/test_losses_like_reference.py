"""The reference's behavioural loss tests (/root/reference/tests/test_models/test_loss.py:17-89), applied to the loss
classes this path registers under the reference's names: zero weights, the ``reduction_override`` contract, empty
inputs, ``avg_factor`` with 'sum' refused."""
import pytest
import torch

from dskd_amd import losses as L


def test_giou_loss_zeros_weight():                                   # test_loss.py:17-25
    pred, target = torch.rand((10, 4)), torch.rand((10, 4))
    assert L.GIoULoss()(pred, target, torch.zeros(10)) == 0.


@pytest.mark.parametrize("loss_class", [L.DistributionFocalLoss, L.MSELoss, L.GIoULoss, L.L1Loss, L.QualityFocalLoss,
                                        L.SmoothL1Loss, L.KnowledgeDistillationKLDivLoss])
def test_loss_with_reduction_override(loss_class):                    # test_loss.py:28-45
    pred, target = torch.rand((10, 4)), (torch.rand((10, 4)),)
    with pytest.raises(AssertionError):     # only None, 'none', 'mean', 'sum' are allowed
        loss_class()(pred, target, None, reduction_override=True)


@pytest.mark.parametrize("loss_class", [L.GIoULoss, L.MSELoss, L.L1Loss, L.SmoothL1Loss])
@pytest.mark.parametrize("input_shape", [(10, 4), (0, 4)])
def test_regression_losses(loss_class, input_shape):                  # test_loss.py:48-86
    pred, target, weight = torch.rand(input_shape), torch.rand(input_shape), torch.rand(input_shape)
    assert isinstance(loss_class()(pred, target), torch.Tensor)
    assert isinstance(loss_class()(pred, target, weight), torch.Tensor)
    assert isinstance(loss_class()(pred, target, reduction_override="mean"), torch.Tensor)
    assert isinstance(loss_class()(pred, target, avg_factor=10), torch.Tensor)
    with pytest.raises(ValueError):         # avg_factor only with reduction None / 'none' / 'mean'
        loss_class()(pred, target, avg_factor=10, reduction_override="sum")
    for reduction_override in [None, "none", "mean"]:
        assert isinstance(loss_class()(pred, target, avg_factor=10, reduction_override=reduction_override), torch.Tensor)
