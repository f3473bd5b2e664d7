"""Pins the oracle itself: LSAP restatement vs the installed scipy (the function the reference
calls) and frozen vectors; MSDA restatement vs its scalar form and the independent copy in
the installed ``transformers`` package (the reference holds no fixture for ext-mmcv's op)."""
import os

import numpy as np
import pytest
import torch

from oracle import msda_ref
from oracle.lsap_ref import linear_sum_assignment as oracle_lsa

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases(rng, n):
    for t in range(n):
        nr, nc = rng.integers(1, 13), rng.integers(1, 13)
        kind = t % 4
        if kind == 0:
            c = rng.integers(0, 4, size=(nr, nc))
        elif kind == 1:
            c = np.round(rng.normal(size=(nr, nc)), 1)
        elif kind == 2:
            c = rng.random((nr, nc))
            c[:, rng.integers(0, nc)] = c[:, 0]
        else:
            c = rng.random((nr, nc))
        yield c.astype(np.float32)


def test_lsap_oracle_vs_scipy_fuzz():
    from scipy.optimize import linear_sum_assignment as sp
    rng = np.random.default_rng(0)
    n = 0
    for c in _cases(rng, 1500):
        a, b = sp(c), oracle_lsa(c)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), c
        n += 1
    for Gn in (1, 5, 17, 60, 100, 300, 310):
        for k in range(6):
            c = (rng.random((300, Gn)) if k % 2 else rng.integers(0, 6, size=(300, Gn))).astype(np.float32)
            a, b = sp(c), oracle_lsa(c)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert n == 1500


def test_lsap_oracle_error_behaviour():
    c = np.random.default_rng(1).random((5, 7)).astype(np.float32)
    c[3, :] = np.inf
    with pytest.raises(ValueError, match="infeasible"):
        oracle_lsa(c)
    c = np.ones((3, 3), dtype=np.float32)
    c[1, 1] = np.nan
    with pytest.raises(ValueError, match="invalid numeric"):
        oracle_lsa(c)
    c[1, 1] = -np.inf
    with pytest.raises(ValueError, match="invalid numeric"):
        oracle_lsa(c)
    r, cc = oracle_lsa(np.zeros((0, 4), dtype=np.float32))
    assert len(r) == 0 and len(cc) == 0


def test_lsap_frozen_vectors():
    """Vectors frozen from scipy 1.15.3 in the build container (tests/golden/lsap_cases.npz)."""
    z = np.load(os.path.join(G, "lsap_cases.npz"))
    n = int(z["n"])
    for k in range(n):
        r, c = oracle_lsa(z[f"cost{k}"])
        assert np.array_equal(r, z[f"row{k}"]) and np.array_equal(c, z[f"col{k}"])


def test_msda_oracle_forms_agree():
    torch.manual_seed(0)
    shapes = [(6, 7), (3, 4), (2, 2), (1, 1)]
    Nv = sum(h * w for h, w in shapes)
    value = torch.randn(2, Nv, 8, 32)
    loc = torch.rand(2, 5, 8, 4, 4, 2) * 1.4 - 0.2
    attn = torch.softmax(torch.randn(2, 5, 8, 16), -1).view(2, 5, 8, 4, 4)
    a = msda_ref.msda_grid_sample(value, shapes, loc, attn)
    b = msda_ref.msda_scalar(value.numpy(), shapes, loc.numpy(), attn.numpy())
    assert np.abs(a.numpy() - b).max() < 2e-6


def test_msda_oracle_vs_transformers_copy():
    mod = pytest.importorskip("transformers.models.deformable_detr.modeling_deformable_detr")
    torch.manual_seed(1)
    shapes = [(5, 6), (3, 3)]
    Nv = sum(h * w for h, w in shapes)
    value = torch.randn(1, Nv, 8, 32)
    loc = torch.rand(1, 7, 8, 2, 4, 2) * 1.2 - 0.1
    attn = torch.softmax(torch.randn(1, 7, 8, 8), -1).view(1, 7, 8, 2, 4)
    a = msda_ref.msda_grid_sample(value, shapes, loc, attn)
    m = mod.MultiScaleDeformableAttention()
    lsi = torch.tensor([0, 30])
    c = m(value, torch.tensor(shapes), shapes, lsi, loc, attn, 64)
    torch.testing.assert_close(a, c, atol=1e-6, rtol=1e-6)


def test_msda_oracle_gradcheck_float64():
    torch.manual_seed(2)
    shapes = [(4, 5), (2, 3)]
    Nv = sum(h * w for h, w in shapes)
    value = torch.randn(1, Nv, 2, 4, dtype=torch.float64, requires_grad=True)
    loc = (torch.rand(1, 3, 2, 2, 2, 2, dtype=torch.float64) * 0.8 + 0.1).requires_grad_(True)
    attn = torch.rand(1, 3, 2, 2, 2, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda v, l, a: msda_ref.msda_grid_sample(v, shapes, l, a), (value, loc, attn),
                                    eps=1e-6, atol=1e-5)
