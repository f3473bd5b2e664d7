"""Parity of every HIP kernel against the CPU oracle, through the C-ABI (-m gpu)."""
import copy

import numpy as np
import pytest
import torch

from dskd_amd import native
from oracle import assign_ref, dskd_losses_ref, msda_ref
from oracle.lsap_ref import linear_sum_assignment as oracle_lsa

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES_SMALL = [(12, 17), (6, 9), (3, 5), (2, 3)]
SHAPES_FULL = [(100, 167), (50, 84), (25, 42), (13, 21)]   # 800x1333 input, BASELINE.json


def _msda_inputs(shapes, B, Nq, seed, spread=1.3, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    Nv = sum(h * w for h, w in shapes)
    value = torch.randn(B, Nv, 8, 32, generator=g)
    loc = torch.rand(B, Nq, 8, len(shapes), 4, 2, generator=g) * spread - (spread - 1) / 2
    attn = torch.softmax(torch.randn(B, Nq, 8, len(shapes) * 4, generator=g), -1).view(B, Nq, 8, len(shapes), 4)
    return value.to(dtype), loc, attn


@pytest.mark.parametrize("B,Nq,seed", [(1, 1, 0), (2, 37, 1), (3, 300, 2), (1, 431, 3)])
def test_msda_fwd_f32_small(B, Nq, seed):
    value, loc, attn = _msda_inputs(SHAPES_SMALL, B, Nq, seed)
    ref = msda_ref.msda_grid_sample(value, SHAPES_SMALL, loc, attn)
    out = native.msda_forward_raw(value.to(DEV), SHAPES_SMALL, loc.to(DEV), attn.to(DEV)).cpu()
    torch.testing.assert_close(out, ref, atol=1e-5, rtol=1e-4)


def test_msda_fwd_edge_locations():
    """Exactly on borders, far outside, NaN: zero padding like grid_sample."""
    value, loc, attn = _msda_inputs(SHAPES_SMALL, 1, 8, 5)
    loc[0, 0] = 0.0
    loc[0, 1] = 1.0
    loc[0, 2] = -3.0
    loc[0, 3] = 7.5
    loc[0, 4, :, :, :, 0] = 0.5 / 17
    loc[0, 5, :, 0] = torch.tensor([0.0, 1.0])
    ref = msda_ref.msda_grid_sample(value, SHAPES_SMALL, loc, attn)
    out = native.msda_forward_raw(value.to(DEV), SHAPES_SMALL, loc.to(DEV), attn.to(DEV)).cpu()
    torch.testing.assert_close(out, ref, atol=1e-5, rtol=1e-4)
    # NaN location contributes nothing in the HIP op (mmcv's bounds test rejects it)
    loc2 = loc.clone()
    loc2[0, 6] = float("nan")
    out2 = native.msda_forward_raw(value.to(DEV), SHAPES_SMALL, loc2.to(DEV), attn.to(DEV)).cpu()
    assert torch.equal(out2[0, 6], torch.zeros(256))
    torch.testing.assert_close(out2[0, :6], ref[0, :6], atol=1e-5, rtol=1e-4)


def test_msda_fwd_bf16():
    value, loc, attn = _msda_inputs(SHAPES_SMALL, 2, 301, 7)
    vb = value.to(torch.bfloat16)
    ref = msda_ref.msda_grid_sample(vb.float(), SHAPES_SMALL, loc, attn)
    out = native.msda_forward_raw(vb.to(DEV), SHAPES_SMALL, loc.to(DEV), attn.to(DEV)).cpu()
    assert out.dtype == torch.bfloat16
    # output rounding to bf16: 2^-8 relative
    torch.testing.assert_close(out.float(), ref, atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_msda_bwd_small(dtype):
    value, loc, attn = _msda_inputs(SHAPES_SMALL, 2, 203, 11, dtype=dtype)
    g = torch.Generator().manual_seed(12)
    go = torch.randn(2, 203, 256, generator=g).to(dtype)
    v = value.float().requires_grad_(True)
    l = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    msda_ref.msda_grid_sample(v, SHAPES_SMALL, l, a).backward(go.float())
    gv, gl, ga = native.msda_backward_raw(value.to(DEV), SHAPES_SMALL, loc.to(DEV), attn.to(DEV), go.to(DEV))
    tol = dict(atol=1e-4, rtol=1e-3) if dtype == torch.float32 else dict(atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(gv.cpu(), v.grad, **tol)
    torch.testing.assert_close(ga.cpu(), a.grad, **tol)
    torch.testing.assert_close(gl.cpu(), l.grad, atol=tol["atol"] * 20, rtol=tol["rtol"])


def test_msda_autograd_function():
    value, loc, attn = _msda_inputs(SHAPES_SMALL, 1, 50, 13)
    v = value.to(DEV).requires_grad_(True)
    l = loc.to(DEV).requires_grad_(True)
    a = attn.to(DEV).requires_grad_(True)
    out = native.ms_deform_attn(v, SHAPES_SMALL, l, a)
    out.square().sum().backward()
    vr, lr, ar = value.clone().requires_grad_(True), loc.clone().requires_grad_(True), attn.clone().requires_grad_(True)
    msda_ref.msda_grid_sample(vr, SHAPES_SMALL, lr, ar).square().sum().backward()
    torch.testing.assert_close(v.grad.cpu(), vr.grad, atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(a.grad.cpu(), ar.grad, atol=1e-4, rtol=1e-3)


def test_msda_full_size_properties():
    """BASELINE size (Nq = Nv = 22223): linearity in value and in attn, decoder-size parity."""
    B = 2
    value, loc, attn = _msda_inputs(SHAPES_FULL, B, 22223, 21, spread=1.05)
    vd, ld, ad = value.to(DEV), loc.to(DEV), attn.to(DEV)
    o1 = native.msda_forward_raw(vd, SHAPES_FULL, ld, ad)
    o2 = native.msda_forward_raw(vd * 2.0, SHAPES_FULL, ld, ad)
    torch.testing.assert_close(o2, o1 * 2.0, atol=1e-5, rtol=1e-5)
    o3 = native.msda_forward_raw(vd, SHAPES_FULL, ld, ad * 0.5)
    torch.testing.assert_close(o3, o1 * 0.5, atol=1e-5, rtol=1e-5)
    # constant value + weights summing to 1 with all samples inside -> constant output
    loc_in = (loc * 0.5 + 0.25).to(DEV)
    oc = native.msda_forward_raw(torch.ones_like(vd), SHAPES_FULL, loc_in, ad)
    torch.testing.assert_close(oc, torch.ones_like(oc), atol=1e-5, rtol=1e-5)
    # oracle on a slice of queries at full value size
    sl = slice(5000, 5600)
    ref = msda_ref.msda_grid_sample(value, SHAPES_FULL, loc[:, sl], attn[:, sl])
    torch.testing.assert_close(o1[:, sl].cpu(), ref, atol=1e-5, rtol=1e-4)
    # backward: sum of grad_value equals sum over valid samples of attn*grad (conservation)
    go = torch.ones(B, 22223, 256, device=DEV)
    gv, gl, ga = native.msda_backward_raw(torch.ones_like(vd), SHAPES_FULL, loc_in, ad, go)
    torch.testing.assert_close(gv.sum(), torch.tensor(float(B * 22223 * 256), device=DEV), rtol=1e-4, atol=1.0)
    # d out / d attn with value == 1 is 32 (channels) for inside samples
    torch.testing.assert_close(ga, torch.full_like(ga, 32.0), atol=1e-3, rtol=1e-4)


def _encoder_like_inputs(shapes, B, seed, sigma_px, dtype=torch.float32):
    """Queries = pixels (Nq == Nv): reference point = pixel centre, offsets ~ N(0, sigma_px) pixels."""
    g = torch.Generator().manual_seed(seed)
    Nv = sum(h * w for h, w in shapes)
    value = torch.randn(B, Nv, 8, 32, generator=g).to(dtype)
    pts = []
    for (H, W) in shapes:
        ys, xs = torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")
        pts.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(pts, 0)[None].expand(B, -1, -1)
    off = torch.randn(B, Nv, 8, len(shapes), 4, 2, generator=g) * sigma_px
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32).view(1, 1, 1, len(shapes), 1, 2)
    loc = (ref[:, :, None, None, None, :] + off / norm).contiguous()
    attn = torch.softmax(torch.randn(B, Nv, 8, len(shapes) * 4, generator=g), -1).view(B, Nv, 8, len(shapes), 4)
    go = torch.randn(B, Nv, 256, generator=g).to(dtype)
    return value, loc, attn, go


@pytest.mark.parametrize("shapes,sigma", [([(25, 42), (13, 21), (7, 11), (4, 6)], 2.0),
                                          ([(40, 70), (20, 35), (10, 18), (5, 9)], 12.0),   # many window misses
                                          ([(64, 96), (32, 48), (16, 24), (8, 12)], 3.0),   # 32-px regions: 8-wave variant
                                          ([(33, 47), (17, 24)], 3.0), ([(9, 5)], 1.0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_msda_bwd_windowed_encoder_shape(shapes, sigma, dtype):
    """Nq == Nv takes the windowed LDS-accumulation kernel: parity with autograd of the oracle,
    including offsets far beyond the window margin (global-atomic fallback)."""
    value, loc, attn, go = _encoder_like_inputs(shapes, 2, 31, sigma, dtype)
    v = value.float().requires_grad_(True)
    l = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    msda_ref.msda_grid_sample(v, shapes, l, a).backward(go.float())
    gv, gl, ga = native.msda_backward_raw(value.to(DEV), shapes, loc.to(DEV), attn.to(DEV), go.to(DEV))
    tol = dict(atol=2e-4, rtol=1e-3) if dtype == torch.float32 else dict(atol=4e-3, rtol=4e-3)
    torch.testing.assert_close(gv.cpu(), v.grad, **tol)
    torch.testing.assert_close(ga.cpu(), a.grad, **tol)


def test_msda_bwd_matrix_core_unusual_attention_weights():
    """csrc/msda_mm.hip accumulates the S image in 16-bit fixed point, which assumes the four attention weights of a
    (query, head, level) are >= 0 and sum to <= 1.5 (softmax outputs are); every other quad of samples must take the
    per-lane path: weights that are negative, large, NaN-free but unnormalised -- against the oracle on the rounded
    inputs, bf16 tolerances."""
    shapes = [(25, 42), (13, 21), (7, 11), (4, 6)]
    value, loc, attn, go = _encoder_like_inputs(shapes, 2, 91, 2.0, torch.bfloat16)
    g = torch.Generator().manual_seed(92)
    attn = attn.clone()
    Nq = attn.shape[1]
    attn[:, : Nq // 3] = torch.randn(attn[:, : Nq // 3].shape, generator=g)                 # signed
    attn[:, Nq // 3: 2 * Nq // 3] = torch.rand(attn[:, Nq // 3: 2 * Nq // 3].shape, generator=g) * 3.0   # sums far above 1.5
    v = value.float().requires_grad_(True)
    l = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    msda_ref.msda_grid_sample(v, shapes, l, a).backward(go.float())
    gv, gl, ga = native.msda_backward_raw(value.to(DEV), shapes, loc.to(DEV), attn.to(DEV), go.to(DEV))
    scale = float(v.grad.abs().max())
    torch.testing.assert_close(gv.cpu(), v.grad, atol=4e-3 * max(scale, 1.0), rtol=4e-3)
    torch.testing.assert_close(ga.cpu(), a.grad, atol=4e-3, rtol=4e-3)
    torch.testing.assert_close(gl.cpu(), l.grad, atol=4e-3 * 20 * max(float(l.grad.abs().max()) / 50.0, 1.0), rtol=4e-3)


def test_msda_bwd_workspace_entry_matches_plain_entry_full_size():
    """BASELINE size, f32: the workspace entry point (dskd_msda_bwd_ws: pull on level 0) against the plain one
    (dskd_msda_bwd: windowed LDS accumulation on every level); grad_loc / grad_attn come from the same gather kernel."""
    value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, 2, 41, 2.5)
    args = (value.to(DEV), SHAPES_FULL, loc.to(DEV), attn.to(DEV), go.to(DEV))
    gv2, gl2, ga2 = native.msda_backward_raw(*args)
    gv1, gl1, ga1 = native.msda_backward_raw(*args, use_workspace=False)
    torch.testing.assert_close(gv2, gv1, atol=2e-4, rtol=1e-3)
    assert torch.equal(gl2, gl1) and torch.equal(ga2, ga1)
    # conservation: sum over value rows of grad_value == sum_q sum_inside-samples attn * grad_out
    torch.testing.assert_close(gv2.sum(dim=(1,)), gv1.sum(dim=(1,)), atol=5e-2, rtol=1e-3)


def _bwd_oracle(value, shapes, loc, attn, go):
    v = value.float().requires_grad_(True)
    l = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    msda_ref.msda_grid_sample(v, shapes, l.nan_to_num(nan=-5.0), a).backward(go.float())
    return v.grad, l.grad, a.grad


@pytest.mark.parametrize("levels", ["01", "0123", "1", "023"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_msda_bwd_pull_levels_vs_oracle(monkeypatch, levels, dtype):
    """The tiled pull kernel (msda_pull.hip) on any subset of the levels, the windowed kernels on the rest: grad_value
    against autograd of the oracle, with border / rejected / NaN / far-away sampling locations in the batch (the
    far ones go through the stray list and the apply kernel), an output buffer full of garbage (it is overwritten, not
    accumulated into) and the workspace header left zeroed."""
    shapes = [(40, 70), (20, 35), (10, 18), (5, 9)]
    value, loc, attn, go = _encoder_like_inputs(shapes, 2, 77, 3.0, dtype)
    loc[0, :7] = torch.tensor([-0.2, 0.0, 0.5, 1.0, 1.3, float("nan"), 0.999]).view(7, 1, 1, 1, 1)
    loc[1, 100:140] += 0.37                                   # far beyond every candidate margin
    gv_ref, _, _ = _bwd_oracle(value, shapes, loc, attn, go)
    monkeypatch.setenv("DSKD_MSDA_PULL_LEVELS", levels)
    lib = native.load()
    B, Nv = value.shape[:2]
    ss, ls, _ = native._geom(shapes)
    vd, ld, ad, gd = value.to(DEV), loc.to(DEV), attn.to(DEV), go.to(DEV)
    need = int(lib.dskd_msda_bwd_workspace(B, Nv, Nv, 8, 4, 4))
    ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
    dt = native.DTYPE_F32 if dtype == torch.float32 else native.DTYPE_BF16
    outs = []
    for _ in range(2):                                        # second call: same workspace, header must have been reset
        gv = torch.full((B, Nv, 8, 32), float("nan"), device=DEV)
        gl, ga = torch.empty_like(ld), torch.empty_like(ad)
        rc = lib.dskd_msda_bwd_ws(vd.data_ptr(), ss, ls, ld.data_ptr(), ad.data_ptr(), gd.data_ptr(), gv.data_ptr(),
                                  gl.data_ptr(), ga.data_ptr(), B, Nv, Nv, 8, 32, 4, 4, dt, ws.data_ptr(), need,
                                  torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.dskd_last_error()
        torch.cuda.synchronize()
        assert int(ws[:64].to(torch.int32).sum()) == 0
        outs.append(gv)
    tol = dict(atol=2e-4, rtol=1e-3) if dtype == torch.float32 else dict(atol=4e-3, rtol=4e-3)
    for gv in outs:
        torch.testing.assert_close(gv.cpu(), gv_ref, **tol)


def test_msda_bwd_pull_stray_list_overflow():
    """A workspace with room for ONE stray entry while thousands of samples leave their candidate ranges: the apply
    kernel ignores the list, walks every tile again and adds the strays directly -- same result."""
    shapes = [(40, 70), (20, 35), (10, 18), (5, 9)]
    value, loc, attn, go = _encoder_like_inputs(shapes, 2, 79, 14.0)
    gv_ref, _, _ = _bwd_oracle(value, shapes, loc, attn, go)
    lib = native.load()
    B, Nv = value.shape[:2]
    ss, ls, _ = native._geom(shapes)
    vd, ld, ad, gd = value.to(DEV), loc.to(DEV), attn.to(DEV), go.to(DEV)
    ws = torch.zeros(64 + 16, dtype=torch.uint8, device=DEV)
    for _ in range(2):
        gv = torch.full((B, Nv, 8, 32), float("nan"), device=DEV)
        gl, ga = torch.empty_like(ld), torch.empty_like(ad)
        rc = lib.dskd_msda_bwd_ws(vd.data_ptr(), ss, ls, ld.data_ptr(), ad.data_ptr(), gd.data_ptr(), gv.data_ptr(),
                                  gl.data_ptr(), ga.data_ptr(), B, Nv, Nv, 8, 32, 4, 4, native.DTYPE_F32, ws.data_ptr(),
                                  ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.dskd_last_error()
        torch.cuda.synchronize()
        assert int(ws[:64].to(torch.int32).sum()) == 0
        torch.testing.assert_close(gv.cpu(), gv_ref, atol=2e-4, rtol=1e-3)
    # argument checks
    assert lib.dskd_msda_bwd_ws(vd.data_ptr(), ss, ls, ld.data_ptr(), ad.data_ptr(), gd.data_ptr(), gv.data_ptr(),
                                gl.data_ptr(), ga.data_ptr(), B, Nv, Nv, 8, 32, 4, 4, native.DTYPE_F32, None, 0,
                                torch.cuda.current_stream().cuda_stream) == -1
    assert lib.dskd_msda_bwd_workspace(2, Nv, Nv, 8, 4, 4) >= 64 + 16 * 4096


def test_msda_bwd_heavy_tailed_gradient():
    """One grad_out element 1e4 times the rest.  The pull kernel accumulates level 0 (75 % of the
    value rows) in f32 registers, so a cell's error is relative to the cell: its rows keep the usual tolerance next to
    the spike.  Levels 1-3 accumulate in 32-bit fixed point scaled by the REGION's bound max|grad_out| * sum|attn|
    (msda.hip), quantum = bound / 1e9 per contribution: inside the spike's region (<= 32 x 32 level-0 pixels) the
    absolute error of a cell is a few quanta times sqrt(contributions) -- stated and checked here -- and every other
    region keeps the usual tolerance."""
    value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, 1, 47, 2.5)
    spike_q = 50 * 167 + 80                                   # a level-0 query in the middle of the image
    go[0, spike_q, 37] = 1.0e4
    gv_ref, _, _ = _bwd_oracle(value, SHAPES_FULL, loc, attn, go)
    gv, _, _ = native.msda_backward_raw(value.to(DEV), SHAPES_FULL, loc.to(DEV), attn.to(DEV), go.to(DEV))
    gv = gv.cpu()
    n0 = 100 * 167
    torch.testing.assert_close(gv[:, :n0], gv_ref[:, :n0], atol=2e-4, rtol=1e-3)
    # fixed-point levels: bound <= 1e4 * sum|attn| over the region's ~700 queries (4..8 of 16 samples each: ~250) = 2.5e6,
    # quantum 2.5e-3; a level-1 / 2 / 3 cell sums ~85 / ~340 / ~1 300 contributions -> error of a few 1e-2 next to the spike
    err = (gv[:, n0:] - gv_ref[:, n0:]).abs()
    assert float(err.max()) < 0.5, float(err.max())
    ok = err <= 2e-4 + 1e-3 * gv_ref[:, n0:].abs()
    assert float(ok.float().mean()) > 0.85, float(ok.float().mean())     # every cell outside the spike's region


def test_msda_bwd_pull_matches_windowed_full_size():
    """BASELINE size, B=2, bf16 (the benchmark's mode): the workspace entry point (pull on level 0; levels 1-3 -- grad_value
    and their samples' grad_loc / grad_attn -- on the matrix-core kernel csrc/msda_mm.hip) against the plain entry point
    (dskd_msda_bwd: all-windowed fixed-point grad_value, every level's dot products in the gather kernel); the dot
    products sum the same 32 products in a different order: equal to f32 rounding, and bit-identical run to run."""
    value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, 2, 49, 2.5, torch.bfloat16)
    args = (value.to(DEV), SHAPES_FULL, loc.to(DEV), attn.to(DEV), go.to(DEV))
    gv2, gl2, ga2 = native.msda_backward_raw(*args)
    gv3, gl3, ga3 = native.msda_backward_raw(*args)
    assert torch.equal(gl2, gl3) and torch.equal(ga2, ga3)
    gv1, gl1, ga1 = native.msda_backward_raw(*args, use_workspace=False)
    torch.testing.assert_close(gv2, gv1, atol=4e-3, rtol=4e-3)
    for a, r in ((gl2, gl1), (ga2, ga1)):
        assert float((a - r).abs().max()) <= 2e-6 * float(r.abs().max()), (float((a - r).abs().max()), float(r.abs().max()))
    # level 0 comes from the same gather kernel with the same arithmetic in both (levels 1-3: matrix-core kernel, f32
    # accumulation of the same exact products in another order)
    assert torch.equal(gl2[..., :1, :, :], gl1[..., :1, :, :]) and torch.equal(ga2[..., :1, :], ga1[..., :1, :])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_msda_bwd_full_size_vs_oracle(dtype):
    """BASELINE size (100x167 ... 13x21, Nq = Nv = 22 223), one image: grad_value, grad_loc and grad_attn of the
    encoder-shape backward against autograd of the oracle on ALL queries and ALL value rows (the oracle needs ~3 s
    for this on the host)."""
    value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, 1, 43, 2.5, dtype)
    v = value.float().requires_grad_(True)
    l = loc.clone().requires_grad_(True)
    a = attn.clone().requires_grad_(True)
    msda_ref.msda_grid_sample(v, SHAPES_FULL, l, a).backward(go.float())
    gv, gl, ga = native.msda_backward_raw(value.to(DEV), SHAPES_FULL, loc.to(DEV), attn.to(DEV), go.to(DEV))
    tol = dict(atol=2e-4, rtol=1e-3) if dtype == torch.float32 else dict(atol=4e-3, rtol=4e-3)
    torch.testing.assert_close(gv.cpu(), v.grad, **tol)
    torch.testing.assert_close(ga.cpu(), a.grad, **tol)
    torch.testing.assert_close(gl.cpu(), l.grad, atol=tol["atol"] * 20, rtol=tol["rtol"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_msda_prepare_vs_module_chain(dtype, oracle_checker):
    """softmax + location prologue kernel (and its backward) against the module's PyTorch chain."""
    g = torch.Generator().manual_seed(17)
    B, Nq = 2, 437
    shapes = [(25, 42), (13, 21), (7, 11), (4, 6)]
    both = (torch.randn(B, Nq, 384, generator=g) * 2).to(dtype)
    ref = torch.rand(B, Nq, 4, 2, generator=g)
    gl = torch.randn(B, Nq, 8, 4, 4, 2, generator=g)
    ga = torch.randn(B, Nq, 8, 4, 4, generator=g)
    bc = both.clone().requires_grad_(True)
    rc = ref.clone().requires_grad_(True)          # the decoder's reference points are differentiable
    loc_r, attn_r = oracle_checker.msda_prepare(bc, rc, shapes, 8, 4, 4)
    (loc_r * gl).sum().backward(retain_graph=True)
    g1 = bc.grad.clone()
    bc.grad = None
    (attn_r * ga).sum().backward()
    g2 = bc.grad.clone()
    bd = both.to(DEV).requires_grad_(True)
    rd = ref.to(DEV).requires_grad_(True)
    loc, attn = native.msda_prepare(bd, rd, shapes, 8, 4, 4)
    torch.testing.assert_close(loc.cpu(), loc_r.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(attn.cpu(), attn_r.detach(), rtol=1e-5, atol=1e-7)
    ((loc * gl.to(DEV)).sum() + (attn * ga.to(DEV)).sum()).backward()
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(bd.grad.float().cpu(), (g1 + g2).float(), **tol)
    torch.testing.assert_close(rd.grad.cpu(), rc.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Nq", [None, 301])          # None: encoder shape (queries == pixels)
def test_msda_fused_prologue_is_bit_identical(dtype, Nq):
    """No-grad forward with the module prologue folded in (dskd_msda_fwd_fused) against prologue
    kernel + sampling kernel: identical bits, and close to the oracle chain."""
    g = torch.Generator().manual_seed(23)
    shapes = [(25, 42), (13, 21), (7, 11), (4, 6)]
    Nv = sum(h * w for h, w in shapes)
    B = 2
    nq = Nv if Nq is None else Nq
    value = torch.randn(B, Nv, 8, 32, generator=g).to(dtype).to(DEV)
    both = (torch.randn(B, nq, 384, generator=g) * 2).to(dtype).to(DEV)
    ref = torch.rand(B, nq, 4, 2, generator=g).to(DEV)
    with torch.no_grad():
        loc, attn = native.msda_prepare(both, ref, shapes, 8, 4, 4)
        two = native.ms_deform_attn(value, shapes, loc, attn)
        one = native.ms_deform_attn_fused(value, shapes, both, ref, 4, 4)
    assert torch.equal(one, two)
    want = msda_ref.msda_grid_sample(value.float().cpu(), shapes, loc.cpu(), attn.cpu())
    tol = dict(atol=1e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=3e-2, rtol=2e-2)
    torch.testing.assert_close(one.float().cpu(), want, **tol)


_WIN_CASES = [([(25, 42), (13, 21), (7, 11), (4, 6)], 2, 2.0),
              ([(25, 42), (13, 21), (7, 11), (4, 6)], 2, 12.0),     # most samples leave the windows
              ([(40, 70), (20, 35), (10, 18), (5, 9)], 3, 6.0),
              ([(17, 16), (9, 8), (5, 4), (3, 2)], 2, 1.0),
              (SHAPES_FULL, 1, 2.5)]


@pytest.mark.parametrize("shapes,B,sigma", _WIN_CASES)
def test_msda_windowed_forward_is_bit_identical(shapes, B, sigma):
    """Windowed forward of the encoder shape (levels 2+3 of one head staged in LDS per 16 x 16-pixel region, the fine
    levels and out-of-window samples through buffer loads): same weights, same sample order, same FMAs as the plain
    kernel, so the bf16 output must be identical -- borders, rejected and far samples included.  The plain kernel is
    reached with the same tensors minus the last query (Nq != Nv is not the encoder shape)."""
    value, loc, attn, _ = _encoder_like_inputs(shapes, B, 71, sigma, torch.bfloat16)
    loc[0, :7] = torch.tensor([-0.2, 0.0, 0.5, 1.0, 1.3, float("nan"), 0.999]).view(7, 1, 1, 1, 1)
    vd, ld, ad = value.to(DEV), loc.to(DEV), attn.to(DEV)
    win = native.msda_forward_raw(vd, shapes, ld, ad)
    plain = native.msda_forward_raw(vd, shapes, ld[:, :-1].contiguous(), ad[:, :-1].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(plain, win[:, :-1])
    # and the result is the oracle's (fp32 evaluation on the rounded inputs)
    want = msda_ref.msda_grid_sample(value.float(), shapes, loc.nan_to_num(nan=-5.0), attn)
    torch.testing.assert_close(win.float().cpu(), want, atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("shapes,B,sigma", _WIN_CASES)
def test_msda_windowed_gather_matches_plain_kernel(shapes, B, sigma):
    """grad_loc / grad_attn of the encoder shape (bf16) against the plain gather kernel (reached with the last query
    dropped): level 0 comes from msda_bwd_win_kernel -- same channels per lane, same DPP reduction, same final
    arithmetic: identical bits, borders / rejected / NaN / far samples included; levels 1-3 come from the matrix-core
    kernel (csrc/msda_mm.hip: the same exact bf16 products summed in f32 by the MFMA instead of FMA chains): equal to
    f32 rounding."""
    value, loc, attn, go = _encoder_like_inputs(shapes, B, 83, sigma, torch.bfloat16)
    loc[0, :7] = torch.tensor([-0.2, 0.0, 0.5, 1.0, 1.3, float("nan"), 0.999]).view(7, 1, 1, 1, 1)
    vd, ld, ad, gd = value.to(DEV), loc.to(DEV), attn.to(DEV), go.to(DEV)
    _, gl1, ga1 = native.msda_backward_raw(vd, shapes, ld, ad, gd)
    _, gl0, ga0 = native.msda_backward_raw(vd, shapes, ld[:, :-1].contiguous(), ad[:, :-1].contiguous(), gd[:, :-1].contiguous())
    torch.cuda.synchronize()
    gl1, ga1 = gl1[:, :-1], ga1[:, :-1]
    same = lambda a, b: torch.equal(a.nan_to_num(nan=12345.0), b.nan_to_num(nan=12345.0))      # noqa: E731
    assert same(ga1[..., :1, :], ga0[..., :1, :]) and same(gl1[..., :1, :, :], gl0[..., :1, :, :])
    for a, r in ((gl1, gl0), (ga1, ga0)):
        a, r = a.nan_to_num(nan=0.0), r.nan_to_num(nan=0.0)
        assert float((a - r).abs().max()) <= 4e-6 * float(r.abs().max()) + 1e-7


# ----------------------------------------------------------------------------- add + dropout + LayerNorm
def test_dropout_masks_change_between_graph_replays():
    """(seed, offset) are launch arguments and are frozen into a captured hipGraph; the kernels also read
    the device epoch word, so every replay draws a new mask once the host has advanced it -- for the fused add+LN
    tail and for the FFN's in-place dropout -- and the backward inside the same replay regenerates the forward's."""
    g = torch.Generator().manual_seed(3)
    rows, D, p = 512, 256, 0.25
    h = (torch.rand(rows, D, generator=g) + 1.0).to(DEV).to(torch.bfloat16).requires_grad_(True)
    res = torch.zeros(rows, D, device=DEV, dtype=torch.bfloat16)
    norm = torch.nn.LayerNorm(D).to(DEV)
    y_ffn = (torch.rand(rows, 1024, generator=g) + 1.0).to(DEV).to(torch.bfloat16)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())

    def region():
        y, _ = native.add_layer_norm(h, res, norm, p)
        (gh,) = torch.autograd.grad(y.float().sum() + (y.float() ** 2).sum(), h)
        d = native.dropout_(y_ffn.clone(), p)
        return y, gh, d
    with torch.cuda.stream(side):
        region()                                             # warm-up outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        y, gh, d = region()
    seen = []
    for _ in range(3):
        native.advance_dropout_epoch(DEV)
        graph.replay()
        torch.cuda.synchronize()
        keep_ffn = d != 0
        keep_ln = gh != 0                                    # d(h) is zero exactly where h was dropped
        frac = keep_ln.float().mean().item()
        assert abs(frac - (1 - p)) < 0.02, frac
        seen.append((keep_ln.clone(), keep_ffn.clone()))
    for i in range(3):
        for j in range(i):
            assert not torch.equal(seen[i][0], seen[j][0]) and not torch.equal(seen[i][1], seen[j][1])
    # without advancing, a replay repeats its masks (that is the failure mode the epoch word removes)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(gh != 0, seen[-1][0]) and torch.equal(d != 0, seen[-1][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("want_q", [False, True])
def test_add_layer_norm_vs_module_chain(dtype, want_q, oracle_checker):
    """identity + out -> LayerNorm -> + query_pos (ext-mmcv BaseTransformerLayer) in one launch
    each way, against the PyTorch chain in fp32 on the same (rounded) inputs.  p = 0."""
    g = torch.Generator().manual_seed(5)
    B, Nv, D = 3, 413, 256                       # 1239 rows: not a multiple of the 4-row workgroup
    h = (torch.randn(B, Nv, D, generator=g) * 1.5).to(dtype)
    res = (torch.randn(B, Nv, D, generator=g) + 0.3).to(dtype)
    pos = torch.randn(1, Nv, D, generator=g)
    norm = torch.nn.LayerNorm(D)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(D, generator=g)); norm.bias.copy_(0.1 * torch.randn(D, generator=g))
    gy = torch.randn(B, Nv, D, generator=g)
    gq = torch.randn(B, Nv, D, generator=g)
    # reference in fp32
    hr, rr = h.float().clone().requires_grad_(True), res.float().clone().requires_grad_(True)
    pr = pos.clone().requires_grad_(True)
    yr, qr = oracle_checker.add_layer_norm(hr, rr, norm, 0.0, pr, want_q)
    ((yr * gy).sum() + ((qr * gq).sum() if want_q else 0)).backward()
    ref_g = (hr.grad, rr.grad, norm.weight.grad.clone(), norm.bias.grad.clone(), pr.grad)
    norm.zero_grad()
    # device
    nd = torch.nn.LayerNorm(D).to(DEV); nd.load_state_dict(norm.state_dict())
    hd, rd = h.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
    pd = pos.to(DEV).requires_grad_(True)
    y, q = native.add_layer_norm(hd, rd, nd, 0.0, pd if want_q else None, want_q)
    assert y.dtype == dtype and (q is None) == (not want_q)
    ((y.float() * gy.to(DEV)).sum() + ((q.float() * gq.to(DEV)).sum() if want_q else 0)).backward()
    tol = dict(rtol=1e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(y.float().cpu(), yr.detach(), **tol)
    if want_q:
        torch.testing.assert_close(q.float().cpu(), qr.detach(), **tol)
    gtol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=5e-2)
    torch.testing.assert_close(hd.grad.float().cpu(), ref_g[0], **gtol)
    torch.testing.assert_close(rd.grad.float().cpu(), ref_g[1], **gtol)
    # column sums over 1239 rows: relative to their magnitude
    wtol = dict(rtol=1e-4, atol=1e-3) if dtype == torch.float32 else dict(rtol=3e-2, atol=0.5)
    torch.testing.assert_close(nd.weight.grad.cpu(), ref_g[2], **wtol)
    torch.testing.assert_close(nd.bias.grad.cpu(), ref_g[3], **wtol)
    if want_q:
        torch.testing.assert_close(pd.grad.cpu(), ref_g[4], **(dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32
                                                               else dict(rtol=2e-2, atol=3e-2)))


def test_add_layer_norm_dropout_mask_through_abi():
    """Dropout inside the fused launch, checked through the raw C-ABI: drop rate, 1/(1-p)
    scaling, determinism in (seed, offset), and backward regenerating the SAME mask."""
    lib = native.load()
    rows, D, p = 2048, 256, 0.3
    g = torch.Generator().manual_seed(9)
    h = (torch.rand(rows, D, generator=g) + 1.0).to(DEV)          # strictly positive: dropped <=> z == res
    res = torch.zeros(rows, D, device=DEV)
    gamma, beta = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    st = torch.cuda.current_stream().cuda_stream

    def fwd(seed, offset, epoch=None):
        y, z = torch.empty_like(h), torch.empty_like(h)
        stats = torch.empty(rows, 2, device=DEV)
        rc = lib.dskd_add_ln_fwd(h.data_ptr(), res.data_ptr(), None, 0, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                 None, z.data_ptr(), stats.data_ptr(), rows, D, 1e-5, p, seed, offset, epoch, native.DTYPE_F32, st)
        assert rc == 0, lib.dskd_last_error()
        return y, z, stats
    y, z, stats = fwd(1234, 7)
    keep = z != 0
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 4 * (p * (1 - p) / (rows * D)) ** 0.5 + 1e-3, frac
    torch.testing.assert_close(z[keep], (h / (1 - p))[keep], rtol=1e-6, atol=0)
    # per-row and per-column drop rates are uniform (no stuck lanes / rows)
    assert (keep.float().mean(0) - (1 - p)).abs().max() < 0.06 and (keep.float().mean(1) - (1 - p)).abs().max() < 0.15
    _, z2, _ = fwd(1234, 7)
    assert torch.equal(z, z2)
    _, z3, _ = fwd(1234, 8)
    assert not torch.equal(z3 != 0, keep)
    # the device epoch word is added to the offset: (7, epoch 1) == (8, no epoch), (7, epoch 0) == (7, no epoch)
    ep = torch.zeros((), dtype=torch.int64, device=DEV)
    assert torch.equal(fwd(1234, 7, ep.data_ptr())[1], z)
    ep.add_(1)
    assert torch.equal(fwd(1234, 7, ep.data_ptr())[1], z3)
    # statistics of z
    torch.testing.assert_close(stats[:, 0], z.mean(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stats[:, 1], (z.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4, atol=1e-5)
    # backward with the same key: d(h) == d(res) * mask / (1 - p)
    dy = torch.randn(rows, D, generator=g).to(DEV)
    dres, dh = torch.empty_like(h), torch.empty_like(h)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    rc = lib.dskd_add_ln_bwd(dy.data_ptr(), None, z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), dres.data_ptr(),
                             dh.data_ptr(), dg.data_ptr(), db.data_ptr(), 1, rows, D, p, 1234, 7, None, native.DTYPE_F32, st)
    assert rc == 0, lib.dskd_last_error()
    assert torch.equal(dh != 0, keep & (dres != 0))
    torch.testing.assert_close(dh[keep], (dres / (1 - p))[keep], rtol=1e-6, atol=0)
    torch.testing.assert_close(db, dy.sum(0), rtol=1e-4, atol=1e-3)
    # argument checks of the ABI
    assert lib.dskd_add_ln_fwd(h.data_ptr(), res.data_ptr(), None, 0, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                               None, None, None, rows, 128, 1e-5, 0.0, 0, 0, None, native.DTYPE_F32, st) == -1
    assert lib.dskd_add_ln_bwd(dy.data_ptr(), None, z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), dres.data_ptr(),
                               None, dg.data_ptr(), db.data_ptr(), 1, rows, D, p, 1, 1, None, native.DTYPE_F32, st) == -1
    # several accumulator copies: the column sums are spread over them
    dg4, db4 = torch.zeros(4, D, device=DEV), torch.zeros(4, D, device=DEV)
    rc = lib.dskd_add_ln_bwd(dy.data_ptr(), None, z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), dres.data_ptr(),
                             dh.data_ptr(), dg4.data_ptr(), db4.data_ptr(), 4, rows, D, p, 1234, 7, None, native.DTYPE_F32, st)
    assert rc == 0
    torch.testing.assert_close(db4.sum(0), db, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dg4.sum(0), dg, rtol=1e-4, atol=1e-3)
    assert int((db4.abs().sum(1) > 0).sum()) == 4


def test_add_layer_norm_full_size_properties():
    """BASELINE size (4 x 22 223 rows, bf16): LayerNorm invariants that need no oracle -- every
    output row has mean beta-weighted 0 / variance 1 for gamma = 1, beta = 0, and scaling the
    inputs by 2 leaves y unchanged."""
    g = torch.Generator().manual_seed(3)
    rows, D = 4 * 22223, 256
    h = torch.randn(rows, D, generator=g).to(torch.bfloat16).to(DEV)
    res = torch.randn(rows, D, generator=g).to(torch.bfloat16).to(DEV)
    norm = torch.nn.LayerNorm(D).to(DEV)
    y, _ = native.add_layer_norm(h, res, norm, 0.0)
    yf = y.float()
    assert yf.mean(1).abs().max() < 2e-2 and (yf.var(1, unbiased=False) - 1).abs().max() < 3e-2
    y2, _ = native.add_layer_norm(h * 2, res * 2, norm, 0.0)
    torch.testing.assert_close(y2.float(), yf, rtol=2e-2, atol=2e-2)


# ----------------------------------------------------------------------------- FFN hidden activation
def test_ffn_dropout_kernels_through_ops():
    """In-place dropout without a stored mask + the one-pass backward that recovers the mask from
    the forward output (ext-mmcv FFN: Linear -> ReLU -> Dropout)."""
    g = torch.Generator().manual_seed(2)
    rows, C, p = 1531, 1024, 0.25
    y = torch.relu(torch.randn(rows, C, generator=g)).to(torch.bfloat16).to(DEV)       # ~50 % zeros from the ReLU
    torch.manual_seed(77)
    native._drop_calls = 0
    yd = native.dropout_(y.clone(), p)
    pos = y > 0
    keep = (yd != 0)
    assert not bool((keep & ~pos).any())
    frac = keep[pos].float().mean().item()
    assert abs(frac - (1 - p)) < 5e-3, frac
    torch.testing.assert_close(yd[keep].float(), (y.float() / (1 - p)).to(torch.bfloat16).float()[keep], rtol=1e-2, atol=0)
    native._drop_calls = 0
    assert torch.equal(native.dropout_(y.clone(), p), yd)                                # same (seed, call) -> same mask
    assert not torch.equal(native.dropout_(y.clone(), p) != 0, keep)                     # next call: another mask
    assert torch.equal(native.dropout_(y.clone(), 0.0), y)
    gr = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(DEV)
    out, colsum = native.relu_dropout_bwd(gr, yd, p)
    ref = torch.where(keep, gr.float() / (1 - p), torch.zeros((), device=DEV))
    torch.testing.assert_close(out.float(), ref.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-6)
    torch.testing.assert_close(colsum, ref.sum(0), rtol=2e-3, atol=0.15)
    for Cc in (256, 512, 2048):
        o2, c2 = native.relu_dropout_bwd(gr[:, :Cc].contiguous(), yd[:, :Cc].contiguous(), p)
        torch.testing.assert_close(c2, ref[:, :Cc].sum(0), rtol=2e-3, atol=0.15) if Cc <= 1024 else None
    lib = native.load()
    assert lib.dskd_relu_dropout_bwd(gr.data_ptr(), yd.data_ptr(), out.data_ptr(), None, 1, rows, 768, p, native.DTYPE_BF16,
                                     torch.cuda.current_stream().cuda_stream) == -1


@pytest.mark.parametrize("C", native.COLSUM_WIDTHS)
def test_colsum_bias_gradient(C):
    """Column sums of a tall bf16 matrix (the bias gradient of a Linear) against fp64 torch."""
    g = torch.Generator().manual_seed(C)
    x = torch.randn(9013, C, generator=g).to(torch.bfloat16)          # >= 8192 rows: 32 accumulator copies
    out = native.colsum(x.to(DEV)).cpu()
    torch.testing.assert_close(out.double(), x.double().sum(0), rtol=1e-4, atol=1e-2)
    assert native.colsum(x[:0].to(DEV)).abs().max() == 0


def test_ffn_inner_matches_torch_chain():
    """Linear + ReLU (+ Dropout p=0) through the fused autograd function, tall bf16 input, against
    the PyTorch chain in fp32 on the same rounded inputs."""
    from dskd_amd.transformer import ffn_inner
    g = torch.Generator().manual_seed(6)
    T, D, Hd = 18000, 256, 1024
    x = torch.randn(T, D, generator=g).to(torch.bfloat16)
    w = (torch.randn(Hd, D, generator=g) * 0.05).to(torch.bfloat16)
    b = (torch.randn(Hd, generator=g) * 0.1).to(torch.bfloat16)
    gy = torch.randn(T, Hd, generator=g).to(torch.bfloat16)
    xr, wr, br = (t.float().clone().requires_grad_(True) for t in (x, w, b))
    torch.relu(torch.nn.functional.linear(xr, wr, br)).backward(gy.float())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ffn_inner(xd, wd, bd, 0.0)
    assert "FFNInner" in type(y.grad_fn.next_functions[0][0]).__name__        # (behind the output view)
    y.backward(gy.to(DEV))
    torch.testing.assert_close(y.float().cpu(), torch.relu(torch.nn.functional.linear(x.float(), w.float(), b.float())),
                               rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(wd.grad.float().cpu(), wr.grad, rtol=3e-2, atol=1.5)      # sums over 18 000 rows (|dW| ~ 100), bf16 result
    torch.testing.assert_close(bd.grad.float().cpu(), br.grad, rtol=3e-2, atol=0.5)


# ----------------------------------------------------------------------------- conv epilogue
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_id,relu", [(False, True), (True, True), (True, False), (False, False)])
def test_bias_act_vs_torch(dtype, with_id, relu, oracle_checker):
    """In-place bias (+ identity) (+ ReLU) pass after a folded convolution, forward and backward,
    against the PyTorch ops of ResNet's Bottleneck on the same (rounded) inputs."""
    g = torch.Generator().manual_seed(4)
    N, C, H, W = 3, 72, 13, 29                                  # odd sizes; C % 8 == 0
    x = torch.randn(N, C, H, W, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    idt = torch.randn(N, C, H, W, generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, generator=g).to(dtype)
    gy = torch.randn(N, C, H, W, generator=g)
    xr, ir = x.float().clone().requires_grad_(True), idt.float().clone().requires_grad_(True)
    yr = oracle_checker.bias_act(xr, b.float(), ir if with_id else None, relu)
    (yr * gy).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    idd = idt.to(DEV).requires_grad_(True)
    xin = xd * 1.0                                              # a non-leaf, like a convolution output
    y = native.bias_act(xin, b.to(DEV), idd if with_id else None, relu)
    assert y.data_ptr() == xin.data_ptr()                       # in place
    (y.float() * gy.to(DEV)).sum().backward()
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(y.float().cpu(), yr.detach(), **tol)
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, **tol)
    if with_id:
        torch.testing.assert_close(idd.grad.float().cpu(), ir.grad, **tol)
    # layouts the kernel does not take fall back to the same arithmetic on the GPU
    xc = x.to(DEV).contiguous()                                 # NCHW-contiguous
    y2 = native.bias_act(xc.clone(), b.to(DEV), None, relu)
    torch.testing.assert_close(y2.float().cpu(), oracle_checker.bias_act(x.float(), b.float(), None, relu), **tol)


# ----------------------------------------------------------------------------- LSAP
def _lsap_device(mats):
    flat = torch.cat([torch.from_numpy(m).reshape(-1) for m in mats]).to(DEV)
    nr = [m.shape[0] for m in mats]
    nc = [m.shape[1] for m in mats]
    offs = np.cumsum([0] + [m.size for m in mats])[:-1].tolist()
    row, col, outs, status = native.lsap_batched(flat, nr, nc, offs)
    row, col, status = row.cpu().numpy(), col.cpu().numpy(), status.cpu().numpy()
    res = []
    for p, m in enumerate(mats):
        n = min(m.shape)
        res.append((row[outs[p]:outs[p] + n], col[outs[p]:outs[p] + n], status[p]))
    return res


def test_lsap_device_bit_exact():
    from scipy.optimize import linear_sum_assignment as sp
    rng = np.random.default_rng(0)
    mats = []
    for t in range(400):
        nr, nc = rng.integers(1, 14), rng.integers(1, 14)
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
        mats.append(c.astype(np.float32))
    for G in (1, 5, 17, 60, 100, 110, 300, 310):
        mats.append(rng.random((300, G)).astype(np.float32))
        mats.append(rng.integers(0, 5, size=(300, G)).astype(np.float32))
    mats.append(rng.random((1024, 3)).astype(np.float32))
    mats.append(rng.random((7, 1024)).astype(np.float32))
    c = rng.random((6, 9)).astype(np.float32)
    c[2, 4] = np.inf
    mats.append(c)
    res = _lsap_device(mats)
    for m, (r, c_, st) in zip(mats, res):
        assert st == 0
        a = sp(m)
        b = oracle_lsa(m)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(r, a[0]) and np.array_equal(c_, a[1]), (m.shape,)


@pytest.mark.parametrize("shapes,kind", [
    ([(300, 17)] * 24, "float"),            # the benchmark's 6 layers x 4 images
    ([(300, 110), (300, 100), (300, 65)], "float"),
    ([(300, 110), (300, 100), (300, 65)], "ties"),
    ([(120, 128), (128, 120), (65, 65)], "ties"),
    ([(300, 17), (300, 1), (1, 300), (300, 64)], "dup"),
])
def test_lsap_several_waves_per_problem(shapes, kind):
    """r4: problems wider than 64 run with one column per thread on several waves (column state in registers, the work matrix
    transposed in LDS when it fits -- these sets fit, the 300 x 310 set of test_lsap_device_bit_exact does not).  Bit-equal
    with scipy AND with the one-wave kernel (dskd_lsap_tune(1)) on float, tie-heavy integer and duplicated-column costs."""
    from scipy.optimize import linear_sum_assignment as sp
    rng = np.random.default_rng(len(shapes) * 7 + len(kind))
    mats = []
    for nr, nc in shapes:
        if kind == "float":
            c = rng.random((nr, nc))
        elif kind == "ties":
            c = rng.integers(0, 4, size=(nr, nc))
        else:
            c = np.round(rng.normal(size=(nr, nc)), 1)
            c[:, -1] = c[:, 0]
        mats.append(c.astype(np.float32))
    lib = native.load()
    runs = {}
    try:
        for mode in (0, 1, 2, 3):       # automatic | one-wave kernel | 1 / 2 columns per thread
            assert lib.dskd_lsap_tune(mode) == 0
            runs[mode] = _lsap_device(mats)
    finally:
        lib.dskd_lsap_tune(0)
    assert lib.dskd_lsap_tune(4) != 0
    for p, m in enumerate(mats):
        a = sp(m)
        for mode, res in runs.items():
            r, c_, st = res[p]
            assert st == 0
            assert np.array_equal(r, a[0]) and np.array_equal(c_, a[1]), (m.shape, mode)
    # errors on the several-waves path: NaN / -inf anywhere, an unreachable row
    bad = rng.random((300, 70)).astype(np.float32); bad[17, 3] = np.nan
    ninf = rng.random((300, 70)).astype(np.float32); ninf[299, 69] = -np.inf
    infeas = rng.random((300, 70)).astype(np.float32); infeas[:, 5] = np.inf
    res = _lsap_device([bad, ninf, infeas, rng.random((300, 70)).astype(np.float32)])
    assert [x[2] for x in res] == [-3, -3, -4, 0]
    for r, c_, st in res[:3]:
        assert (r >= 0).all() and (r < 300).all() and (c_ >= 0).all() and (c_ < 70).all()


def test_lsap_device_errors():
    rng = np.random.default_rng(1)
    bad = rng.random((5, 7)).astype(np.float32)
    bad[1, 1] = np.nan
    ninf = rng.random((5, 7)).astype(np.float32)
    ninf[0, 0] = -np.inf
    infeasible = rng.random((5, 7)).astype(np.float32)
    infeasible[3, :] = np.inf
    ok = rng.random((5, 7)).astype(np.float32)
    res = _lsap_device([bad, ninf, infeasible, ok])
    assert [r[2] for r in res] == [-3, -3, -4, 0]
    for r, c, st in res[:3]:          # failed problems still hand back in-range indices
        assert (r >= 0).all() and (r < 5).all() and (c >= 0).all() and (c < 7).all()
    with pytest.raises(ValueError, match="invalid numeric"):
        native.raise_for_lsap_status(torch.tensor([0, -3]))
    with pytest.raises(ValueError, match="infeasible"):
        native.raise_for_lsap_status(torch.tensor([-4]))


# ----------------------------------------------------------------------------- cost
def test_match_cost_vs_oracle():
    g = torch.Generator().manual_seed(3)
    P, Q, C = 5, 300, 80
    Gs = [17, 1, 0, 60, 9]
    bbox = torch.rand(P, Q, 4, generator=g) * torch.tensor([1.0, 1.0, 0.5, 0.5])
    cls = torch.randn(P, Q, C, generator=g) * 3
    gts, labs, start, wh = [], [], [0], []
    for p in range(P):
        w, h = 1333.0 - 10 * p, 800.0 - 3 * p
        xy = torch.rand(Gs[p], 2, generator=g) * torch.tensor([0.6 * w, 0.6 * h])
        sz = torch.rand(Gs[p], 2, generator=g) * torch.tensor([0.35 * w, 0.35 * h]) + 8
        gts.append(torch.cat([xy, xy + sz], 1))
        labs.append(torch.randint(0, C, (Gs[p],), generator=g))
        start.append(start[-1] + Gs[p])
        wh.append((w, h))
    gt = torch.cat(gts)
    lab = torch.cat(labs)
    cost = native.match_cost(bbox.to(DEV), cls.to(DEV), gt.to(DEV), lab.to(DEV), start, wh, 2.0, 5.0, 2.0).cpu()
    for p in range(P):
        if Gs[p] == 0:
            continue
        ref = assign_ref.cost_matrix(bbox[p], cls[p], gts[p], labs[p], wh[p][0], wh[p][1])
        got = cost[Q * start[p]: Q * start[p + 1]].view(Q, Gs[p])
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


# ----------------------------------------------------------------------------- DSKD losses
def _loss_inputs(B, L, seed, n_t=10, n_gt=7, C=80, D=256):
    g = torch.Generator().manual_seed(seed)
    N = B * 300
    hs_s = torch.randn(N, D, generator=g)
    hs_t = hs_s + 0.1 * torch.randn(N, D, generator=g)
    labels = torch.full((N,), C, dtype=torch.long)
    keep, lab_t = [], []
    for b in range(B):
        perm = torch.randperm(300, generator=g)
        tl = torch.randint(0, L, (n_t,), generator=g)
        labels[b * 300 + perm[:n_t].sort().values] = tl           # one student query per teacher box
        labels[b * 300 + perm[n_t:n_t + n_gt]] = torch.randint(L, C, (n_gt,), generator=g)
        keep.append(b * 300 + torch.randperm(300, generator=g)[:n_t])
        lab_t.append(tl)
    prev = torch.zeros(C, dtype=torch.bool)
    prev[:L] = True
    return hs_s, hs_t, labels, torch.cat(keep), torch.cat(lab_t), prev


@pytest.mark.parametrize("B,L", [(1, 40), (4, 70), (2, 5)])
def test_proto_corr_vs_oracle(B, L):
    hs_s, hs_t, labels, keep, lab_t, prev = _loss_inputs(B, L, 100 + B)
    x = hs_s.clone().requires_grad_(True)
    ref = dskd_losses_ref.proto_corr_loss(x, labels, prev, hs_t, keep, lab_t, L, 1.0)
    ref.backward()
    xd = hs_s.to(DEV).requires_grad_(True)
    out = native.proto_corr_loss(xd, labels.to(DEV), prev.to(DEV), hs_t.to(DEV), keep.to(DEV), lab_t.to(DEV), L, 1.0)
    out.backward()
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=1e-3, atol=1e-7)


def _fg_inputs(B, shapes, seed, n_t, img_hw):
    g = torch.Generator().manual_seed(seed)
    hs_s, hs_t, labels, keep, lab_t, prev = _loss_inputs(B, 40, seed, n_t=n_t)
    fs = [torch.randn(B, 256, h, w, generator=g) for h, w in shapes]
    ft = [f + 0.3 * torch.randn(f.shape, generator=g) for f in fs]
    boxes = []
    for b in range(B):
        H, W = img_hw[b]
        xy = torch.rand(n_t, 2, generator=g) * torch.tensor([0.6 * W, 0.6 * H])
        sz = torch.rand(n_t, 2, generator=g) * torch.tensor([0.35 * W, 0.35 * H]) + 8
        bx = torch.cat([xy, xy + sz], 1)
        bx[:, 0::2].clamp_(0, W)
        bx[:, 1::2].clamp_(0, H)
        boxes.append(bx)
    return fs, ft, boxes, hs_s, hs_t, labels, keep, prev


@pytest.mark.parametrize("B,shapes,n_t", [(1, [(13, 21), (7, 11)], 3), (2, [(25, 42), (13, 21), (7, 11), (4, 6)], 6),
                                          (1, [(100, 70), (50, 35), (33, 9), (17, 3)], 5),      # 8 / 4 / 4 / 2 waves per strip
                                          (1, [(130, 5), (13, 21)], 3)])                       # H > 128: LDS-strip kernel
def test_fgkd_vs_oracle(B, shapes, n_t):
    """The KL of two near-equal softmaxes is O(d^2) computed from O(log H) terms: the
    reference's own fp32 evaluation carries ~1% rounding noise at these magnitudes (asserted
    below against its float64 evaluation).  The HIP kernel is held to rtol 1e-4 of the float64
    evaluation of the oracle (SURVEY.md section 8d tolerance) and the gradient to rtol 1e-3."""
    img_hw = [(200 - 7 * b, 333 - 5 * b) for b in range(B)]
    fs, ft, boxes, hs_s, hs_t, labels, keep, prev = _fg_inputs(B, shapes, 200 + B, n_t, img_hw)
    x32 = hs_s.clone().requires_grad_(True)
    ref32 = dskd_losses_ref.fgkd_loss(fs, ft, boxes, img_hw, hs_t, keep, x32, labels, prev, 2.0, 1.0)
    ref32.backward()
    x = hs_s.double().requires_grad_(True)
    ref = dskd_losses_ref.fgkd_loss([f.double() for f in fs], [f.double() for f in ft], [b.double() for b in boxes],
                                    img_hw, hs_t.double(), keep, x, labels, prev, 2.0, 1.0)
    ref.backward()
    torch.testing.assert_close(ref32.detach().double(), ref.detach(), rtol=5e-2, atol=0)   # reference noise floor
    xd = hs_s.to(DEV).requires_grad_(True)
    out, status = native.fgkd_loss([f.to(DEV) for f in fs], [f.to(DEV) for f in ft], [b.to(DEV) for b in boxes],
                                   img_hw, hs_t.to(DEV), keep.to(DEV), xd, labels.to(DEV), prev.to(DEV), 2.0, 1.0,
                                   return_status=True)
    out.backward()
    assert int(status.item()) == 0
    torch.testing.assert_close(out.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=1e-9)
    torch.testing.assert_close(xd.grad.cpu().double(), x.grad, rtol=1e-3, atol=1e-9)
    torch.testing.assert_close(xd.grad.cpu(), x32.grad, rtol=1e-3, atol=1e-8)


def test_fgkd_full_size_properties():
    """BASELINE feature sizes: identical teacher/student features give exactly zero loss and
    zero gradient (KL of equal distributions), and no boxes gives zero as well."""
    B = 2
    img_hw = [(800, 1333)] * B
    fs, ft, boxes, hs_s, hs_t, labels, keep, prev = _fg_inputs(B, SHAPES_FULL, 300, 10, img_hw)
    fsd = [f.to(DEV) for f in fs]
    xd = hs_s.to(DEV).requires_grad_(True)
    out = native.fgkd_loss(fsd, fsd, [b.to(DEV) for b in boxes], img_hw, hs_t.to(DEV), keep.to(DEV), xd,
                           labels.to(DEV), prev.to(DEV), 2.0, 1.0)
    out.backward()
    assert abs(float(out.detach())) < 1e-6
    assert float(xd.grad.abs().max()) < 1e-5
    # real case is positive and finite
    out2 = native.fgkd_loss(fsd, [f.to(DEV) for f in ft], [b.to(DEV) for b in boxes], img_hw, hs_t.to(DEV),
                            keep.to(DEV), xd, labels.to(DEV), prev.to(DEV), 2.0, 1.0)
    assert float(out2.detach()) > 0 and np.isfinite(float(out2.detach()))


def test_dskd_losses_ignore_out_of_range_keepid():
    """A teacher keepid outside [0, B*Q) is an IndexError in the reference; the kernels must never read out of bounds
    (a GPU fault can take the node down): such a detection contributes nothing, every other one is unchanged."""
    hs_s, hs_t, labels, keep, lab_t, prev = _loss_inputs(2, 70, 131)
    L = 70
    N = hs_s.shape[0]
    ok = native.proto_corr_loss(hs_s.to(DEV), labels.to(DEV), prev.to(DEV), hs_t.to(DEV), keep.to(DEV), lab_t.to(DEV), L, 1.0)
    # the same detections plus two whose query index is far outside the batch, with labels that already have a prototype
    bad_keep = torch.cat([keep, torch.tensor([N + 225, 10 ** 9])])
    bad_lab = torch.cat([lab_t, lab_t[:2]])
    bad = native.proto_corr_loss(hs_s.to(DEV), labels.to(DEV), prev.to(DEV), hs_t.to(DEV), bad_keep.to(DEV), bad_lab.to(DEV),
                                 L, 1.0)
    torch.cuda.synchronize()
    torch.testing.assert_close(bad, ok, rtol=1e-6, atol=0)


# --------------------------------------------------------------------------- fused MFMA FFN (csrc/ffn_mfma.hip)
def _ffn_inputs(T, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, 256, generator=g).bfloat16()
    w1 = (torch.randn(1024, 256, generator=g) / 16).bfloat16()
    b1 = (torch.randn(1024, generator=g) * 0.1).bfloat16()
    w2 = (torch.randn(256, 1024, generator=g) / 32).bfloat16()
    b2 = (torch.randn(256, generator=g) * 0.1).bfloat16()
    gy = torch.randn(T, 256, generator=g).bfloat16()
    return x, w1, b1, w2, b2, gy


def _close(a, ref, tol):
    """max |a - ref| <= tol * max |ref|: bf16 results of f32 accumulations, one rounding (2^-9 relative) per element
    plus accumulation-order noise; 8e-3 of the largest magnitude is ~4 bf16 ulps there."""
    return float((a.float().cpu() - ref).abs().max()) <= tol * float(ref.abs().max())


@pytest.mark.parametrize("T", [1, 200, 4096 + 37, 88892])
def test_ffn_fused_vs_float_reference(T):
    """T = 88 892 = 4 x 22 223 is the benchmark's token count (ragged last workgroup at the real size).
    dskd_ffn_fwd / dskd_ffn_bwd without dropout against the FFN chain of the reference (ext-mmcv FFN.layers:
    Linear -> ReLU -> Dropout -> Linear) evaluated in fp32 on the CPU from the same bf16-rounded inputs; ragged token
    counts (last workgroup / last wave partly or wholly dead)."""
    x, w1, b1, w2, b2, gy = _ffn_inputs(T)
    dx, dgy = x.to(DEV), gy.to(DEV)
    pf, pb = native.ffn_pack(w1.to(DEV), w2.to(DEV))
    guard = torch.full((T + 64, 1024), 7.0, dtype=torch.bfloat16, device=DEV)      # rows >= T must stay untouched
    y, h = native.ffn_fwd_raw(dx, pf, b1.to(DEV), b2.to(DEV), 0.0, True)
    y_eval, none = native.ffn_fwd_raw(dx, pf, b1.to(DEV), b2.to(DEV), 0.0, False)
    assert none is None and torch.equal(y_eval, y)
    href = torch.relu(x.float() @ w1.float().t() + b1.float())
    assert _close(h, href, 8e-3)
    assert _close(y, h.float().cpu() @ w2.float().t() + b2.float(), 8e-3)
    gh, gx, cs = native.ffn_bwd_raw(dgy, h, pb, 0.0, want_colsum=True)
    ghref = (gy.float() @ w2.float()) * (h.float().cpu() != 0)
    assert _close(gh, ghref, 8e-3)
    assert _close(gx, gh.float().cpu() @ w1.float(), 8e-3)
    csref = ghref.sum(0)                                               # grad of b1: f32 sums of the unrounded gradient
    assert float((cs.cpu() - csref).abs().max()) <= 2e-3 * float(ghref.abs().sum(0).max()) + 1e-6
    gh2, gx2 = native.ffn_bwd_raw(dgy, h, pb, 0.0)                      # without the column sums: same tensors
    assert torch.equal(gh2, gh) and torch.equal(gx2, gx)
    assert bool((gh[h == 0] == 0).all())
    torch.cuda.synchronize()
    assert bool((guard == 7.0).all())


@pytest.mark.parametrize("rows,C", [(1200, 256), (7200, 384), (1, 1024), (300, 80), (16383, 2048), (129, 72)])
def test_colsum_short_vs_float_reference(rows, C):
    """dskd_colsum_short (the bias gradient ``grad.sum(0)`` of the decoder's / head branches' nn.Linear layers, one launch)
    against the f32 CPU sum of the same bf16 values; the short GEMM path of the same layers (dskd_gemm_nt on [rows, K] x
    [N, K]^T + bias, optional ReLU) against the f32 product.  Row counts below, at and off the 128-row-lane stride."""
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).bfloat16()
    xd = x.to(DEV)
    assert native.colsum_short_ok(xd)
    guard = torch.full((4096,), 3.0, dtype=torch.bfloat16, device=DEV)
    out = native.colsum_short(xd)
    ref = x.float().sum(0)
    assert out.dtype == torch.bfloat16 and out.shape == (C,)
    assert float((out.float().cpu() - ref).abs().max()) <= 6e-3 * float(ref.abs().max()) + 1e-3
    if C % 64 == 0:
        w = (torch.randn(128, C, generator=g) / C ** 0.5).bfloat16()
        b = torch.randn(128, generator=g).bfloat16()
        assert native.gemm_nt_2d_ok(xd, w.to(DEV), b.to(DEV))
        for relu in (False, True):
            y = native.gemm_nt_2d(xd, w.to(DEV), b.to(DEV), relu)
            yr = x.float() @ w.float().t() + b.float()
            assert _close(y, torch.relu(yr) if relu else yr, 8e-3)
    torch.cuda.synchronize()
    assert bool((guard == 3.0).all())


def test_lin256_prepack_serves_fresh_images_only():
    """native.Lin256Prepack (one dskd_lin256_pack_many launch for all weights of a step) hands lin256_pack the same bytes as
    a pack on the spot -- forward and transposed form, a [256, 256] weight and a [384, 256] joint buffer -- and ONLY while
    its stamp matches the owner's epoch: after the sources were rewritten without a refresh the stale image is not used."""
    g = torch.Generator().manual_seed(9)
    w = torch.randn(256, 256, generator=g).bfloat16().to(DEV)
    j = torch.randn(384, 256, generator=g).bfloat16().to(DEV)

    def on_the_spot(t, transposed):
        native._prepacked.clear()
        return native.lin256_pack(t, transposed)

    want = {(id(w), False): on_the_spot(w, False), (id(w), True): on_the_spot(w, True), (id(j), False): on_the_spot(j, False)}
    epoch = [0]
    pre = native.Lin256Prepack([w, j], epoch)
    assert pre.n == 3
    epoch[0] += 1
    pre.refresh()
    for (t, tr) in ((w, False), (w, True), (j, False)):
        got = native.lin256_pack(t, tr)
        assert got.data_ptr() == pre.images[(t.data_ptr(), tuple(t.shape), tr)].data_ptr()       # the persistent image
        assert torch.equal(got, want[(id(t), tr)])
    # the owner rewrites the sources (next step's cast) and has not refreshed yet: no stale image
    w.mul_(2.0)
    epoch[0] += 1
    fresh = native.lin256_pack(w, False)
    assert fresh.data_ptr() != pre.images[(w.data_ptr(), (256, 256), False)].data_ptr()
    assert torch.equal(fresh.float(), want[(id(w), False)].float() * 2)
    pre.refresh()
    assert torch.equal(native.lin256_pack(w, False).float(), want[(id(w), False)].float() * 2)
    x = torch.randn(17000, 256, generator=g).bfloat16().to(DEV)
    y = native.lin256(x, native.lin256_pack(j, False), 384)
    assert _close(y, x.float().cpu() @ j.float().cpu().t(), 8e-3)
    pre.drop()
    assert not native._prepacked


def test_ffn_fused_dropout_is_the_mask_of_dskd_dropout_fwd():
    """Training forward: the dropped hidden activation equals the GEMM chain's (addmm + ReLU, then dskd_dropout_fwd
    under the same key) -- identical zero pattern, values to bf16 rounding (the chain rounds twice) -- the rate is p,
    and the backward scales the surviving gradients by 1 / (1 - p)."""
    T, p = 20000 + 11, 0.1
    x, w1, b1, w2, b2, gy = [t.to(DEV) for t in _ffn_inputs(T, seed=3)]
    pf, pb = native.ffn_pack(w1, w2)
    _, h0 = native.ffn_fwd_raw(x, pf, b1, b2, 0.0, True)
    native._drop_calls = 4321
    y, h = native.ffn_fwd_raw(x, pf, b1, b2, p, True)
    chain = torch._addmm_activation(b1, x, w1.t())
    native._drop_calls = 4321
    native.dropout_(chain, p)
    assert torch.equal(h != 0, chain != 0)
    assert float((h.float() - chain.float()).abs().max()) <= 1.6e-2 * float(chain.float().abs().max())
    active = h0 != 0
    rate = float(((h == 0) & active).sum()) / float(active.sum())
    assert abs(rate - p) < 2e-3
    kept = h != 0
    assert float((h.float()[kept] - h0.float()[kept] / (1 - p)).abs().max()) <= 1.6e-2 * float(h0.float().abs().max()) / (1 - p)
    assert _close(y, h.float().cpu() @ w2.float().cpu().t() + b2.float().cpu(), 8e-3)
    gh, gx = native.ffn_bwd_raw(gy, h, pb, p)
    ghref = (gy.float().cpu() @ w2.float().cpu()) * (h.float().cpu() != 0) / (1 - p)
    assert _close(gh, ghref, 8e-3) and _close(gx, gh.float().cpu() @ w1.float().cpu(), 8e-3)
    native.advance_dropout_epoch(DEV)                                   # what a graph replay does between steps
    native._drop_calls = 4321
    _, h2 = native.ffn_fwd_raw(x, pf, b1, b2, p, True)
    assert not torch.equal(h2 != 0, h != 0)


def test_ffn_fused_refuses_other_sizes():
    w1 = torch.zeros(512, 256, dtype=torch.bfloat16, device=DEV)
    w2 = torch.zeros(256, 512, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(native.NativeError):
        native.ffn_pack(w1, w2)
    x, w1, b1, w2, b2, _ = [t.to(DEV) for t in _ffn_inputs(8)]
    pf, _ = native.ffn_pack(w1, w2, want_bwd=False)
    with pytest.raises(native.NativeError):
        native.ffn_fwd_raw(x, pf, b1, b2, 0.1, False)                   # dropout without H: the backward could not see the mask


def test_ffn_module_fused_equals_gemm_chain(monkeypatch):
    """transformer.FFN on a tall bf16 activation: the fused MFMA path (default) against the library GEMM chain
    (transformer.FFN_FUSED = False) -- output, input gradient and all four parameter gradients, dropout off."""
    from dskd_amd import transformer
    from dskd_amd.transformer import FFN
    torch.manual_seed(1)
    ffn = FFN(256, 1024, ffn_drop=0.0).to(DEV)
    x = torch.randn(2, 9000, 256, device=DEV)
    up = torch.randn(2, 9000, 256, device=DEV)
    res = {}
    for mode in ("fused", "chain"):
        if mode == "chain":
            monkeypatch.setattr(transformer, "FFN_FUSED", False)
        xi = x.clone().requires_grad_(True)
        for q in ffn.parameters():
            q.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = ffn.core(xi, final_dropout=False)
        out.float().mul(up).sum().backward()
        res[mode] = [out.float(), xi.grad.float()] + [q.grad.float().clone() for q in ffn.parameters()]
    assert res["fused"][0].shape == (2, 9000, 256)
    names = ["out", "dx"] + [n for n, _ in ffn.named_parameters()]
    for n, a, b in zip(names, res["fused"], res["chain"]):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)
        assert err < 2e-2, (n, err)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        monkeypatch.setattr(transformer, "FFN_FUSED", True)
        assert float((ffn.core(x, final_dropout=False).float() - res["fused"][0]).abs().max()) <= 2e-2 * float(res["fused"][0].abs().max())


# --------------------------------------------------------------------------- GroupNorm of the neck (csrc/gn.hip)
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hw", [(100, 167), (13, 21), (1, 3)])
def test_group_norm_cl_vs_torch(dtype, hw, relu):
    """dskd_gn_fwd / dskd_gn_bwd on a channels_last activation against F.group_norm + autograd evaluated in fp32 on the CPU
    from the same (rounded) inputs: output, input gradient, affine gradients; the incoming gradient is a level's slice of a
    concatenated token tensor (own batch stride), as the encoder's backward hands it over.  Tolerance: f32 1e-4 of the
    largest magnitude (summation order), bf16 1.6e-2 (two bf16 ulps: the result and the incoming gradient are rounded)."""
    import torch.nn as nn
    import torch.nn.functional as F
    H, W = hw
    B = 2
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(B, 256, H, W, generator=g) * (1 + torch.arange(256).view(1, 256, 1, 1) % 5) + 0.5).to(dtype)
    gn = nn.GroupNorm(32, 256)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(256, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(256, generator=g) * 0.3)
    extra = 37                                           # rows of "other levels" around this one in the token tensor
    up_tok = torch.randn(B, H * W + extra, 256, generator=g).to(dtype)
    up = up_tok[:, 5:5 + H * W].transpose(1, 2).reshape(B, 256, H, W)          # logical NCHW view of the slice
    xr = x.float().requires_grad_(True)
    yr = F.group_norm(xr, 32, gn.weight, gn.bias, gn.eps)
    if relu:                                             # ConvModule(conv, GN, ReLU) of the GFL towers
        yr = torch.relu(yr)
    gxr, gwr, gbr = torch.autograd.grad(yr, (xr, gn.weight, gn.bias), up.float())

    gnd = nn.GroupNorm(32, 256).to(DEV)
    gnd.load_state_dict(gn.state_dict())
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert native.group_norm_cl_ok(xd, gnd)
    y = native.group_norm_cl(xd, gnd, relu=relu)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    upd_tok = up_tok.to(DEV)
    upd = upd_tok[:, 5:5 + H * W].transpose(1, 2).unflatten(2, (H, W))         # channels_last rows with a batch stride
    gx, gw, gb = torch.autograd.grad(y, (xd, gnd.weight, gnd.bias), upd)
    tol = 1e-4 if dtype == torch.float32 else 1.6e-2
    for name, a, r in (("y", y, yr.detach()), ("dx", gx, gxr), ("dgamma", gw, gwr), ("dbeta", gb, gbr)):
        err = float((a.float().cpu() - r).abs().max())
        assert err <= tol * float(r.abs().max()) + 1e-6, (name, err, float(r.abs().max()))
    with torch.no_grad():                                # inference (the frozen teacher): no statistics saved
        assert torch.equal(native.group_norm_cl(xd.detach(), gnd, relu=relu), y)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_nchw_f32_of_channels_last_maps(dtype):
    """native.nchw_f32 (dskd_nhwc_to_nchw_f32): exactly ``t.contiguous().float()`` for channels_last maps, ragged sizes."""
    for H, W in ((100, 167), (13, 21), (1, 1), (3, 11)):
        t = torch.randn(2, 256, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
        out = native.nchw_f32(t)
        assert out.is_contiguous() and out.dtype == torch.float32 and torch.equal(out, t.contiguous().float())
    t = torch.randn(2, 256, 5, 7, device=DEV)                          # already NCHW: ATen path, same result
    assert torch.equal(native.nchw_f32(t), t)


# --------------------------------------------------------------------------- tall Linear with 256 inputs (lin256_kernel)
@pytest.mark.parametrize("T,N", [(16384 + 77, 256), (20000, 384), (16384, 32), (17000, 512), (88892, 256), (88892, 384)])
def test_lin256_vs_float_reference(T, N):
    """dskd_lin256_fwd (and the transposed pack used for dX) against x @ W^T + b evaluated in fp32 on the CPU from the same
    bf16 inputs; ragged token counts; with and without bias / ReLU.  Tolerance 8e-3 of the largest magnitude."""
    g = torch.Generator().manual_seed(N)
    x = torch.randn(T, 256, generator=g).bfloat16()
    w = (torch.randn(N, 256, generator=g) / 16).bfloat16()
    b = (torch.randn(N, generator=g) * 0.1).bfloat16()
    dx, dw, db = x.to(DEV), w.to(DEV), b.to(DEV)
    assert native.lin256_ok(dx, N, 256)
    pk = native.lin256_pack(dw)
    guard = torch.full((64, N), 3.0, dtype=torch.bfloat16, device=DEV)
    y = native.lin256(dx, pk, N, db, relu=False)
    ref = x.float() @ w.float().t() + b.float()
    assert _close(y, ref, 8e-3)
    yr = native.lin256(dx, pk, N, None, relu=True)
    assert _close(yr, torch.relu(x.float() @ w.float().t()), 8e-3) and bool((yr >= 0).all())
    if N == 256:                                                        # dX = g @ W through the transposed image
        pkt = native.lin256_pack(dw, transposed=True)
        gx = native.lin256(dx, pkt, 256)
        assert _close(gx, x.float() @ w.float(), 8e-3)
    torch.cuda.synchronize()
    assert bool((guard == 3.0).all())


def test_tall_linear_autograd_uses_the_mfma_kernel_and_matches_the_library(monkeypatch):
    """transformer.tall_linear on a tall 256-wide bf16 activation: lin256 path (default) against the hipBLASLt path
    (native.LIN256_ENABLED = False) -- output, dX, dW, db."""
    from dskd_amd.transformer import tall_linear
    torch.manual_seed(4)
    x = torch.randn(2, 9000, 256, device=DEV).bfloat16()
    w = (torch.randn(256, 256, device=DEV) / 16).bfloat16()
    b = (torch.randn(256, device=DEV) * 0.1).bfloat16()
    up = torch.randn(2, 9000, 256, device=DEV).bfloat16()
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setattr(native, "LIN256_ENABLED", False)
        xi, wi, bi = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = tall_linear(xi, wi, bi)
        gx, gw, gb = torch.autograd.grad(y, (xi, wi, bi), up)
        res.append([t.float() for t in (y.detach(), gx, gw, gb)])
    for a, r in zip(*res):
        assert float((a - r).abs().max()) <= 1.6e-2 * float(r.abs().max())


def test_add_pos_equals_the_mixed_dtype_add():
    """native.add_pos (dskd_add_pos): bit-identical to ``(x + pos).to(bf16)`` with pos in f32, full and broadcast tables,
    and the same gradients."""
    x = torch.randn(2, 3001, 256, device=DEV).bfloat16()
    for pos in (torch.randn(2, 3001, 256, device=DEV), torch.randn(1, 3001, 256, device=DEV)):
        xi, pi = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
        q = native.add_pos(xi, pi)
        ref = (x.float() + pos).to(torch.bfloat16)
        assert torch.equal(q, ref)
        up = torch.randn_like(q)
        gx, gp = torch.autograd.grad(q, (xi, pi), up)
        assert torch.equal(gx, up) and gp.shape == pos.shape
        refp = up.float() if pos.shape[0] == 2 else up.float().sum(0, keepdim=True)
        assert float((gp - refp).abs().max()) <= 1e-6 * float(refp.abs().max()) + 1e-6


# --------------------------------------------------------------------------- 1x1 convolution + epilogue (csrc/gemm_nt.hip)
@pytest.mark.parametrize("B,K,N,H,W,stride,with_res,relu", [
    (2, 64, 64, 23, 37, 1, False, True),        # layer1 conv1 of the first block (N = 64 tile variant), ragged M
    (2, 64, 256, 23, 37, 1, True, True),        # conv3 + identity + ReLU
    (1, 256, 128, 20, 33, 1, False, True),
    (2, 256, 512, 21, 35, 2, False, False),     # downsample branch: stride 2, no activation, odd input size
    (1, 1024, 2048, 13, 21, 2, False, False),
    (1, 2048, 512, 13, 21, 1, False, True),     # K = 2048: 32 stages
    (2, 512, 256, 9, 11, 1, False, False),      # ChannelMapper lateral (no bias)
    (1, 128, 192, 5, 7, 1, True, False),        # N = 192: 64-wide tiles, M < one tile
])
def test_conv1x1_mfma_vs_float_reference(B, K, N, H, W, stride, with_res, relu):
    """dskd_gemm_nt behind native.conv1x1: ``act(conv2d(x, w, stride) + bias (+ identity))`` and its input gradient against
    F.conv2d + autograd evaluated in fp32 on the CPU from the same bf16-rounded inputs (the chain of
    mmdet/models/backbones/resnet.py:271-303 with the BN folded).  Tolerance: 8e-3 of the largest magnitude (bf16 output of
    an f32 accumulation, as for the other MFMA kernels); dW / d(bias) / d(identity) come from the library and ATen."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(K + N + stride)
    x = torch.randn(B, K, H, W, generator=g).bfloat16()
    w = (torch.randn(N, K, 1, 1, generator=g) / K ** 0.5).bfloat16()
    b = (torch.randn(N, generator=g) * 0.3).bfloat16() if N != 256 or with_res else None
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn(B, N, Ho, Wo, generator=g).bfloat16() if with_res else None
    up = torch.randn(B, N, Ho, Wo, generator=g).bfloat16()

    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    rr = res.float().requires_grad_(True) if with_res else None
    yr = F.conv2d(xr, wr, None if b is None else b.float(), stride=stride)
    if with_res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    gr = torch.autograd.grad(yr, [xr, wr] + ([rr] if with_res else []), up.float())

    conv = torch.nn.Conv2d(K, N, 1, stride=stride, bias=False)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    rd = res.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
    assert native.conv1x1_ok(xd, wd, conv)
    guard = torch.full((4096,), 5.0, dtype=torch.bfloat16, device=DEV)
    y = native.conv1x1(xd, wd, None if b is None else b.to(DEV), rd, relu, stride)
    assert y.shape == (B, N, Ho, Wo) and y.is_contiguous(memory_format=torch.channels_last)
    assert _close(y, yr.detach(), 8e-3)
    gd = torch.autograd.grad(y, [xd, wd] + ([rd] if with_res else []), up.to(DEV).contiguous(memory_format=torch.channels_last))
    for name, a, r in zip(("dx", "dw", "dres"), gd, gr):
        assert _close(a, r, 1.2e-2), (name, float((a.float().cpu() - r).abs().max()), float(r.abs().max()))
    torch.cuda.synchronize()
    assert bool((guard == 5.0).all())


def test_conv1x1_mfma_full_size_layer1():
    """The benchmark's largest 1x1 convolution (layer1 conv3 at B = 4, 800 x 1333: 267 200 tokens, 64 -> 256 with identity
    and ReLU): every output against the f32 CPU evaluation."""
    g = torch.Generator().manual_seed(9)
    B, K, N, H, W = 4, 64, 256, 200, 334
    x = torch.randn(B, H, W, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / 8).bfloat16()
    b = (torch.randn(N, generator=g) * 0.3).bfloat16()
    res = torch.randn(B, H, W, N, generator=g).bfloat16()
    ref = torch.relu(x.float().view(-1, K) @ w.float().t() + b.float() + res.float().view(-1, N))
    xd = x.to(DEV).permute(0, 3, 1, 2)
    rd = res.to(DEV).permute(0, 3, 1, 2)
    y = native.conv1x1(xd, w.to(DEV).view(N, K, 1, 1), b.to(DEV), rd, True, 1)
    assert _close(y.permute(0, 2, 3, 1).reshape(-1, N), ref, 8e-3)


# --------------------------------------------------------------------------- Swin window attention (csrc/winattn.hip)
def _window_attention_reference(qkv, bias, mask, nH, scale):
    """WindowMSA.forward of the reference between qkv and proj (mmdet/models/backbones/swin.py:81-126) in plain fp32 ops:
    q * scale @ k^T + bias (+ mask per window) -> softmax -> @ v -> [Bw, N, C]."""
    Bw, N, _ = qkv.shape
    q, k, v = qkv.view(Bw, N, 3, nH, 32).permute(2, 0, 3, 1, 4)
    attn = (q * scale) @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(Bw // nW, nW, nH, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(Bw, nH, N, N)
    return (attn.softmax(-1) @ v).transpose(1, 2).reshape(Bw, N, nH * 32)


@pytest.mark.parametrize("nH,nW,images,shifted", [(3, 12, 2, True), (6, 6, 1, False), (24, 4, 3, True), (12, 1, 5, False)])
def test_window_attention_vs_float_reference(nH, nW, images, shifted):
    """dskd_winattn_fwd / dskd_winattn_bwd (native.window_attention) against the reference formula evaluated in fp32 on the
    CPU from the same bf16 inputs: output, d(qkv) and the relative-position-bias gradient; shifted layers with several
    distinct masks (blocks of -100 as ShiftWindowMSA builds them), ragged window counts (tasks not a multiple of the
    waves per workgroup).  Tolerances: output 8e-3 of the largest magnitude (bf16 result of f32 accumulation; P is rounded
    to bf16 before P V as in every fused attention), gradients 2e-2 (dS is rounded to bf16 before the dQ / dK products)."""
    N, C = 49, nH * 32
    Bw = nW * images
    g = torch.Generator().manual_seed(nH * 7 + nW)
    qkv = (torch.randn(Bw, N, 3 * C, generator=g) * 1.5).bfloat16()
    bias = torch.randn(nH, N, N, generator=g) * 0.5
    up = torch.randn(Bw, N, C, generator=g).bfloat16()
    scale = 32 ** -0.5
    mask = None
    if shifted:                               # label maps like the reference's img_mask, a few distinct ones
        mask = torch.zeros(nW, N, N)
        for w in range(nW):
            kind = w % 4
            lab = torch.zeros(7, 7)
            if kind in (1, 3):
                lab[:, 4:] += 1
            if kind in (2, 3):
                lab[4:, :] += 2
            lab = lab.view(-1)
            mask[w] = (lab[None, :] != lab[:, None]).float() * -100.0
    qr, br = qkv.float().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = _window_attention_reference(qr, br, mask, nH, scale)
    gq, gb = torch.autograd.grad(ref, (qr, br), up.float())

    qd = qkv.to(DEV).requires_grad_(True)
    bd = bias.to(DEV).requires_grad_(True)
    types = wtype = None
    if shifted:
        t, inv = torch.unique(mask.view(nW, -1), dim=0, return_inverse=True)
        types, wtype = t.view(-1, N, N).to(DEV), inv.to(torch.int32).to(DEV)
        assert types.shape[0] == min(4, nW)
    assert native.window_attention_ok(qd, nH, N, 0.0)
    out = native.window_attention(qd, bd, types, wtype, nH, scale)
    assert out.shape == (Bw, N, C) and out.dtype == torch.bfloat16
    assert _close(out, ref.detach(), 8e-3)
    dq, db = torch.autograd.grad(out, (qd, bd), up.to(DEV))
    assert _close(dq, gq, 2e-2), float((dq.float().cpu() - gq).abs().max()) / float(gq.abs().max())
    assert _close(db, gb, 2e-2), float((db.float().cpu() - gb).abs().max()) / float(gb.abs().max())


@pytest.mark.parametrize("B,C,N,H,W,stride", [
    (2, 128, 128, 21, 35, 1),        # ragged last stage (1 470 pixels), borders on every side
    (1, 256, 128, 17, 19, 2),        # stride 2, odd input: Ho = 9, Wo = 10
    (3, 128, 256, 16, 12, 2),        # stride 2, even input (the last tap column / row falls outside on one side only)
    (2, 256, 256, 9, 8, 1),
    (4, 512, 512, 25, 42, 1),        # ResNet stage 4 at the benchmark's size: 36 x 4 tiles, several splits
    (1, 128, 128, 1, 1, 1),          # one pixel: only the centre tap sees data
])
def test_conv3x3_weight_gradient_vs_float_reference(B, C, N, H, W, stride):
    """dskd_conv3x3_wgrad (native.conv3x3_wgrad: the split-K MFMA kernel over a virtual [pixels, 9 C] operand) against
    ``torch.nn.grad.conv2d_weight`` in fp32 on the CPU from the same bf16 tensors; deterministic (two calls bit-equal, whatever
    the scratch held).  Tolerance: bf16 result of an f32 accumulation over up to 4 200 pixels, 8e-3 of the largest entry."""
    g = torch.Generator().manual_seed(B * 100 + C + stride)
    x = torch.randn(B, C, H, W, generator=g).bfloat16()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    gy = torch.randn(B, N, Ho, Wo, generator=g).bfloat16()
    ref = torch.nn.grad.conv2d_weight(x.float(), (N, C, 3, 3), gy.float(), stride=stride, padding=1)
    cl = torch.channels_last
    xd, gd = x.to(DEV).contiguous(memory_format=cl), gy.to(DEV).contiguous(memory_format=cl)
    assert native.conv3x3_wgrad_ok(gd, xd, stride)
    dw = native.conv3x3_wgrad(gd, xd, stride)
    assert dw.shape == (N, C, 3, 3) and dw.dtype == torch.bfloat16 and dw.is_contiguous(memory_format=cl)
    assert _close(dw, ref, 8e-3), float((dw.float().cpu() - ref).abs().max()) / float(ref.abs().max())
    for ws in native._tn_scratch.values():
        ws.fill_(0x7F)
    assert torch.equal(native.conv3x3_wgrad(gd, xd, stride), dw)
    # dskd_conv3x3_wgrad_bias: the same dW bit for bit, plus the sums of g over batch and pixels (the folded-BN bias gradient)
    dw2, db = native.conv3x3_wgrad(gd, xd, stride, want_bias=True)
    assert torch.equal(dw2, dw) and db.shape == (N,) and db.dtype == torch.bfloat16
    bref = gy.float().sum((0, 2, 3))
    assert _close(db, bref, 8e-3), float((db.float().cpu() - bref).abs().max()) / float(bref.abs().max())
    # shapes the kernel is not built for are refused, not mis-computed
    assert not native.conv3x3_wgrad_ok(gd[:, :64].contiguous(memory_format=cl), xd, stride)
    assert native.load().dskd_conv3x3_wgrad_scratch_bytes(B, H, W, 64, N, stride) == -1
    assert native.load().dskd_conv3x3_wgrad_scratch_bytes(B, H, W, C, N, 3) == -1


def test_weight_transposes_of_a_stage_in_one_launch():
    """dskd_weight_t_many (native.WeightTransposes): the operands of the input-gradient launches -- w^T of 1x1 weights, the
    tap-flipped channel-swapped 3x3 weight -- for a list of convolutions at once, bit-equal with the torch expressions the
    Bottleneck backward used to evaluate per convolution; ineligible entries (odd channel counts, f32) come back as None; a
    second call with the same addresses reuses the device table."""
    g = torch.Generator().manual_seed(5)
    cl = torch.channels_last
    ws = [torch.randn(128, 256, 1, 1, generator=g).bfloat16().to(DEV).contiguous(memory_format=cl),
          torch.randn(128, 128, 3, 3, generator=g).bfloat16().to(DEV).contiguous(memory_format=cl),
          torch.randn(512, 128, 1, 1, generator=g).bfloat16().to(DEV),
          torch.randn(256, 64, 3, 3, generator=g).bfloat16().to(DEV).contiguous(memory_format=cl),
          torch.randn(96, 64, 1, 1, generator=g).bfloat16().to(DEV),                  # 96 rows: not eligible
          torch.randn(64, 64, 1, 1, generator=g).to(DEV)]                             # f32: not eligible
    tr = native.WeightTransposes()
    for _ in range(2):
        outs = tr.run(ws)
        assert outs[4] is None and outs[5] is None
        for w, o in zip(ws[:4], outs[:4]):
            if w.shape[2] == 1:
                assert o.shape == (w.shape[1], w.shape[0]) and torch.equal(o, w.reshape(w.shape[0], w.shape[1]).t().contiguous())
            else:
                ref = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=cl)
                assert o.shape == ref.shape and o.is_contiguous(memory_format=cl) and torch.equal(o, ref)


def _self_attention_reference(qk, v, H, keep=None, p=0.0):
    """softmax(q k^T / sqrt(32)) (* keep / (1 - p)) @ v in fp32; qk [B, L, 2 E], v [B, L, E] -> [B, L, E]."""
    B, L, E = v.shape
    q, k = qk[..., :E], qk[..., E:]
    q, k, vh = (t.reshape(B, L, H, 32).permute(0, 2, 1, 3) for t in (q, k, v))
    P = ((q @ k.transpose(-1, -2)) * 32 ** -0.5).softmax(-1)
    if keep is not None:
        P = P * keep / (1.0 - p)
    return (P @ vh).permute(0, 2, 1, 3).reshape(B, L, E), P


@pytest.mark.parametrize("B,L,batch_first", [(4, 300, True), (2, 300, False), (1, 37, True), (3, 320, True), (2, 32, False)])
def test_decoder_self_attention_vs_float_reference(B, L, batch_first):
    """dskd_attn_fwd / dskd_attn_bwd (native.self_attention, the core of the decoder's MultiheadAttention) against the
    formula in fp32 on the CPU from the same bf16 inputs, for both token layouts, the training shape (300 queries) and ragged
    / full / single tiles.  Tolerances as for the window attention: output 8e-3 of the largest magnitude (P rounded to bf16
    before P V), gradients 2e-2 (dS rounded to bf16 before the dQ / dK products)."""
    H, E = 8, 256
    g = torch.Generator().manual_seed(L * 3 + B)
    qk = (torch.randn(B, L, 2 * E, generator=g) * 1.5).bfloat16()
    v = torch.randn(B, L, E, generator=g).bfloat16()
    up = torch.randn(B, L, E, generator=g).bfloat16()
    qr, vr = qk.float().requires_grad_(True), v.float().requires_grad_(True)
    ref, _ = _self_attention_reference(qr, vr, H)
    gq, gv = torch.autograd.grad(ref, (qr, vr), up.float())

    lay = (lambda t: t) if batch_first else (lambda t: t.transpose(0, 1).contiguous())
    qd, vd = lay(qk).to(DEV).requires_grad_(True), lay(v).to(DEV).requires_grad_(True)
    assert native.self_attention_ok(qd, vd, H)
    out = native.self_attention(qd, vd, H, 0.0, batch_first=batch_first)
    assert out.shape == vd.shape and out.dtype == torch.bfloat16
    assert _close((out if batch_first else out.transpose(0, 1)).detach(), ref.detach(), 8e-3)
    dq, dv = torch.autograd.grad(out, (qd, vd), lay(up).to(DEV))
    if not batch_first:
        dq, dv = dq.transpose(0, 1), dv.transpose(0, 1)
    assert _close(dq, gq, 2e-2), float((dq.float().cpu() - gq).abs().max()) / float(gq.abs().max())
    assert _close(dv, gv, 2e-2), float((dv.float().cpu() - gv).abs().max()) / float(gv.abs().max())
    with torch.no_grad():                # inference: no statistics buffer
        assert torch.equal(native.self_attention(qd.detach(), vd.detach(), H, 0.0, batch_first=batch_first), out)


def test_decoder_self_attention_dropout_mask_is_the_same_in_forward_and_backward():
    """Attention dropout (p = 0.1 in the benchmark config): the mask is a counter hash of (image, head, query, key) keyed by
    (seed, offset + epoch).  Recovered here through the C-ABI with one-hot value probes (32 keys per launch, the SAME key
    every launch): the kept entries equal P / (1 - p) of the fp32 reference, the dropped fraction is p, another offset or
    epoch draws another mask; and the backward -- which regenerates the mask in BOTH of its orientations (key-tile waves for
    dK / dV, query-tile waves for dQ) -- matches autograd through the reference with that recovered mask."""
    import ctypes as C
    B, H, L, E, p = 2, 8, 96, 256, 0.25
    g = torch.Generator().manual_seed(11)
    qk = (torch.randn(B, L, 2 * E, generator=g)).bfloat16().to(DEV)
    lib, st = native.load(), (C.c_int64 * 8)(L * 2 * E, 2 * E, L * 2 * E, 2 * E, L * E, E, L * E, E)
    epoch = torch.zeros((), dtype=torch.int64, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream

    def probe(seed, offset):
        Pd = torch.zeros(B, H, L, L)
        for t in range(L // 32):
            v = torch.zeros(B, L, H, 32)
            v[:, 32 * t + torch.arange(32), :, torch.arange(32)] = 1.0       # key 32 t + d lights channel d of every head
            vd, out = v.view(B, L, E).bfloat16().to(DEV), torch.empty(B, L, E, dtype=torch.bfloat16, device=DEV)
            rc = lib.dskd_attn_fwd(qk.data_ptr(), qk.data_ptr() + 2 * E, vd.data_ptr(), out.data_ptr(), None, B, H, L, 32, st,
                                   32 ** -0.5, p, seed, offset, epoch.data_ptr(), native.DTYPE_BF16, stream)
            assert rc == 0
            Pd[..., 32 * t:32 * t + 32] = out.float().cpu().view(B, L, H, 32).permute(0, 2, 1, 3)
        return Pd

    Pd = probe(1234, 7)
    _, P = _self_attention_reference(qk.float().cpu(), torch.zeros(B, L, E), H)
    keep = Pd != 0
    frac = 1.0 - keep.float().mean().item()
    assert abs(frac - p) < 0.01, frac
    assert (keep.float().mean((-1, -2)) > 0.6).all()                    # every (image, head) has its own mask, none degenerate
    assert float((Pd - P / (1 - p) * keep).abs().max()) < 8e-3 * float(P.max() / (1 - p))
    assert torch.equal(probe(1234, 7), Pd)
    assert not torch.equal(probe(1234, 8) != 0, keep) and not torch.equal(probe(1235, 7) != 0, keep)
    epoch.add_(1 << 32)
    assert not torch.equal(probe(1234, 7) != 0, keep)
    epoch.zero_()

    v = torch.randn(B, L, E, generator=g).bfloat16()
    up = torch.randn(B, L, E, generator=g).bfloat16()
    qr, vr = qk.float().cpu().requires_grad_(True), v.float().requires_grad_(True)
    ref, _ = _self_attention_reference(qr, vr, H, keep.float(), p)
    gq, gv = torch.autograd.grad(ref, (qr, vr), up.float())
    vd, upd = v.to(DEV), up.to(DEV)
    out = torch.empty(B, L, E, dtype=torch.bfloat16, device=DEV)
    stats = torch.empty(B, H, L, 2, device=DEV)
    delta = torch.empty(B, H, L, device=DEV)
    dqk, dv = torch.full_like(qk, float("nan")), torch.full_like(vd, float("nan"))
    assert lib.dskd_attn_fwd(qk.data_ptr(), qk.data_ptr() + 2 * E, vd.data_ptr(), out.data_ptr(), stats.data_ptr(), B, H, L, 32,
                             st, 32 ** -0.5, p, 1234, 7, epoch.data_ptr(), native.DTYPE_BF16, stream) == 0
    assert _close(out, ref.detach(), 8e-3)
    assert lib.dskd_attn_bwd(qk.data_ptr(), qk.data_ptr() + 2 * E, vd.data_ptr(), out.data_ptr(), upd.data_ptr(),
                             stats.data_ptr(), delta.data_ptr(), dqk.data_ptr(), dqk.data_ptr() + 2 * E, dv.data_ptr(), B, H, L,
                             32, st, 32 ** -0.5, p, 1234, 7, epoch.data_ptr(), native.DTYPE_BF16, stream) == 0
    assert _close(dqk, gq, 2e-2), float((dqk.float().cpu() - gq).abs().max()) / float(gq.abs().max())
    assert _close(dv, gv, 2e-2), float((dv.float().cpu() - gv).abs().max()) / float(gv.abs().max())
    # argument checks: wrong head dimension, too many tokens, odd strides
    assert lib.dskd_attn_fwd(qk.data_ptr(), qk.data_ptr(), vd.data_ptr(), out.data_ptr(), None, B, H, L, 64, st, 1.0, 0.0, 0, 0,
                             None, native.DTYPE_BF16, stream) != 0
    assert lib.dskd_attn_fwd(qk.data_ptr(), qk.data_ptr(), vd.data_ptr(), out.data_ptr(), None, B, H, 321, 32, st, 1.0, 0.0, 0, 0,
                             None, native.DTYPE_BF16, stream) != 0
    bad = (C.c_int64 * 8)(L * 2 * E, 2 * E + 4, L * 2 * E, 2 * E, L * E, E, L * E, E)
    assert lib.dskd_attn_fwd(qk.data_ptr(), qk.data_ptr(), vd.data_ptr(), out.data_ptr(), None, B, H, L, 32, bad, 1.0, 0.0, 0, 0,
                             None, native.DTYPE_BF16, stream) != 0


@pytest.mark.parametrize("shift", [0, 3])
def test_shift_window_msa_module_uses_the_mfma_kernel(shift):
    """swin.ShiftWindowMSA on the GPU under bf16 autocast (window attention through csrc/winattn.hip) against the same
    module on the CPU in fp32 (SDPA path), padded resolution (9 x 13 tokens): output and every parameter gradient."""
    from dskd_amd import swin
    torch.manual_seed(5 + shift)
    att = swin.ShiftWindowMSA(96, 3, 7, shift_size=shift).eval()
    with torch.no_grad():
        att.w_msa.relative_position_bias_table.normal_(std=0.5)
    x = torch.randn(2, 9 * 13, 96)
    up = torch.randn(2, 9 * 13, 96)
    xc = x.clone().requires_grad_(True)
    yc = att(xc, (9, 13))
    gc = torch.autograd.grad(yc, [xc] + list(att.parameters()), up)
    att_d = copy.deepcopy(att).to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    calls = []
    orig = native.window_attention
    native.window_attention = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yd = att_d(xd, (9, 13))
    finally:
        native.window_attention = orig
    assert calls, "the MFMA window-attention kernel was not used"
    gd = torch.autograd.grad(yd, [xd] + list(att_d.parameters()), up.to(DEV))
    assert _close(yd, yc.detach(), 3e-2)
    for a, r in zip(gd, gc):
        assert _close(a, r, 5e-2), (tuple(r.shape), float((a.float().cpu() - r).abs().max()) / float(r.abs().max()))


@pytest.mark.parametrize("B,C,N,H,W,stride,relu", [
    (2, 64, 64, 23, 37, 1, True),           # layer1 conv2, odd sizes (borders on every side)
    (1, 128, 128, 20, 33, 2, True),         # first block of a stage: stride 2, odd input width
    (2, 256, 256, 9, 11, 1, True),
    (1, 512, 512, 13, 21, 1, False),
    (1, 128, 192, 5, 4, 1, False),          # N = 192: the 64-wide tile variant; tiny map
    (3, 64, 128, 1, 1, 1, True),            # a 1 x 1 map: only the centre tap is inside the image
])
def test_conv3x3_mfma_vs_float_reference(B, C, N, H, W, stride, relu):
    """dskd_conv3x3 behind native.conv3x3: ``act(conv2d(x, w, stride, padding=1) + bias)`` and its input gradient against
    F.conv2d + autograd in fp32 on the CPU from the same bf16-rounded inputs (conv2 -> bn2 -> relu of the Bottleneck,
    mmdet/models/backbones/resnet.py:283-288, BN folded); zero padding on every border, stride 2, ragged M."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(C + N + H)
    x = torch.randn(B, C, H, W, generator=g).bfloat16()
    w = (torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5).bfloat16()
    b = (torch.randn(N, generator=g) * 0.3).bfloat16()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    up = torch.randn(B, N, Ho, Wo, generator=g).bfloat16()
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.float(), stride=stride, padding=1)
    if relu:
        yr = torch.relu(yr)
    gr = torch.autograd.grad(yr, [xr, wr], up.float())
    conv = torch.nn.Conv2d(C, N, 3, stride=stride, padding=1, bias=False)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert native.conv3x3_ok(xd, wd, conv)
    guard = torch.full((4096,), 5.0, dtype=torch.bfloat16, device=DEV)
    y = native.conv3x3(xd, wd, b.to(DEV), None, relu, stride)
    assert y.shape == (B, N, Ho, Wo) and y.is_contiguous(memory_format=torch.channels_last)
    assert _close(y, yr.detach(), 8e-3)
    gd = torch.autograd.grad(y, [xd, wd], up.to(DEV).contiguous(memory_format=torch.channels_last))
    for name, a, r in zip(("dx", "dw"), gd, gr):
        assert _close(a, r, 1.2e-2), (name, float((a.float().cpu() - r).abs().max()), float(r.abs().max()))
    torch.cuda.synchronize()
    assert bool((guard == 5.0).all())


@pytest.mark.parametrize("Cin,P,H,W,stride,down", [(512, 128, 20, 27, 1, False), (256, 128, 21, 30, 2, True),
                                                   (1024, 256, 9, 14, 1, False)])
def test_fused_bottleneck_chain_vs_float_reference(Cin, P, H, W, stride, down):
    """native.bottleneck (one autograd node per Bottleneck; ReLU masks and the identity-path gradient add folded into the
    input-gradient GEMMs: dskd_gemm_nt_dx / dskd_conv3x3_dx) on a chain of TWO blocks -- the second one hands the first a
    gradient that is already masked and tagged -- against the same chain in fp32 on the CPU (F.conv2d + autograd) from the
    same bf16-rounded inputs: output, input gradient and every weight gradient.  Reference:
    mmdet/models/backbones/resnet.py:271-303 (BatchNorms folded into weight + bias)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(Cin + P + H)
    B, N = 2, 4 * P

    def mk(co, ci, k):
        return (torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5).bfloat16()

    x = torch.randn(B, Cin, H, W, generator=g).relu().bfloat16()
    blocks = [dict(w1=mk(P, Cin, 1), w2=mk(P, P, 3), w3=mk(N, P, 1), wd=mk(N, Cin, 1) if down else None, s=stride),
              dict(w1=mk(P, N, 1), w2=mk(P, P, 3), w3=mk(N, P, 1), wd=None, s=1)]
    if not down:
        assert Cin == N
    for blk in blocks:
        for k in ("b1", "b2", "b3", "bd"):
            co = P if k in ("b1", "b2") else N
            blk[k] = (torch.randn(co, generator=g) * 0.1).bfloat16()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    up = torch.randn(B, N, Ho, Wo, generator=g).bfloat16()

    # fp32 reference
    xr = x.float().requires_grad_(True)
    leaves, h = [xr], xr
    for blk in blocks:
        ws = {k: blk[k].float().requires_grad_(True) for k in ("w1", "w2", "w3", "wd") if blk[k] is not None}
        leaves += [ws[k] for k in ("w1", "w2", "w3", "wd") if k in ws]
        o = F.relu(F.conv2d(h, ws["w1"], blk["b1"].float()))
        o = F.relu(F.conv2d(o, ws["w2"], blk["b2"].float(), stride=blk["s"], padding=1))
        idn = F.conv2d(h, ws["wd"], blk["bd"].float(), stride=blk["s"]) if "wd" in ws else h
        h = F.relu(F.conv2d(o, ws["w3"], blk["b3"].float()) + idn)
    gr = torch.autograd.grad(h, leaves, up.float())

    def dev_w(w):
        return w.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)

    def run(fused):
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        leaves_d, hd = [xd], xd
        for bi, blk in enumerate(blocks):
            ws = {k: dev_w(blk[k]) for k in ("w1", "w2", "w3", "wd") if blk[k] is not None}
            leaves_d += [ws[k] for k in ("w1", "w2", "w3", "wd") if k in ws]
            bs = {k: blk[k].to(DEV) for k in ("b1", "b2", "b3", "bd")}
            if fused:
                hd = native.bottleneck(hd, ws["w1"], bs["b1"], ws["w2"], bs["b2"], ws["w3"], bs["b3"], ws.get("wd"),
                                       bs["bd"] if "wd" in ws else None, blk["s"], x_is_relu=True)
            else:
                o = native.conv1x1(hd, ws["w1"], bs["b1"], None, True, 1)
                o = native.conv3x3(o, ws["w2"], bs["b2"], None, True, blk["s"])
                idn = native.conv1x1(hd, ws["wd"], bs["bd"], None, False, blk["s"]) if "wd" in ws else hd
                hd = native.conv1x1(o, ws["w3"], bs["b3"], idn, True, 1)
        gd = torch.autograd.grad(hd, leaves_d, up.to(DEV).contiguous(memory_format=torch.channels_last))
        return hd.detach(), gd

    guard = torch.full((4096,), 5.0, dtype=torch.bfloat16, device=DEV)
    y_f, g_f = run(True)
    y_u, g_u = run(False)
    assert y_f.shape == (B, N, Ho, Wo) and torch.equal(y_f, y_u)              # the same forward launches
    assert float((y_f.float().cpu() - h.detach()).norm() / h.detach().norm()) <= 3e-2
    xmask = (x.float() > 0).float()
    for i, (a, u, r) in enumerate(zip(g_f, g_u, gr)):
        assert a.shape == r.shape
        if i == 0:          # the fused node returns the input gradient already masked by x > 0 (see the class docstring)
            r, u = r * xmask, u * xmask.to(DEV)
        # against the unfused chain on the same kernels: only the roundings the fusion removes may differ
        assert _close(a, u.float().cpu(), 1.0e-2), (i, float((a.float() - u.float()).abs().max()), float(u.abs().max()))
        # against fp32 only as a sanity bound, in the Frobenius norm: six ReLUs deep with every intermediate rounded to
        # bf16, a fraction f ~ 0.3 % of the masks flips where a rounded pre-activation crosses zero, and over the identity
        # path a flip toggles a whole upstream gradient entry: relative error ~ sqrt(f) (7 % measured, the unfused chain
        # exactly the same).  Each convolution alone is pinned against fp32 in the tests above.
        rel_f = float((a.float().cpu() - r).norm() / r.norm())
        rel_u = float((u.float().cpu() - r).norm() / r.norm())
        assert rel_f <= 0.15 and rel_f <= 1.1 * rel_u + 5e-3, (i, rel_f, rel_u)
    # x is a ReLU output here: the fused input gradient is zero wherever x is (the mask of the producing layer)
    assert bool((g_f[0][x.to(DEV).contiguous(memory_format=torch.channels_last) <= 0] == 0).all())
    torch.cuda.synchronize()
    assert bool((guard == 5.0).all())


@pytest.mark.parametrize("dtype,want_q", [(torch.float32, True), (torch.bfloat16, True), (torch.bfloat16, False)])
def test_add_layer_norm_fork_sums_the_two_gradients_in_the_kernel(dtype, want_q):
    """native.add_layer_norm(fork=True) hands y out as two autograd outputs; the gradients of the two consumers reach
    dskd_add_ln_bwd2 as dy / dy2 and are summed there (f32) -- same d(h), d(res), d(gamma), d(beta) as the single-output form,
    where autograd adds them with a launch of its own.  One consumer may also be absent (gradient None)."""
    g = torch.Generator().manual_seed(9)
    rows = 3 * 37
    h = torch.randn(3, 37, 256, generator=g).to(DEV, dtype)
    res = torch.randn(3, 37, 256, generator=g).to(DEV, dtype)
    pos = torch.randn(1, 37, 256, generator=g).to(DEV)
    a, b, c = (torch.randn(3, 37, 256, generator=g).to(DEV, dtype) for _ in range(3))
    norm = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.normal_()

    def run(fork, drop_second=False):
        hh, rr = h.clone().requires_grad_(True), res.clone().requires_grad_(True)
        norm.zero_grad()
        if fork:
            y1, y2, q = native.add_layer_norm(hh, rr, norm, 0.0, pos if want_q else None, want_q, fork=True)
        else:
            y1, q = native.add_layer_norm(hh, rr, norm, 0.0, pos if want_q else None, want_q)
            y2 = y1
        loss = (y1.float() * a.float()).sum()
        if not drop_second:
            loss = loss + (y2.float() * b.float()).sum()
        if want_q:
            loss = loss + (q.float() * c.float()).sum()
        loss.backward()
        return [hh.grad.float(), rr.grad.float(), norm.weight.grad.float().clone(), norm.bias.grad.float().clone()]

    tol = 1e-5 if dtype == torch.float32 else 2e-2       # bf16: autograd's own sum rounds dy1 + dy2 to bf16 first
    for drop in (False, True):
        ref, got = run(False, drop), run(True, drop)
        for r_, g_ in zip(ref, got):
            assert float((r_ - g_).abs().max()) <= tol * float(r_.abs().max()) + 1e-6, (dtype, want_q, drop)
    assert rows == h.shape[0] * h.shape[1]


def test_multi_tensor_cast_scale_vs_torch():
    """dskd_cast_scale_many (native.MultiCast): f32 -> bf16 and bf16 -> f32 of a list of tensors in one launch, with and
    without a per-output-channel scale, against the PyTorch expression -- bit-exact (one rounding, same order: the product
    is formed in f32 and rounded once).  Sizes that are no multiple of 8, tensors longer than one 8 192-element chunk,
    channels_last weights, a destination whose address only allows scalar stores (odd offset in a flat buffer)."""
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 256), (70,), (70, 256), (64, 64, 3, 3), (128, 256, 1, 1), (3, 5), (1000, 37)]
    srcs = [torch.randn(sh, generator=g).to(DEV) for sh in shapes]
    srcs[3] = srcs[3].contiguous(memory_format=torch.channels_last)
    srcs[4] = srcs[4].contiguous(memory_format=torch.channels_last)
    scales = [None, None, torch.rand(70, generator=g).to(DEV) + 0.5, torch.rand(64, generator=g).to(DEV) + 0.5,
              torch.rand(128, generator=g).to(DEV) + 0.5, torch.rand(3, generator=g).to(DEV), None]
    dsts = [torch.empty_like(s_, dtype=torch.bfloat16) for s_ in srcs]
    assert native.MultiCast.ok(srcs, dsts, scales, 0)
    mc = native.MultiCast(0)
    mc.run(srcs, dsts, scales)
    mc.run(srcs, dsts, scales)                      # second call: cached table
    for s_, d, sc in zip(srcs, dsts, scales):
        ref = s_ if sc is None else s_ * sc.view(-1, *[1] * (s_.dim() - 1))
        assert torch.equal(d, ref.to(torch.bfloat16)), s_.shape
    # the way back, into odd offsets of one flat buffer (scalar path) and into fresh tensors (vector path)
    gsrc = [d.clone() for d in dsts]
    flat = torch.full((sum(t.numel() for t in gsrc) + 16,), 7.0, device=DEV)
    outs, off = [], 1
    for t in gsrc:
        outs.append(flat[off:off + t.numel()].as_strided(t.shape, t.stride()))
        off += t.numel()
    fresh = [torch.empty_like(t, dtype=torch.float32) for t in gsrc]
    mb = native.MultiCast(1)
    for target in (outs, fresh):
        assert native.MultiCast.ok(gsrc, target, scales, 1)
        mb.run(gsrc, target, scales)
        for t, o, sc in zip(gsrc, target, scales):
            ref = t.float() if sc is None else t.float() * sc.view(-1, *[1] * (t.dim() - 1))
            assert torch.equal(o, ref), t.shape
    assert float(flat[0]) == 7.0 and bool((flat[off:] == 7.0).all())
    assert not native.MultiCast.ok(srcs, [d.float() for d in dsts], scales, 0)          # wrong destination dtype


@pytest.mark.parametrize("M,N,K,conv3", [(1000 + 37, 256, 512, False), (4200, 512, 1024, False), (333, 256, 64, False),
                                         (2 * 13 * 21, 256, 128, True), (3 * 25 * 42, 512, 512, True)])
def test_gemm_tile_configurations_agree_with_float_reference(M, N, K, conv3):
    """Every tile configuration of csrc/gemm_nt.hip behind dskd_gemm_nt_ws / dskd_conv3x3_ws -- gemm_nt_kernel with the
    register and the LDS epilogue, the six big tiles of gemm_big_kernel with and without the split-K remainder (forced
    through dskd_gemm_nt_tune, 3 and 7 splits) and the automatic choice -- against the f32 CPU product of the same bf16
    inputs: forward form (bias + residual + ReLU) and input-gradient form (gate + residual).  Ragged M (rows past the end
    untouched), K from one stage to 72.  Tolerance 8e-3 of the largest magnitude (~4 bf16 ulps)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(M + N + K)
    lib = native.load()
    if conv3:
        C = K
        B, H, W = (2, 13, 21) if M == 2 * 13 * 21 else (3, 25, 42)
        x = torch.randn(B, C, H, W, generator=g).bfloat16()
        w = (torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5).bfloat16()
    else:
        x = torch.randn(M, K, generator=g).bfloat16()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, generator=g).bfloat16()
    res = torch.randn(M, N, generator=g).bfloat16()
    gate = torch.randn(M, N, generator=g).bfloat16()
    if conv3:
        pre = F.conv2d(x.float(), w.float(), None, padding=1).permute(0, 2, 3, 1).reshape(M, N)
    else:
        pre = x.float() @ w.float().t()
    ref_f = torch.relu(pre + bias.float() + res.float())
    ref_d = torch.where(gate.float() > 0, pre + res.float(), torch.zeros(()))
    cl = dict(memory_format=torch.channels_last)
    xd = x.to(DEV).contiguous(**cl) if conv3 else x.to(DEV)
    wd = w.to(DEV).contiguous(**cl) if conv3 else w.to(DEV)
    bd, rd, gd = bias.to(DEV), res.to(DEV), gate.to(DEV)
    guard = 3.0
    try:
        for cfg, sp in [(-1, 0), (0, 0), (7, 0), (8, 0), (9, 0)] + [(c, s) for c in range(1, 7) for s in (1, 3, 7)]:
            assert lib.dskd_gemm_nt_tune(cfg, sp) == 0
            bn = 64 * (1, 2, 2, 4, 1, 2, 2)[cfg] if 1 <= cfg <= 6 else 64
            for form in ("fwd", "dx"):
                out = torch.full((M + 8, N), guard, dtype=torch.bfloat16, device=DEV)
                if N % bn:
                    with pytest.raises(native.NativeError):
                        native.gemm_nt_raw(xd, wd, bd, rd, M, N, K, True, out) if not conv3 else \
                            native.conv3x3_raw(xd, wd, bd, rd.view(B, H, W, N).permute(0, 3, 1, 2), True, 1,
                                               out=out[:M].view(B, H, W, N).permute(0, 3, 1, 2))
                    break
                if conv3:
                    o4 = out[:M].view(B, H, W, N).permute(0, 3, 1, 2)
                    r4, g4 = rd.view(B, H, W, N).permute(0, 3, 1, 2), gd.view(B, H, W, N).permute(0, 3, 1, 2)
                    if form == "fwd":
                        native.conv3x3_raw(xd, wd, bd, r4, True, 1, out=o4)
                    else:
                        native.conv3x3_raw(xd, wd, None, r4, False, 1, out=o4, gate=g4)
                elif form == "fwd":
                    native.gemm_nt_raw(xd, wd, bd, rd, M, N, K, True, out)
                else:
                    native.gemm_nt_dx_raw(xd, wd, rd, gd, M, N, K, out)
                ref = ref_f if form == "fwd" else ref_d
                assert _close(out[:M], ref, 8e-3), (cfg, sp, form, float((out[:M].float().cpu() - ref).abs().max()))
                assert bool((out[M:] == guard).all()), (cfg, sp, form)
    finally:
        lib.dskd_gemm_nt_tune(-1, 0)


@pytest.mark.parametrize("side_first", [True, False])
def test_fused_bottleneck_output_with_a_second_consumer(side_first):
    """A stage output y that feeds the next Bottleneck AND another consumer (the neck's lateral convolution, the feature
    loss): autograd sums the two gradients of y, in place into whichever arrived first.  When the next block's tagged
    ("already masked by y > 0") gradient arrives first the sum keeps the Python object and with it the tag: the
    producer must mask again, because the other addend is not masked.  The side consumer is built before / after the
    second block (both arrival orders) with a gradient that is large exactly where y == 0; compared with the unfused
    chain on the same kernels (which always masks), input and weight gradients of the FIRST block."""
    g = torch.Generator().manual_seed(11 + side_first)
    B, Cin, P, H, W = 2, 256, 64, 19, 23
    N = 4 * P

    def mk(co, ci, k):
        return (torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5).bfloat16()

    x = torch.randn(B, Cin, H, W, generator=g).relu().bfloat16()
    blocks = [dict(w1=mk(P, Cin, 1), w2=mk(P, P, 3), w3=mk(N, P, 1)), dict(w1=mk(P, N, 1), w2=mk(P, P, 3), w3=mk(N, P, 1))]
    for blk in blocks:
        for k, co in (("b1", P), ("b2", P), ("b3", N)):
            blk[k] = (torch.randn(co, generator=g) * 0.1 - (0.3 if k == "b3" else 0.0)).bfloat16()       # many zeros in y
    up = torch.randn(B, N, H, W, generator=g).bfloat16()
    side_w = (torch.randn(B, N, H, W, generator=g) * 4.0).bfloat16()

    def run(fused):
        cl = dict(memory_format=torch.channels_last)
        xd = x.to(DEV).contiguous(**cl).requires_grad_(True)
        leaves, hd, side, y_first = [xd], xd, None, None
        for bi, blk in enumerate(blocks):
            ws = {k: blk[k].to(DEV).contiguous(**cl).requires_grad_(True) for k in ("w1", "w2", "w3")}
            bs = {k: blk[k].to(DEV) for k in ("b1", "b2", "b3")}
            if bi == 0:
                leaves += [ws["w1"], ws["w2"], ws["w3"]]
            if bi == 1 and side_first:
                side = (hd * side_w.to(DEV).contiguous(**cl)).float().sum()
            if fused:
                hd = native.bottleneck(hd, ws["w1"], bs["b1"], ws["w2"], bs["b2"], ws["w3"], bs["b3"], None, None, 1,
                                       x_is_relu=True)
            else:
                o = native.conv1x1(hd, ws["w1"], bs["b1"], None, True, 1)
                o = native.conv3x3(o, ws["w2"], bs["b2"], None, True, 1)
                hd = native.conv1x1(o, ws["w3"], bs["b3"], hd, True, 1)
            if bi == 0:
                y_first = hd
        if not side_first:
            side = (y_first * side_w.to(DEV).contiguous(**cl)).float().sum()
        total = (hd * up.to(DEV).contiguous(**cl)).float().sum() + side
        return y_first.detach(), torch.autograd.grad(total, leaves)

    y_f, g_f = run(True)
    y_u, g_u = run(False)
    assert torch.equal(y_f, y_u) and float((y_f == 0).float().mean()) > 0.2
    xmask = (x.to(DEV) > 0)
    for i, (a, u) in enumerate(zip(g_f, g_u)):
        if i == 0:
            u = u * xmask
        assert _close(a, u.float().cpu(), 1.0e-2), (i, float((a.float() - u.float()).abs().max()), float(u.abs().max()))


# --------------------------------------------------------------------------- weight-gradient GEMM (gemm_tn_kernel)
@pytest.mark.parametrize("M,N,K", [(1024, 128, 128), (5000 + 37, 256, 384), (88892, 256, 256), (20011, 1024, 256),
                                   (16800, 256, 1024), (4200, 512, 2048)])
def test_gemm_tn_vs_float_reference(M, N, K):
    """dskd_gemm_tn: ``g^T @ x`` (dW = dY^T X) in f32 against the f32 CPU product of the same bf16 inputs; token counts that
    are no multiple of the 32-token stage or of the split, row strides larger than the used columns, accumulation onto
    an existing buffer.  Tolerance 2e-3 of the largest magnitude (f32 accumulation in a different order; atomics)."""
    g = torch.Generator().manual_seed(M % 97 + N + K)
    gm = torch.randn(M, N + 64, generator=g).bfloat16()
    xm = torch.randn(M, K, generator=g).bfloat16()
    ref = gm[:, :N].float().t() @ xm.float()
    gd, xd = gm.to(DEV)[:, :N], xm.to(DEV)
    assert native.gemm_tn_ok(gd, xd)
    out = native.gemm_tn(gd, xd)
    assert out.dtype == torch.float32 and out.shape == (N, K)
    assert _close(out, ref, 2e-3), float((out.cpu() - ref).abs().max()) / float(ref.abs().max())
    out2 = native.gemm_tn(gd, xd, out=out.clone())                      # accumulates
    assert _close(out2, 2 * ref, 2e-3)
    # the bf16 forms.  dskd_gemm_tn_bf16: split-K planes in a scratch + a fixed-order reduction: deterministic (two calls
    # bit-equal) and independent of what the scratch held before
    skey = (gd.device, torch.cuda.current_stream(gd.device).cuda_stream)
    native._tn_scratch.pop(skey, None)
    b1 = native.gemm_tn_bf16(gd, xd)
    native._tn_scratch[skey].fill_(0x7F)                           # NaN-ish garbage in every plane
    b2 = native.gemm_tn_bf16(gd, xd)
    assert b1.dtype == torch.bfloat16 and _close(b1, ref, 6e-3) and torch.equal(b1, b2)
    # ... with the bias gradient (column sums of g) as a by-product of the same two launches (dskd_gemm_tn_bias_bf16): the
    # weight gradient is bit-equal with the plain form, the sums match the f32 column sums of the bf16 input
    native._tn_scratch[skey].fill_(0x7F)
    b3, db = native.gemm_tn_bf16(gd, xd, want_bias=True)
    assert torch.equal(b3, b1) and db.shape == (N,) and db.dtype == torch.bfloat16
    ref_db = gm[:, :N].float().sum(0)
    assert _close(db, ref_db, 6e-3), float((db.float().cpu() - ref_db).abs().max()) / float(ref_db.abs().max())
    assert torch.equal(native.gemm_tn_bf16(gd, xd, want_bias=True)[1], db)
    # the atomic form (persistent accumulator + dskd_cvt_clear): same values, and the accumulator is zero again
    a1 = native.gemm_tn_bf16_atomic(gd, xd)
    a2 = native.gemm_tn_bf16_atomic(gd, xd)
    assert _close(a1, ref, 6e-3) and _close(a2, ref, 6e-3)
    assert float(native._tn_acc[(N, K, gd.device, skey[1])].abs().max()) == 0.0
