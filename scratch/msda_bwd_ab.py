"""Encoder-shape MSDA backward at the BASELINE size (B=4, bf16): per-variant time with HIP events.
  python scratch/msda_bwd_ab.py            (variants via DSKD_MSDA_PULL_LEVELS / DSKD_MSDA_BWD / DSKD_MSDA_PULL_MARGIN)
Offsets: grid-initialised (as the benchmark's random-init model: |offset| <= 4 px per level) or N(0, sigma)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + "/tests")
import torch  # noqa: E402

from dskd_amd import native  # noqa: E402
from test_gpu_kernels import SHAPES_FULL, _encoder_like_inputs  # noqa: E402

B = int(os.environ.get("AB_B", "4"))
dtype = torch.bfloat16 if os.environ.get("AB_DTYPE", "bf16") == "bf16" else torch.float32
value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, B, 41, float(os.environ.get("AB_SIGMA", "2.5")), dtype)
if os.environ.get("AB_GRID", "1") == "1":
    # the module's initialisation (sampling_offsets.bias = grid_init, weight = 0): head h points along direction h of 8,
    # point p at distance p + 1 pixels of the level -- what the benchmark's random-init model samples
    import math
    th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
    d = torch.stack([th.cos(), th.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]
    off = d.view(8, 1, 1, 2) * torch.arange(1, 5, dtype=torch.float32).view(1, 1, 4, 1)          # [8, 1, 4, 2] pixels
    pts = []
    for (H, W) in SHAPES_FULL:
        ys, xs = torch.meshgrid((torch.arange(H) + 0.5) / H, (torch.arange(W) + 0.5) / W, indexing="ij")
        pts.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
    ref = torch.cat(pts, 0)
    norm = torch.tensor([[w, h] for h, w in SHAPES_FULL], dtype=torch.float32).view(1, 1, 4, 1, 2)
    loc = (ref.view(1, -1, 1, 1, 1, 2) + (off.expand(8, 4, 4, 2).reshape(1, 1, 8, 4, 4, 2) / norm)).expand(B, -1, -1, -1, -1, -1).contiguous()
args = (value.cuda(), SHAPES_FULL, loc.cuda(), attn.cuda(), go.cuda())
if os.environ.get("AB_PPROF"):            # library built with -DDSKD_PULL_PROFILE: per-phase clock sums of the pull kernel
    import ctypes
    lib = native.load()
    lib.dskd_debug_pull_prof.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 32)()
    native.msda_backward_raw(*args)
    lib.dskd_debug_pull_prof(buf, 1)
    N = 5
    for _ in range(N):
        native.msda_backward_raw(*args)
    lib.dskd_debug_pull_prof(buf, 1)
    names = ["tables", "phaseA", "phaseB", "prefix", "place", "reduce"]
    for lv in range(4):
        tot = sum(buf[lv * 8 + k] for k in range(6))
        if tot:
            print("level", lv, " ".join(f"{names[k]}={100.0 * buf[lv * 8 + k] / tot:.1f}%" for k in range(6)),
                  f"total={tot / N:.3e} clocks summed over workgroups")
    sys.exit(0)
only = os.environ.get("AB_ONLY")
if only is not None:                      # one variant (for rocprofv3): environment as given by the caller
    for _ in range(10):
        native.msda_backward_raw(*args)
    torch.cuda.synchronize()
    sys.exit(0)


def timed(label, env):
    for k in ("DSKD_MSDA_PULL_LEVELS", "DSKD_MSDA_BWD", "DSKD_MSDA_PULL_MARGIN"):
        os.environ.pop(k, None)
    if os.environ.get("AB_MARGIN"):
        os.environ["DSKD_MSDA_PULL_MARGIN"] = os.environ["AB_MARGIN"]
    os.environ.update(env)
    for _ in range(3):
        out = native.msda_backward_raw(*args)
    torch.cuda.synchronize()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = native.msda_backward_raw(*args)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:34s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us per backward (incl. allocation / zero fill)", flush=True)
    return out


ref = timed("windowed (round 1)", {"DSKD_MSDA_BWD": "win"})
for label, env in [("pull 0+1, margin 6 (default)", {}), ("pull 0+1, margin 5", {"DSKD_MSDA_PULL_MARGIN": "5"}),
                   ("pull 0", {"DSKD_MSDA_PULL_LEVELS": "0"}), ("pull 1", {"DSKD_MSDA_PULL_LEVELS": "1"}),
                   ("pull 0+1+2+3", {"DSKD_MSDA_PULL_LEVELS": "0123"}), ("no pull (ws entry)", {"DSKD_MSDA_PULL_LEVELS": "none"})]:
    out = timed(label, env)
    err = float((out[0] - ref[0]).abs().max())
    print(f"{'':34s} max |grad_value - windowed| = {err:.3e}", flush=True)
