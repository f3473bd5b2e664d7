"""Layout experiment for the MSDA forward gather (DSKD_MSDA_VALUE_LAYOUT, experiment-only kernel variants):
is the gather bound by 128-B L2 -> L1 line fills of which the bf16 [B, Nv, heads, 32] layout uses half?

  std   [B, Nv, 8, 32] bf16   one corner of one head = 64 B = half a line, 4 lines per sample
  f32   [B, Nv, 8, 32] f32    one corner of one head = 128 B = one line,   4 lines per sample
  hm    [B, 8, Nv, 32] bf16   the corner pair (x0, x0+1) of a row is contiguous: 3 lines per sample on average
  pair  [B, 8, Nv, 2, 32] bf16 (pixel, right neighbour): the corner pair is one aligned line, 2 lines per sample

B=4, 100x167 .. 13x21, encoder-like locations (sigma 2.5 px).  Prints HIP-event time per launch and
equality of hm / pair with std; writes gpurun_out/msda_layout_ab.json.  N_ITER=2 for counter passes."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + "/tests")
import torch  # noqa: E402

from dskd_amd import native  # noqa: E402
from test_gpu_kernels import SHAPES_FULL, _encoder_like_inputs  # noqa: E402

N_ITER = int(os.environ.get("N_ITER", "20"))
B = 4
value, loc, attn, _ = _encoder_like_inputs(SHAPES_FULL, B, 41, 2.5, torch.bfloat16)
Nv = value.shape[1]
v = value.cuda()
locd, attd = loc.cuda(), attn.cuda()


def timed(fn, n=N_ITER):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


def make_hm():
    return v.permute(0, 2, 1, 3).contiguous()


def make_pair():
    big = torch.zeros(B, 8, Nv, 2, 32, dtype=v.dtype, device=v.device)
    vp = v.permute(0, 2, 1, 3)
    big[:, :, :, 0] = vp
    big[:, :, :-1, 1] = vp[:, :, 1:]
    return big


res = {}
os.environ.pop("DSKD_MSDA_VALUE_LAYOUT", None)
out0 = native.msda_forward_raw(v, SHAPES_FULL, locd, attd)
res["std_bf16_us"] = timed(lambda: native.msda_forward_raw(v, SHAPES_FULL, locd, attd))
vf = v.float()
res["std_f32_us"] = timed(lambda: native.msda_forward_raw(vf, SHAPES_FULL, locd, attd))
del vf

hm = make_hm()
os.environ["DSKD_MSDA_VALUE_LAYOUT"] = "hm"
hm_view = hm.view(B, Nv, 8, 32)
out1 = native.msda_forward_raw(hm_view, SHAPES_FULL, locd, attd)
res["hm_equal"] = bool(torch.equal(out1, out0))
res["hm_bf16_us"] = timed(lambda: native.msda_forward_raw(hm_view, SHAPES_FULL, locd, attd))

big = make_pair()
os.environ["DSKD_MSDA_VALUE_LAYOUT"] = "pair"
pair_view = big.view(-1)[: B * Nv * 256].view(B, Nv, 8, 32)      # the kernel indexes the whole [B, 8, Nv, 2, 32] buffer
out2 = native.msda_forward_raw(pair_view, SHAPES_FULL, locd, attd)
res["pair_equal"] = bool(torch.equal(out2, out0))
res["pair_maxdiff"] = float((out2.float() - out0.float()).abs().max())
res["pair_bf16_us"] = timed(lambda: native.msda_forward_raw(pair_view, SHAPES_FULL, locd, attd))
os.environ.pop("DSKD_MSDA_VALUE_LAYOUT")

res["make_hm_us(torch permute copy)"] = timed(make_hm, 5)
res["make_pair_us(torch, 3 kernels)"] = timed(make_pair, 5)
print(json.dumps(res, indent=1), flush=True)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
if N_ITER >= 10:
    with open(os.path.join(R, "gpurun_out", "msda_layout_ab.json"), "w") as f:
        json.dump(res, f, indent=1)
