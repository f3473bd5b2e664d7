"""A/B of the phased staging of the bf16 MSDA gather kernels (DSKD_MSDA_PHASES = 1 | 2 | 4) at the
BASELINE shape (B=4, 100x167 .. 13x21, bf16): HIP-event time per launch of the forward, the fused
no-grad forward and the backward (grad_loc/grad_attn kernel + the three windowed grad_value
kernels), and bit-equality of every phase count with the single-phase result.
Writes gpurun_out/msda_phases.json."""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + "/tests")
import torch  # noqa: E402

from dskd_amd import native  # noqa: E402
from test_gpu_kernels import SHAPES_FULL, _encoder_like_inputs  # noqa: E402

B = 4
value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, B, 41, 2.5, torch.bfloat16)
args = (value.cuda(), SHAPES_FULL, loc.cuda(), attn.cuda())
g = go.cuda()
gen = torch.Generator().manual_seed(3)
Nv = value.shape[1]
both = (torch.randn(B, Nv, 384, generator=gen) * 2).to(torch.bfloat16).cuda()
ref = torch.rand(B, Nv, 4, 2, generator=gen).cuda()
dec = (loc[:, :300].contiguous().cuda(), attn[:, :300].contiguous().cuda())
gd = go[:, :300].contiguous().cuda()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


res, base = {}, None
for ph in (1, 2, 4, 1):
    os.environ["DSKD_MSDA_PHASES"] = str(ph)
    out = native.msda_forward_raw(*args)
    fused = native.ms_deform_attn_fused(args[0], SHAPES_FULL, both, ref, 4, 4)
    gv, gl, ga = native.msda_backward_raw(*args, g)
    od = native.msda_forward_raw(args[0], SHAPES_FULL, *dec)
    torch.cuda.synchronize()
    cur = (out, fused, gl, ga, od)
    if base is None:
        base, gv0 = cur, gv
    eq = [bool(torch.equal(a, b)) for a, b in zip(cur, base)]
    gv_err = float((gv - gv0).abs().max())
    # backward timing includes the torch.zeros of grad_value and two empty_like (allocator only)
    t = {"fwd_us": timed(lambda: native.msda_forward_raw(*args)),
         "fwd_fused_us": timed(lambda: native.ms_deform_attn_fused(args[0], SHAPES_FULL, both, ref, 4, 4)),
         "bwd_us": timed(lambda: native.msda_backward_raw(*args, g)),
         "fwd_dec_us": timed(lambda: native.msda_forward_raw(args[0], SHAPES_FULL, *dec)),
         "equal_to_ph1(out,fused,grad_loc,grad_attn,out_dec)": eq, "grad_value_maxdiff": gv_err}
    res.setdefault(f"phases_{ph}", []).append(t)
    print(ph, json.dumps(t), flush=True)

os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
with open(os.path.join(R, "gpurun_out", "msda_phases.json"), "w") as f:
    json.dump(res, f, indent=1)
