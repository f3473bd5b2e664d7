"""Encoder-shape MSDA backward (B=4, bf16): time per call with HIP events, for A/B runs of DSKD_MSDA_PULL_STREAM; under
rocprofv3 --kernel-trace the csv shows whether the pull launch overlaps the gather / matrix-core launches."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, R + "/tests")
import torch  # noqa: E402

from dskd_amd import native  # noqa: E402
from test_gpu_kernels import SHAPES_FULL, _encoder_like_inputs  # noqa: E402

B = 4
value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, B, 41, 2.5, torch.bfloat16)
args = (value.cuda(), SHAPES_FULL, loc.cuda(), attn.cuda())
g = go.cuda()
n = int(os.environ.get("N", "50"))
for _ in range(5):
    native.msda_backward_raw(*args, g)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    native.msda_backward_raw(*args, g)
b.record()
torch.cuda.synchronize()
print(f"DSKD_MSDA_PULL_STREAM={os.environ.get('DSKD_MSDA_PULL_STREAM', '(unset)')}: {a.elapsed_time(b) / n * 1e3:.1f} us per backward call")
