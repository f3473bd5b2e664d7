import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import torch
from dskd_amd import native
from test_gpu_kernels import SHAPES_FULL, _encoder_like_inputs
DEV = torch.device("cuda:0")
value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, 2, 49, 2.5, torch.bfloat16)
args = (value.to(DEV), SHAPES_FULL, loc.to(DEV), attn.to(DEV), go.to(DEV))
a = native.msda_backward_raw(*args)
b = native.msda_backward_raw(*args)
os.environ["DSKD_MSDA_BWD"] = "r2"
c = native.msda_backward_raw(*args)
for name, i in (("gv", 0), ("gl", 1), ("ga", 2)):
    x, y, z = a[i], b[i], c[i]
    print(name, "nan", int(torch.isnan(x).sum()), int(torch.isnan(z).sum()), "run-to-run max", float((x - y).abs().nan_to_num().max()),
          "vs r2 max", float((x - z).abs().nan_to_num().max()), "scale", float(z.abs().nan_to_num().max()))
d = (a[1] - c[1]).abs().nan_to_num()
idx = (d > 1e-3 * float(c[1].abs().max())).nonzero()
print("gl mismatches", idx.shape[0], idx[:10].tolist())
d = (a[1] - b[1]).abs()
idx = (d > 0).nonzero()
print("gl run-to-run mismatches", idx.shape[0], idx[:10].tolist())
if idx.shape[0]:
    i = tuple(idx[0].tolist())
    print(a[1][i], b[1][i], c[1][i])
