"""Which part of the student's head breaks hipGraph capture of its backward?  Debug only.
usage: graph_region_dbg.py enc|dec   (env toggles select kernel variants)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import torch
import test_gpu_model as T
from dskd_amd import native
from dskd_amd.utils import GraphedFunction
from torch.nn.utils.stateless import _reparametrize_module
region = sys.argv[1]
dev = torch.device("cuda:0")
cfg, m = T._build(seed=13, num_query=100)
m.to(dev).train()
tr = m.bbox_head.transformer
shapes = [(24, 32), (12, 16), (6, 8), (3, 4)]
Nv = sum(h * w for h, w in shapes)
g = torch.Generator().manual_seed(1)
B = 2
if os.environ.get("DBG_TORCH_LN"):
    def ln(h, res, norm, p=0.0, pos=None, want_q=False):
        y = torch.nn.functional.layer_norm((res + h).float(), (256,), norm.weight, norm.bias, norm.eps).to(h.dtype)
        return y, ((y + pos.to(y.dtype)) if want_q else None)
    native.add_layer_norm = ln
if region == "enc":
    mod = tr.encoder
    x = torch.randn(B, Nv, 256, generator=g).to(dev).requires_grad_(True)
    pos = torch.randn(B, Nv, 256, generator=g).to(dev).requires_grad_(True)
    vr = torch.ones(B, 4, 2, device=dev)
    ref = tr.get_reference_points(shapes, vr, dev)
    def body(x, pos):
        return (mod(query=x, key=None, value=None, query_pos=pos, query_key_padding_mask=None, spatial_shapes=shapes,
                    reference_points=ref, level_start_index=None, valid_ratios=vr, tokens_batch_first=True),)
    args = [x, pos]
else:
    mod = tr.decoder
    mem = torch.randn(B, Nv, 256, generator=g).to(dev).to(torch.bfloat16).requires_grad_(True)
    q = torch.randn(100, B, 256, generator=g).to(dev).requires_grad_(True)
    qp = torch.randn(100, B, 256, generator=g).to(dev).requires_grad_(True)
    refp = torch.rand(B, 100, 2, generator=g).to(dev)
    vr = torch.ones(B, 4, 2, device=dev)
    def body(mem, q, qp):
        hs, _ = mod(query=q, key=None, value=mem, query_pos=qp, key_padding_mask=None, reference_points=refp,
                    spatial_shapes=shapes, level_start_index=None, valid_ratios=vr, reg_branches=None, value_batch_first=True)
        return (hs,)
    args = [mem, q, qp]
names = [n for n, p in mod.named_parameters() if p.requires_grad]
params = [p for n, p in mod.named_parameters() if p.requires_grad]
views = params
na = len(args)
def fn(*a):
    with _reparametrize_module(mod, dict(zip(names, a[na:])), tie_weights=False, strict=False):
        return body(*a[:na])
amp = None if os.environ.get("DBG_FP32") else torch.bfloat16
for _ in range(2):        # eager steps first
    with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
        out = fn(*args, *views)
    torch.autograd.grad(out[0].float().sum(), args + views, allow_unused=True)
torch.cuda.synchronize(); print("eager ok", flush=True)
os.environ["DSKD_GRAPH_TRACE"] = "1"
gf = GraphedFunction(fn, args, views, verify=True, autocast_dtype=amp)
out = gf(*args, *views)
out[0].float().sum().backward()
torch.cuda.synchronize(); print("GRAPH OK", region, flush=True)
