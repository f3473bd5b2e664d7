import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from dskd_amd import native
dev = "cuda:0"
def timeit(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = 88892
h = torch.randn(T, 256, device=dev, dtype=torch.bfloat16); res = torch.randn_like(h)
pos = torch.randn(22223, 256, device=dev)
norm = torch.nn.LayerNorm(256).to(dev)
with torch.no_grad():
    print("add_ln fwd (inference, no q): %.1f us" % timeit(lambda: native.add_layer_norm(h, res, norm, 0.0)))
    print("add_ln fwd (inference, +q):   %.1f us" % timeit(lambda: native.add_layer_norm(h, res, norm, 0.0, pos, True)))
hg, rg = h.clone().requires_grad_(True), res.clone().requires_grad_(True)
def fb(p, q):
    y, qq = native.add_layer_norm(hg, rg, norm, p, pos if q else None, q)
    ((y.float().sum() + (qq.float().sum() if q else 0))).backward()
    hg.grad = None; rg.grad = None
y, qq = native.add_layer_norm(hg, rg, norm, 0.1, pos, True)
gy = torch.randn_like(y)
def bw():
    torch.autograd.grad((y, qq), (hg, rg), (gy, gy), retain_graph=True)
print("add_ln fwd (train p=0.1, +q): %.1f us" % timeit(lambda: native.add_layer_norm(hg, rg, norm, 0.1, pos, True)))
print("add_ln bwd (p=0.1, dq):       %.1f us (incl. zeros + casts)" % timeit(bw))
g = torch.randn(T, 1024, device=dev, dtype=torch.bfloat16)
print("colsum 256:  %.1f us   colsum 1024: %.1f us" % (timeit(lambda: native.colsum(h)), timeit(lambda: native.colsum(g))))
yd = torch.relu(torch.randn(T, 1024, device=dev, dtype=torch.bfloat16))
print("relu_dropout_bwd 1024: %.1f us   dropout_fwd 1024: %.1f us" % (timeit(lambda: native.relu_dropout_bwd(g, yd, 0.1)), timeit(lambda: native.dropout_(yd, 0.1))))
x4 = torch.randn(4, 256, 200, 334, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
b4 = torch.randn(256, device=dev, dtype=torch.bfloat16)
print("bias_act 4x256x200x334 (+id, relu): %.1f us" % timeit(lambda: native.bias_act(x4, b4, x4, True)))
