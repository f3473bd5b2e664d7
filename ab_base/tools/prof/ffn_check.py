"""Fused MFMA FFN (csrc/ffn_mfma.hip) vs a float reference and vs the GEMM chain; timings."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dskd_amd import native

dev = "cuda"
torch.manual_seed(0)
T = int(os.environ.get("T", 88892))
x = torch.randn(T, 256, device=dev).bfloat16()
w1 = (torch.randn(1024, 256, device=dev) / 16).bfloat16()
b1 = (torch.randn(1024, device=dev) * 0.1).bfloat16()
w2 = (torch.randn(256, 1024, device=dev) / 32).bfloat16()
b2 = (torch.randn(256, device=dev) * 0.1).bfloat16()
gy = torch.randn(T, 256, device=dev).bfloat16()
pf, pb = native.ffn_pack(w1, w2)

def relerr(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())

# --- p = 0 against a float reference on the same rounded inputs
y, h = native.ffn_fwd_raw(x, pf, b1, b2, 0.0, True)
href = torch.relu(x.float() @ w1.float().t() + b1.float())
print("H   vs float ref:", relerr(h, href))
yref = h.float() @ w2.float().t() + b2.float()
print("Y   vs float ref (on stored H):", relerr(y, yref))
y_e, _ = native.ffn_fwd_raw(x, pf, b1, b2, 0.0, False)
print("eval kernel == train kernel:", bool(torch.equal(y_e, y)))
gh, gx = native.ffn_bwd_raw(gy, h, pb, 0.0)
ghref = (gy.float() @ w2.float()) * (h != 0)
print("gH  vs float ref:", relerr(gh, ghref))
gxref = gh.float() @ w1.float()
print("gX  vs float ref (on stored gH):", relerr(gx, gxref))
# ragged tail: nothing written past T (buffers are exactly T rows; checked by a guard allocation pattern)
# --- dropout: same mask as dskd_dropout_fwd under the same key
p = 0.1
native._drop_calls = 1000
yd, hd = native.ffn_fwd_raw(x, pf, b1, b2, p, True)
chain = torch._addmm_activation(b1, x, w1.t())
native._drop_calls = 1000
native.dropout_(chain, p)
same_mask = ((hd != 0) == (chain != 0))
print("dropout: mask agreement with the chain:", float(same_mask.float().mean()), " drop rate among active:",
      float(((hd == 0) & (h != 0)).float().sum() / (h != 0).float().sum()))
print("dropout: H vs chain H:", relerr(hd, chain))
ghd, gxd = native.ffn_bwd_raw(gy, hd, pb, p)
ghc, _ = native.relu_dropout_bwd(gy @ w2, chain, p, want_colsum=False)
print("dropout: gH vs chain:", relerr(ghd, ghc), " gX vs chain:", relerr(gxd, ghc @ w1))

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def chain_fwd():
    hh = torch._addmm_activation(b1, x, w1.t()); native.dropout_(hh, p); return torch.addmm(b2, hh, w2.t())
def chain_bwd():
    g1, _ = native.relu_dropout_bwd(gy @ w2, hd, p, want_colsum=False); return g1 @ w1
print("us  pack            :", round(timeit(lambda: native.ffn_pack(w1, w2)), 1))
print("us  fused fwd train :", round(timeit(lambda: native.ffn_fwd_raw(x, pf, b1, b2, p, True)), 1))
print("us  fused fwd p=0 H :", round(timeit(lambda: native.ffn_fwd_raw(x, pf, b1, b2, 0.0, True)), 1))
print("us  fused fwd eval  :", round(timeit(lambda: native.ffn_fwd_raw(x, pf, b1, b2, 0.0, False)), 1))
print("us  chain fwd       :", round(timeit(chain_fwd), 1))
print("us  fused bwd       :", round(timeit(lambda: native.ffn_bwd_raw(gy, hd, pb, p)), 1))
print("us  chain bwd       :", round(timeit(chain_bwd), 1))
