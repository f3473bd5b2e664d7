import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dskd_amd import native
dev = "cuda"
for T in (128, 200, 256, 4133):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(T, 256, generator=g).bfloat16().to(dev)
    w1 = (torch.randn(1024, 256, generator=g) / 16).bfloat16().to(dev); b1 = (torch.randn(1024, generator=g) * 0.1).bfloat16().to(dev)
    w2 = (torch.randn(256, 1024, generator=g) / 32).bfloat16().to(dev); b2 = (torch.randn(256, generator=g) * 0.1).bfloat16().to(dev)
    gy = torch.randn(T, 256, generator=g).bfloat16().to(dev)
    pf, pb = native.ffn_pack(w1, w2)
    y, h = native.ffn_fwd_raw(x, pf, b1, b2, 0.0, True)
    for rep in range(3):
        gh, gx, cs = native.ffn_bwd_raw(gy, h, pb, 0.0, want_colsum=True)
        ghref = (gy.float() @ w2.float()) * (h != 0)
        err = (gh.float() - ghref).abs()
        rows = (err.max(1).values > 0.02 * ghref.abs().max()).nonzero().flatten().tolist()
        cols = (err.max(0).values > 0.02 * ghref.abs().max()).nonzero().flatten().tolist()
        print(T, rep, "gh max err", float(err.max()), "bad rows", rows[:12], len(rows), "bad cols", cols[:12], len(cols),
              " cs err", float((cs - ghref.sum(0)).abs().max()), " gx err", float((gx.float() - gh.float() @ w1.float()).abs().max()))
