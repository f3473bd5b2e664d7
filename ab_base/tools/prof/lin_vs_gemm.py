import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dskd_amd import native
native.load()
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = 88892
x = torch.randn(T, 256, device=dev).bfloat16()
for N in (256, 384, 128, 64):
    w = (torch.randn(N, 256, device=dev) / 16).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    y = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    tg = timeit(lambda: native.gemm_nt_raw(x, w, b, None, T, N, 256, False, y))
    tl = float("nan")
    if N % 32 == 0 and N >= 32:
        pk = native.lin256_pack(w)
        tl = timeit(lambda: native.lin256(x, pk, N, b))
        err = float((native.lin256(x, pk, N, b).float() - y.float()).abs().max())
    ta = timeit(lambda: torch.addmm(b, x, w.t()))
    print(f"T={T} N={N}: gemm_nt {tg:.1f} us, lin256 {tl:.1f} us, addmm {ta:.1f} us, maxdiff {err:.3g}")
# K = 1024 -> 256 and 256 -> 1024 (the FFN's two GEMMs as plain GEMMs, for reference)
for (N, K) in ((1024, 256), (256, 1024)):
    xx = torch.randn(T, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / 16).bfloat16()
    y = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    tg = timeit(lambda: native.gemm_nt_raw(xx, w, None, None, T, N, K, False, y))
    ta = timeit(lambda: torch.mm(xx, w.t()))
    print(f"T={T} N={N} K={K}: gemm_nt {tg:.1f} us ({2.0*T*N*K/tg/1e6:.0f} TF/s), mm {ta:.1f} us")
# decoder-size GEMMs (1200 tokens)
for (M, N, K) in ((1200, 256, 256), (7200, 256, 256), (1200, 1024, 256), (1200, 256, 1024)):
    xx = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / 16).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tg = timeit(lambda: native.gemm_nt_raw(xx, w, b, None, M, N, K, False, y))
    ta = timeit(lambda: torch.addmm(b, xx, w.t()))
    print(f"M={M} N={N} K={K}: gemm_nt {tg:.1f} us, addmm {ta:.1f} us")
