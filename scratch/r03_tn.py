import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dskd_amd import native
from dskd_amd.transformer import _token_chunk
native.load()
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(88892, 256, 256), (88892, 384, 256), (88892, 1024, 256), (88892, 256, 1024), (267200, 256, 128), (267200, 128, 256),
                  (66800, 512, 128), (66800, 128, 512), (16800, 256, 1024), (16800, 1024, 256), (4200, 512, 2048), (4200, 2048, 512), (66800, 256, 512)]:
    g = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    out = torch.zeros(N, K, device=dev)
    t = timeit(lambda: native.gemm_tn(g, x, out=out))
    tz = timeit(lambda: native.gemm_tn(g, x))
    c = _token_chunk(M, N * K)
    def lib():
        if c:
            nb = M // c
            return torch.bmm(g.view(nb, c, N).transpose(1, 2), x.view(nb, c, K)).sum(0, dtype=torch.float32)
        return g.t() @ x
    tl = timeit(lib)
    tm = timeit(lambda: g.t() @ x)
    print(f"M={M} N={N} K={K}: gemm_tn {t:.1f} us (+zero fill {tz:.1f}) {2.0*M*N*K/t/1e6:.0f} TF/s {2.0*M*(N+K)/t/1e3:.0f} GB/s | bmm+sum {tl:.1f} us | mm {tm:.1f} us")
