"""gemm_tn per layer shape of the step: us, TFLOP/s (scratch; B=4 800x1333 token counts)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dskd_amd import native
native.load()
dev = torch.device("cuda:0")
shapes = [("enc.ffn.dW1", 88892, 1024, 256), ("enc.ffn.dW2", 88892, 256, 1024), ("enc.lin256", 88892, 256, 256),
          ("enc.so_aw", 88892, 384, 256), ("l2.conv1", 66800, 128, 512), ("l2.conv3", 66800, 512, 128),
          ("l2.conv1a", 267200, 128, 256), ("l3.conv1", 16800, 256, 1024), ("l3.conv3", 16800, 1024, 256),
          ("l3.conv1a", 66800, 256, 512), ("l4.conv1", 4200, 512, 2048), ("l4.conv3", 4200, 2048, 512),
          ("l4.conv1a", 16800, 512, 1024), ("neck0", 66800, 256, 512), ("neck1", 16800, 256, 1024), ("neck2", 4200, 256, 2048),
          ("dec.lin", 1200, 256, 256), ("dec.ffn", 1200, 1024, 256)]
tot = 0.0
for name, M, N, K in shapes:
    g = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    if not native.gemm_tn_ok(g, x):
        print(f"{name:12s} M={M:6d} N={N:4d} K={K:4d}: not taken"); continue
    res = []
    for fn in (native.gemm_tn_bf16, native.gemm_tn_bf16_atomic):
        for _ in range(3): fn(g, x)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn(g, x)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    us = res[0]
    print(f"{name:12s} M={M:6d} N={N:4d} K={K:4d}: planes+reduce {us:7.1f} us ({2.0 * M * N * K / us / 1e6:5.0f} TF/s)   atomics+cvt_clear {res[1]:7.1f} us")
