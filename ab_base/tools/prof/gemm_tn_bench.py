"""gemm_tn per layer shape of the step: us, TFLOP/s (B=4 800x1333 token counts)."""
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
    lib = native.load()
    # interleaved rounds, minimum per variant (the chip's clock sags under sustained load: whatever runs first looks faster)
    variants = ((native.gemm_tn_bf16, -1), (native.gemm_tn_bf16, -2), (native.gemm_tn_bf16, -3), (native.gemm_tn_bf16_atomic, -1))
    res = [1e9] * len(variants)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for rnd in range(4):
        for vi, (fn, tile) in enumerate(variants):
            lib.dskd_gemm_nt_tune(tile, 0)          # -1: automatic, -2: 128 x 128 tiles, -3: 256 x 128 tiles
            fn(g, x)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10): fn(g, x)
            e1.record(); torch.cuda.synchronize()
            if rnd:
                res[vi] = min(res[vi], e0.elapsed_time(e1) / 10 * 1e3)
    lib.dskd_gemm_nt_tune(-1, 0)
    us = res[0]
    print(f"{name:12s} M={M:6d} N={N:4d} K={K:4d}: planes+reduce {us:7.1f} us ({2.0 * M * N * K / us / 1e6:5.0f} TF/s)   "
          f"128x128 {res[1]:6.1f}  256x128 {res[2]:6.1f}   atomics+cvt_clear {res[3]:7.1f} us")
