"""dW of the class branch ([7 200, 80]^T [7 200, 256], BASELINE configs[1]): one library GEMM against token chunks as bmm
batches against the split-K kernel on the gradient padded to 128 columns.  Device time per variant from torch.profiler
(the host cannot issue these launches as fast as the GPU runs them)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

from dskd_amd import native  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (M, N, K) in [(7200, 80, 256), (7200, 70, 256), (7200, 4, 256), (7200, 68, 256)]:
    gy = torch.randn(M, N, generator=g).bfloat16().to(dev)
    x = torch.randn(M, K, generator=g).bfloat16().to(dev)
    variants = {
        "one GEMM": lambda: gy.t() @ x,
        "bmm 16 x 450 + sum": lambda: torch.bmm(gy.view(16, 450, N).transpose(1, 2), x.view(16, 450, K)).sum(0),
        "pad to 128 + split-K": lambda: native.gemm_tn_bf16(torch.nn.functional.pad(gy, (0, 128 - N)), x)[:N],
    }
    ref = (gy.float().t() @ x.float())
    for name, fn in variants.items():
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        dev_us = sum(e.self_device_time_total for e in prof.key_averages()) / 10
        err = float((out.float() - ref).abs().max()) / float(ref.abs().max())
        print(f"N={N:3d}: {name:22s} {dev_us:7.1f} us of kernels per call   rel err {err:.1e}")
