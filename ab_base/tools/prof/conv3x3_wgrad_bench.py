"""3x3 weight gradients of ResNet-50 at B=4, 800x1333 (bf16, channels_last): native.conv3x3_wgrad (split-K MFMA kernel +
reduction) against aten::convolution_backward (MIOpen igemm_wrw + its workspace helpers), rounds interleaved, wall time of
back-to-back calls (so the library's helper launches count)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_FIND_MODE", "1")
import torch
from dskd_amd import native
dev = torch.device("cuda:0")
SWEEP = len(sys.argv) > 1 and sys.argv[1] == "sweep"      # also forced tiles x splits through dskd_gemm_nt_tune(-2 | -3, splits)
SPLITS = (1, 2, 3, 4, 7, 14, 28, 56)
cl = torch.channels_last
shapes = [("l2.conv2a", 128, 200, 334, 2, 1), ("l2.conv2", 128, 100, 167, 1, 3), ("l3.conv2a", 256, 100, 167, 2, 1),
          ("l3.conv2", 256, 50, 84, 1, 5), ("l4.conv2a", 512, 50, 84, 2, 1), ("l4.conv2", 512, 25, 42, 1, 2)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = [0.0, 0.0]
for name, C, H, W, s, cnt in shapes:
    x = torch.randn(4, C, H, W, device=dev).bfloat16().contiguous(memory_format=cl)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    g = torch.randn(4, C, Ho, Wo, device=dev).bfloat16().contiguous(memory_format=cl)
    w = torch.randn(C, C, 3, 3, device=dev).bfloat16().contiguous(memory_format=cl)
    lib = native.load()
    def own(cfg, sp):
        def f():
            lib.dskd_gemm_nt_tune(cfg, sp)
            r = native.conv3x3_wgrad(g, x, s)
            lib.dskd_gemm_nt_tune(-1, 0)
            return r
        return f
    variants = [("auto", own(-1, 0))] + ([(f"{'128' if c == -2 else '256'}x128/s{sp}", own(c, sp)) for c in (-2, -3) for sp in SPLITS
                                          if not (c == -3 and C % 256)] if SWEEP else [])
    fns = [v[1] for v in variants] + [lambda: torch.ops.aten.convolution_backward(g, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]]
    a, b = fns[0](), fns[-1]()
    err = float((a.float() - b.float()).abs().max() / b.float().abs().max())
    best = [1e9] * len(fns)
    for _ in range(4):
        for i, fn in enumerate(fns):
            fn(); e0.record()
            for _ in range(8):
                fn()
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / 8 * 1e3)
    fl = 2.0 * 4 * Ho * Wo * C * 9 * C
    print(f"{name:10s} own {best[0]:6.1f} us ({fl / best[0] / 1e6:4.0f} TF)   library {best[-1]:6.1f} us   max diff {err:.1e}   "
          + "  ".join(f"{v[0]} {t:5.1f}" for v, t in zip(variants[1:], best[1:-1])), flush=True)
    tot[0] += best[0] * cnt; tot[1] += best[-1] * cnt
print(f"13 convolutions of one backward: own {tot[0] / 1e3:.3f} ms, library {tot[1] / 1e3:.3f} ms")
