"""Microbenchmark + agreement check of the tile configurations of csrc/gemm_nt.hip (gemm_nt_kernel = cfg 0, gemm_big_kernel
cfg 1..6, with and without the split-K remainder) on every 1x1 / 3x3 convolution shape of ResNet-50 + ChannelMapper at
B=4, 800x1333 (bf16, channels_last): forward (bias + residual + ReLU) and the input-gradient form (gate + residual).
Interleaved rounds in one process, random data (cdna_hip_programming.md rules 24, 25).
Usage: python tools/prof/gemm_tiles_bench.py [auto | epi]      ("auto": only cfg 0 against the automatic choice; "epi": the
small tile with the register epilogue (cfg 7), the LDS epilogue (cfg 8), the LDS epilogue without the early residual / gate
reads (cfg 9))"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dskd_amd import native
lib = native.load()
dev = torch.device("cuda:0")
B = 4
AUTO_ONLY = len(sys.argv) > 1 and sys.argv[1] == "auto"
EPI = len(sys.argv) > 1 and sys.argv[1] == "epi"
# (name, K, N, H, W, stride, residual, relu, count per model forward)
shapes1 = [("l1.conv1a", 64, 64, 200, 334, 1, 0, 1, 1), ("l1.conv1", 256, 64, 200, 334, 1, 0, 1, 2),
           ("l1.conv3", 64, 256, 200, 334, 1, 1, 1, 3), ("l1.down", 64, 256, 200, 334, 1, 0, 0, 1),
           ("l2.conv1a", 256, 128, 200, 334, 1, 0, 1, 1), ("l2.conv1", 512, 128, 100, 167, 1, 0, 1, 3),
           ("l2.conv3", 128, 512, 100, 167, 1, 1, 1, 4), ("l2.down", 256, 512, 200, 334, 2, 0, 0, 1),
           ("l3.conv1a", 512, 256, 100, 167, 1, 0, 1, 1), ("l3.conv1", 1024, 256, 50, 84, 1, 0, 1, 5),
           ("l3.conv3", 256, 1024, 50, 84, 1, 1, 1, 6), ("l3.down", 512, 1024, 100, 167, 2, 0, 0, 1),
           ("l4.conv1a", 1024, 512, 50, 84, 1, 0, 1, 1), ("l4.conv1", 2048, 512, 25, 42, 1, 0, 1, 2),
           ("l4.conv3", 512, 2048, 25, 42, 1, 1, 1, 3), ("l4.down", 1024, 2048, 50, 84, 2, 0, 0, 1),
           ("neck0", 512, 256, 100, 167, 1, 0, 0, 1), ("neck1", 1024, 256, 50, 84, 1, 0, 0, 1),
           ("neck2", 2048, 256, 25, 42, 1, 0, 0, 1)]
# (name, C, H, W, stride, count)
shapes3 = [("l1.conv2", 64, 200, 334, 1, 3), ("l2.conv2a", 128, 200, 334, 2, 1), ("l2.conv2", 128, 100, 167, 1, 3),
           ("l3.conv2a", 256, 100, 167, 2, 1), ("l3.conv2", 256, 50, 84, 1, 5), ("l4.conv2a", 512, 50, 84, 2, 1),
           ("l4.conv2", 512, 25, 42, 1, 2)]
CFGS = [(0, 0)] + [(c, s) for c in range(1, 7) for s in (0, 1)]
NAMES = {0: "64x128", 1: "128x128/2", 2: "256x128/3", 3: "128x256/3", 4: "256x64/2", 5: "128x128/3", 6: "256x128/2"}


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def bench(fn, variants, rounds=3, n=8):
    """variants: list of (cfg, splits); returns {variant: best us} with the rounds interleaved."""
    best = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ok = []
    for v in variants:
        lib.dskd_gemm_nt_tune(*v)
        try:
            fn(); fn()
            ok.append(v)
        except native.NativeError:
            pass
    torch.cuda.synchronize()
    for _ in range(rounds):
        for v in ok:
            lib.dskd_gemm_nt_tune(*v)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / n * 1e3
            best[v] = min(best.get(v, 1e9), t)
    lib.dskd_gemm_nt_tune(-1, 0)
    return best


def report(name, flops, fn, out, tot):
    variants = [(0, 0), (-1, 0)] if AUTO_ONLY else [(0, 0), (7, 0), (8, 0), (9, 0), (-1, 0)] if EPI else CFGS + [(-1, 0)]
    lib.dskd_gemm_nt_tune(0, 0)
    fn()
    ref = out.float().clone()
    worst = 0.0
    for v in variants[1:]:
        lib.dskd_gemm_nt_tune(*v)
        out.zero_()
        try:
            fn()
        except native.NativeError:
            continue
        worst = max(worst, float((out.float() - ref).abs().max()) / float(ref.abs().max()))
    t = bench(fn, variants)
    t0, ta = t[(0, 0)], t[(-1, 0)]
    cols = ""
    if EPI:
        k = [t.get((c, 0)) for c in (7, 8, 9)]
        cols += " register / LDS / LDS-no-prefetch epilogue " + " ".join("   -  " if v is None else f"{v:6.1f}" for v in k) + " |"
        tot[2] += min([v for v in k if v is not None] + [t0])
    if not AUTO_ONLY and not EPI:
        for c in range(1, 7):
            a, b = t.get((c, 0)), t.get((c, 1))
            cols += "    -  " if a is None else f" {min(a, b):5.1f}{'*' if a < b else ' '}"
        bestv = min((v for v in t if v[0] >= 0), key=lambda v: t[v])
        cols += f" | best {NAMES[bestv[0]]}{'+split' if bestv[1] == 0 and bestv[0] else ''} {t[bestv]:5.1f}"
    print(f"{name:14s} small {t0:6.1f} ({flops / t0 / 1e6:4.0f} TF) |{cols} | auto {ta:6.1f} ({flops / ta / 1e6:4.0f} TF)  maxdiff {worst:.1e}",
          flush=True)
    tot[0] += t0; tot[1] += ta


print("columns: us per launch; big tiles 1..6 = " + ", ".join(NAMES[c] for c in range(1, 7)) + " (* = faster WITH the split-K remainder)")
tot_f, tot_d, tot_3, tot_3d = [0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0]
g = torch.Generator(device=dev).manual_seed(0)
for name, K, N, H, W, s, res, relu, cnt in shapes1:
    x = cl(torch.randn(B, K, H, W, device=dev, generator=g).bfloat16())
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g).bfloat16()
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    idt = cl(torch.randn(B, N, Ho, Wo, device=dev, generator=g).bfloat16()) if res else None
    M = B * Ho * Wo
    y = cl(torch.empty(B, N, Ho, Wo, device=dev, dtype=torch.bfloat16))
    args = (0, 0, 0, 0, 0) if s == 1 else (s, Ho, Wo, H, W)
    t = [0, 0, 0]
    report(name, 2.0 * M * N * K, lambda: native.gemm_nt_raw(x, w, b, idt, M, N, K, bool(relu), y, *args), y, t)
    tot_f[0] += t[0] * cnt; tot_f[1] += t[1] * cnt; tot_f[2] += t[2] * cnt
    if s == 1:
        gy = cl(torch.randn(B, N, Ho, Wo, device=dev, generator=g).bfloat16())
        wt = w.t().contiguous()
        gx = torch.empty_like(x)
        gres = cl(torch.randn(B, K, H, W, device=dev, generator=g).bfloat16()) if name.endswith("conv1") else None
        t = [0, 0, 0]
        report(name + ".dX", 2.0 * M * N * K, lambda: native.gemm_nt_dx_raw(gy, wt, gres, x, M, K, N, gx), gx, t)
        tot_d[0] += t[0] * cnt; tot_d[1] += t[1] * cnt; tot_d[2] += t[2] * cnt
    del x, y, idt
for name, C, H, W, s, cnt in shapes3:
    x = cl(torch.randn(B, C, H, W, device=dev, generator=g).bfloat16())
    w = cl((torch.randn(C, C, 3, 3, device=dev, generator=g) / (9 * C) ** 0.5).bfloat16())
    b = torch.randn(C, device=dev, generator=g).bfloat16()
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    y = cl(torch.empty(B, C, Ho, Wo, device=dev, dtype=torch.bfloat16))
    fl = 2.0 * B * Ho * Wo * C * 9 * C
    t = [0, 0, 0]
    report(name, fl, lambda: native.conv3x3_raw(x, w, b, None, True, s, out=y), y, t)
    tot_3[0] += t[0] * cnt; tot_3[1] += t[1] * cnt; tot_3[2] += t[2] * cnt
    if s == 1:
        gy = cl(torch.randn(B, C, H, W, device=dev, generator=g).bfloat16())
        gx = torch.empty_like(x)
        t = [0, 0, 0]
        report(name + ".dX", fl, lambda: native.conv3x3_raw(gy, w, None, None, False, 1, out=gx, gate=x), gx, t)
        tot_3d[0] += t[0] * cnt; tot_3d[1] += t[1] * cnt; tot_3d[2] += t[2] * cnt
for lab, t in (("1x1 forward, one model", tot_f), ("1x1 dX (stride-1 layers)", tot_d), ("3x3 forward, one model", tot_3),
               ("3x3 dX (stride-1 layers)", tot_3d)):
    print(f"{lab}: small tile {t[0] / 1e3:.3f} ms, best epilogue {t[2] / 1e3:.3f} ms, automatic choice {t[1] / 1e3:.3f} ms")
