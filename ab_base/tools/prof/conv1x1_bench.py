"""Microbenchmark: every 1x1 convolution shape of ResNet-50 + ChannelMapper at B=4, 800x1333 (bf16, channels_last):
dskd_gemm_nt (conv + bias + residual + ReLU in one launch) against F.conv2d + native.bias_act, forward and dX."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_FIND_MODE", "1")
import torch, torch.nn.functional as F
from dskd_amd import native
native.load()
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
B = 4
# (name, K, N, H, W, stride, residual, relu, count per model)
shapes = [("l1.conv1a", 64, 64, 200, 334, 1, 0, 1, 1), ("l1.conv1", 256, 64, 200, 334, 1, 0, 1, 2),
          ("l1.conv3", 64, 256, 200, 334, 1, 1, 1, 3), ("l1.down", 64, 256, 200, 334, 1, 0, 0, 1),
          ("l2.conv1a", 256, 128, 200, 334, 1, 0, 1, 1), ("l2.conv1", 512, 128, 100, 167, 1, 0, 1, 3),
          ("l2.conv3", 128, 512, 100, 167, 1, 1, 1, 4), ("l2.down", 256, 512, 200, 334, 2, 0, 0, 1),
          ("l3.conv1a", 512, 256, 100, 167, 1, 0, 1, 1), ("l3.conv1", 1024, 256, 50, 84, 1, 0, 1, 5),
          ("l3.conv3", 256, 1024, 50, 84, 1, 1, 1, 6), ("l3.down", 512, 1024, 100, 167, 2, 0, 0, 1),
          ("l4.conv1a", 1024, 512, 50, 84, 1, 0, 1, 1), ("l4.conv1", 2048, 512, 25, 42, 1, 0, 1, 2),
          ("l4.conv3", 512, 2048, 25, 42, 1, 1, 1, 3), ("l4.down", 1024, 2048, 50, 84, 2, 0, 0, 1),
          ("neck0", 512, 256, 100, 167, 1, 0, 0, 1), ("neck1", 1024, 256, 50, 84, 1, 0, 0, 1),
          ("neck2", 2048, 256, 25, 42, 1, 0, 0, 1)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot_a = tot_b = tot_da = tot_db = 0.0
print(f"{'layer':10s} {'K':>5s} {'N':>5s} {'M':>7s}  own_us  lib_us  TF/s  GB/s | dX own  dX lib")
for name, K, N, H, W, s, res, relu, cnt in shapes:
    x = torch.randn(B, K, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(N, K, 1, 1, device=dev) / K ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(N, device=dev).bfloat16()
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    idt = torch.randn(B, N, Ho, Wo, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) if res else None
    M = B * Ho * Wo
    with torch.no_grad():
        ta = timeit(lambda: native.conv1x1(x, w, b, idt, relu, s))
        tb = timeit(lambda: native.bias_act(F.conv2d(x, w, None, stride=s), b, idt, bool(relu)))
        y1 = native.conv1x1(x, w, b, idt, relu, s)
        y2 = native.bias_act(F.conv2d(x, w, None, stride=s), b, idt, bool(relu))
        err = float((y1.float() - y2.float()).abs().max()) / float(y2.float().abs().max())
        g = torch.randn(B, N, Ho, Wo, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        tda = tdb = 0.0
        if s == 1:
            wt = w.view(N, K).t().contiguous()
            gx = torch.empty_like(x)
            tda = timeit(lambda: native.gemm_nt_raw(g, wt, None, None, M, K, N, False, gx))
            tdb = timeit(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                                     [True, False, False])[0])
    flops = 2.0 * M * N * K
    byts = 2.0 * (M * (K if s == 1 else K) + M * N * (2 if res else 1) + N * K)
    print(f"{name:10s} {K:5d} {N:5d} {M:7d}  {ta:6.1f}  {tb:6.1f}  {flops / ta / 1e6:5.0f}  {byts / ta / 1e3:5.0f} | {tda:6.1f}  {tdb:6.1f}   relerr {err:.1e}")
    tot_a += ta * cnt; tot_b += tb * cnt; tot_da += tda * cnt; tot_db += tdb * cnt
print(f"one model forward, all 1x1: own {tot_a / 1e3:.2f} ms, library + bias_act {tot_b / 1e3:.2f} ms;  dX (stride-1 layers, all stages): own {tot_da / 1e3:.2f} ms, library {tot_db / 1e3:.2f} ms")
