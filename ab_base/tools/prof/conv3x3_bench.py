"""Microbenchmark: the 3x3 convolutions of ResNet-50 at B=4, 800x1333 (bf16 channels_last): dskd_conv3x3 (conv + bias + ReLU)
against F.conv2d + native.bias_act, forward and dX."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_FIND_MODE", "1")
import torch, torch.nn.functional as F
from dskd_amd import native
native.load()
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
B = 4
shapes = [("l1.conv2", 64, 200, 334, 1, 3), ("l2.conv2a", 128, 200, 334, 2, 1), ("l2.conv2", 128, 100, 167, 1, 3),
          ("l3.conv2a", 256, 100, 167, 2, 1), ("l3.conv2", 256, 50, 84, 1, 5), ("l4.conv2a", 512, 50, 84, 2, 1),
          ("l4.conv2", 512, 25, 42, 1, 2)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ta_tot = tb_tot = tda = tdb = 0.0
for name, C, H, W, s, cnt in shapes:
    x = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device=dev).bfloat16()
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    with torch.no_grad():
        ta = timeit(lambda: native.conv3x3_raw(x, w, b, None, True, s))
        tb = timeit(lambda: native.bias_act(F.conv2d(x, w, None, stride=s, padding=1), b, None, True))
        y1 = native.conv3x3_raw(x, w, b, None, True, s); y2 = native.bias_act(F.conv2d(x, w, None, stride=s, padding=1), b, None, True)
        err = float((y1.float() - y2.float()).abs().max()) / float(y2.float().abs().max())
        g = torch.randn(B, C, Ho, Wo, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        da = db = 0.0
        if s == 1:
            wt = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
            da = timeit(lambda: native.conv3x3_raw(g, wt, None, None, False, 1))
            db = timeit(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0])
    fl = 2.0 * B * Ho * Wo * C * C * 9
    print(f"{name:10s} C={C:4d} {H}x{W} s={s}: own {ta:6.1f} us ({fl/ta/1e6:4.0f} TF/s)  lib+bias_act {tb:6.1f} us | dX own {da:6.1f} lib {db:6.1f}  relerr {err:.1e}")
    ta_tot += ta * cnt; tb_tot += tb * cnt; tda += da * cnt; tdb += db * cnt
print(f"one model forward, all 3x3: own {ta_tot/1e3:.2f} ms, library + bias_act {tb_tot/1e3:.2f} ms; dX (stride 1): own {tda/1e3:.2f} ms, library {tdb/1e3:.2f} ms")
