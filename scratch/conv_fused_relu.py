"""conv + bias + relu: separate ATen ops vs aten::miopen_convolution_relu (MIOpen fusion API)."""
import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = "cuda:0"
def timeit(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 4
shapes = [(64, 64, 3, 200, 334), (64, 256, 1, 200, 334), (256, 64, 1, 200, 334), (128, 128, 3, 100, 167), (128, 512, 1, 100, 167),
          (512, 128, 1, 100, 167), (256, 256, 3, 50, 84), (256, 1024, 1, 50, 84), (1024, 256, 1, 50, 84), (512, 512, 3, 25, 42)]
for ci, co, k, H, W in shapes:
    x = torch.randn(B, ci, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, k, k, device=dev, dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device=dev, dtype=torch.bfloat16)
    pad = k // 2
    with torch.no_grad():
        t_conv = timeit(lambda: F.conv2d(x, w, None, 1, pad))
        t_sep = timeit(lambda: F.relu(F.conv2d(x, w, b, 1, pad), inplace=True))
        try:
            y1 = torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [pad, pad], [1, 1], 1)
            y0 = F.relu(F.conv2d(x, w, b, 1, pad))
            err = float((y1.float() - y0.float()).abs().max())
            t_fused = timeit(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [pad, pad], [1, 1], 1))
            cl = y1.is_contiguous(memory_format=torch.channels_last)
        except Exception as e:
            t_fused, err, cl = float("nan"), str(e)[:80], None
    print(f"{ci:5d}->{co:5d} k{k} @{H}x{W}: conv only {t_conv:7.1f}  conv+bias+relu {t_sep:7.1f}  miopen fused {t_fused:7.1f} us  maxdiff {err} cl={cl}", flush=True)
