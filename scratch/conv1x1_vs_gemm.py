"""1x1 convolutions of ResNet-50 at 800x1333 (B=4, bf16, channels_last): MIOpen vs hipBLASLt GEMM."""
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
shapes = [  # (Cin, Cout, H, W)
    (64, 64, 200, 334), (64, 256, 200, 334), (256, 64, 200, 334),
    (256, 128, 200, 334), (128, 512, 100, 167), (512, 128, 100, 167),
    (512, 256, 100, 167), (256, 1024, 50, 84), (1024, 256, 50, 84),
    (1024, 512, 50, 84), (512, 2048, 25, 42), (2048, 512, 25, 42),
]
tot_c = tot_g = tot_cb = tot_gb = 0
for ci, co, H, W in shapes:
    x = torch.randn(B, ci, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(co, ci, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(co, device=dev, dtype=torch.bfloat16)
    gy = torch.randn(B, co, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x2 = x.detach().permute(0, 2, 3, 1).reshape(-1, ci).requires_grad_(True)      # view of channels_last memory
    w2 = w.detach().view(co, ci).requires_grad_(True)
    gy2 = gy.permute(0, 2, 3, 1).reshape(-1, co)
    with torch.no_grad():
        tc = timeit(lambda: F.relu(F.conv2d(x, w, b), inplace=True))
        tg = timeit(lambda: torch._addmm_activation(b, x2, w2.t()))
    def cb():
        y = F.relu(F.conv2d(x, w, b), inplace=True); y.backward(gy); x.grad = None; w.grad = None
    def gb():
        y = torch.relu_(F.linear(x2, w2, b)); y.backward(gy2); x2.grad = None; w2.grad = None
    tcb, tgb = timeit(cb, n=10, w=3), timeit(gb, n=10, w=3)
    tot_c += tc; tot_g += tg; tot_cb += tcb; tot_gb += tgb
    print(f"{ci:5d}->{co:5d} @{H}x{W}: fwd conv {tc:7.1f} us  gemm {tg:7.1f} us | fwd+bwd conv {tcb:7.1f} us  gemm {tgb:7.1f} us", flush=True)
print(f"sum fwd conv {tot_c:.0f} gemm {tot_g:.0f} | fwd+bwd conv {tot_cb:.0f} gemm {tot_gb:.0f}")
