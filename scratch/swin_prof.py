import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from torch.profiler import profile, ProfilerActivity
from dskd_amd.swin import SwinTransformer
dev = "cuda:0"
m = SwinTransformer(embed_dims=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), out_indices=(1, 2, 3), drop_path_rate=0.2).to(dev).train()
x = torch.randn(4, 3, 800, 1333, device=dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outs = m(x)
    sum(o.float().pow(2).mean() for o in outs).backward()
for _ in range(3): step()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): step()
e1.record(); torch.cuda.synchronize()
print("Swin-T fwd+bwd B=4 800x1333 bf16:", e0.elapsed_time(e1) / 5, "ms")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=70))
