import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from dskd_amd import native
from dskd_amd.runner import build_optimizer
dev=torch.device('cuda:0')
cfg, model = bench.build_models(dev, 111, None)
model = model.to(memory_format=torch.channels_last); model.teacher_model.to(memory_format=torch.channels_last)
model.lazy_log=True
model.bbox_head.graph_head = False      # op-level attribution: replayed graphs hide the ops
opt = build_optimizer(model, cfg.optimizer[0])
data, synth = bench.make_batch(4, cfg.num_prev, 111, dev)
data["img"]=data["img"].contiguous(memory_format=torch.channels_last)
ahead = model.teacher_ahead()
for _ in range(3): bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
    torch.cuda.synchronize()
os.makedirs('gpurun_out', exist_ok=True)
with open('gpurun_out/torch_prof_shapes.txt','w') as f:
    f.write(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=40, max_shapes_column_width=90))
with open('gpurun_out/torch_prof_ops.txt','w') as f:
    f.write(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=50))
print("done")
with open('gpurun_out/torch_prof_cpu.txt','w') as f:
    f.write(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=70, max_name_column_width=60))
evs = prof.key_averages(group_by_input_shape=True)
rows = sorted(evs, key=lambda e: -getattr(e, "self_device_time_total", 0))
with open('gpurun_out/torch_prof_tsv.txt', 'w') as f:
    for e in rows[:600]:
        f.write(f"{e.key[:60]}\t{e.self_device_time_total/1e3:.3f}ms\t{e.count}\t{str(e.input_shapes)[:160]}\n")
