"""Which ATen ops (name, input shapes) launch the elementwise / copy / reduce kernels of one step, and from where?"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import collections
import torch, bench
from torch.profiler import profile, ProfilerActivity
from dskd_amd.runner import build_optimizer
dev = torch.device('cuda:0')
cfg, model = bench.build_models(dev, 111, None)
model = model.to(memory_format=torch.channels_last); model.teacher_model.to(memory_format=torch.channels_last)
model.lazy_log = True
model.bbox_head.graph_head = False      # op-level attribution: replayed graphs hide the ops
opt = build_optimizer(model, cfg.optimizer[0])
data, synth = bench.make_batch(4, cfg.num_prev, 111, dev)
data["img"] = data["img"].contiguous(memory_format=torch.channels_last)
ahead = model.teacher_ahead()
for _ in range(3):
    bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0, set()])
tot = 0.0
for ev in prof.events():
    if not ev.kernels:
        continue
    for k in ev.kernels:
        n = k.name
        if n.startswith("void at::native") or "at::native" in n or "Memcpy" in n or "Memset" in n:
            stack = [s for s in (ev.stack or []) if "dskd_amd" in s or "bench.py" in s]
            where = stack[0].split("/")[-1][:70] if stack else "?"
            key = (ev.name, str(ev.input_shapes)[:90], where)
            agg[key][0] += k.duration
            agg[key][1] += 1
            agg[key][2].add(n[:60])
            tot += k.duration
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
os.makedirs("gpurun_out", exist_ok=True)
with open(os.environ.get("ATEN_TAIL_OUT", "gpurun_out/aten_tail.txt"), "w") as f:
    f.write(f"ATen / memcpy kernels of one step: {tot / 1e3:.2f} ms\n")
    for (name, shapes, where), (t, n, ks) in rows[:120]:
        f.write(f"{t / 1e3:7.3f} ms {n:4d}  {name:32s} {shapes:90s} {where}\n")
print(open(os.environ.get("ATEN_TAIL_OUT", "gpurun_out/aten_tail.txt")).read()[:6000])
