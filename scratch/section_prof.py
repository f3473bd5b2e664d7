"""Device time and kernel count per forward section of the student step (torch profiler ranges)."""
import sys, os, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ.setdefault("MIOPEN_FIND_MODE", "1")
import torch, bench
from torch.profiler import profile, ProfilerActivity, record_function
from dskd_amd.runner import build_optimizer
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg, model = bench.build_models(dev, 111, None)
model = model.to(memory_format=torch.channels_last); model.teacher_model.to(memory_format=torch.channels_last)
model.lazy_log = True
opt = build_optimizer(model, cfg.optimizer[0])
data, synth = bench.make_batch(4, cfg.num_prev, 111, dev)
data["img"] = data["img"].contiguous(memory_format=torch.channels_last)
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function("SEC:" + tag):
            return f(*a, **k)
    setattr(obj, name, g)
head = model.bbox_head
wrap(model.backbone, "forward", "s.backbone")
wrap(model.neck, "forward", "s.neck")
wrap(head.transformer.encoder, "forward", "s.encoder")
wrap(head.transformer.decoder, "forward", "s.decoder")
wrap(head, "get_targets_all_layers", "s.targets")
wrap(head, "loss_layers_dense", "s.det_losses")
wrap(head, "loss", "s.loss_total")
wrap(head, "forward", "s.head_forward_total")
wrap(opt, "step", "optimizer")
ahead = model.teacher_ahead(); ahead.use_graphs = False
wrap(ahead, "finish", "teacher.decode")
wrap(ahead, "launch", "teacher.forward")
for _ in range(4): bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
    torch.cuda.synchronize()
evs = prof.events()
kern = [e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA]
secs = [e for e in evs if e.name.startswith("SEC:") and e.device_type == torch.autograd.DeviceType.CPU]
# attribute kernels to sections through correlation: kernel launch (cpu) time inside the section's cpu range
launches = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and e.kernels]
tot = collections.defaultdict(lambda: [0.0, 0])
allk = [0.0, 0]
for e in launches:
    t = e.time_range.start
    dur = sum(k.duration for k in e.kernels); n = len(e.kernels)
    allk[0] += dur; allk[1] += n
    for s in secs:
        if s.time_range.start <= t <= s.time_range.end:
            tot[s.name][0] += dur; tot[s.name][1] += n
print(f"all kernels launched from op calls: {allk[0]/1e3:.2f} ms, {allk[1]} kernels")
for k, (d, n) in sorted(tot.items()):
    print(f"  {k:28s} {d/1e3:7.2f} ms  {n:5d} kernels")
