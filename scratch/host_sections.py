"""Host (enqueue) time per section of the training step -- no synchronisation inside the step."""
import sys, os, time, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ.setdefault("MIOPEN_FIND_MODE", "1")
import torch, bench
from dskd_amd.runner import build_optimizer
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg, model = bench.build_models(dev, 111, None)
model = model.to(memory_format=torch.channels_last); model.teacher_model.to(memory_format=torch.channels_last)
model.lazy_log = True
opt = build_optimizer(model, cfg.optimizer[0])
data, synth = bench.make_batch(4, cfg.num_prev, 111, dev)
data["img"] = data["img"].contiguous(memory_format=torch.channels_last)
T = collections.defaultdict(float)
def timed(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[tag] += time.perf_counter() - t; return r
    setattr(obj, name, g)
head = model.bbox_head
timed(model, "extract_feat", "student backbone+neck fwd")
timed(head, "forward", "student transformer+heads fwd")
timed(head, "loss", "targets + losses")
timed(head.transformer.encoder, "forward", "  (encoder)")
timed(head.transformer.decoder, "forward", "  (decoder)")
ahead = model.teacher_ahead()
timed(ahead, "finish", "teacher decode/finish")
timed(ahead, "launch", "teacher enqueue (next batch)")
_bw = torch.Tensor.backward
def bw(self, *a, **k):
    t = time.perf_counter(); r = _bw(self, *a, **k); T["backward"] += time.perf_counter() - t; return r
torch.Tensor.backward = bw
timed(opt, "step", "optimizer.step")
_clip = torch.nn.utils.clip_grad_norm_
def clip(*a, **k):
    t = time.perf_counter(); r = _clip(*a, **k); T["clip_grad_norm"] += time.perf_counter() - t; return r
torch.nn.utils.clip_grad_norm_ = clip
for _ in range(5): bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
torch.cuda.synchronize(); T.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N): bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"host loop {host/N*1e3:.1f} ms/step, with final sync {tot/N*1e3:.1f} ms/step")
for k, v in T.items(): print(f"  {k:36s} {v/N*1e3:6.2f} ms")
