"""Where do the calls of a few torch ops of one training step come from (file:line inside dskd_amd / bench.py), and how big
are they?  Usage: python tools/prof/op_sites.py [op ...]   (default: cat stack; ".to" / ".contiguous" / ".float" ...: Tensor methods that
returned NEW memory)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from dskd_amd.runner import build_optimizer
ops = sys.argv[1:] or ["cat", "stack"]
dev = torch.device("cuda:0")
cfg, model = bench.build_models(dev, 111, None)
model = model.to(memory_format=torch.channels_last); model.teacher_model.to(memory_format=torch.channels_last)
model.lazy_log = True
model.bbox_head.graph_head = False
opt = build_optimizer(model, cfg.optimizer[0])
data, synth = bench.make_batch(4, cfg.num_prev, 111, dev)
data["img"] = data["img"].contiguous(memory_format=torch.channels_last)
ahead = model.teacher_ahead()
for _ in range(3):
    bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
torch.cuda.synchronize()
counts = collections.Counter()
sizes = collections.defaultdict(int)
orig = {}
def wrap(name):
    f = getattr(torch, name)
    orig[name] = f
    def g(*a, **k):
        site = "?"
        for fr in reversed(traceback.extract_stack()[:-1]):
            if "dskd_amd" in fr.filename or fr.filename.endswith("bench.py"):
                site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                break
        out = f(*a, **k)
        counts[(name, site)] += 1
        if torch.is_tensor(out):
            sizes[(name, site)] = max(sizes[(name, site)], out.numel() * out.element_size())
        return out
    setattr(torch, name, g)
def wrap_method(name):
    f = getattr(torch.Tensor, name)
    def g(self, *a, **k):
        out = f(self, *a, **k)
        if torch.is_tensor(out) and out is not self and out.is_cuda and (name.endswith("_") or out.data_ptr() != self.data_ptr()):
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "dskd_amd" in fr.filename or fr.filename.endswith("bench.py"):
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            counts[("." + name, site)] += 1
            sizes[("." + name, site)] = max(sizes[("." + name, site)], out.numel() * out.element_size())
        return out
    setattr(torch.Tensor, name, g)
for o in ops:
    if o.startswith("."):
        wrap_method(o[1:])
    else:
        wrap(o)
bench.train_step(model, model, opt, data, synth, torch.bfloat16, ahead=ahead)
torch.cuda.synchronize()
for (name, site), c in counts.most_common(40):
    print(f"{c:4d}  torch.{name:6s} {site:45s} largest result {sizes[(name, site)] / 1e3:10.1f} KB")
