"""Which convolution of a teacher Bottleneck is not bit-reproducible at the small test image, and which kernel runs it?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_model as T
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
cfg, m = T._build(seed=13)
m.to(dev).train()
tb = m.teacher_model.backbone
g = torch.Generator().manual_seed(1)
for lname, cin, hw in (("layer2", 512, (24, 32)), ("layer3", 1024, (12, 16)), ("layer4", 2048, (6, 8))):
    blk = getattr(tb, lname)[1]
    x = torch.randn(2, cin, *hw, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    convs = [(n, mod) for n, mod in blk.named_modules() if isinstance(mod, torch.nn.Conv2d)]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        outs = [blk(x).clone() for _ in range(8)]
        nd = sum(1 for o in outs[1:] if not torch.equal(o, outs[0]))
        print(lname, "block[1] runs differing from the first:", nd, "of 7")
        for n, conv in convs:
            xi = torch.randn(2, conv.in_channels, *hw, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
            o = [conv(xi).clone() for _ in range(8)]
            nd = sum(1 for t in o[1:] if not torch.equal(t, o[0]))
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                conv(xi); torch.cuda.synchronize()
            ks = [e.key[:90] for e in prof.key_averages() if e.self_device_time_total > 0]
            print(f"   {lname}.1.{n} k={conv.kernel_size} in={conv.in_channels} out={conv.out_channels}: differing {nd}/7; kernels: {ks}")
