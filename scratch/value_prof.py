"""Per-phase cycle breakdown of the windowed grad_value kernels (library built with
-DDSKD_VALUE_PROFILE).  Slots: 0 init/zero, 1 prepass, 2 main loop (wave 0), 3 barrier wait, 4 flush."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import torch
from dskd_amd import native
from test_gpu_kernels import _encoder_like_inputs, SHAPES_FULL
value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, 4, 41, 2.5, torch.bfloat16)
args = (value.cuda(), SHAPES_FULL, loc.cuda(), attn.cuda())
g = go.cuda()
lib = native.load()
buf = (ctypes.c_ulonglong * 64)()
native.msda_backward_raw(*args, g)
lib.dskd_debug_value_prof(buf, 1)
N = 5
for _ in range(N):
    native.msda_backward_raw(*args, g)
lib.dskd_debug_value_prof(buf, 1)
names = ["init", "prepass", "loop", "barrier", "flush"]
for v in range(3):
    tot = sum(buf[v * 8 + k] for k in range(5))
    print("variant", v, " ".join(f"{names[k]}={buf[v*8+k]/N/1e5:.1f}" for k in range(5)), f"total={tot/N/1e5:.1f} (x1e5 ticks of 100MHz = ms summed over blocks)")
