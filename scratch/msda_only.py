import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import torch
from dskd_amd import native
from test_gpu_kernels import _encoder_like_inputs, SHAPES_FULL
value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, 4, 41, 2.5, torch.bfloat16)
args = (value.cuda(), SHAPES_FULL, loc.cuda(), attn.cuda())
g = go.cuda()
for _ in range(5):
    native.msda_forward_raw(*args)
    native.msda_backward_raw(*args, g)
torch.cuda.synchronize()
