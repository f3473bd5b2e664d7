import sys, os; sys.path.insert(0,'.')
import torch
sys.path.insert(0,'tests')
from dskd_amd import native
from test_gpu_kernels import _encoder_like_inputs, SHAPES_FULL
value, loc, attn, go = _encoder_like_inputs(SHAPES_FULL, 1, 41, 2.5)
args = (value.cuda(), SHAPES_FULL, loc.cuda(), attn.cuda(), go.cuda())
def t(n=5):
    native.msda_backward_raw(*args); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): native.msda_backward_raw(*args)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for dbg in ("0",):
    os.environ["DSKD_DBG"]=dbg
    print("dbg",dbg,"(1=skip flush,2=skip ds_add)", round(t(),1),"us")
os.environ["DSKD_DBG"]="0"; os.environ["DSKD_MSDA_BWD"]="v1"
print("v1", round(t(),1))
