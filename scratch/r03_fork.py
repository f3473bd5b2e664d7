import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scratch"))
import torch
import msda_only as M
from dskd_amd import native
def t(n=20):
    for _ in range(3): native.msda_backward_raw(*M.args, M.g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): native.msda_backward_raw(*M.args, M.g)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("encoder backward per call (events, back to back):", round(t(), 1), "us; fork =", os.environ.get("DSKD_TMP_FORK"))
