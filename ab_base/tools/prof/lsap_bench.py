"""dskd_lsap_batched: the register-resident kernel (default for problems wider than 64) with 1 / 2 columns per thread
(dskd_lsap_tune(2 | 3)) against the one-wave kernel of round 1 (dskd_lsap_tune(1)), interleaved, us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dskd_amd import native
lib = native.load()
rng = np.random.default_rng(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for label, shapes in (("24 x (300 x 17)  [bench]", [(300, 17)] * 24), ("24 x (300 x 40)", [(300, 40)] * 24), ("1 x (300 x 64)", [(300, 64)]),
                      ("1 x (300 x 110)", [(300, 110)]), ("24 x (300 x 110)", [(300, 110)] * 24), ("1 x (300 x 310)", [(300, 310)])):
    mats = [rng.random(s).astype(np.float32) for s in shapes]
    flat = torch.cat([torch.from_numpy(m).reshape(-1) for m in mats]).cuda()
    nr, nc = [m.shape[0] for m in mats], [m.shape[1] for m in mats]
    offs = np.cumsum([0] + [m.size for m in mats])[:-1].tolist()
    MODES = (0, 1, 2, 3)
    best = {m: 1e9 for m in MODES}
    for _ in range(4):
        for mode in MODES:
            lib.dskd_lsap_tune(mode)
            native.lsap_batched(flat, nr, nc, offs); e0.record()
            for _ in range(5):
                native.lsap_batched(flat, nr, nc, offs)
            e1.record(); torch.cuda.synchronize()
            best[mode] = min(best[mode], e0.elapsed_time(e1) / 5 * 1e3)
    lib.dskd_lsap_tune(0)
    print(f"{label:26s} auto {best[0]:7.1f} us   one wave (r1) {best[1]:7.1f}   columns per thread 1: {best[2]:7.1f}  2: {best[3]:7.1f}", flush=True)
