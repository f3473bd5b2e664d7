"""MSDeformAttn forward, encoder shape (B=4, bf16): the product against the two ablation builds in ONE
process, rounds interleaved: -DDSKD_FWD_ABLATE=1 = every corner load, no unpack / FMA work; =2 = staging + arithmetic, no
corner loads.  Windowed kernel (queries == pixels) and the plain kernel (one query dropped, so the generic path runs).
Usage: python tools/prof/msda_fwd_ablation.py PRODUCT.so LOADS_ONLY.so MATH_ONLY.so"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, R + "/tools/prof")
import torch
from dskd_amd import native
import msda_only                      # builds the encoder-shape inputs (and runs 5 warm-up launches of the default library)
libs = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    lib.dskd_msda_fwd.restype, lib.dskd_msda_fwd.argtypes = native._SIGNATURES["dskd_msda_fwd"]
    libs.append((os.path.basename(path), lib))
value, shapes, loc, attn = msda_only.args
B, Nv = value.shape[0], value.shape[1]
ss, ls, _ = native._geom(shapes)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for label, Nq in (("windowed kernel (student encoder)", Nv), ("plain kernel (generic path; the teacher's fused forward is this loop)", Nv - 1)):
    l2, a2 = loc[:, :Nq].contiguous(), attn[:, :Nq].contiguous()
    out = torch.empty(B, Nq, 256, dtype=torch.bfloat16, device="cuda")
    fns = [(n, (lambda lib=lib: lib.dskd_msda_fwd(value.data_ptr(), ss, ls, l2.data_ptr(), a2.data_ptr(), out.data_ptr(), B, Nv, Nq, 8,
                                                  32, 4, 4, native.DTYPE_BF16, st))) for n, lib in libs]
    best = [1e9] * len(fns)
    for _ in range(5):
        for i, (_, fn) in enumerate(fns):
            assert fn() == 0
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) * 100)
    print(label + ":  " + "   ".join(f"{n} {t:6.1f} us" for (n, _), t in zip(fns, best)), flush=True)
