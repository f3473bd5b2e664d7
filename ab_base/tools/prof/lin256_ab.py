"""lin256_kernel ([88 892, 256] x [256, N]^T + bias) of several builds of the library in ONE process, rounds interleaved
(cdna_hip_programming.md rules 24, 25).  Usage: python tools/prof/lin256_ab.py LIB [LIB ...]   (paths of libdskd_*.so;
tools/prof/build_variant.sh NAME -D... builds them)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dskd_amd import native
libs = []
for path in sys.argv[1:] or [native.lib_path()]:
    lib = C.CDLL(os.path.abspath(path))
    for name in ("dskd_lin256_fwd", "dskd_lin256_pack", "dskd_lin256_packed_bytes"):
        getattr(lib, name).restype, getattr(lib, name).argtypes = native._SIGNATURES[name]
    libs.append((os.path.basename(path), lib))
dev, T = torch.device("cuda:0"), 88892
x = torch.randn(T, 256, device=dev).bfloat16()
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for N in (256, 384, 128):
    w = (torch.randn(N, 256, device=dev) / 16).bfloat16()
    b = (torch.randn(N, device=dev) * 0.1).bfloat16()
    ref = torch.addmm(b.float(), x.float(), w.float().t())
    runs = []
    for name, lib in libs:
        pk = torch.empty(int(lib.dskd_lin256_packed_bytes(N)), dtype=torch.uint8, device=dev)
        assert lib.dskd_lin256_pack(w.data_ptr(), pk.data_ptr(), N, 256, 0, native.DTYPE_BF16, st) == 0
        y = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
        fn = (lambda lib=lib, pk=pk, y=y: lib.dskd_lin256_fwd(x.data_ptr(), pk.data_ptr(), b.data_ptr(), y.data_ptr(), T, N, 256, 0,
                                                              native.DTYPE_BF16, st))
        assert fn() == 0
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        runs.append([name, fn, 1e9, err])
    for _ in range(5):
        for r in runs:
            r[1](); e0.record()
            for _ in range(10):
                r[1]()
            e1.record(); torch.cuda.synchronize()
            r[2] = min(r[2], e0.elapsed_time(e1) * 100)
    mb = T * (256 + N) * 2 / 1e6
    print(f"N={N}: " + "   ".join(f"{n} {t:6.1f} us ({mb / t / 1e6 * 1e6 / 1e3:5.2f} TB/s, err {e:.1e})" for n, _, t, e in runs), flush=True)
