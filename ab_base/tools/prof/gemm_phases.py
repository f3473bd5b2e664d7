"""Per-phase shader clocks of gemm_nt_kernel / gemm_big_kernel (build: tools/prof/build_variant.sh gemmprof -DDSKD_GEMM_PROFILE,
run with DSKD_HIP_LIB=tools/prof/libs/libdskd_gemmprof.so): every workgroup's wave 0 stamps start / first stage landed / loop
end / epilogue end (s_memtime) + the 100 MHz wall clock at start and end, into a buffer nothing else reads."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dskd_amd import native
lib = native.load()
lib.dskd_gemm_nt_profile.restype, lib.dskd_gemm_nt_profile.argtypes = C.c_int, [C.c_void_p]
dev = torch.device("cuda:0")
B = 4


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def run(name, K, N, H, W, res, cfgs, conv3=False):
    g = torch.Generator(device=dev).manual_seed(0)
    M = B * H * W
    if conv3:
        x = cl(torch.randn(B, K, H, W, device=dev, generator=g).bfloat16())
        w = cl((torch.randn(N, K, 3, 3, device=dev, generator=g) / (9 * K) ** 0.5).bfloat16())
    else:
        x = cl(torch.randn(B, K, H, W, device=dev, generator=g).bfloat16())
        w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g).bfloat16()
    idt = cl(torch.randn(B, N, H, W, device=dev, generator=g).bfloat16()) if res else None
    y = cl(torch.empty(B, N, H, W, device=dev, dtype=torch.bfloat16))
    buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)

    def fn():
        if conv3:
            native.conv3x3_raw(x, w, b, idt, True, 1, out=y)
        else:
            native.gemm_nt_raw(x, w, b, idt, M, N, K, True, y)
    for cfg, sp in cfgs:
        lib.dskd_gemm_nt_tune(cfg, sp)
        lib.dskd_gemm_nt_profile(None)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        buf.zero_()
        lib.dskd_gemm_nt_profile(buf.data_ptr())
        fn()
        torch.cuda.synchronize()
        lib.dskd_gemm_nt_profile(None)
        p = buf.view(-1, 8).cpu()
        p = p[p[:, 0] != 0]
        n = p.shape[0]
        t0, t1, t2, t3, t4 = (p[:, i].double() for i in range(5))
        big = cfg != 0
        pro, loop = (t1 - t0), (t2 - t1)
        epi = (t4 - (t3 if big else t2))
        bar = (t3 - t2) if big else torch.zeros_like(t0)
        wall0, wall1 = p[:, 6].double(), p[:, 7].double()
        span = (wall1.max() - wall0.min()) / 100.0           # us
        life = ((wall1 - wall0) / 100.0)
        start_spread = (wall0.max() - wall0.min()) / 100.0
        clk = ((t4 - t0) / ((wall1 - wall0).clamp(min=1) * 10.0)).median()    # cycles per ns = GHz
        cu = ((p[:, 5] >> 32) * 4096 + ((p[:, 5] & 0xFFFFFFFF) >> 8 & 0xF) + (((p[:, 5] & 0xFFFFFFFF) >> 13) & 0x7) * 16)
        ncu = len(set(cu.tolist()))
        nk = (9 * K if conv3 else K) // 64
        print(f"{name:10s} cfg {cfg} sp {sp}: {us:6.1f} us/launch | {n:4d} WGs on {ncu:3d} (xcc,se,cu) ids, in-kernel span {span:5.1f} us, "
              f"starts within {start_spread:4.1f} us, WG life med {life.median():5.1f} max {life.max():5.1f} us, clock {clk:4.2f} GHz | "
              f"cycles med: prologue {pro.median():6.0f}  loop {loop.median():7.0f} ({loop.median() / nk:5.0f}/stage)  "
              f"barrier {bar.median():5.0f}  epilogue {epi.median():6.0f}", flush=True)
    lib.dskd_gemm_nt_tune(-1, 0)


cf = [(0, 0), (1, 1), (2, 1), (2, 0), (5, 1)]
run("l3.conv1a", 512, 256, 100, 167, 0, cf)
run("l4.conv1a", 1024, 512, 50, 84, 0, cf)
run("l3.conv3", 256, 1024, 50, 84, 1, cf)
run("l3.conv2", 256, 256, 50, 84, 0, cf, conv3=True)
run("l2.conv2", 128, 128, 100, 167, 0, [(0, 0), (1, 1), (2, 1)], conv3=True)
