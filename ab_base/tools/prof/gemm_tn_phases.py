"""Where a workgroup of gemm_tn_kernel spends its cycles (build: tools/prof/build_variant.sh gemmprof -DDSKD_GEMM_PROFILE; run
with DSKD_HIP_LIB=tools/prof/libs/libdskd_gemmprof.so): wave 0's totals of the DMA wait (s_waitcnt vmcnt), the barrier and
the fragment waits (lgkmcnt) over the stage loop."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dskd_amd import native
lib = native.load()
lib.dskd_gemm_nt_profile.restype, lib.dskd_gemm_nt_profile.argtypes = C.c_int, [C.c_void_p]
dev = torch.device("cuda:0")
for name, M, N, K in [("enc.ffn.dW1", 88892, 1024, 256), ("enc.ffn.dW2", 88892, 256, 1024), ("enc.lin256", 88892, 256, 256),
                      ("l3.conv1a", 66800, 256, 512), ("l3.conv1", 16800, 256, 1024), ("l4.conv1a", 16800, 512, 1024)]:
    g = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    for _ in range(3):
        native.gemm_tn_bf16(g, x)
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    lib.dskd_gemm_nt_profile(buf.data_ptr())
    native.gemm_tn_bf16(g, x)
    torch.cuda.synchronize()
    lib.dskd_gemm_nt_profile(None)
    p = buf.view(-1, 8).cpu()
    p = p[p[:, 0] != 0]
    tot = (p[:, 2] - p[:, 0]).double()
    pro = (p[:, 1] - p[:, 0]).double()
    nst = (p[:, 5] >> 40).double()
    bar = (p[:, 5] & 0xFFFFFFFFFF).double()
    wait, lg = p[:, 3].double(), p[:, 7].double()
    loop = tot - pro
    print(f"{name:12s} {p.shape[0]:4d} WGs, stages/WG {nst.median():5.0f}: cycles per stage (median): total {(loop / nst).median():6.0f}  "
          f"DMA wait {(wait / nst).median():6.0f}  barrier {(bar / nst).median():6.0f}  fragment waits {(lg / nst).median():6.0f}  "
          f"| prologue {pro.median():6.0f}, MFMA floor 512")
