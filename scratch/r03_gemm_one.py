import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dskd_amd import native
native.load()
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in sys.argv[1:4]]
x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(5): native.gemm_nt_raw(x, w, None, None, M, N, K, False, y)
torch.cuda.synchronize()
