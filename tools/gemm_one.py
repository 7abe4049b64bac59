"""One GEMM shape, a few launches: the target of PMC passes (tools/gpu_r2_e.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_amd import ops
M, N, K, ta, tb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "T", sys.argv[5] == "T"
a = torch.randn((K, M) if ta else (M, K), device="cuda")
b = torch.randn((N, K) if tb else (K, N), device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(5):
    ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
torch.cuda.synchronize()
