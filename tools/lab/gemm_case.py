"""Run one GEMM case a few times (for rocprofv3 kernel-trace / PMC passes).
usage: gemm_case.py LAYOUT M N K [split] [a_dt] [b_dt] [c_dt]   (dt in bf16|f32)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
lay = sys.argv[1]; M, N, K = [int(v) for v in sys.argv[2:5]]
split = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dts = {"bf16": torch.bfloat16, "f32": torch.float32}
adt = dts[sys.argv[6]] if len(sys.argv) > 6 else torch.bfloat16
bdt = dts[sys.argv[7]] if len(sys.argv) > 7 else torch.bfloat16
cdt = dts[sys.argv[8]] if len(sys.argv) > 8 else torch.bfloat16
if lay == "NT":   A = torch.randn(M, K, device=dev).to(adt); B = torch.randn(N, K, device=dev).to(bdt); lda, ldb, code = K, K, L.GEMM_NT
elif lay == "NN": A = torch.randn(M, K, device=dev).to(adt); B = torch.randn(K, N, device=dev).to(bdt); lda, ldb, code = K, N, L.GEMM_NN
else:             A = torch.randn(K, M, device=dev).to(adt); B = torch.randn(K, N, device=dev).to(bdt); lda, ldb, code = M, N, L.GEMM_TN
Cm = torch.empty(M, N, device=dev, dtype=cdt)
def run(): ops.gemm(code, A, 0, lda, B, 0, ldb, Cm, N, M, N, K, compute=L.BF16, split_k=split)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"{lay} {M}x{N}x{K} split={split}: {us:.1f} us  {2.0*M*N*K/us/1e6:.1f} TF/s")
