import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N, C = 4, 2048, 384
da = torch.randn(B, N, C, device=dev); db = torch.randn(B, N, C, device=dev)
s = torch.tensor([0.0, 1.25, 1.25, 0.0], device=dev)
out = torch.empty_like(da); out16 = torch.empty(B, N, C, device=dev, dtype=torch.bfloat16)
L.check(L.load().csts_add2_scaled_copy(da.data_ptr(), L.F32, db.data_ptr(), L.F32, out.data_ptr(), out16.data_ptr(), s.data_ptr(), N * C, da.numel(), torch.cuda.current_stream().cuda_stream), "x")
ref = ops.scale_rows(out, s, N, B * N, C, out_dt=L.BF16)
torch.cuda.synchronize()
print("equal:", torch.equal(ref, out16), (ref.float() - out16.float()).abs().max().item(), torch.equal(out, da + db))
