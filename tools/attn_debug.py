import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csts_amd import ops, lib as L
from oracle import csts_oracle as O
DEV = torch.device("cuda:0")
def rnd(*s, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*s, generator=g).to(DEV)
for compute in (L.F32,):
    B, T, Hh, Ww, Cc, H = 2, 2, 4, 4, 768, 8
    HW = Hh * Ww; N = T * HW + T
    dt = torch.float32
    qkv = rnd(B, N, 3 * Cc, seed=3).to(dt).requires_grad_(True)
    meta = (B, N, Cc, H, [T, Hh, Ww], "plain", (1, 1, 1), (1, 1, 1), False, False, L.MASK_SPATIAL, T, HW, compute)
    o, lse = ops.attention_inner(qkv, *([None] * 9), meta)
    go = rnd(*o.shape, seed=4).to(dt)
    o.backward(go)
    qr = qkv.detach().float().requires_grad_(True)
    t = qr.reshape(B, N, 3, H, Cc // H).permute(2, 0, 3, 1, 4)
    of, attn = O.attention_core(t[0], t[1], t[2], (Cc // H) ** -0.5, O.spatial_mask(T, HW, DEV))
    of = of.transpose(1, 2).reshape(B, N, Cc)
    of.backward(go.float())
    d = (o - of).abs().reshape(B, N, H, -1).amax(-1)
    print("fwd max diff per (b, token, head):"); print((d > 1e-3).int()[0].T)
    g = (qkv.grad - qr.grad).abs().reshape(B, N, 3, H, -1).amax(-1)
    for i, nm in enumerate("qkv"):
        print("d" + nm, float(g[:, :, i].max())); print((g[0, :, i] > 1e-3).int().T)
    print("lse sample", lse[0, 0, :8] * 0.6931472)
    with torch.no_grad():
        of2, _ = O.attention_core(t[0], t[1], t[2], (Cc // H) ** -0.5, None)
        of2 = of2.transpose(1, 2).reshape(B, N, Cc)
        print("vs masked oracle", float((o - of).abs().max()), " vs unmasked oracle", float((o - of2).abs().max()))
        print(o[0, :4, :4]); print(of[0, :4, :4]); print(of2[0, :4, :4])
        sc = (t[0] @ t[1].transpose(-1, -2)) * (Cc // H) ** -0.5
        print("true lse masked", torch.logsumexp(sc - O.spatial_mask(T, HW, DEV), -1)[0, 0, :8])
        print("true lse unmasked", torch.logsumexp(sc, -1)[0, 0, :8])
