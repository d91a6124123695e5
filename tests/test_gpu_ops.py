"""GPU parity tests, op level: every HIP kernel (through the C ABI, via csts_amd.ops) against a plain PyTorch fp32
reference of the same op on the same seeded inputs.  Tolerances: fp32 mode (exact-fp32 MFMA) 2e-5 rel-L2;
bf16 mode 2e-2 rel-L2 (operands rounded to bf16, fp32 accumulation)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():  # collected everywhere, run only on the GPU box
    pytest.skip("needs a GPU", allow_module_level=True)

from csts_amd import lib as L          # noqa: E402
from csts_amd import ops               # noqa: E402
from oracle import csts_oracle as O    # noqa: E402

DEV = torch.device("cuda:0")
TOL = {L.F32: 2e-5, L.BF16: 2e-2}


def rnd(*shape, seed=0, scale=1.0, dt=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dt)


def tdt(c):
    return torch.float32 if c == L.F32 else torch.bfloat16


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("compute", [L.F32, L.BF16])
@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (1040, 768, 264), (128, 128, 32), (16, 768, 4096), (65, 97, 40)])
def test_gemm_layouts(compute, M, N, K):
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    dt = tdt(compute)
    Ad = A.to(dt)
    ref = Ad.float() @ B.t()
    out = torch.empty(M, N, device=DEV)
    ops.gemm(L.GEMM_NT, Ad, 0, K, B, 0, K, out, N, M, N, K, compute=compute)
    assert rel_l2(out, ref) < TOL[compute]
    # NN: A[M,K] @ Bn[K,N]
    Bn = rnd(K, N, seed=3)
    ops.gemm(L.GEMM_NN, Ad, 0, K, Bn, 0, N, out, N, M, N, K, compute=compute)
    assert rel_l2(out, Ad.float() @ Bn) < TOL[compute]
    # TN: At[K,M]^T @ Bn[K,N], split-k with atomics
    At = rnd(K, M, seed=4).to(dt)
    for split in (1, 3):
        out2 = torch.zeros(M, N, device=DEV)
        ops.gemm(L.GEMM_TN, At, 0, M, Bn, 0, N, out2, N, M, N, K, compute=compute, split_k=split)
        assert rel_l2(out2, At.float().t() @ Bn) < TOL[compute]


@pytest.mark.parametrize("compute", [L.F32, L.BF16])
def test_gemm_epilogues(compute):
    M, N, K = 520, 384, 96
    dt = tdt(compute)
    A, W, b = rnd(M, K, seed=1).to(dt), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3)
    res = rnd(260, N, seed=4)
    rs = torch.tensor([0.0, 1.25], device=DEV)
    pre = A.float() @ W.t() + b
    out = torch.empty(M, N, device=DEV, dtype=dt)
    aux = torch.empty(M, N, device=DEV, dtype=dt)
    ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out, N, M, N, K, compute=compute, bias=b, epilogue=L.EPI_GELU, aux=aux)
    assert rel_l2(aux.float(), pre) < TOL[compute] and rel_l2(out.float(), F.gelu(pre)) < TOL[compute]
    out32 = torch.empty(M, N, device=DEV)
    ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out32, N, M, N, K, compute=compute, bias=b, residual=res, ldr=N, res_row_mod=260,
             row_scale=rs, rows_per_scale=260)
    ref = pre * rs.repeat_interleave(260)[:, None] + res.repeat(2, 1)
    assert rel_l2(out32, ref) < TOL[compute]
    # DGELU: (A @ Wn) * gelu'(h)
    h = rnd(M, N, seed=7).to(dt)
    Wn = rnd(K, N, seed=8, scale=0.2)
    ops.gemm(L.GEMM_NN, A, 0, K, Wn, 0, N, out32, N, M, N, K, compute=compute, epilogue=L.EPI_DGELU, aux=h)
    hf = h.float().requires_grad_(True)
    F.gelu(hf).sum().backward()
    assert rel_l2(out32, (A.float() @ Wn) * hf.grad) < TOL[compute]


@pytest.mark.parametrize("algo", [0, 2, 312, 313, 314, 322, 323, 324, 1322, 342, 343])
@pytest.mark.parametrize("M,N,K", [(520, 384, 96), (2080, 160, 448), (777, 288, 160), (4096, 1536, 384), (300, 96, 1040),
                                   (33000, 192, 64)])
def test_gemm_nt_persistent_kernel(algo, M, N, K):
    """gemm3 (persistent LDS-DMA NT kernel, every forced tile/stage variant and the library heuristic) against fp32 torch:
    ragged M, N % 128 != 0, K % 64 in {0, 16, 32}, all epilogues, bf16 and fp32 outputs (reference nn.Linear + GELU +
    drop-path + residual: attention.py:130,159,238-248; common.py:26-34)."""
    dt = torch.bfloat16
    A, W, b = rnd(M, K, seed=1).to(dt), rnd(N, K, seed=2, scale=0.2).to(dt), rnd(N, seed=3)
    pre = A.float() @ W.float().t() + b
    out = torch.full((M + 1, N), 7.0, device=DEV, dtype=dt)   # a guard row behind the output
    aux = torch.full((M + 1, N), 7.0, device=DEV, dtype=dt)
    ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out, N, M, N, K, compute=L.BF16, bias=b, epilogue=L.EPI_GELU, aux=aux, algo=algo)
    assert rel_l2(aux[:M].float(), pre) < 6e-3 and rel_l2(out[:M].float(), F.gelu(pre)) < 6e-3
    assert (out[M] == 7).all() and (aux[M] == 7).all()
    ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out, N, M, N, K, compute=L.BF16, algo=algo)            # no bias, bf16 out
    assert rel_l2(out[:M].float(), pre - b) < 6e-3
    per = (M + 1) // 2
    res = rnd(per, N, seed=4)
    rs = torch.tensor([0.0, 1.25], device=DEV)
    out32 = torch.full((M + 1, N), 7.0, device=DEV)
    ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out32, N, M, N, K, compute=L.BF16, bias=b, residual=res, ldr=N, res_row_mod=per,
             row_scale=rs, rows_per_scale=per, algo=algo)
    ref = pre * rs.repeat_interleave(per)[:M, None] + res.repeat(2, 1)[:M]
    assert rel_l2(out32[:M], ref) < 1e-4 and (out32[M] == 7).all()
    h = rnd(M, N, seed=7).to(dt)                                                                    # DGELU, fp32 out
    ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out32, N, M, N, K, compute=L.BF16, epilogue=L.EPI_DGELU, aux=h, algo=algo)
    hf = h.float().requires_grad_(True)
    F.gelu(hf).sum().backward()
    assert rel_l2(out32[:M], (pre - b) * hf.grad) < 1e-3


@pytest.mark.parametrize("algo", [0, 462, 463, 473])
@pytest.mark.parametrize("rps", [512, 100])
def test_gemm4_fp32_residual_form(algo, rps):
    """gemm4 FORM 5 (fp32 C = (A W^T + bias) * drop-path scale + fp32 residual on unswapped MFMA operands: attention.py:238-248 proj / fc2
    with the residual add): forced 2- and 3-stage variants and the library's own pick (which routes this shape there) against fp32
    torch; per-sample scales whose boundaries fall INSIDE a 32-row unit (rps = 100) take the per-row path of the epilogue."""
    M, N, K = 2048, 384, 192
    A, W, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.2).bfloat16(), rnd(N, seed=3)
    res = rnd(M, N, seed=4)
    rs = rnd((M + rps - 1) // rps, seed=5).abs() + 0.5
    out = torch.full((M + 1, N), 7.0, device=DEV)
    ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out, N, M, N, K, compute=L.BF16, bias=b, residual=res, ldr=N, row_scale=rs, rows_per_scale=rps, algo=algo)
    ref = (A.float() @ W.float().t() + b) * rs.repeat_interleave(rps)[:M, None] + res
    assert rel_l2(out[:M], ref) < 1e-4 and (out[M] == 7).all()
    out2 = torch.full((M + 1, N), 7.0, device=DEV)
    ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out2, N, M, N, K, compute=L.BF16, residual=res, ldr=N, algo=algo)          # no bias, no scale
    assert rel_l2(out2[:M], A.float() @ W.float().t() + res) < 1e-4 and (out2[M] == 7).all()
    # narrow outputs: N = 96 on 128-column tiles, the last 32-column block of every tile is outside the matrix
    Mn, Nn = 1024, 96
    An, Wn, bn, rn = rnd(Mn, K, seed=6).bfloat16(), rnd(Nn, K, seed=7, scale=0.2).bfloat16(), rnd(Nn, seed=8), rnd(Mn, Nn, seed=9)
    on = torch.full((Mn + 1, Nn), 7.0, device=DEV)
    ops.gemm(L.GEMM_NT, An, 0, K, Wn, 0, K, on, Nn, Mn, Nn, K, compute=L.BF16, bias=bn, residual=rn, ldr=Nn, algo=0 if algo == 0 else 433)
    assert rel_l2(on[:Mn], An.float() @ Wn.float().t() + bn + rn) < 1e-4 and (on[Mn] == 7).all()
    if algo != 0:                                    # the same arithmetic in the same order as the LDS-staged epilogue of gemm2: the same bits
        base = torch.empty(M, N, device=DEV)
        ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, base, N, M, N, K, compute=L.BF16, bias=b, residual=res, ldr=N, row_scale=rs, rows_per_scale=rps, algo=2)
        assert torch.equal(base, out[:M])


@pytest.mark.parametrize("algo", [0, 472, 473, 474])
def test_gemm4_64_row_tiles(algo):
    """gemm4 on 64 x 192 tiles / 4-wave workgroups (round 5: the M = 8192 / 2048 problems of the 384- / 768-channel stages, which have at most
    128 tiles of 128 x 192): every specialised epilogue form (bias, bias + GELU with the kept pre-activation, plain, x gelu') forced
    and as the library's own pick for this shape, against fp32 torch and bit for bit against the register-staged kernel."""
    M, N, K = 2048, 768, 768
    dt = torch.bfloat16
    A, W, b = rnd(M, K, seed=1).to(dt), rnd(N, K, seed=2, scale=0.1).to(dt), rnd(N, seed=3)
    pre = A.float() @ W.float().t() + b

    def run(alg, **kw):
        out = torch.full((M + 1, N), 7.0, device=DEV, dtype=dt)
        ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out, N, M, N, K, compute=L.BF16, algo=alg, **kw)
        assert (out[M] == 7).all()
        return out[:M]

    aux, aux2 = torch.empty(M, N, device=DEV, dtype=dt), torch.empty(M, N, device=DEV, dtype=dt)
    got = run(algo, bias=b, epilogue=L.EPI_GELU, aux=aux)
    assert rel_l2(aux.float(), pre) < 6e-3 and rel_l2(got.float(), F.gelu(pre)) < 6e-3
    assert torch.equal(got, run(2, bias=b, epilogue=L.EPI_GELU, aux=aux2)) and torch.equal(aux, aux2)
    got = run(algo, bias=b)
    assert rel_l2(got.float(), pre) < 6e-3 and torch.equal(got, run(2, bias=b))
    got = run(algo)
    assert rel_l2(got.float(), pre - b) < 6e-3 and torch.equal(got, run(2))
    h = rnd(M, N, seed=7).to(dt)
    got = run(algo, epilogue=L.EPI_DGELU, aux=h)
    hf = h.float().requires_grad_(True)
    F.gelu(hf).sum().backward()
    assert rel_l2(got.float(), (pre - b) * hf.grad) < 6e-3 and torch.equal(got, run(2, epilogue=L.EPI_DGELU, aux=h))


@pytest.mark.parametrize("algo", [0, 500])
@pytest.mark.parametrize("M,N,K", [(16384, 96, 96), (16416, 384, 96), (16384, 192, 96), (20000 // 32 * 32, 96, 192), (16384, 288, 192),
                                   (16384, 768, 96), (16384 + 32 * 7, 96, 384), (16384, 288, 384)])
def test_gemm5_streaming_thin_kernel(algo, M, N, K):
    """gemm5 (weights resident in LDS, one LDS-DMA stream per wave, register epilogue on column pairs): every epilogue form it carries
    -- bias / bias + GELU with the kept pre-activation / x gelu'(aux) / fp32 residual stream with per-sample drop-path scales / plain fp32
    (reference nn.Linear + GELU + drop-path + residual: attention.py:130,159,238-248; common.py:26-34) -- forced (algo 500) and as the
    library's own pick for these shapes, against fp32 torch AND bit for bit against the tiled kernel (same arithmetic, same order);
    one / two / three / eight column tiles, K = 96, 192 and 384 (the four-chunk ring; no x gelu' form there), unit counts that do and do not
    divide over the waves."""
    dt = torch.bfloat16
    A, W, b = rnd(M, K, seed=1).to(dt), rnd(N, K, seed=2, scale=0.2).to(dt), rnd(N, seed=3)
    pre = A.float() @ W.float().t() + b

    def run(alg, **kw):
        out = torch.full((M + 1, N), 7.0, device=DEV, dtype=kw.pop("odt", dt))
        ops.gemm(L.GEMM_NT, A, 0, K, W, 0, K, out, N, M, N, K, compute=L.BF16, algo=alg, **kw)
        assert (out[M] == 7).all()
        return out[:M]

    aux = torch.full((M + 1, N), 7.0, device=DEV, dtype=dt)
    got = run(algo, bias=b, epilogue=L.EPI_GELU, aux=aux)
    assert rel_l2(aux[:M].float(), pre) < 6e-3 and rel_l2(got.float(), F.gelu(pre)) < 6e-3 and (aux[M] == 7).all()
    aux2 = torch.empty(M, N, device=DEV, dtype=dt)
    assert torch.equal(got, run(2, bias=b, epilogue=L.EPI_GELU, aux=aux2)) and torch.equal(aux[:M], aux2)
    got = run(algo, bias=b, epilogue=L.EPI_GELU)                                  # inference: pre-activation not kept
    assert rel_l2(got.float(), F.gelu(pre)) < 6e-3
    got = run(algo)                                                               # no bias, bf16 out
    assert rel_l2(got.float(), pre - b) < 6e-3 and torch.equal(got, run(2))
    got = run(algo, bias=b)
    assert rel_l2(got.float(), pre) < 6e-3 and torch.equal(got, run(2, bias=b))
    if not (K == 384 and algo == 500):
        h = rnd(M, N, seed=7).to(dt)                                              # data gradient through GELU
        got = run(algo, epilogue=L.EPI_DGELU, aux=h)
        hf = h.float().requires_grad_(True)
        F.gelu(hf).sum().backward()
        assert rel_l2(got.float(), (pre - b) * hf.grad) < 6e-3 and torch.equal(got, run(2, epilogue=L.EPI_DGELU, aux=h))
    res = rnd(M, N, seed=4)                                                       # the fp32 residual stream
    rps = 4096 + 32
    rs = rnd((M + rps - 1) // rps, seed=5).abs() + 0.5
    got = run(algo, odt=torch.float32, bias=b, residual=res, ldr=N, row_scale=rs, rows_per_scale=rps)
    assert rel_l2(got, pre * rs.repeat_interleave(rps)[:M, None] + res) < 1e-4
    assert torch.equal(got, run(2, odt=torch.float32, bias=b, residual=res, ldr=N, row_scale=rs, rows_per_scale=rps))
    got = run(algo, odt=torch.float32, residual=res, ldr=N)
    assert rel_l2(got, pre - b + res) < 1e-4 and torch.equal(got, run(2, odt=torch.float32, residual=res, ldr=N))
    got = run(algo, odt=torch.float32, bias=b)                                    # plain fp32 output
    assert rel_l2(got, pre) < 1e-4 and torch.equal(got, run(2, odt=torch.float32, bias=b))
    if algo == 0 and K != 384:                                                    # the library does route these shapes to gemm5 (K = 384: opt-in)
        a = L.GemmArgs()
        a.layout, a.M, a.N, a.K, a.compute = L.GEMM_NT, M, N, K, L.BF16
        a.A, a.a_dt, a.lda, a.B, a.b_dt, a.ldb = A.data_ptr(), L.BF16, K, W.data_ptr(), L.BF16, K
        o = torch.empty(M, N, device=DEV, dtype=dt)
        a.C, a.c_dt, a.ldc = o.data_ptr(), L.BF16, N
        import ctypes as C
        buf, ns = C.create_string_buffer(160), C.c_int(0)
        assert ops._lib().csts_gemm_kernel_name(C.byref(a), buf, 160, C.byref(ns)) == 0
        assert buf.value.decode().startswith("gemm5_kernel<"), buf.value


@pytest.mark.parametrize("compute", [L.F32, L.BF16])
def test_linear_and_mlp_autograd(compute):
    dt = tdt(compute)
    B, N, Cc, Hd, Co = 2, 130, 96, 384, 192
    x = rnd(B, N, Cc, seed=1).to(dt).requires_grad_(True)
    W1, b1 = rnd(Hd, Cc, seed=2, scale=0.1).requires_grad_(True), rnd(Hd, seed=3, scale=0.1).requires_grad_(True)
    W2, b2 = rnd(Co, Hd, seed=4, scale=0.1).requires_grad_(True), rnd(Co, seed=5, scale=0.1).requires_grad_(True)
    res = rnd(B, N, Co, seed=6).requires_grad_(True)
    rs = torch.tensor([1.0 / 0.8, 0.0], device=DEV)
    y = ops.mlp(x, W1, b1, W2, b2, residual=res, row_scale=rs, rows_per_scale=N, act_dt=compute, out_dt=L.F32, compute=compute)
    gy = rnd(B, N, Co, seed=9)
    y.backward(gy)
    got = [y.detach(), x.grad, W1.grad, b1.grad, W2.grad, b2.grad, res.grad]
    xr = x.detach().float().requires_grad_(True)
    P = [t.detach().clone().requires_grad_(True) for t in (W1, b1, W2, b2, res)]
    yr = F.linear(F.gelu(F.linear(xr, P[0], P[1])), P[2], P[3]) * rs[:, None, None] + P[4]
    yr.backward(gy)
    ref = [yr.detach(), xr.grad, P[0].grad, P[1].grad, P[2].grad, P[3].grad, P[4].grad]
    for i, (a, b) in enumerate(zip(got, ref)):
        assert rel_l2(a.float(), b) < TOL[compute] * 2, i
    # plain linear with both operands differentiable (sim matrix use)
    a_, b_ = rnd(5, 256, seed=1).requires_grad_(True), rnd(5, 256, seed=2).requires_grad_(True)
    s = ops.linear(a_, b_, None, out_dt=L.F32, compute=L.F32)
    s.backward(torch.ones_like(s))
    assert rel_l2(s, a_.detach() @ b_.detach().t()) < 1e-5
    assert rel_l2(a_.grad, torch.ones(5, 5, device=DEV) @ b_.detach()) < 1e-5


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("Cc", [96, 192, 384, 768])
@pytest.mark.parametrize("out_dt", [L.F32, L.BF16])
def test_layernorm(Cc, out_dt):
    x = rnd(3, 77, Cc, seed=1, scale=2.0).requires_grad_(True)
    g, b = (1 + 0.1 * rnd(Cc, seed=2)).requires_grad_(True), (0.1 * rnd(Cc, seed=3)).requires_grad_(True)
    y = ops.layer_norm(x, g, b, 1e-6, out_dt)
    gy = rnd(3, 77, Cc, seed=4).to(tdt(out_dt))
    y.backward(gy)
    xr, gr, br = [t.detach().clone().requires_grad_(True) for t in (x, g, b)]
    yr = F.layer_norm(xr, (Cc,), gr, br, 1e-6)
    yr.backward(gy.float())
    tol = 2e-5 if out_dt == L.F32 else 1e-2
    assert rel_l2(y.float(), yr) < tol
    assert rel_l2(x.grad, xr.grad) < 1e-4 and rel_l2(g.grad, gr.grad) < 1e-4 and rel_l2(b.grad, br.grad) < 1e-4


# ------------------------------------------------------------------------------------------------ attention inner
def _ref_attn_inner(qkv, P, B, N, Cc, H, thw, kind, sq, skv, has_q, has_kv, mask):
    HD = Cc // H
    t = qkv.reshape(B, N, 3, H, HD).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    if kind == "dec":
        q, _ = O.upsample_conv_ln(q, thw, P["wq"], sq, P["gq"], P["bq"])
    elif has_q:
        q, _ = O.pool_conv_ln(q, thw, P["wq"], sq, P["gq"], P["bq"])
    if has_kv:
        k, _ = O.pool_conv_ln(k, thw, P["wk"], skv, P["gk"], P["bk"])
        v, _ = O.pool_conv_ln(v, thw, P["wv"], skv, P["gv"], P["bv"])
    m = O.spatial_mask(thw[0], thw[1] * thw[2], qkv.device) if mask else None
    o, _ = O.attention_core(q, k, v, HD ** -0.5, m)
    return o.transpose(1, 2).reshape(B, q.shape[2], Cc)


CASES = [
    # name, B, thw, C, H, kind, sq, skv, has_q, has_kv, mask
    ("enc_nopoolq", 2, (2, 16, 16), 96, 1, "enc", (1, 1, 1), (1, 8, 8), False, True, False),
    ("enc_poolq", 2, (2, 16, 16), 192, 2, "enc", (1, 2, 2), (1, 4, 4), True, True, False),
    ("enc_kv111", 1, (4, 8, 8), 384, 4, "enc", (1, 1, 1), (1, 1, 1), False, True, False),
    ("enc_odd", 1, (3, 7, 5), 192, 2, "enc", (1, 2, 2), (1, 2, 2), True, True, False),
    ("dec_122", 2, (2, 8, 8), 192, 2, "dec", (1, 2, 2), (1, 2, 2), True, True, False),
    ("dec_211", 1, (2, 16, 16), 192, 2, "dec", (2, 1, 1), (1, 16, 16), True, True, False),
    ("dec_hd192", 1, (2, 8, 8), 384, 2, "dec", (1, 2, 2), (1, 4, 4), True, True, False),
    ("temporal", 2, (2, 2, 2), 768, 8, "plain", (1, 1, 1), (1, 1, 1), False, False, False),
]


@pytest.mark.parametrize("compute", [L.F32, L.BF16])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_attention_inner(case, compute):
    name, B, thw, Cc, H, kind, sq, skv, has_q, has_kv, mask = case
    HD = Cc // H
    N = thw[0] * thw[1] * thw[2]
    dt = tdt(compute)
    qkv = rnd(B, N, 3 * Cc, seed=1).to(dt).requires_grad_(True)
    P = {}
    for i, s in enumerate("qkv"):
        P["w" + s] = rnd(HD, 1, 3, 3, 3, seed=10 + i, scale=0.2).requires_grad_(True)
        P["g" + s] = (1 + 0.1 * rnd(HD, seed=20 + i)).requires_grad_(True)
        P["b" + s] = (0.1 * rnd(HD, seed=30 + i)).requires_grad_(True)
    use_q = has_q or kind == "dec"
    meta = (B, N, Cc, H, list(thw), kind, sq, skv, use_q, has_kv, L.MASK_NONE, 0, 0, compute)
    args = [P["wq"], P["gq"], P["bq"]] if use_q else [None, None, None]
    args += [P["wk"], P["gk"], P["bk"], P["wv"], P["gv"], P["bv"]] if has_kv else [None] * 6
    o, lse = ops.attention_inner(qkv, *args, meta)
    go = rnd(*o.shape, seed=5).to(dt)
    o.backward(go)
    qr = qkv.detach().float().requires_grad_(True)
    Pr = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    orf = _ref_attn_inner(qr, Pr, B, N, Cc, H, list(thw), kind, sq, skv, has_q, has_kv, mask)
    orf.backward(go.float())
    tol = 3e-5 if compute == L.F32 else 3e-2
    assert rel_l2(o.float(), orf) < tol, "o"
    assert rel_l2(qkv.grad.float(), qr.grad) < tol * 2, "dqkv"
    used = (["q"] if use_q else []) + (["k", "v"] if has_kv else [])
    for s in used:
        for pfx in "wgb":
            if pfx + s == "bk":
                # d/d(norm_k.bias) is exactly 0 (a constant key shift moves every score of a query equally and
                # softmax is shift-invariant): both sides are rounding noise, compare against the scale of d(gamma_k)
                assert float(P["bk"].grad.norm()) < (1e-3 if compute == L.F32 else 0.5) * float(Pr["gk"].grad.norm()) + 1e-6
                continue
            assert rel_l2(P[pfx + s].grad, Pr[pfx + s].grad) < tol * 3, pfx + s


# ------------------------------------------------------------------------------ k|v only for the rows the pools read
KV_CASES = [  # (name, B, thw, C, heads, kind, stride_q, stride_kv)
    ("enc_s8", 2, (2, 32, 32), 96, 1, "enc", (1, 1, 1), (1, 8, 8)),
    ("enc_s4_poolq", 1, (2, 16, 16), 192, 2, "enc", (1, 2, 2), (1, 4, 4)),
    ("enc_s8_ragged", 1, (4, 14, 14), 96, 1, "enc", (1, 1, 1), (1, 8, 8)),      # BASELINE config 1 grid: last tap outside
    ("enc_s4_odd", 1, (3, 9, 13), 192, 2, "enc", (1, 1, 1), (1, 4, 4)),
    ("dec_s16", 1, (2, 32, 32), 192, 2, "dec", (2, 1, 1), (1, 16, 16)),
    ("dec_s4_hd192", 1, (2, 8, 8), 384, 2, "dec", (1, 2, 2), (1, 4, 4)),
]


def _fine_index(ch, s):
    return ((ch + 1) // 3) * s + (ch + 1) % 3 - 1


@pytest.mark.parametrize("case", KV_CASES, ids=[c[0] for c in KV_CASES])
def test_kv_rows_gather_scatter(case):
    """csts_rows_gather / csts_rows_scatter_add against index arithmetic in torch; the compact grid holds exactly the rows a
    3x3x3 pool with stride (1, s, s), padding 1 reads (attention.py:11-49)."""
    import ctypes as C
    name, B, thw, Cc, H, kind, sq, skv = case
    T, Hh, Ww = thw
    Hc, Wc = ops.kv_compact_dims(thw, skv)
    hs = [_fine_index(c, skv[1]) for c in range(Hc)]
    ws_ = [_fine_index(c, skv[2]) for c in range(Wc)]
    read_h = sorted({o * skv[1] + k for o in range((Hh - 1) // skv[1] + 1) for k in (-1, 0, 1) if 0 <= o * skv[1] + k < Hh})
    read_w = sorted({o * skv[2] + k for o in range((Ww - 1) // skv[2] + 1) for k in (-1, 0, 1) if 0 <= o * skv[2] + k < Ww})
    assert hs == read_h and ws_ == read_w
    g = ops._kv_rows_geom(B, Cc, thw, skv)
    for dt in (torch.float32, torch.bfloat16):
        x = rnd(B, T, Hh, Ww, Cc, seed=2).to(dt)
        out = torch.empty(B, T, Hc, Wc, Cc, device=DEV, dtype=dt)
        L.check(L.load().csts_rows_gather(C.byref(g), x.data_ptr(), ops._dt(x), out.data_ptr(), torch.cuda.current_stream().cuda_stream), "gather")
        ref = x[:, :, hs][:, :, :, ws_]
        assert torch.equal(out, ref)
        src = rnd(B, T, Hc, Wc, Cc, seed=3)
        dst = rnd(B, T, Hh, Ww, Cc, seed=4).to(dt)
        want = dst.float().clone()
        want[:, :, torch.tensor(hs, device=DEV)[:, None], torch.tensor(ws_, device=DEV)[None, :]] += src
        L.check(L.load().csts_rows_scatter_add(C.byref(g), src.data_ptr(), L.F32, dst.data_ptr(), ops._dt(dst), torch.cuda.current_stream().cuda_stream), "scatter")
        assert torch.equal(dst, want.to(dt))


@pytest.mark.parametrize("compute", [L.F32, L.BF16])
@pytest.mark.parametrize("case", KV_CASES, ids=[c[0] for c in KV_CASES])
def test_qkv_compact_equals_full_qkv(case, compute, monkeypatch):
    """qkv Linear + pooled attention through the compact k|v path (ops.qkv_compact -> attention_inner(kvc=...)) against the
    full path (ops.linear -> attention_inner): the rows of a Linear are independent and the pools read the same taps, so the
    forward is BIT-IDENTICAL in both compute modes; gradients agree to rounding (the data gradient adds its two parts in another
    order) and match torch autograd of the reference formulation to the usual bars."""
    name, B, thw, Cc, H, kind, sq, skv = case
    monkeypatch.setattr(ops, "KV_COMPACT_MIN_STRIDE", 3)      # the model takes this path from stride 8 up (CSTS_KV_COMPACT); any stride >= 3 is valid
    HD = Cc // H
    N = thw[0] * thw[1] * thw[2]
    dt = tdt(compute)
    use_q = kind == "dec" or sq != (1, 1, 1)
    meta = (B, N, Cc, H, list(thw), kind, sq, skv, use_q, True, L.MASK_NONE, 0, 0, compute)

    def run(compact):
        x = rnd(B, N, Cc, seed=1).to(dt).requires_grad_(True)
        W = rnd(3 * Cc, Cc, seed=2, scale=Cc ** -0.5).requires_grad_(True)
        b = (0.1 * rnd(3 * Cc, seed=3)).requires_grad_(True)
        P = {}
        for i, s_ in enumerate("qkv"):
            P["w" + s_] = rnd(HD, 1, 3, 3, 3, seed=10 + i, scale=0.2).requires_grad_(True)
            P["g" + s_] = (1 + 0.1 * rnd(HD, seed=20 + i)).requires_grad_(True)
            P["b" + s_] = (0.1 * rnd(HD, seed=30 + i)).requires_grad_(True)
        args = ([P["wq"], P["gq"], P["bq"]] if use_q else [None, None, None]) + [P["wk"], P["gk"], P["bk"], P["wv"], P["gv"], P["bv"]]
        w16 = W.detach().to(torch.bfloat16) if compute == L.BF16 else None
        if compact:
            assert ops.kv_compact_ok(thw, skv, True)
            q, kvc = ops.qkv_compact(x, W, b, thw, skv, out_dt=compute, compute=compute, w16=w16)
            assert q.shape == (B, N, Cc) and kvc.shape[2] == 2 * Cc and kvc.shape[1] < N
            o, _ = ops.attention_inner(q, *args, meta, kvc=kvc)
        else:
            qkv = ops.linear(x, W, b, out_dt=compute, compute=compute, w16=w16)
            o, _ = ops.attention_inner(qkv, *args, meta)
        go = rnd(*o.shape, seed=5).to(dt)
        o.backward(go)
        grads = {"x": x.grad, "W": W.grad, "b": b.grad}
        grads.update({k: v.grad for k, v in P.items() if v.grad is not None})
        return o.detach(), grads

    o_c, g_c = run(True)
    o_f, g_f = run(False)
    assert set(g_c) == set(g_f)
    # same taps, same arithmetic per row; the only difference allowed is the GEMM's own k-split on a tiny compact grid
    # (M <= 128 rows take a deterministic split-K: another fp32 summation order)
    Nc = thw[0] * ops.kv_compact_dims(thw, skv)[0] * ops.kv_compact_dims(thw, skv)[1]
    if B * Nc > 128:
        assert torch.equal(o_c, o_f), "forward must be bit-identical"
    else:
        assert rel_l2(o_c.float(), o_f.float()) < (1e-6 if compute == L.F32 else 4e-3)
    tol = 2e-5 if compute == L.F32 else 2e-2
    for k in g_f:
        if k == "bk":        # exactly zero in exact arithmetic (softmax shift invariance): rounding noise on both sides
            continue
        assert rel_l2(g_c[k].float(), g_f[k].float()) < tol, (k, rel_l2(g_c[k].float(), g_f[k].float()))


@pytest.mark.parametrize("compute", [L.F32, L.BF16])
def test_attention_spatial_mask_and_probs(compute):
    B, T, Hh, Ww, Cc, H = 2, 2, 4, 4, 768, 8
    HW = Hh * Ww
    N = T * HW + T
    dt = tdt(compute)
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
    tol = 3e-5 if compute == L.F32 else 3e-2
    assert rel_l2(o.float(), of) < tol and rel_l2(qkv.grad.float(), qr.grad) < 2 * tol
    probs = ops.attention_probs(qkv.detach(), B, N, Cc, H, lse, L.MASK_SPATIAL, T, HW)
    assert rel_l2(probs, attn) < (1e-4 if compute == L.F32 else 3e-2)


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 2, 1024, 128), (1, 2, 1500, 100), (1, 1, 4000, 32), (2, 1, 2304, 128), (1, 4, 1100, 8)])
def test_attention_backward_one_pass_short_kv(B, H, Nq, Nk):
    """csts_attn_bwd with N_k <= 128, head_dim 96, bf16 takes the one-pass kernel (dQ, dK, dV from one sweep over Q / dO;
    decoder blocks, attention.py:365-392): ragged query tiles, ragged / tiny key counts, q read in place inside a qkv
    buffer (token stride 3C), several query splits -> against fp32 torch autograd of softmax(q k^T / sqrt(hd)) v."""
    import ctypes as C
    hd = 96
    Cc = H * hd
    qkv = rnd(B, Nq, 3 * Cc, seed=1).to(torch.bfloat16)
    k, v = rnd(B, Nk, Cc, seed=2).to(torch.bfloat16), rnd(B, Nk, Cc, seed=3).to(torch.bfloat16)
    do = rnd(B, Nq, Cc, seed=4).to(torch.bfloat16)
    o = torch.empty(B, Nq, Cc, device=DEV, dtype=torch.bfloat16)
    lse, delta = torch.empty(B, H, Nq, device=DEV), torch.empty(B, H, Nq, device=DEV)
    dqkv = torch.full((B, Nq + 1, 3 * Cc), 7.0, device=DEV, dtype=torch.bfloat16)     # dq written in place, guard row behind
    dk, dv = torch.empty_like(k), torch.empty_like(v)
    a = L.AttnArgs()
    a.Q, a.K, a.V, a.O, a.LSE = qkv.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
    a.dO, a.delta, a.dQ, a.dK, a.dV = do.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.dtype, a.B, a.H, a.Nq, a.Nk, a.head_dim = L.BF16, B, H, Nq, Nk, hd
    a.q_strides = (C.c_int64 * 3)(Nq * 3 * Cc, 3 * Cc, hd)
    a.dq_strides = (C.c_int64 * 3)((Nq + 1) * 3 * Cc, 3 * Cc, hd)
    so, sk = (C.c_int64 * 3)(Nq * Cc, Cc, hd), (C.c_int64 * 3)(Nk * Cc, Cc, hd)
    a.o_strides = so; a.do_strides = so
    a.k_strides = sk; a.v_strides = sk; a.dk_strides = sk; a.dv_strides = sk
    a.scale = hd ** -0.5
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.csts_attn_fwd(C.byref(a), st), "fwd")
    ws = torch.empty(max(16, lib.csts_attn_bwd_workspace(C.byref(a))), dtype=torch.uint8, device=DEV)
    L.check(lib.csts_attn_bwd(C.byref(a), ws.data_ptr(), ws.numel(), st), "bwd")
    torch.cuda.synchronize()
    qf = qkv[:, :, :Cc].float().reshape(B, Nq, H, hd).transpose(1, 2).detach().requires_grad_(True)
    kf = k.float().reshape(B, Nk, H, hd).transpose(1, 2).detach().requires_grad_(True)
    vf = v.float().reshape(B, Nk, H, hd).transpose(1, 2).detach().requires_grad_(True)
    of = torch.softmax(qf @ kf.transpose(-1, -2) * hd ** -0.5, -1) @ vf
    of.backward(do.float().reshape(B, Nq, H, hd).transpose(1, 2))
    back = lambda t, n: t.transpose(1, 2).reshape(B, n, Cc)
    assert rel_l2(o.float(), back(of, Nq)) < 1e-2
    assert rel_l2(dqkv[:, :Nq, :Cc].float(), back(qf.grad, Nq)) < 1.5e-2, "dq"
    assert rel_l2(dk.float(), back(kf.grad, Nk)) < 1.5e-2, "dk"
    assert rel_l2(dv.float(), back(vf.grad, Nk)) < 1.5e-2, "dv"
    assert (dqkv[:, Nq] == 7).all() and (dqkv[:, :Nq, Cc:] == 7).all()      # nothing outside the dq slot was touched


@pytest.mark.parametrize("fthw,st", [((3, 8, 12), (1, 2, 2)), ((4, 6, 4), (2, 2, 2)), ((2, 16, 16), (4, 2, 2)), ((3, 7, 5), (1, 2, 2))])
def test_transposed_stencil_stride2_blocks(fthw, st):
    """bf16 transposed depthwise convolution with stride 2 along H and W (2 x 2 output blocks per thread; odd grids fall back
    to the generic kernel) against torch.conv_transpose3d semantics of attention_upsample / the pool backward
    (attention.py:251-289, :11-49): fine grid = the given one, coarse = floor((fine-1)/stride)+1, coarse read in place
    inside a 3C-wide buffer."""
    import ctypes as C
    B, Cc, HD = 2, 192, 96
    cthw = [(f - 1) // s_ + 1 for f, s_ in zip(fthw, st)]
    Nf, Nc = fthw[0] * fthw[1] * fthw[2], cthw[0] * cthw[1] * cthw[2]
    coarse = rnd(B, Nc, 3 * Cc, seed=1).to(torch.bfloat16)
    w = rnd(HD, 27, seed=2, scale=0.3)
    fine = torch.full((B, Nf + 1, Cc), 7.0, device=DEV, dtype=torch.bfloat16)
    g = L.DwconvGeom()
    g.B, g.C, g.HD = B, Cc, HD
    g.Tf, g.Hf, g.Wf = fthw; g.Tc, g.Hc, g.Wc = cthw; g.st, g.sh, g.sw = st
    g.fine_batch_stride, g.fine_token_stride = (Nf + 1) * Cc, Cc
    g.coarse_batch_stride, g.coarse_token_stride = Nc * 3 * Cc, 3 * Cc
    lib = L.load()
    L.check(lib.csts_dwconv_transposed(C.byref(g), coarse.data_ptr() + 2 * Cc, L.BF16, w.data_ptr(), fine.data_ptr(), L.BF16,
                                       torch.cuda.current_stream().cuda_stream), "transposed")     # the middle (k) slot
    torch.cuda.synchronize()
    x = coarse[:, :, Cc:2 * Cc].float().reshape(B, *cthw, Cc).permute(0, 4, 1, 2, 3)
    wt = w.reshape(HD, 1, 3, 3, 3).repeat(Cc // HD, 1, 1, 1, 1)
    op = [f - ((c - 1) * s_ - 2 + 3) for f, c, s_ in zip(fthw, cthw, st)]     # output_padding that restores the fine grid
    ref = F.conv_transpose3d(x, wt, stride=st, padding=1, output_padding=op, groups=Cc)
    ref = ref.permute(0, 2, 3, 4, 1).reshape(B, Nf, Cc)
    assert rel_l2(fine[:, :Nf].float(), ref) < 6e-3
    assert (fine[:, Nf] == 7).all()


def test_attention_long_kv_online_softmax():
    """N_kv = 1000 (not a multiple of the 64-key tile), large score range: exercises the online-softmax rescale."""
    B, N, Cc, H = 1, 1000, 96, 1
    qkv = rnd(B, N, 3 * Cc, seed=11, scale=3.0).requires_grad_(True)
    meta = (B, N, Cc, H, [10, 10, 10], "plain", (1, 1, 1), (1, 1, 1), False, False, L.MASK_NONE, 0, 0, L.F32)
    o, _ = ops.attention_inner(qkv, *([None] * 9), meta)
    o.backward(torch.ones_like(o))
    qr = qkv.detach().clone().requires_grad_(True)
    t = qr.reshape(B, N, 3, H, Cc).permute(2, 0, 3, 1, 4)
    of, _ = O.attention_core(t[0], t[1], t[2], Cc ** -0.5)
    of = of.transpose(1, 2).reshape(B, N, Cc)
    of.backward(torch.ones_like(of))
    assert rel_l2(o, of) < 3e-5 and rel_l2(qkv.grad, qr.grad) < 1e-4


# ------------------------------------------------------------------------------------------------ resampling
@pytest.mark.parametrize("thw,stride", [((2, 8, 8), (1, 2, 2)), ((3, 7, 5), (1, 2, 2)), ((4, 6, 6), (2, 2, 2))])
def test_maxpool_skip(thw, stride):
    x = rnd(2, thw[0] * thw[1] * thw[2], 96, seed=1).requires_grad_(True)
    y = ops.maxpool_skip(x, thw, stride)
    gy = rnd(*y.shape, seed=2)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_(True)
    yr, _ = O.maxpool_skip(xr, list(thw), stride)
    yr.backward(gy)
    assert torch.equal(y, yr) and rel_l2(x.grad, xr.grad) < 1e-6


@pytest.mark.parametrize("thw,stride", [((2, 8, 8), (1, 2, 2)), ((4, 4, 4), (2, 1, 1)), ((3, 5, 7), (1, 2, 2))])
def test_trilinear(thw, stride):
    x = rnd(2, thw[0] * thw[1] * thw[2], 96, seed=1).requires_grad_(True)
    y = ops.trilinear(x, thw, stride)
    gy = rnd(*y.shape, seed=2)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_(True)
    yr, _ = O.trilinear_skip(xr, list(thw), stride)
    yr.backward(gy)
    assert rel_l2(y, yr) < 1e-6 and rel_l2(x.grad, xr.grad) < 1e-6


# ------------------------------------------------------------------------------------------------ embed / fusion / head
@pytest.mark.parametrize("compute", [L.F32, L.BF16])
@pytest.mark.parametrize("cin", [3, 1])
def test_patch_embed(compute, cin):
    x = rnd(2, cin, 4, 32, 32, seed=1)
    W = rnd(96, cin, 3, 7, 7, seed=2, scale=0.05).requires_grad_(True)
    b = rnd(96, seed=3, scale=0.1).requires_grad_(True)
    ps = rnd(1, 64, 96, seed=4, scale=0.1).requires_grad_(True)
    pt = rnd(1, 2, 96, seed=5, scale=0.1).requires_grad_(True)
    y = ops.patch_embed(x, W, b, ps, pt, (3, 7, 7), (2, 4, 4), (1, 3, 3), compute, compute)
    gy = rnd(*y.shape, seed=6)
    y.backward(gy)
    Pr = [t.detach().clone().requires_grad_(True) for t in (W, b, ps, pt)]
    yr = O.patch_embed(x, Pr[0], Pr[1], (2, 4, 4), (1, 3, 3)) + Pr[2].repeat(1, 2, 1) + torch.repeat_interleave(Pr[3], 64, dim=1)
    yr.backward(gy)
    tol = TOL[compute]
    assert rel_l2(y, yr) < tol
    for a, r in zip((W, b, ps, pt), Pr):
        assert rel_l2(a.grad, r.grad) < 2 * tol


@pytest.mark.parametrize("compute", [L.F32, L.BF16])
def test_fusion_conv(compute):
    B, T, Cc = 2, 2, 192
    x = rnd(B, T * 64, Cc, seed=1).requires_grad_(True)
    W = rnd(Cc, Cc, 1, 8, 8, seed=2, scale=0.01).requires_grad_(True)
    b = rnd(Cc, seed=3, scale=0.1).requires_grad_(True)
    y = ops.fusion_conv(x, W, b, T, 64, compute, compute)
    gy = rnd(*y.shape, seed=4)
    y.backward(gy)
    Pr = [t.detach().clone().requires_grad_(True) for t in (x, W, b)]
    yr = O.fusion_conv(Pr[0], [T, 8, 8], Pr[1], Pr[2])
    yr.backward(gy)
    tol = TOL[compute]
    assert rel_l2(y, yr) < tol
    for a, r in zip((x, W, b), Pr):
        assert rel_l2(a.grad, r.grad) < 2 * tol


def test_classifier_head_and_glue():
    B, thw = 2, (2, 8, 8)
    N = thw[0] * thw[1] * thw[2]
    feat = rnd(B, 2 * N, 96, seed=1).requires_grad_(True)
    en = rnd(B, N, 96, seed=2).requires_grad_(True)
    w = rnd(1, 96, 1, 1, 1, seed=3, scale=0.1).requires_grad_(True)
    b = rnd(1, seed=4).requires_grad_(True)
    lg = ops.classifier_head(feat, en, w, b, thw)
    gl = rnd(*lg.shape, seed=5)
    lg.backward(gl)
    Pr = [t.detach().clone().requires_grad_(True) for t in (feat, en, w, b)]
    f5 = Pr[0].reshape(B, 4, 8, 8, 96).permute(0, 4, 1, 2, 3)
    e5 = Pr[1].reshape(B, 2, 8, 8, 96).permute(0, 4, 1, 2, 3)
    lr = F.conv3d(f5 + F.interpolate(e5, size=(4, 8, 8), mode="trilinear"), Pr[2], Pr[3])
    lr.backward(gl)
    assert rel_l2(lg, lr) < 1e-5
    for a, r in zip((feat, en, w, b), Pr):
        assert rel_l2(a.grad, r.grad) < 1e-4
    # reweight / token mean / add
    x = rnd(B, 4 * 64, 96, seed=6).requires_grad_(True)
    wt = rnd(B, 4, 96, seed=7).requires_grad_(True)
    y = ops.add(ops.reweight(x, wt, 4, 64), en.detach().repeat(1, 2, 1))
    m = ops.token_mean(y)
    m.backward(torch.ones_like(m))
    xr, wr = x.detach().clone().requires_grad_(True), wt.detach().clone().requires_grad_(True)
    yr = (xr.reshape(B, 4, 64, 96) * wr[:, :, None, :]).reshape(B, 256, 96) + en.detach().repeat(1, 2, 1)
    mr = yr.mean(dim=1)
    mr.backward(torch.ones_like(mr))
    assert rel_l2(m, mr) < 1e-6 and rel_l2(x.grad, xr.grad) < 1e-6 and rel_l2(wt.grad, wr.grad) < 1e-5


def test_losses_against_oracle():
    logits = (rnd(3, 1, 4, 64, 64, seed=1) * 3).requires_grad_(True)
    tgt = torch.rand(3, 4, 64, 64, generator=torch.Generator().manual_seed(2)).to(DEV)
    tgt = tgt / tgt.sum(dim=(-1, -2), keepdim=True)
    v, a = rnd(3, 256, seed=3).requires_grad_(True), rnd(3, 256, seed=4).requires_grad_(True)
    p = ops.frame_softmax(logits, 2.0)
    kl = ops.kldiv(p, tgt)
    nce = ops.egonce(ops.sim_matrix(v, a))
    (kl + 0.05 * nce).backward()
    lr, vr, ar = [t.detach().clone().requires_grad_(True) for t in (logits, v, a)]
    loss_r, kl_r, nce_r = O.csts_loss(lr, vr, ar, tgt, 0.05)
    loss_r.backward()
    assert abs(float(kl) - float(kl_r)) < 1e-5 and abs(float(nce) - float(nce_r)) < 1e-4
    assert rel_l2(p, O.frame_softmax(lr.detach(), 2.0)) < 1e-5
    assert rel_l2(logits.grad, lr.grad) < 1e-4 and rel_l2(v.grad, vr.grad) < 1e-4 and rel_l2(a.grad, ar.grad) < 1e-4
    # uniform-prior KLDiv (target None)
    assert abs(float(ops.kldiv(p.detach(), None)) - float(
        ((p.detach().reshape(3, 4, -1) * torch.log(p.detach().reshape(3, 4, -1) + 1e-10)).sum(-1) + math.log(4096)).sum(-1).div(
            4 * math.log(4096)).mean())) < 1e-5


@pytest.mark.parametrize("M,Cin,Hd,Cout,out16", [(20000, 192, 768, 96, False), (12304, 96, 288, 96, True), (4112, 384, 96, 192, True)])
def test_grouped_weight_gradients_match_inline(M, Cin, Hd, Cout, out16):
    """Linear / MLP weight and bias gradients through the grouped end-of-backward launch (several token chunks with
    split-K partial slabs, fp32 and bf16 dY) == the in-line split-K GEMMs == torch, bf16 mode.  The thin layers (features in multiples
    of 96, bf16 dY) take csts_wgrad_grouped5 -- 96 x 96 tiles, one item per wave, token counts that are multiples of 16 only, several
    chunks, the fused bias gradient of the n0 == 0 tiles; the fp32-dY layers and the rest stay on the 128-wide classes."""
    x = rnd(M, Cin, seed=1).bfloat16()
    W1, b1 = rnd(Hd, Cin, seed=2, scale=0.05), rnd(Hd, seed=3, scale=0.1)
    W2, b2 = rnd(Cout, Hd, seed=4, scale=0.05), rnd(Cout, seed=5, scale=0.1)
    res = None if out16 else rnd(M, Cout, seed=6)
    dy = rnd(M, Cout, seed=7).to(torch.bfloat16 if out16 else torch.float32)

    def run(grouped):
        old = ops.GROUP_WGRADS
        ops.GROUP_WGRADS = "always" if grouped else "never"
        try:
            ps = [t.clone().requires_grad_() for t in (W1, b1, W2, b2)]
            y = ops.mlp(x, ps[0], ps[1], ps[2], ps[3], residual=res, act_dt=L.BF16, out_dt=L.BF16 if out16 else L.F32, compute=L.BF16)
            y.backward(dy)
            torch.cuda.synchronize()
            return [p.grad.clone() for p in ps]
        finally:
            ops.GROUP_WGRADS = old

    g_grouped, g_inline = run(True), run(False)
    xr = x.float()
    w = [t.clone().requires_grad_() for t in (W1, b1, W2, b2)]
    yr = F.linear(F.gelu(F.linear(xr, w[0].bfloat16().float(), w[1])), w[2].bfloat16().float(), w[3])
    (yr if res is None else yr + res).backward(dy.float())
    for a, b_, r in zip(g_grouped, g_inline, w):
        assert rel_l2(a, b_) < 2e-3
        assert rel_l2(a, r.grad) < 2e-2


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_grouped_stencil_weight_gradients(dt):
    """csts_dwconv_wgrad_grouped (every pool's first stage in ONE launch; raw C-ABI calls) against csts_dwconv_wgrad with dweight
    NULL per problem (same products, longer token chunks: equal to fp32 summation order) -- mixed geometries in one call: stride 1 / 2 / 8 / compact (1,3,3),
    head dims 96 / 192 / 32, ragged grids, operands read in place inside 3C-wide buffers; the row sums then equal the gradient of
    torch's grouped conv3d (attention.py:104-116)."""
    import ctypes as C
    lib = L.load()
    cdt = L.F32 if dt == torch.float32 else L.BF16
    st_ = torch.cuda.current_stream().cuda_stream
    cases = [  # B, C, HD, fine thw, stride
        (2, 384, 96, (4, 8, 8), (1, 1, 1)), (2, 192, 96, (4, 9, 7), (1, 2, 2)), (1, 96, 96, (2, 16, 16), (1, 8, 8)),
        (2, 768, 96, (4, 4, 4), (1, 1, 1)), (1, 192, 192, (2, 6, 10), (1, 2, 2)), (2, 64, 32, (3, 11, 8), (1, 3, 3)),
        (3, 96, 96, (1, 5, 5), (1, 1, 1)),
    ]
    probs = []
    for i, (B, Cc, HD, fthw, st) in enumerate(cases):
        cthw = [(f - 1) // s_ + 1 for f, s_ in zip(fthw, st)]
        Nf, Nc = fthw[0] * fthw[1] * fthw[2], cthw[0] * cthw[1] * cthw[2]
        fine = rnd(B, Nf, 3 * Cc, seed=10 + i).to(dt)            # the conv reads slot 1 of a qkv-shaped buffer
        coarse = rnd(B, Nc, Cc, seed=40 + i).to(dt)
        g = L.DwconvGeom()
        g.B, g.C, g.HD = B, Cc, HD
        g.Tf, g.Hf, g.Wf = fthw; g.Tc, g.Hc, g.Wc = cthw; g.st, g.sh, g.sw = st
        g.fine_batch_stride, g.fine_token_stride = Nf * 3 * Cc, 3 * Cc
        g.coarse_batch_stride, g.coarse_token_stride = Nc * Cc, Cc
        wsz = lib.csts_dwconv_wgrad_workspace(C.byref(g))
        gsz = lib.csts_dwconv_wgrad_grouped_workspace(C.byref(g))       # fewer, longer token chunks than the single launch
        assert 0 < gsz <= wsz and gsz % (HD * 27 * 4) == 0
        ws_one = torch.zeros(wsz // 4, dtype=torch.float32, device=DEV)
        ws_grp = torch.full((gsz // 4,), float("nan"), dtype=torch.float32, device=DEV)
        fptr = fine.data_ptr() + Cc * fine.element_size()
        L.check(lib.csts_dwconv_wgrad(C.byref(g), fptr, cdt, coarse.data_ptr(), cdt, None, ws_one.data_ptr(), wsz, st_), "single")
        probs.append((g, fine, fptr, coarse, ws_one, ws_grp, (B, Cc, HD, fthw, cthw, st, Nf, Nc)))
    items = (L.DwconvWgradItem * len(probs))()
    for i, (g, fine, fptr, coarse, ws_one, ws_grp, _) in enumerate(probs):
        items[i].geom, items[i].fine, items[i].coarse, items[i].workspace = g, fptr, coarse.data_ptr(), ws_grp.data_ptr()
    image = (C.c_uint8 * (L.DWCONV_WGRAD_TABLE_ENTRY * len(probs)))()
    nblocks = C.c_int(0)
    L.check(lib.csts_dwconv_wgrad_grouped_plan(items, len(probs), image, len(image), C.byref(nblocks)), "plan")
    assert nblocks.value > 0
    table = torch.frombuffer(bytearray(bytes(image)), dtype=torch.uint8).to(DEV)
    L.check(lib.csts_dwconv_wgrad_grouped(table.data_ptr(), len(probs), nblocks.value, cdt, st_), "grouped")
    torch.cuda.synchronize()
    for g, fine, fptr, coarse, ws_one, ws_grp, (B, Cc, HD, fthw, cthw, st, Nf, Nc) in probs:
        dw = ws_grp.view(-1, HD * 27).sum(0).view(HD, 27)
        assert bool(torch.isfinite(ws_grp).all())                       # every partial row of the plan was written
        assert rel_l2(dw, ws_one.view(-1, HD * 27).sum(0).view(HD, 27)) < 1e-6, (Cc, HD, fthw, st)
        x = fine[:, :, Cc:2 * Cc].float().reshape(B, *fthw, Cc).permute(0, 4, 1, 2, 3).requires_grad_(False)
        w = torch.zeros(Cc, 1, 3, 3, 3, device=DEV, requires_grad=True)
        y = F.conv3d(x, w, stride=st, padding=1, groups=Cc)
        assert tuple(y.shape[2:]) == tuple(cthw)
        y.backward(coarse.float().reshape(B, *cthw, Cc).permute(0, 4, 1, 2, 3))
        ref = w.grad.view(Cc // HD, HD, 27).sum(0)               # the pools share one (hd, 27) weight over the heads
        assert rel_l2(dw, ref) < (2e-5 if dt == torch.float32 else 2e-5), (Cc, HD, fthw, st)
    # a table that is too small, and an empty call, are refused
    assert lib.csts_dwconv_wgrad_grouped_plan(items, len(probs), image, 16, C.byref(nblocks)) != 0
    assert lib.csts_dwconv_wgrad_grouped(table.data_ptr(), 0, 1, cdt, st_) != 0


# ------------------------------------------------------------------------------------------------ optimizer
def test_fused_adamw_matches_torch_clip_plus_adamw():
    """csts_adamw_step == clip_grad_norm_(1.0) + torch.optim.AdamW(eps 1e-8) (train_avgaze_net.py:101-109,
    optimizer.py:85-93) on ragged tensor sizes (chunk boundaries, unaligned tails, a skipped tensor), per-tensor
    weight decay, and the bf16 shadow it maintains."""
    from csts_amd.optim import FusedAdamW, CHUNK
    shapes = [(768, 300), (96,), (5,), (CHUNK + 1,), (1, 409, 96), (131, 7), (2 * CHUNK,), (33,)]
    pa = [rnd(*s, seed=10 + i, scale=0.5).requires_grad_() for i, s in enumerate(shapes)]
    pb = [p.detach().clone().requires_grad_() for p in pa]
    decay = [0, 4, 5, 6]
    ga = [{"params": [pa[i] for i in decay], "weight_decay": 0.05},
          {"params": [pa[i] for i in range(len(pa)) if i not in decay], "weight_decay": 0.0}]
    gb = [{"params": [pb[i] for i in decay], "weight_decay": 0.05},
          {"params": [pb[i] for i in range(len(pb)) if i not in decay], "weight_decay": 0.0}]
    shadow = torch.empty(shapes[0], dtype=torch.bfloat16, device=DEV)
    fused = FusedAdamW(ga, lr=1e-3, eps=1e-8, max_grad_norm=1.0, shadows={id(pa[0]): shadow})
    ref = torch.optim.AdamW(gb, lr=1e-3, eps=1e-8, weight_decay=0.05)
    for step, gscale in enumerate([1.0, 1e-4, 3.0, 0.02]):     # clipped, un-clipped, clipped, un-clipped
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 7:
                a.grad, b.grad = None, None                     # a parameter without gradient is skipped (like torch)
                continue
            g = rnd(*shapes[i], seed=100 * step + i, scale=gscale)
            a.grad, b.grad = g.clone(), g.clone()
        lr = 1e-3 * (1 + step)
        for grp in fused.param_groups:
            grp["lr"].fill_(lr)
        for grp in ref.param_groups:
            grp["lr"] = lr
        norm_ref = torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ref.step()
        fused.step()
        assert abs(float(fused.grad_norm) - float(norm_ref)) < 1e-5 * float(norm_ref)
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert rel_l2(a.detach(), b.detach()) < 2e-6, (step, i)
    assert torch.equal(shadow, pa[0].detach().bfloat16())
    assert torch.equal(pa[7].detach(), pb[7].detach())
    m_ref = ref.state[pb[3]]["exp_avg"]
    assert rel_l2(fused._m[[id(p) for p in fused.params].index(id(pa[3]))].view_as(m_ref), m_ref) < 1e-6


@pytest.mark.parametrize("a_dtype", [torch.bfloat16, torch.float32])
def test_factored_adamw_matches_dense_gradient(a_dtype):
    """csts_factored_sqnorm + csts_adamw_factored (FusedAdamW.set_factored): a weight whose gradient is dW = dY^T A with few token
    rows (the (1,8,8) fusion convs, custom_multimodal_builder.py:227-229) is updated from the FACTORS -- the clip norm from T x T
    Gram matrices, g formed on the fly in fp32 -- and must equal clip_grad_norm_ + torch AdamW on the materialised dW; other
    parameters of the same optimizer take the ordinary path in the same step.  Also: loss-scaled factors and a skipped step."""
    from csts_amd.optim import FusedAdamW
    N, K, T2 = 48, 1024, 32
    shapes = [(N, 4, 16, 16), (64, 2, 8, 32), (300,), (96, 32)]          # two "fusion convs" (N x K = 48 x 1024, 64 x 512) + ordinary ones
    # 16-bit operands take the MFMA form: dY is rounded to the 16-bit type like in the TN GEMM that materialises dW otherwise
    dy16 = (lambda t: t.to(a_dtype).float()) if a_dtype != torch.float32 else (lambda t: t)
    pa = [rnd(*s, seed=10 + i, scale=0.5).requires_grad_() for i, s in enumerate(shapes)]
    pb = [p.detach().clone().requires_grad_() for p in pa]
    ga = [{"params": [pa[0], pa[1], pa[3]], "weight_decay": 0.05}, {"params": [pa[2]], "weight_decay": 0.0}]
    gb = [{"params": [pb[0], pb[1], pb[3]], "weight_decay": 0.05}, {"params": [pb[2]], "weight_decay": 0.0}]
    shadow = torch.empty(shapes[0], dtype=torch.bfloat16, device=DEV)
    fused = FusedAdamW(ga, lr=1e-3, eps=1e-8, max_grad_norm=1.0, shadows={id(pa[0]): shadow})
    ref = torch.optim.AdamW(gb, lr=1e-3, eps=1e-8, weight_decay=0.05)
    for step, gscale in enumerate([1.0, 1e-3, 2.0]):                      # clipped, un-clipped, clipped
        facs = []
        for i, Tn in ((0, T2), (1, 8)):
            n, k = shapes[i][0], pa[i].numel() // shapes[i][0]
            dy = rnd(Tn, n, seed=50 * step + i, scale=gscale)
            a = rnd(Tn, k, seed=70 * step + i).to(a_dtype)
            facs.append((pa[i], dy, a))
            pb[i].grad = (dy16(dy).t() @ a.float()).view(shapes[i])
            pa[i].grad = None
        for i in (2, 3):
            g = rnd(*shapes[i], seed=100 * step + i, scale=gscale)
            pa[i].grad, pb[i].grad = g.clone(), g.clone()
        fused.set_factored(facs)
        norm_ref = torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ref.step()
        fused.step()
        assert abs(float(fused.grad_norm) - float(norm_ref)) < 2e-5 * float(norm_ref), (step, float(fused.grad_norm), float(norm_ref))
        for i, (a_, b_) in enumerate(zip(pa, pb)):
            assert rel_l2(a_.detach(), b_.detach()) < 3e-6, (step, i)
    assert torch.equal(shadow, pa[0].detach().bfloat16())
    fused.set_factored(None)
    # loss scaling: factors of the SCALED loss; an inf in a factor skips the whole step
    fs = FusedAdamW([{"params": [pa[0], pa[2]], "weight_decay": 0.0}], lr=1e-3, max_grad_norm=1.0, loss_scaling=True, init_scale=1024.0)
    before = [p.detach().clone() for p in (pa[0], pa[2])]
    dy, a = rnd(T2, N, seed=1), rnd(T2, K, seed=2).to(a_dtype)
    ref2 = torch.optim.AdamW([pb[0], pb[2]], lr=1e-3, eps=1e-8, weight_decay=0.0)
    pb[0].data.copy_(pa[0].data); pb[2].data.copy_(pa[2].data)
    g2 = rnd(300, seed=3)
    pb[0].grad, pb[2].grad = (dy16(dy).t() @ a.float()).view(shapes[0]), g2.clone()
    pa[0].grad, pa[2].grad = None, g2 * 1024.0
    fs.set_factored([(pa[0], dy * 1024.0, a)])
    nr = torch.nn.utils.clip_grad_norm_([pb[0], pb[2]], 1.0)
    ref2.step(); fs.step()
    assert abs(float(fs.grad_norm) - float(nr)) < 2e-5 * float(nr)
    assert rel_l2(pa[0].detach(), pb[0].detach()) < 3e-6 and rel_l2(pa[2].detach(), pb[2].detach()) < 3e-6
    keep = [pa[0].detach().clone(), pa[2].detach().clone()]
    bad = dy * 1024.0
    bad[3, 5] = float("inf")
    fs.set_factored([(pa[0], bad, a)])
    fs.step()
    assert float(fs.state_t[3]) == 1.0 and float(fs.loss_scale) == 512.0
    assert torch.equal(pa[0].detach(), keep[0]) and torch.equal(pa[2].detach(), keep[1])


# ------------------------------------------------------------------------------------------------ evaluation metric
def test_adaptive_f1_on_device():
    """csts_adaptive_f1 (min-max rescale folded in) == the reference's adaptive_f1 (fixture generated by importing
    slowfast/utils/metrics.py) == the oracle, on all three threshold tables; counts are integers -> exact up to the
    fp32 means."""
    import numpy as np
    from csts_amd import metrics as M
    from conftest import GOLDEN
    import os
    g = np.load(os.path.join(GOLDEN, "metrics_f1.npz"))
    logits, labels = torch.from_numpy(g["logits"]).to(DEV), torch.from_numpy(g["labels"]).to(DEV)
    hm = O.synthetic_batch(3, 8, 256, seed=55)["labels_hm"].to(DEV)
    preds = ops.frame_softmax(logits, 2.0)
    for ds in ("ego4d_av_gaze_forecast", "aria_av_gaze_forecast", "ego4d_av_gaze"):
        f1, rec, prec, thr = M.adaptive_f1(preds, hm, labels, ds, rescale=True)
        ref = g[ds]
        assert abs(f1 - ref[0]) < 2e-6 and abs(rec - ref[1]) < 2e-6 and abs(prec - ref[2]) < 2e-6 and abs(thr - ref[3]) < 1e-12, (ds, f1, ref)
        f1b, *_ = M.adaptive_f1(O.minmax_rescale(preds.cpu()).to(DEV), hm, labels, ds, rescale=False)
        assert abs(f1b - ref[0]) < 2e-6
    with pytest.raises(NotImplementedError):
        M.adaptive_f1(preds, hm, labels, "kinetics")


def test_fused_adamw_state_dict_is_torch_adamw_format():
    """FusedAdamW.state_dict() / load_state_dict() speak torch.optim.AdamW's layout (what a reference .pyth stores as
    "optimizer_state"): moments move both ways and the next step agrees."""
    from csts_amd.optim import FusedAdamW
    shapes = [(64, 40), (96,), (3, 5, 7)]
    pa = [rnd(*s_, seed=20 + i, scale=0.5).requires_grad_() for i, s_ in enumerate(shapes)]
    pb = [p.detach().clone().requires_grad_() for p in pa]
    ref = torch.optim.AdamW([{"params": [pb[0], pb[2]], "weight_decay": 0.05}, {"params": [pb[1]], "weight_decay": 0.0}],
                            lr=2e-3, eps=1e-8)
    fused = FusedAdamW([{"params": [pa[0], pa[2]], "weight_decay": 0.05}, {"params": [pa[1]], "weight_decay": 0.0}],
                       lr=1e-5, eps=1e-8, max_grad_norm=0.0)
    for step in range(2):                      # give the torch optimizer some history
        for i, b in enumerate(pb):
            b.grad = rnd(*shapes[i], seed=300 + 10 * step + i)
        ref.step()
    for a, b in zip(pa, pb):
        a.data.copy_(b.data)
    fused.load_state_dict(ref.state_dict())    # torch -> fused
    for i, (a, b) in enumerate(zip(pa, pb)):
        g = rnd(*shapes[i], seed=900 + i)
        a.grad, b.grad = g.clone(), g.clone()
    ref.step(); fused.step()
    for a, b in zip(pa, pb):
        assert rel_l2(a.detach(), b.detach()) < 2e-6
    ref2 = torch.optim.AdamW([{"params": [pb[0], pb[2]], "weight_decay": 0.05}, {"params": [pb[1]], "weight_decay": 0.0}],
                             lr=1.0, eps=1e-8)
    ref2.load_state_dict(fused.state_dict())   # fused -> torch
    assert abs(ref2.param_groups[0]["lr"] - 2e-3) < 1e-9
    assert rel_l2(ref2.state[pb[2]]["exp_avg"], ref.state[pb[2]]["exp_avg"]) < 1e-6
    assert float(ref2.state[pb[0]]["step"]) == 3.0


# ------------------------------------------------------------------------------------------------ input pipeline
def test_input_pipeline_on_device():
    """csts_frames_normalize / csts_stft_logpower / csts_audio_windows / csts_gaze_heatmaps against the oracle's
    restatement of the dataset code (slowfast/datasets/utils.py:290-307, data/preprocess.py:276-290,
    ego4d_avgaze_forecast.py:214-219,318-326,404-422) and, for the STFT, against torch.stft."""
    import numpy as np
    from csts_amd import inputs as I
    g = torch.Generator().manual_seed(5)
    B, T = 2, 8
    fr = torch.randint(0, 256, (B, T, 32, 48, 3), generator=g, dtype=torch.uint8)
    out = I.normalize_frames(fr.to(DEV))
    for b in range(B):
        assert torch.allclose(out[b].cpu(), O.frames_normalize(fr[b]), atol=1e-6)
    wav = 0.1 * torch.randn(B, 24000 * 2 + 37, generator=g) + 0.05 * torch.sin(torch.arange(24000 * 2 + 37) * 0.11)[None]
    spec = I.stft_logpower(wav.to(DEV)).cpu()
    for b in range(B):
        ref = torch.from_numpy(O.stft_logpower(wav[b].numpy()))
        assert spec[b].shape == ref.shape
        assert (spec[b] - ref).abs().max() < 2e-3             # log domain; tiny powers amplify fp32 rounding
        tref = torch.stft(wav[b], n_fft=511, hop_length=120, win_length=240, window=torch.hann_window(240), center=True,
                          pad_mode="constant", return_complex=True)
        assert (spec[b] - torch.log(tref.abs() ** 2 + 1e-6)).abs().max() < 2e-3
    idx = torch.tensor([[3.0, 10.5, 22.0, 40.0, 57.5, 70.0, 88.0, 89.9]] * B)
    win = I.audio_windows(spec.to(DEV), idx.to(DEV), 90.0).cpu()
    for b in range(B):
        ref, _ = O.audio_windows(spec[b].numpy(), idx[b].numpy(), 90.0)
        assert np.array_equal(win[b].numpy(), ref)
    labels = torch.tensor([[[0.5, 0.5, 0], [0.0, 0.98, 0], [1.5, 0.5, 0], [0.03, 0.01, 1], [0.999, 0.999, 0], [0.25, 0.75, 0],
                            [0.5078125, 0.4921875, 0], [-0.2, 0.3, 0]]] * B)
    hm = I.gaze_heatmaps(labels.to(DEV)).cpu()
    for b in range(B):
        ref = torch.from_numpy(O.gaze_heatmaps(labels[b].numpy(), T))
        assert torch.allclose(hm[b], ref, atol=1e-7, rtol=1e-4)
    assert torch.allclose(hm.sum(dim=(-1, -2)), torch.ones(B, T), atol=1e-5)
    batch = I.assemble_batch(fr.to(DEV), wav.to(DEV), idx.to(DEV), 90.0, labels.to(DEV))
    assert batch["video"].shape == (B, 3, T, 32, 48) and batch["audio"].shape == (B, 1, T, 256, 256)


# ------------------------------------------------------------------------------------------------ audio -> pixel attention
@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-5)])
@pytest.mark.parametrize("B,T,HW,C,H", [(2, 4, 64, 768, 8), (1, 2, 4, 192, 2), (2, 3, 49, 384, 2)])
def test_audio_attn_forward_backward(dt, tol, B, T, HW, C, H):
    """csts_audio_attn_{fwd,bwd} + csts_rowdot2 (MVIT.SPATIAL_AUDIO_ATTN: av_attention.py:356-370 and
    custom_multimodal_builder.py:438-440) against the same computation in plain fp32 torch ops with the reference's
    additive -1e8 mask, autograd for the gradient.  Both sides read the same (possibly bf16-rounded) qkv values."""
    N = T * HW + T
    hd = C // H
    qkv = rnd(B, N, 3 * C, seed=5, scale=1.5).to(dt)
    x = rnd(B, T * HW, C, seed=6)
    gy = rnd(B, T * HW, C, seed=7)
    # ---- torch reference
    qr = qkv.float().clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    q4 = qr.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    off = torch.full((N, N), 1e8, device=DEV)
    for t in range(T):
        off[HW * t:HW * (t + 1), HW * t:HW * (t + 1)] = 0
        off[HW * t:HW * (t + 1), T * HW + t] = 0
        off[T * HW + t, HW * t:HW * (t + 1)] = 0
        off[T * HW + t, T * HW + t] = 0
    attn = ((q4[0] @ q4[1].transpose(-2, -1)) * hd ** -0.5 - off).softmax(dim=-1)
    aa = torch.stack([attn[:, :, T * HW + t, HW * t:HW * (t + 1)] for t in range(T)], dim=2)
    resc = (aa - aa.min(dim=-1, keepdim=True)[0]) / (aa.max(dim=-1, keepdim=True)[0] - aa.min(dim=-1, keepdim=True)[0] + 1e-8)
    y_ref = xr * resc.mean(dim=1).reshape(B, T * HW, 1)
    y_ref.backward(gy)
    # ---- HIP
    qh = qkv.clone().requires_grad_(True)
    xh = x.clone().requires_grad_(True)
    wmap, aah = ops.audio_attn(qh, T, HW, C, H)
    y = ops.row_weight(xh, wmap)
    y.backward(gy)
    assert rel_l2(aah, resc.detach()) < 1e-4 and rel_l2(y, y_ref.detach()) < 1e-4
    assert rel_l2(xh.grad, xr.grad) < 1e-4
    gq = qh.grad.float()
    assert torch.count_nonzero(gq[:, :, 2 * C:]) == 0 and torch.count_nonzero(gq[:, :T * HW, :C]) == 0   # v and video-q untouched
    assert rel_l2(gq, qr.grad) < (1e-3 if dt == torch.float32 else 2e-2)      # bf16: the gradient itself is stored in bf16


def test_tap_sums_both_gradients_and_bf16_copy_is_version_guarded():
    """ops.tap: two aliases of a residual-stream tensor; backward = ONE kernel d_a + d_b (+ bf16 copy).  The bf16 copy
    attached to a gradient tensor is dropped as soon as the tensor is modified in place (what autograd's own accumulation
    does to the first-arrived gradient of a multi-consumer tensor)."""
    x = rnd(3, 50, 96, seed=1).requires_grad_(True)
    a, b = ops.tap(x, L.BF16)
    assert torch.equal(a, x) and torch.equal(b, x)
    ga, gb = rnd(3, 50, 96, seed=2), rnd(3, 50, 96, seed=3)
    seen = {}
    x.register_hook(lambda g: seen.setdefault("g", g))
    (a * ga).sum().backward(retain_graph=True, inputs=[x])
    assert torch.equal(x.grad, ga)                       # only one branch carried a gradient
    x.grad = None
    seen.clear()
    ((a * ga).sum() + (b * gb).sum()).backward()
    assert torch.equal(x.grad, ga + gb)
    g = seen["g"]
    d16 = ops._grad16(g, L.BF16)
    assert d16 is not None and torch.equal(d16, (ga + gb).to(torch.bfloat16))
    g.add_(1.0)                                          # in-place change -> the copy is stale
    assert ops._grad16(g, L.BF16) is None


def _attn_bwd_outputs():
    """csts_attn_bwd (bf16, hd 96) on shapes that take every combination of the FAST / generic dQ and dK/dV kernels (called
    in-process and in a child process with the FAST forms switched off)."""
    import ctypes as C
    outs = []
    hd = 96
    for (B, H, Nq, Nk) in [(1, 2, 512, 256), (2, 1, 384, 192), (1, 2, 192, 100), (1, 1, 1024, 640), (1, 3, 320, 64)]:
        Cc = H * hd
        q = rnd(B, Nq, Cc, seed=21).to(torch.bfloat16)
        k, v = rnd(B, Nk, Cc, seed=22).to(torch.bfloat16), rnd(B, Nk, Cc, seed=23).to(torch.bfloat16)
        do = rnd(B, Nq, Cc, seed=24).to(torch.bfloat16)
        o = torch.empty_like(q)
        lse, delta = torch.empty(B, H, Nq, device=DEV), torch.empty(B, H, Nq, device=DEV)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        a = L.AttnArgs()
        a.Q, a.K, a.V, a.O, a.LSE = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
        a.dO, a.delta, a.dQ, a.dK, a.dV = do.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
        a.dtype, a.B, a.H, a.Nq, a.Nk, a.head_dim = L.BF16, B, H, Nq, Nk, hd
        sq, sk = (C.c_int64 * 3)(Nq * Cc, Cc, hd), (C.c_int64 * 3)(Nk * Cc, Cc, hd)
        a.q_strides = sq; a.o_strides = sq; a.do_strides = sq; a.dq_strides = sq
        a.k_strides = sk; a.v_strides = sk; a.dk_strides = sk; a.dv_strides = sk
        a.scale = hd ** -0.5
        lib = L.load()
        st = torch.cuda.current_stream().cuda_stream
        L.check(lib.csts_attn_fwd(C.byref(a), st), "fwd")
        ws = torch.empty(max(16, lib.csts_attn_bwd_workspace(C.byref(a))), dtype=torch.uint8, device=DEV)
        L.check(lib.csts_attn_bwd(C.byref(a), ws.data_ptr(), ws.numel(), st), "bwd")
        torch.cuda.synchronize()
        qf = q.float().reshape(B, Nq, H, hd).transpose(1, 2).detach().requires_grad_(True)
        kf = k.float().reshape(B, Nk, H, hd).transpose(1, 2).detach().requires_grad_(True)
        vf = v.float().reshape(B, Nk, H, hd).transpose(1, 2).detach().requires_grad_(True)
        of = torch.softmax(qf @ kf.transpose(-1, -2) * hd ** -0.5, -1) @ vf
        of.backward(do.float().reshape(B, Nq, H, hd).transpose(1, 2))
        back = lambda t, n: t.transpose(1, 2).reshape(B, n, Cc)
        for got, ref, what in [(dq, back(qf.grad, Nq), "dq"), (dk, back(kf.grad, Nk), "dk"), (dv, back(vf.grad, Nk), "dv")]:
            assert rel_l2(got.float(), ref) < 2e-2, (what, B, H, Nq, Nk)
        assert rel_l2(o.float(), back(of, Nq)) < 1e-2, ("o", B, H, Nq, Nk)
        outs += [o.cpu(), lse.cpu(), delta.cpu(), dq.cpu(), dk.cpu(), dv.cpu()]
    return outs


def test_attention_backward_fast_forms_match_generic(tmp_path):
    """attn_fwd_fast_kernel / attn_dq_fast_kernel / the FAST form of attn_dkv_kernel (LDS-DMA tiles, MFMA slot stream) against the
    generic kernels (a child process with CSTS_ATTN_{FWD,DQ,DKV}_FAST=0) on shapes that mix whole and ragged tiles on either side; both
    are also checked against fp32 autograd inside _attn_bwd_outputs.  Same products in the same order per 32-row unit: the
    results agree to bf16 rounding of partial sums that are split differently (128- vs 64-query tiles)."""
    import os
    import subprocess
    import sys
    fast = _attn_bwd_outputs()
    out = str(tmp_path / "generic.pt")
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_ops as t; torch.save(t._attn_bwd_outputs(), %r)"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), out))
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSTS_ATTN_FWD_FAST="0", CSTS_ATTN_DQ_FAST="0", CSTS_ATTN_DKV_FAST="0"),
                   check=True, timeout=600)
    generic = torch.load(out)
    assert len(fast) == len(generic)
    for i, (a, b) in enumerate(zip(fast, generic)):
        assert rel_l2(a.float(), b.float()) < 4e-3, i


@pytest.mark.parametrize("B,H,Nq,Nk,hd", [(1, 2, 4096, 2048, 96), (2, 1, 8192, 512, 96), (1, 4, 1000, 2048, 96), (1, 2, 2048, 1024, 192),
                                           (1, 1, 16384, 4096, 96), (1, 2, 192, 448, 96), (1, 1, 640, 100, 96)])
def test_attention_bf16_backward_many_key_tiles(B, H, Nq, Nk, hd):
    """csts_attn_fwd / csts_attn_bwd in bf16 at the key counts of the benchmarked grids (N_k = 512 ... 2048 at 16x256^2, 4096 at
    32x256^2): keys span many LDS tiles (online softmax over them), queries are split over workgroups, dK / dV partials go
    through attn_dkv_reduce -- against fp32 torch autograd of softmax(q k^T / sqrt(hd)) v on the same bf16-rounded inputs.
    q is read in place inside a qkv buffer (token stride 3C), dq written in place into its slot."""
    import ctypes as C
    Cc = H * hd
    qkv = rnd(B, Nq, 3 * Cc, seed=1).to(torch.bfloat16)
    k, v = rnd(B, Nk, Cc, seed=2).to(torch.bfloat16), rnd(B, Nk, Cc, seed=3).to(torch.bfloat16)
    do = rnd(B, Nq, Cc, seed=4).to(torch.bfloat16)
    o = torch.empty(B, Nq, Cc, device=DEV, dtype=torch.bfloat16)
    lse, delta = torch.empty(B, H, Nq, device=DEV), torch.empty(B, H, Nq, device=DEV)
    dqkv = torch.full((B, Nq + 1, 3 * Cc), 7.0, device=DEV, dtype=torch.bfloat16)
    dk, dv = torch.empty_like(k), torch.empty_like(v)
    a = L.AttnArgs()
    a.Q, a.K, a.V, a.O, a.LSE = qkv.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
    a.dO, a.delta, a.dQ, a.dK, a.dV = do.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.dtype, a.B, a.H, a.Nq, a.Nk, a.head_dim = L.BF16, B, H, Nq, Nk, hd
    a.q_strides = (C.c_int64 * 3)(Nq * 3 * Cc, 3 * Cc, hd)
    a.dq_strides = (C.c_int64 * 3)((Nq + 1) * 3 * Cc, 3 * Cc, hd)
    so, sk = (C.c_int64 * 3)(Nq * Cc, Cc, hd), (C.c_int64 * 3)(Nk * Cc, Cc, hd)
    a.o_strides = so; a.do_strides = so
    a.k_strides = sk; a.v_strides = sk; a.dk_strides = sk; a.dv_strides = sk
    a.scale = hd ** -0.5
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.csts_attn_fwd(C.byref(a), st), "fwd")
    ws = torch.empty(max(16, lib.csts_attn_bwd_workspace(C.byref(a))), dtype=torch.uint8, device=DEV)
    L.check(lib.csts_attn_bwd(C.byref(a), ws.data_ptr(), ws.numel(), st), "bwd")
    torch.cuda.synchronize()
    qf = qkv[:, :, :Cc].float().reshape(B, Nq, H, hd).transpose(1, 2).detach().requires_grad_(True)
    kf = k.float().reshape(B, Nk, H, hd).transpose(1, 2).detach().requires_grad_(True)
    vf = v.float().reshape(B, Nk, H, hd).transpose(1, 2).detach().requires_grad_(True)
    of = torch.softmax(qf @ kf.transpose(-1, -2) * hd ** -0.5, -1) @ vf
    of.backward(do.float().reshape(B, Nq, H, hd).transpose(1, 2))
    back = lambda t, n: t.transpose(1, 2).reshape(B, n, Cc)
    assert rel_l2(o.float(), back(of, Nq)) < 1e-2, "o"
    assert rel_l2(dqkv[:, :Nq, :Cc].float(), back(qf.grad, Nq)) < 2e-2, "dq"
    assert rel_l2(dk.float(), back(kf.grad, Nk)) < 2e-2, "dk"
    assert rel_l2(dv.float(), back(vf.grad, Nk)) < 2e-2, "dv"
    assert (dqkv[:, Nq] == 7).all() and (dqkv[:, :Nq, Cc:] == 7).all()


# ------------------------------------------------------------------------------------------------ round-2 additions
@pytest.mark.parametrize("M,N,K", [(4, 256, 768), (4, 768, 256), (256, 768, 4), (4, 4, 256), (7, 33, 5), (61, 130, 16), (3, 2000, 100)])
def test_tiny_fp32_gemm_path(M, N, K):
    """fp32 GEMMs with few outputs or few reduction steps (EgoNCE projections / similarity: gemm_tiny_kernel) in all three
    layouts, with bias, against torch fp64."""
    A, Bt, Bn, At = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(K, N, seed=3), rnd(K, M, seed=4)
    bias = rnd(N, seed=5)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(L.GEMM_NT, A, 0, K, Bt, 0, K, out, N, M, N, K, compute=L.F32, bias=bias)
    assert rel_l2(out, (A.double() @ Bt.double().t() + bias.double()).float()) < 2e-6
    ops.gemm(L.GEMM_NN, A, 0, K, Bn, 0, N, out, N, M, N, K, compute=L.F32)
    assert rel_l2(out, (A.double() @ Bn.double()).float()) < 2e-6
    ops.gemm(L.GEMM_TN, At, 0, M, Bn, 0, N, out, N, M, N, K, compute=L.F32, split_k=3)      # a split request is ignored here
    assert rel_l2(out, (At.double().t() @ Bn.double()).float()) < 2e-6
    out_b = out.clone()
    ops.gemm(L.GEMM_TN, At, 0, M, Bn, 0, N, out, N, M, N, K, compute=L.F32)
    assert torch.equal(out, out_b)                                                        # fixed summation order


@pytest.mark.parametrize("Cc", [96, 384, 768])
def test_layernorm_backward_two_gradients_and_scaled_copy(Cc):
    """csts_layernorm_bwd_ex: dy + dy2 read by the kernel == LayerNorm backward of the summed gradient; the bf16 copy times
    a per-sample scale == what csts_scale_rows makes of dx, bit for bit; dx itself stays unscaled."""
    B, N = 3, 70
    x = rnd(B, N, Cc, seed=1)
    gamma, beta = rnd(Cc, seed=2) * 0.1 + 1.0, rnd(Cc, seed=3) * 0.1
    dya, dyb = rnd(B, N, Cc, seed=4, dt=torch.bfloat16), rnd(B, N, Cc, seed=5, dt=torch.bfloat16)
    dpass = rnd(B, N, Cc, seed=6)
    xr = x.clone().requires_grad_(True)
    y = F.layer_norm(xr, (Cc,), gamma, beta, 1e-6)
    y.backward(dya.float() + dyb.float())
    ref_dx = xr.grad + dpass
    mean = x.mean(-1).reshape(-1).contiguous()
    rstd = (1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-6)).reshape(-1).contiguous()
    scale = torch.tensor([0.0, 1.0 / 0.8, 1.0 / 0.8], device=DEV)
    dx, dgb = ops._ln_bwd_call(dya, x, gamma, mean, rstd, dpass, B * N, Cc, "csts_layernorm_bwd_ex", want16=True,
                               copy_scale=(scale, N), dy2=dyb)
    assert rel_l2(dx, ref_dx) < 2e-5
    assert rel_l2(dgb[:Cc], (((dya.float() + dyb.float()) * ((x - x.mean(-1, keepdim=True)) * rstd.view(B, N, 1))).sum((0, 1)))) < 2e-5
    d16 = ops._grad16(dx, L.BF16, (scale, N))
    assert d16 is not None and ops._grad16(dx, L.BF16) is None and ops._grad16(dx, L.BF16, (scale, N + 1)) is None
    assert torch.equal(d16, ops.scale_rows(dx, scale, N, B * N, Cc, out_dt=L.BF16))


def test_tap_copy_scaled_for_the_consumer_branch():
    """csts_add2_scaled_copy: fp32 sum unscaled, bf16 copy = bf16(sum * per-sample scale) == csts_scale_rows of the sum."""
    B, N, Cc = 4, 130, 96
    a, b = rnd(B, N, Cc, seed=1), rnd(B, N, Cc, seed=2)
    s = torch.tensor([0.0, 1.25, 1.25, 0.0], device=DEV)
    out = torch.empty_like(a)
    out16 = torch.empty(B, N, Cc, device=DEV, dtype=torch.bfloat16)
    L.check(L.load().csts_add2_scaled_copy(a.data_ptr(), L.F32, b.data_ptr(), L.F32, out.data_ptr(), out16.data_ptr(), s.data_ptr(),
                                           N * Cc, a.numel(), torch.cuda.current_stream().cuda_stream), "csts_add2_scaled_copy")
    assert torch.equal(out, a + b)
    assert torch.equal(out16, ops.scale_rows(out, s, N, B * N, Cc, out_dt=L.BF16))


def test_narrow_colsum_and_token_mean():
    """colsum_narrow_kernel (N <= 256, many rows; weighted = the classifier's weight gradient) and the 16 x 16 token_mean."""
    M, N = 40000, 96
    X, w = rnd(M, N, seed=1), rnd(M, seed=2)
    assert rel_l2(ops.colsum(X, 1, M, N, row_weight=w).view(-1), (X.double() * w.double()[:, None]).sum(0).float()) < 1e-5
    Xb = X.to(torch.bfloat16)
    assert rel_l2(ops.colsum(Xb, 1, M, N).view(-1), Xb.double().sum(0).float()) < 1e-5
    t = rnd(4, 2048, 768, seed=3)
    assert rel_l2(ops.token_mean(t), t.double().mean(1).float()) < 1e-5
    t2 = rnd(3, 77, 40, seed=4)
    assert rel_l2(ops.token_mean(t2), t2.double().mean(1).float()) < 1e-5


def test_fusion_conv_bf16_shadow_is_bit_identical():
    """The fusion convs read a bf16 shadow of their 37.7 M-parameter weight in bf16 mode: same output and gradients as reading
    the fp32 master (the GEMM rounds the operand to bf16 while staging either way)."""
    B, T, HW, Cc, Cout = 2, 4, 16, 96, 64
    x = rnd(B, T * HW, Cc, seed=1)
    W = (rnd(Cout, Cc, 1, 4, 4, seed=2) * 0.05)
    b = rnd(Cout, seed=3)
    outs = []
    for w16 in (None, W.to(torch.bfloat16)):
        xr, Wr = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
        y = ops.fusion_conv(xr, Wr, b, T, HW, L.BF16, L.BF16, w16=w16)
        y.backward(rnd(*y.shape, seed=4))
        outs.append((y.detach(), xr.grad, Wr.grad))
    for u, v in zip(*outs):
        assert torch.equal(u, v)


@pytest.mark.parametrize("Cc", [96, 200, 768])
def test_layernorm_with_folded_skip_add(Cc):
    """layer_norm(x, addend=a): LayerNorm(x + a), the sum as the pass-through output, the same gradient for x and a."""
    B, N = 2, 45
    x, a = rnd(B, N, Cc, seed=1).requires_grad_(True), rnd(B, N, Cc, seed=2).requires_grad_(True)
    gamma, beta = (rnd(Cc, seed=3) * 0.1 + 1.0).requires_grad_(True), (rnd(Cc, seed=4) * 0.1).requires_grad_(True)
    y, xs = ops.layer_norm(x, gamma, beta, 1e-6, L.F32, passthrough=True, addend=a)
    xr, ar = x.detach().clone().requires_grad_(True), a.detach().clone().requires_grad_(True)
    gr, br = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    sr = xr + ar
    yr = F.layer_norm(sr, (Cc,), gr, br, 1e-6)
    assert torch.equal(xs, sr) and rel_l2(y, yr) < 2e-6
    gy, gs = rnd(B, N, Cc, seed=5), rnd(B, N, Cc, seed=6)
    ((y * gy).sum() + (xs * gs).sum()).backward()
    ((yr * gy).sum() + (sr * gs).sum()).backward()
    assert rel_l2(x.grad, xr.grad) < 2e-5 and torch.equal(x.grad, a.grad)
    assert rel_l2(gamma.grad, gr.grad) < 2e-5 and rel_l2(beta.grad, br.grad) < 2e-5


@pytest.mark.parametrize("compute", [L.F32, L.BF16])
@pytest.mark.parametrize("thw,stride,Cin,Cout", [((2, 4, 8), (1, 2, 2), 64, 96), ((4, 8, 8), (2, 1, 1), 96, 192), ((1, 2, 4), (2, 2, 2), 32, 200)])
def test_linear_with_upsampled_residual_in_the_epilogue(compute, thw, stride, Cin, Cout):
    """Decoder skip (attention.py:463-471: x_res = Upsample(trilinear)(x), x = x_res + proj(attn)): the proj GEMM's epilogue
    up-samples the coarse skip itself (csts_gemm_args.res_up).  Forward == trilinear kernel -> GEMM with a plain residual BIT
    FOR BIT (same arithmetic, term for term) and == torch's F.interpolate within the op tolerance; backward: the skip's
    gradient is the adjoint up-sampling of dy, the other gradients are unchanged."""
    B = 2
    out = [t * s for t, s in zip(thw, stride)]
    Nc, Nf = thw[0] * thw[1] * thw[2], out[0] * out[1] * out[2]
    dt = tdt(compute)
    o = rnd(B, Nf, Cin, seed=1).to(dt).requires_grad_(True)
    W = rnd(Cout, Cin, seed=2, scale=0.1).requires_grad_(True)
    b = rnd(Cout, seed=3).requires_grad_(True)
    skip = rnd(B, Nc, Cout, seed=4).requires_grad_(True)
    rs = torch.tensor([0.0, 1.25], device=DEV)
    dy = rnd(B, Nf, Cout, seed=5)

    def run(fused):
        for t in (o, W, b, skip):
            t.grad = None
        if fused:
            y = ops.linear(o, W, b, residual=skip, row_scale=rs, rows_per_scale=Nf, out_dt=L.F32, compute=compute, res_up=(list(thw), out))
        else:
            y = ops.linear(o, W, b, residual=ops.trilinear(skip, thw, stride), row_scale=rs, rows_per_scale=Nf, out_dt=L.F32, compute=compute)
        y.backward(dy)
        ops.flush_deferred()
        torch.cuda.synchronize()
        return y.detach().clone(), [t.grad.detach().clone() for t in (o, W, b, skip)]

    y1, g1 = run(True)
    y0, g0 = run(False)
    assert torch.equal(y1, y0)
    for a, c in zip(g1, g0):
        assert torch.equal(a, c)
    up = F.interpolate(skip.detach().view(B, *thw, Cout).permute(0, 4, 1, 2, 3), scale_factor=tuple(float(s) for s in stride), mode="trilinear",
                       align_corners=False).permute(0, 2, 3, 4, 1).reshape(B, Nf, Cout)
    ref = (o.detach().float() @ W.detach().t() + b.detach()) * rs.repeat_interleave(Nf).view(B, Nf, 1) + up
    assert rel_l2(y1, ref) < TOL[compute]
    with pytest.raises(L.CstsError):       # fine grid sizes must be powers of two
        ops.linear(o, W, b, residual=skip, out_dt=L.F32, compute=compute, res_up=(list(thw), [out[0], out[1], out[2] + 1]))


def _resample_outputs():
    """maxpool_skip / trilinear forward + backward on CSTS-like and ragged grids (called in-process and in a child process)."""
    outs = []
    for thw, stride in [((2, 8, 8), (1, 2, 2)), ((3, 7, 5), (1, 2, 2)), ((2, 16, 4), (1, 2, 2))]:
        x = rnd(2, thw[0] * thw[1] * thw[2], 96, seed=11).requires_grad_(True)
        y = ops.maxpool_skip(x, thw, stride)
        y.backward(rnd(*y.shape, seed=12))
        outs += [y.detach(), x.grad.detach()]
    for thw, stride in [((2, 8, 8), (1, 2, 2)), ((4, 4, 4), (2, 1, 1)), ((3, 5, 7), (1, 2, 2)), ((2, 3, 4), (2, 2, 2)), ((1, 1, 6), (2, 2, 1))]:
        x = rnd(2, thw[0] * thw[1] * thw[2], 64, seed=13).requires_grad_(True)
        y = ops.trilinear(x, thw, stride)
        y.backward(rnd(*y.shape, seed=14))
        outs += [y.detach(), x.grad.detach()]
    return [o.cpu() for o in outs]


def test_resampling_specialisations_match_generic(tmp_path):
    """The closed-form scale-1/2 resampling kernels (maxpool133_fwd / _bwd, trilinear_bwd_fast) against the generic kernels
    they replace (a child process with CSTS_RESAMPLE_FAST=0): bit-identical outputs and gradients, ragged grids included."""
    import os
    import subprocess
    import sys
    fast = _resample_outputs()
    out = str(tmp_path / "generic.pt")
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_ops as t; torch.save(t._resample_outputs(), %r)"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), out))
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CSTS_RESAMPLE_FAST="0"), check=True, timeout=600)
    generic = torch.load(out)
    assert len(fast) == len(generic)
    for i, (a, b) in enumerate(zip(fast, generic)):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("compute", [L.F32, L.BF16])
@pytest.mark.parametrize("M,K,Hd,N", [(8192, 384, 1536, 384), (1000, 96, 384, 192)])
def test_mlp_inference_forward_does_not_keep_the_preactivation_and_matches_training_forward(compute, M, K, Hd, N):
    """An inference forward (no input requires a gradient) skips the M x 4C pre-activation write of fc1 (ops.MlpFn); the output
    must equal the training-mode forward bit for bit (same GELU flavour, same kernels up to the epilogue's extra store)."""
    dt = tdt(compute)
    x = rnd(M, K, seed=1).to(dt)
    W1, b1 = rnd(Hd, K, seed=2, scale=K ** -0.5), 0.1 * rnd(Hd, seed=3)
    W2, b2 = rnd(N, Hd, seed=4, scale=Hd ** -0.5), 0.1 * rnd(N, seed=5)
    res = rnd(M, N, seed=6)
    w16 = (W1.to(torch.bfloat16), W2.to(torch.bfloat16)) if compute == L.BF16 else (None, None)
    kw = dict(residual=res, act_dt=compute, out_dt=L.F32, compute=compute, w16_1=w16[0], w16_2=w16[1])
    with torch.no_grad():
        y_inf = ops.mlp(x, W1, b1, W2, b2, **kw)
    xg = x.clone().requires_grad_(True)
    y_tr = ops.mlp(xg, W1.clone().requires_grad_(True), b1, W2, b2, **kw)
    assert torch.equal(y_inf, y_tr.detach())
    ref = F.linear(F.gelu(F.linear(x.float(), W1 if compute == L.F32 else w16[0].float(), b1)), W2 if compute == L.F32 else w16[1].float(), b2) + res
    assert rel_l2(y_inf, ref) < (2e-5 if compute == L.F32 else 1e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_token_cat_and_split_equal_torch(dt):
    """ops.cat_tokens / ops.split_tokens (csts_copy_token_segments) == torch.cat(dim=1) / x[:, :n], x[:, n:], values and gradients,
    bit for bit (custom_multimodal_builder.py:421-423,432,448,454-455); an unused half of a split gets a zero gradient."""
    B, Na, Nb, Cc = 3, 37, 5, 96
    a = rnd(B, Na, Cc, seed=1).to(dt).requires_grad_(True)
    b = rnd(B, Nb, Cc, seed=2).to(dt).requires_grad_(True)
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    y = ops.cat_tokens(a, b)
    y2 = torch.cat([a2, b2], dim=1)
    assert isinstance(y.grad_fn, torch.autograd.function.BackwardCFunction) and torch.equal(y, y2)
    g = rnd(B, Na + Nb, Cc, seed=3).to(dt)
    y.backward(g); y2.backward(g)
    assert torch.equal(a.grad, a2.grad) and torch.equal(b.grad, b2.grad)
    x = rnd(B, Na + Nb, Cc, seed=4).to(dt).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    p, q = ops.split_tokens(x, Na)
    assert p.is_contiguous() and q.is_contiguous() and torch.equal(p, x2[:, :Na]) and torch.equal(q, x2[:, Na:])
    gp, gq = rnd(B, Na, Cc, seed=5).to(dt), rnd(B, Nb, Cc, seed=6).to(dt)
    (p * gp).sum().backward(retain_graph=True)         # only the first half is used
    (x2[:, :Na] * gp).sum().backward()
    assert torch.equal(x.grad, x2.grad)
    x.grad = None; x2.grad = None
    p, q = ops.split_tokens(x, Na)
    ((p * gp).sum() + (q * gq).sum()).backward()
    ((x2[:, :Na] * gp).sum() + (x2[:, Na:] * gq).sum()).backward()
    assert torch.equal(x.grad, x2.grad)


@pytest.mark.parametrize("T", [96, 256])
def test_factored_adamw_many_token_rows(T):
    """The data-parallel chain gathers the fusion-conv factors of W ranks: T = W * B * T' rows, 256 at eight ranks.  With 16-bit operands
    FusedAdamW keeps the gradient factored up to 256 rows (MFMA form of the update; clip norm from the two T x T Gram matrices through the
    library GEMM): same update and same norm as clip_grad_norm_ + torch AdamW on the materialised dW; more rows (or fp32 operands beyond
    64) are refused by factored_ok so that the caller materialises."""
    from csts_amd.optim import FusedAdamW
    N, K = 64, 2048
    pa = [rnd(N, K, seed=3, scale=0.5).requires_grad_(), rnd(300, seed=4, scale=0.5).requires_grad_()]
    pb = [p.detach().clone().requires_grad_() for p in pa]
    fused = FusedAdamW([{"params": pa, "weight_decay": 0.05}], lr=1e-3, eps=1e-8, max_grad_norm=1.0)
    ref = torch.optim.AdamW([{"params": pb, "weight_decay": 0.05}], lr=1e-3, eps=1e-8)
    assert FusedAdamW.factored_ok(pa[0], T) and not FusedAdamW.factored_ok(pa[0], 257) and not FusedAdamW.factored_ok(pa[0], T, a16=False)
    for step, gscale in enumerate([1.0, 1e-3]):                          # clipped, un-clipped
        dy = rnd(T, N, seed=20 + step, scale=gscale)
        a = rnd(T, K, seed=30 + step).to(torch.bfloat16)
        pb[0].grad = dy.bfloat16().float().t() @ a.float()
        g = rnd(300, seed=40 + step, scale=gscale)
        pa[0].grad, pa[1].grad, pb[1].grad = None, g.clone(), g.clone()
        fused.set_factored([(pa[0], dy, a)])
        norm_ref = torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ref.step()
        fused.step()
        assert abs(float(fused.grad_norm) - float(norm_ref)) < 1e-4 * float(norm_ref), (step, float(fused.grad_norm), float(norm_ref))
        for i, (a_, b_) in enumerate(zip(pa, pb)):
            assert rel_l2(a_.detach(), b_.detach()) < 3e-6, (step, i)
