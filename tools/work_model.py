"""Algorithmic work of every non-GEMM C-ABI entry point, from the arguments of the call:
`work(name, args)` -> (bytes, flop) or None.  "Algorithmic" = every operand read once and every result written once
(DESIGN.md section 4's table), so that bytes / duration / 8 TB/s and flop / duration / 2.5 PF are roofline fractions anybody can
recompute.  Used by bench.py (OpTimer: `roofline.per_entry`) and tools/mfma_util.py; measurement infrastructure, not the product.

args are the ctypes arguments as the Python binding passes them (csts_amd/lib.py SYMBOLS): struct pointers arrive as byref()
objects (`._obj` is the struct), device pointers as ints / None."""
import ctypes as C

ESZ = {0: 4, 1: 2}        # CSTS_F32, CSTS_BF16


def _obj(a):
    return a._obj if hasattr(a, "_obj") else a.contents


def _has(p):
    if p is None:
        return False
    if isinstance(p, int):
        return p != 0
    v = getattr(p, "value", p)
    return bool(v)


def _ln_fwd(a):
    x_dt, y_dt, rows, Cc = a[1], a[5], a[8], a[9]
    return rows * Cc * (ESZ[x_dt] + ESZ[y_dt]) + rows * 8, 0


def _ln_fwd_add(a):
    x_dt, y_dt, rows, Cc = a[3], a[7], a[10], a[11]
    return rows * Cc * (3 * ESZ[x_dt] + ESZ[y_dt]) + rows * 8, 0


def _ln_bwd(a):       # csts_layernorm_bwd(dy, dy_dt, x, x_dt, gamma, mean, rstd, dx, dx_dt, addend, dx16, dgamma, dbeta, ws, ws_bytes, rows, C)
    rows, Cc = a[15], a[16]
    b = rows * Cc * (ESZ[a[1]] + ESZ[a[3]] + ESZ[a[8]] + (ESZ[a[8]] if _has(a[9]) else 0) + (2 if _has(a[10]) else 0)) + rows * 8
    return b, 0


def _ln_bwd_ex(a):    # (dy, dy2, dy_dt, x, x_dt, gamma, mean, rstd, dx, dx_dt, addend, dx16, cs, rps, dgamma, dbeta, ws, ws_bytes, rows, C)
    rows, Cc = a[18], a[19]
    b = rows * Cc * (ESZ[a[2]] * (2 if _has(a[1]) else 1) + ESZ[a[4]] + ESZ[a[9]] + (ESZ[a[9]] if _has(a[10]) else 0)
                     + (2 if _has(a[11]) else 0)) + rows * 8
    return b, 0


def _ln_bwd2(a):      # two stacked tensors: (dy, dy_dt, x, x_dt, g0, g1, mean, rstd, dx, dx_dt, dgb0, dgb1, ws, ws_bytes, rows, C)
    rows, Cc = a[14], a[15]
    return 2 * (rows * Cc * (ESZ[a[1]] + ESZ[a[3]] + ESZ[a[9]]) + rows * 8), 0


def _geom_tokens(g):
    return g.B * g.Tf * g.Hf * g.Wf, g.B * g.Tc * g.Hc * g.Wc


def _dwconv(a, slots=1):          # (geom, src, src_dt, w, dst, dst_dt): fine grid + coarse grid once each, 27 taps per output
    g = _obj(a[0])
    nf, nc = _geom_tokens(g)
    dt = a[2]
    return slots * (nf + nc) * g.C * ESZ[dt], slots * 54 * nc * g.C


def _dwconv_tr(a, slots=1):       # transposed form: every FINE cell is an output with <= 27 taps; counted as 27 per coarse cell
    return _dwconv(a, slots)


def _dwconv_wgrad(a, slots=1):    # (geom, fine, f_dt, coarse, c_dt, dw, ws, ws_bytes)
    g = _obj(a[0])
    nf, nc = _geom_tokens(g)
    return slots * (nf * ESZ[a[2]] + nc * ESZ[a[4]]) * g.C, slots * 54 * nc * g.C


def _pool_ln(a):
    p = _obj(a[0])
    g = p.geom
    nf, nc = _geom_tokens(g)
    e = ESZ[p.dt]
    return p.nslots * ((nf + 2 * nc) * g.C * e + nc * (g.C // g.HD) * 8), p.nslots * 54 * nc * g.C


def _pool(a, reads_argmax=False):  # (PoolGeom, x, dt, ...)
    g = _obj(a[0])
    ni, no = g.B * g.Ti * g.Hi * g.Wi, g.B * g.To * g.Ho * g.Wo
    return (ni + no) * g.C * ESZ[a[2]] + (no * g.C * 4 if reads_argmax else 0), 0


def _attn(a, bwd):
    p = _obj(a[0])
    e = ESZ[p.dtype]
    Cc = p.H * p.head_dim
    qo = p.B * p.Nq * Cc * e
    kv = p.B * p.Nk * Cc * e
    stats = p.B * p.H * p.Nq * 4
    mm = 2.0 * p.B * p.H * p.Nq * p.Nk * p.head_dim
    if bwd:      # reads q, o, dO, k, v, lse; writes dq, dk, dv, delta; five matrix products (S, dP, dV, dK, dQ)
        return 4 * qo + 4 * kv + 2 * stats, 5 * mm
    return 2 * qo + 2 * kv + stats, 2 * mm


def _adamw(a):
    o = _obj(a[0])
    return None       # element count lives in device tables: bench.py passes it in (see work_adamw)


def work_adamw(n_params, n_shadowed):
    """clip + AdamW + bf16 shadows: g read twice (norm pass, update pass), p / m / v read + written, shadow written."""
    return n_params * (8 + 24) + n_shadowed * 2, 0


TABLE = {
    "csts_layernorm_fwd": _ln_fwd,
    "csts_layernorm_fwd_add": _ln_fwd_add,
    "csts_layernorm_bwd": _ln_bwd,
    "csts_layernorm_bwd_ex": _ln_bwd_ex,
    "csts_layernorm_bwd2": _ln_bwd2,
    "csts_dwconv_strided": _dwconv,
    "csts_dwconv_transposed": _dwconv_tr,
    "csts_dwconv_transposed2": lambda a: _dwconv_tr(a, 2),
    "csts_dwconv_wgrad": _dwconv_wgrad,
    "csts_dwconv_wgrad2": lambda a: _dwconv_wgrad(a, 2),
    "csts_pool_ln_fwd": _pool_ln,
    "csts_maxpool_fwd": lambda a: _pool(a),
    "csts_maxpool_bwd": lambda a: _pool(a, True),
    "csts_trilinear_fwd": lambda a: _pool(a),
    "csts_trilinear_bwd": lambda a: _pool(a),
    "csts_attn_fwd": lambda a: _attn(a, False),
    "csts_attn_bwd": lambda a: _attn(a, True),
}


def work(name, args):
    fn = TABLE.get(name)
    if fn is None:
        return None
    try:
        return fn(args)
    except Exception:       # an argument layout this model does not know: no figure rather than a wrong one
        return None
