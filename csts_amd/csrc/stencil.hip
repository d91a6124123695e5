// Token-major stencils of the CSTS path (K4, K5, K7, K8 of SURVEY.md 2.3).  All tensors stay in the
// (B, T*H*W, C) token layout the GEMMs produce -- the reference's (B*h, hd, T, H, W) permute+contiguous
// round trips (attention.py:31,37) do not exist here; (t, h, w) is recovered from the token index.
//
//  dwconv_strided    : depthwise Conv3d k=3 p=1 stride s  (pool_q/k/v fwd, attention.py:104-116;
//                      also the data-gradient of the ConvTranspose3d upsample_q)
//  dwconv_transposed : gather form of the transposed conv (upsample_q fwd, attention.py:344-348, and the
//                      data-gradient of the pools) -- no atomics
//  dwconv_wgrad      : dW[c][tap] = sum fine[o*s-1+tap] * coarse[o], two-stage deterministic reduction
//  maxpool / trilinear fwd+bwd : residual-path resampling (attention.py:193-195,240 ; :463-467,471 ;
//                      custom_multimodal_builder.py:479)
// HBM-bound kernels: channels are the fastest index, consecutive lanes -> consecutive channels.
#include "common.h"

namespace {

struct Geom {
  int B, C, HD;
  int Tf, Hf, Wf;   // fine grid   (conv input  / transposed-conv output)
  int Tc, Hc, Wc;   // coarse grid (conv output / transposed-conv input)
  int st, sh, sw;
  int64_t f_bs, f_ts;  // fine tensor: batch stride, token stride (elements)
  int64_t c_bs, c_ts;  // coarse tensor strides
};

constexpr int VEC = 4;

__device__ __forceinline__ void ld4(const void* p, int dt, int64_t i, float (&o)[4]) {
  if (dt == CSTS_F32) {
    float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  } else {
    bf16x4 a = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p) + i);
    o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)a[2]; o[3] = (float)a[3];
  }
}
__device__ __forceinline__ void st4(void* p, int dt, int64_t i, const float (&o)[4]) {
  if (dt == CSTS_F32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    bf16x4 a;
    a[0] = (bf16)o[0]; a[1] = (bf16)o[1]; a[2] = (bf16)o[2]; a[3] = (bf16)o[3];
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p) + i) = a;
  }
}

// weights staged as wl[tap][HD] (fp32) so that lanes read consecutive channels
__device__ __forceinline__ void stage_weights(const float* __restrict__ w, float* wl, int HD) {
  for (int i = threadIdx.x; i < HD * 27; i += blockDim.x) {
    const int c = i / 27, k = i - c * 27;
    wl[k * HD + c] = w[i];
  }
  __syncthreads();
}

// coarse[b,o,c] = sum_k fine[b, o*s-1+k, c] * w[c%HD][k]
__global__ __launch_bounds__(256) void dwconv_strided_kernel(Geom g, const void* __restrict__ fine, int f_dt,
                                                             const float* __restrict__ w, void* __restrict__ coarse,
                                                             int c_dt) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  stage_weights(w, wl, g.HD);
  const int CQ = g.C / VEC;
  const int64_t ntok = (int64_t)g.Tc * g.Hc * g.Wc;
  const int64_t total = (int64_t)g.B * ntok * CQ;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % CQ);
    const int64_t bt = idx / CQ;
    const int b = (int)(bt / ntok);
    int o = (int)(bt - (int64_t)b * ntok);
    const int ow = o % g.Wc; o /= g.Wc;
    const int oh = o % g.Hc;
    const int ot = o / g.Hc;
    const int c = cq * VEC, cw = c % g.HD;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int t = ot * g.st - 1 + kt;
      if (t < 0 || t >= g.Tf) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int h = oh * g.sh - 1 + kh;
        if (h < 0 || h >= g.Hf) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int x = ow * g.sw - 1 + kw;
          if (x < 0 || x >= g.Wf) continue;
          float v[4];
          ld4(fine, f_dt, b * g.f_bs + ((int64_t)(t * g.Hf + h) * g.Wf + x) * g.f_ts + c, v);
          const float4 ww = *reinterpret_cast<const float4*>(&wl[(kt * 9 + kh * 3 + kw) * g.HD + cw]);
          acc[0] += v[0] * ww.x; acc[1] += v[1] * ww.y; acc[2] += v[2] * ww.z; acc[3] += v[3] * ww.w;
        }
      }
    }
    st4(coarse, c_dt, b * g.c_bs + (bt - (int64_t)b * ntok) * g.c_ts + c, acc);
  }
}

// fine[b,f,c] = sum_{k : (f+1-k) % s == 0} coarse[b, (f+1-k)/s, c] * w[c%HD][k]
__global__ __launch_bounds__(256) void dwconv_transposed_kernel(Geom g, const void* __restrict__ coarse, int c_dt,
                                                                const float* __restrict__ w, void* __restrict__ fine,
                                                                int f_dt) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  stage_weights(w, wl, g.HD);
  const int CQ = g.C / VEC;
  const int64_t ntok = (int64_t)g.Tf * g.Hf * g.Wf;
  const int64_t total = (int64_t)g.B * ntok * CQ;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int cq = (int)(idx % CQ);
    const int64_t bt = idx / CQ;
    const int b = (int)(bt / ntok);
    int f = (int)(bt - (int64_t)b * ntok);
    const int fw = f % g.Wf; f /= g.Wf;
    const int fh = f % g.Hf;
    const int ft = f / g.Hf;
    const int c = cq * VEC, cw = c % g.HD;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int nt = ft + 1 - kt;
      if (nt < 0 || nt % g.st != 0) continue;
      const int ot = nt / g.st;
      if (ot >= g.Tc) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int nh = fh + 1 - kh;
        if (nh < 0 || nh % g.sh != 0) continue;
        const int oh = nh / g.sh;
        if (oh >= g.Hc) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int nw = fw + 1 - kw;
          if (nw < 0 || nw % g.sw != 0) continue;
          const int ow = nw / g.sw;
          if (ow >= g.Wc) continue;
          float v[4];
          ld4(coarse, c_dt, b * g.c_bs + ((int64_t)(ot * g.Hc + oh) * g.Wc + ow) * g.c_ts + c, v);
          const float4 ww = *reinterpret_cast<const float4*>(&wl[(kt * 9 + kh * 3 + kw) * g.HD + cw]);
          acc[0] += v[0] * ww.x; acc[1] += v[1] * ww.y; acc[2] += v[2] * ww.z; acc[3] += v[3] * ww.w;
        }
      }
    }
    st4(fine, f_dt, b * g.f_bs + (bt - (int64_t)b * ntok) * g.f_ts + c, acc);
  }
}

// partial dW: block = (slab of blockDim.x channels) x (blockDim.y token lanes), chunk of coarse tokens -> ws[part][HD*27]
// token lanes and heads of a slab are folded in LDS in a fixed order (bitwise reproducible).
__global__ void dwconv_wgrad_kernel(Geom g, const void* __restrict__ fine, int f_dt, const void* __restrict__ coarse,
                                    int c_dt, float* __restrict__ ws, int64_t chunk) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [blockDim.x][27]
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c = blockIdx.x * blockDim.x + tx;
  const int64_t ntok = (int64_t)g.Tc * g.Hc * g.Wc;
  const int64_t total = (int64_t)g.B * ntok;
  const int64_t beg = (int64_t)blockIdx.y * chunk, end = min(total, beg + chunk);
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  for (int64_t bt = beg + ty; bt < end; bt += blockDim.y) {
    const int b = (int)(bt / ntok);
    int o = (int)(bt - (int64_t)b * ntok);
    const float cv = ld_as_f32(coarse, c_dt, b * g.c_bs + (int64_t)o * g.c_ts + c);
    const int ow = o % g.Wc; o /= g.Wc;
    const int oh = o % g.Hc;
    const int ot = o / g.Hc;
    const int64_t fb = b * g.f_bs + c;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int t = ot * g.st - 1 + kt;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int h = oh * g.sh - 1 + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int x = ow * g.sw - 1 + kw;
          const bool ok = t >= 0 && t < g.Tf && h >= 0 && h < g.Hf && x >= 0 && x < g.Wf;
          const float fv = ok ? ld_as_f32(fine, f_dt, fb + ((int64_t)(t * g.Hf + h) * g.Wf + x) * g.f_ts) : 0.f;
          acc[kt * 9 + kh * 3 + kw] += fv * cv;
        }
      }
    }
  }
  // fold token lanes (fixed order), then heads
  for (int l = 0; l < (int)blockDim.y; ++l) {
    if (ty == l) {
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        if (l == 0) red[tx * 27 + k] = acc[k];
        else red[tx * 27 + k] += acc[k];
      }
    }
    __syncthreads();
  }
  const int heads_in_slab = blockDim.x / g.HD;
  float* out = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * g.HD * 27;
  const int tid = ty * blockDim.x + tx, nthr = blockDim.x * blockDim.y;
  for (int i = tid; i < g.HD * 27; i += nthr) {
    const int cc = i / 27, k = i - cc * 27;
    float s2 = 0.f;
    for (int hh = 0; hh < heads_in_slab; ++hh) s2 += red[(cc + hh * g.HD) * 27 + k];
    out[i] = s2;
  }
}

// ------------------------------------------------------------------ max-pool skip
struct PoolGeom {
  int B, C;
  int Ti, Hi, Wi, To, Ho, Wo;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
};

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(PoolGeom g, const void* __restrict__ x, int dt,
                                                          void* __restrict__ y, uint8_t* __restrict__ arg) {
  const int CQ = g.C / VEC;
  const int64_t ntok = (int64_t)g.To * g.Ho * g.Wo;
  const int64_t total = (int64_t)g.B * ntok * CQ;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % CQ) * VEC;
    const int64_t bt = idx / CQ;
    const int b = (int)(bt / ntok);
    int o = (int)(bt - (int64_t)b * ntok);
    const int ow = o % g.Wo; o /= g.Wo;
    const int oh = o % g.Ho;
    const int ot = o / g.Ho;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
    for (int a = 0; a < g.kt; ++a) {
      const int t = ot * g.st - g.pt + a;
      if (t < 0 || t >= g.Ti) continue;
      for (int e = 0; e < g.kh; ++e) {
        const int h = oh * g.sh - g.ph + e;
        if (h < 0 || h >= g.Hi) continue;
        for (int f = 0; f < g.kw; ++f) {
          const int w = ow * g.sw - g.pw + f;
          if (w < 0 || w >= g.Wi) continue;
          float v[4];
          ld4(x, dt, ((int64_t)b * g.Ti * g.Hi * g.Wi + (int64_t)(t * g.Hi + h) * g.Wi + w) * g.C + c, v);
          const int tap = (a * g.kh + e) * g.kw + f;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (first || v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = tap; }  // first max wins; NaN propagates
          first = false;
        }
      }
    }
    st4(y, dt, bt * g.C + c, best);
    if (arg) *reinterpret_cast<uchar4*>(arg + bt * g.C + c) = make_uchar4((uint8_t)bi[0], (uint8_t)bi[1], (uint8_t)bi[2], (uint8_t)bi[3]);
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(PoolGeom g, const void* __restrict__ dy, int dt,
                                                          const uint8_t* __restrict__ arg, void* __restrict__ dx) {
  const int CQ = g.C / VEC;
  const int64_t ntok = (int64_t)g.Ti * g.Hi * g.Wi;
  const int64_t ntoko = (int64_t)g.To * g.Ho * g.Wo;
  const int64_t total = (int64_t)g.B * ntok * CQ;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % CQ) * VEC;
    const int64_t bt = idx / CQ;
    const int b = (int)(bt / ntok);
    int i = (int)(bt - (int64_t)b * ntok);
    const int w = i % g.Wi; i /= g.Wi;
    const int h = i % g.Hi;
    const int t = i / g.Hi;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < g.kt; ++a) {
      const int nt = t + g.pt - a;
      if (nt < 0 || nt % g.st != 0 || nt / g.st >= g.To) continue;
      for (int e = 0; e < g.kh; ++e) {
        const int nh = h + g.ph - e;
        if (nh < 0 || nh % g.sh != 0 || nh / g.sh >= g.Ho) continue;
        for (int f = 0; f < g.kw; ++f) {
          const int nw = w + g.pw - f;
          if (nw < 0 || nw % g.sw != 0 || nw / g.sw >= g.Wo) continue;
          const int64_t o = ((int64_t)b * ntoko + (int64_t)((nt / g.st) * g.Ho + nh / g.sh) * g.Wo + nw / g.sw) * g.C + c;
          const uchar4 am = *reinterpret_cast<const uchar4*>(arg + o);
          const uint8_t tap = (uint8_t)((a * g.kh + e) * g.kw + f);
          float v[4];
          ld4(dy, dt, o, v);
          if (am.x == tap) s[0] += v[0];
          if (am.y == tap) s[1] += v[1];
          if (am.z == tap) s[2] += v[2];
          if (am.w == tap) s[3] += v[3];
        }
      }
    }
    st4(dx, dt, bt * g.C + c, s);
  }
}

// ------------------------------------------------------------------ trilinear (align_corners=False)
struct UpGeom {
  int B, C;
  int Ti, Hi, Wi, To, Ho, Wo;
};
// source coordinate of output index o (torch area_pixel_compute_source_index, align_corners=False)
__device__ __forceinline__ void src_index(int o, int in, int out, int& i0, int& i1, float& l0, float& l1) {
  const float scale = (float)in / (float)out;
  float s = ((float)o + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void trilinear_fwd_kernel(UpGeom g, const void* __restrict__ x, int x_dt,
                                                            const void* __restrict__ addend, int a_dt,
                                                            void* __restrict__ y, int y_dt) {
  const int CQ = g.C / VEC;
  const int64_t ntok = (int64_t)g.To * g.Ho * g.Wo, ntoki = (int64_t)g.Ti * g.Hi * g.Wi;
  const int64_t total = (int64_t)g.B * ntok * CQ;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % CQ) * VEC;
    const int64_t bt = idx / CQ;
    const int b = (int)(bt / ntok);
    int o = (int)(bt - (int64_t)b * ntok);
    const int ow = o % g.Wo; o /= g.Wo;
    const int oh = o % g.Ho;
    const int ot = o / g.Ho;
    int ti[2], hi[2], wi[2];
    float tl[2], hl[2], wl[2];
    src_index(ot, g.Ti, g.To, ti[0], ti[1], tl[0], tl[1]);
    src_index(oh, g.Hi, g.Ho, hi[0], hi[1], hl[0], hl[1]);
    src_index(ow, g.Wi, g.Wo, wi[0], wi[1], wl[0], wl[1]);
    const int64_t base = (int64_t)b * ntoki;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (tl[a] == 0.f) continue;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (hl[e] == 0.f) continue;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          if (wl[f] == 0.f) continue;
          const float wgt = tl[a] * hl[e] * wl[f];
          float v[4];
          ld4(x, x_dt, (base + (int64_t)(ti[a] * g.Hi + hi[e]) * g.Wi + wi[f]) * g.C + c, v);
          acc[0] += wgt * v[0]; acc[1] += wgt * v[1]; acc[2] += wgt * v[2]; acc[3] += wgt * v[3];
        }
      }
    }
    if (addend) {
      float v[4];
      ld4(addend, a_dt, bt * g.C + c, v);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    st4(y, y_dt, bt * g.C + c, acc);
  }
}

// 1-D adjoint weight: how much output o reads from input i
__device__ __forceinline__ float adj_w(int o, int i, int in, int out) {
  int i0, i1; float l0, l1;
  src_index(o, in, out, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

__global__ __launch_bounds__(256) void trilinear_bwd_kernel(UpGeom g, const void* __restrict__ dy, int dy_dt,
                                                            void* __restrict__ dx, int dx_dt) {
  const int CQ = g.C / VEC;
  const int64_t ntok = (int64_t)g.Ti * g.Hi * g.Wi, ntoko = (int64_t)g.To * g.Ho * g.Wo;
  const int64_t total = (int64_t)g.B * ntok * CQ;
  const int rt = g.To / g.Ti, rh = g.Ho / g.Hi, rw = g.Wo / g.Wi;  // integer up-factors
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % CQ) * VEC;
    const int64_t bt = idx / CQ;
    const int b = (int)(bt / ntok);
    int i = (int)(bt - (int64_t)b * ntok);
    const int w = i % g.Wi; i /= g.Wi;
    const int h = i % g.Hi;
    const int t = i / g.Hi;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ot = max(0, rt * t - rt); ot <= min(g.To - 1, rt * t + 2 * rt - 1); ++ot) {
      const float wt = adj_w(ot, t, g.Ti, g.To);
      if (wt == 0.f) continue;
      for (int oh = max(0, rh * h - rh); oh <= min(g.Ho - 1, rh * h + 2 * rh - 1); ++oh) {
        const float wh = adj_w(oh, h, g.Hi, g.Ho);
        if (wh == 0.f) continue;
        for (int ow = max(0, rw * w - rw); ow <= min(g.Wo - 1, rw * w + 2 * rw - 1); ++ow) {
          const float ww = adj_w(ow, w, g.Wi, g.Wo);
          if (ww == 0.f) continue;
          const float wgt = wt * wh * ww;
          float v[4];
          ld4(dy, dy_dt, ((int64_t)b * ntoko + (int64_t)(ot * g.Ho + oh) * g.Wo + ow) * g.C + c, v);
          s[0] += wgt * v[0]; s[1] += wgt * v[1]; s[2] += wgt * v[2]; s[3] += wgt * v[3];
        }
      }
    }
    st4(dx, dx_dt, bt * g.C + c, s);
  }
}

int grid_for(int64_t total) { return (int)std::min<int64_t>(cdiv(total, 256), 256 * 16); }

int fill_geom(const csts_dwconv_geom* a, Geom& g) {
  g.B = a->B; g.C = a->C; g.HD = a->HD;
  g.Tf = a->Tf; g.Hf = a->Hf; g.Wf = a->Wf; g.Tc = a->Tc; g.Hc = a->Hc; g.Wc = a->Wc;
  g.st = a->st; g.sh = a->sh; g.sw = a->sw;
  g.f_bs = a->fine_batch_stride; g.f_ts = a->fine_token_stride;
  g.c_bs = a->coarse_batch_stride; g.c_ts = a->coarse_token_stride;
  return 0;
}

}  // namespace

#define CHECK_GEOM(a)                                                                                         \
  CSTS_REQUIRE((a) != nullptr, "null geometry");                                                              \
  CSTS_REQUIRE((a)->B > 0 && (a)->C > 0 && (a)->HD > 0 && (a)->C % (a)->HD == 0 && (a)->HD % 4 == 0, "bad channels"); \
  CSTS_REQUIRE((a)->HD <= 192, "head_dim > 192 unsupported");                                                \
  CSTS_REQUIRE((a)->st >= 1 && (a)->sh >= 1 && (a)->sw >= 1, "bad stride");                                 \
  CSTS_REQUIRE((a)->Tc == ((a)->Tf - 1) / (a)->st + 1 && (a)->Hc == ((a)->Hf - 1) / (a)->sh + 1 &&          \
                   (a)->Wc == ((a)->Wf - 1) / (a)->sw + 1,                                                   \
               "coarse grid must equal floor((fine-1)/stride)+1");                                           \
  CSTS_REQUIRE((a)->fine_token_stride % 4 == 0 && (a)->coarse_token_stride % 4 == 0 &&                       \
                   (a)->fine_batch_stride % 4 == 0 && (a)->coarse_batch_stride % 4 == 0,                     \
               "strides must be multiples of 4 elements")

extern "C" int csts_dwconv_strided(const csts_dwconv_geom* a, const void* fine, int fine_dt, const float* weight,
                                   void* coarse, int coarse_dt, hipStream_t stream) {
  CHECK_GEOM(a);
  CSTS_REQUIRE(fine && weight && coarse, "null pointer");
  CSTS_REQUIRE(((uintptr_t)fine & 7) == 0 && ((uintptr_t)coarse & 7) == 0, "tensors must be 8-byte aligned");
  Geom g; fill_geom(a, g);
  const int64_t total = (int64_t)g.B * g.Tc * g.Hc * g.Wc * (g.C / VEC);
  hipLaunchKernelGGL(dwconv_strided_kernel, dim3(grid_for(total)), dim3(256), (size_t)g.HD * 27 * 4, stream, g, fine,
                     fine_dt, weight, coarse, coarse_dt);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_dwconv_transposed(const csts_dwconv_geom* a, const void* coarse, int coarse_dt, const float* weight,
                                      void* fine, int fine_dt, hipStream_t stream) {
  CHECK_GEOM(a);
  CSTS_REQUIRE(fine && weight && coarse, "null pointer");
  CSTS_REQUIRE(((uintptr_t)fine & 7) == 0 && ((uintptr_t)coarse & 7) == 0, "tensors must be 8-byte aligned");
  Geom g; fill_geom(a, g);
  const int64_t total = (int64_t)g.B * g.Tf * g.Hf * g.Wf * (g.C / VEC);
  hipLaunchKernelGGL(dwconv_transposed_kernel, dim3(grid_for(total)), dim3(256), (size_t)g.HD * 27 * 4, stream, g, coarse,
                     coarse_dt, weight, fine, fine_dt);
  CSTS_LAUNCH_CHECK();
  return 0;
}

static void wgrad_plan(const csts_dwconv_geom* a, int& slab, int& nslab, int64_t& chunk, int64_t& nchunk) {
  int k = std::max(1, 192 / a->HD);    // a slab (= thread block) holds k whole heads, <= 192 channels
  while (k > 1 && a->C % (a->HD * k) != 0) --k;
  slab = a->HD * k;
  nslab = a->C / slab;
  const int64_t total = (int64_t)a->B * a->Tc * a->Hc * a->Wc;
  nchunk = std::max<int64_t>(1, std::min<int64_t>(2048 / nslab, cdiv(total, 16)));
  chunk = cdiv(total, nchunk);
  nchunk = cdiv(total, chunk);
}

extern "C" size_t csts_dwconv_wgrad_workspace(const csts_dwconv_geom* a) {
  if (!a || a->C <= 0) return 0;
  int slab, nslab; int64_t chunk, nchunk;
  wgrad_plan(a, slab, nslab, chunk, nchunk);
  return (size_t)nchunk * nslab * a->HD * 27 * sizeof(float);
}

extern "C" int csts_dwconv_wgrad(const csts_dwconv_geom* a, const void* fine, int fine_dt, const void* coarse,
                                 int coarse_dt, float* dweight, void* workspace, size_t ws_bytes, hipStream_t stream) {
  CHECK_GEOM(a);
  CSTS_REQUIRE(fine && coarse && dweight && workspace, "null pointer");
  int slab, nslab; int64_t chunk, nchunk;
  wgrad_plan(a, slab, nslab, chunk, nchunk);
  CSTS_REQUIRE(a->C % slab == 0 && slab % a->HD == 0, "channel slab must hold whole heads");
  CSTS_REQUIRE(ws_bytes >= (size_t)nchunk * nslab * a->HD * 27 * sizeof(float), "workspace too small");
  Geom g; fill_geom(a, g);
  float* ws = reinterpret_cast<float*>(workspace);
  const int lanes = (int)std::max<int64_t>(1, std::min<int64_t>(4, chunk / 4));   // token lanes per channel
  hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(nslab, (unsigned)nchunk), dim3(slab, lanes), (size_t)slab * 27 * 4, stream, g,
                     fine, fine_dt, coarse, coarse_dt, ws, chunk);
  CSTS_LAUNCH_CHECK();

  csts_reduce_rows_launch(ws, dweight, nchunk * nslab, (int64_t)a->HD * 27, 1.f, stream);
  CSTS_LAUNCH_CHECK();
  return 0;
}

static int fill_pool(const csts_pool_geom* a, PoolGeom& g) {
  g.B = a->B; g.C = a->C; g.Ti = a->Ti; g.Hi = a->Hi; g.Wi = a->Wi;
  g.st = a->st; g.sh = a->sh; g.sw = a->sw;
  g.kt = a->st > 1 ? a->st + 1 : 1; g.kh = a->sh > 1 ? a->sh + 1 : 1; g.kw = a->sw > 1 ? a->sw + 1 : 1;
  g.pt = g.kt / 2; g.ph = g.kh / 2; g.pw = g.kw / 2;
  g.To = (a->Ti + 2 * g.pt - g.kt) / a->st + 1;
  g.Ho = (a->Hi + 2 * g.ph - g.kh) / a->sh + 1;
  g.Wo = (a->Wi + 2 * g.pw - g.kw) / a->sw + 1;
  return 0;
}

extern "C" int csts_maxpool_fwd(const csts_pool_geom* a, const void* x, int dt, void* y, uint8_t* argmax,
                                hipStream_t stream) {
  CSTS_REQUIRE(a && x && y, "null pointer");
  CSTS_REQUIRE(a->B > 0 && a->C > 0 && a->C % 4 == 0 && a->st >= 1 && a->sh >= 1 && a->sw >= 1, "bad geometry (C % 4)");
  PoolGeom g; fill_pool(a, g);
  CSTS_REQUIRE(g.To == a->To && g.Ho == a->Ho && g.Wo == a->Wo, "output grid mismatch");
  const int64_t total = (int64_t)g.B * g.To * g.Ho * g.Wo * (g.C / VEC);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g, x, dt, y, argmax);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_maxpool_bwd(const csts_pool_geom* a, const void* dy, int dt, const uint8_t* argmax, void* dx,
                                hipStream_t stream) {
  CSTS_REQUIRE(a && dy && argmax && dx, "null pointer");
  PoolGeom g; fill_pool(a, g);
  CSTS_REQUIRE(g.To == a->To && g.Ho == a->Ho && g.Wo == a->Wo, "output grid mismatch");
  const int64_t total = (int64_t)g.B * g.Ti * g.Hi * g.Wi * (g.C / VEC);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g, dy, dt, argmax, dx);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_trilinear_fwd(const csts_pool_geom* a, const void* x, int x_dt, const void* addend, int addend_dt,
                                  void* y, int y_dt, hipStream_t stream) {
  CSTS_REQUIRE(a && x && y, "null pointer");
  CSTS_REQUIRE(a->C % 4 == 0, "C % 4");
  CSTS_REQUIRE(a->To == a->Ti * a->st && a->Ho == a->Hi * a->sh && a->Wo == a->Wi * a->sw, "output grid must be input*scale");
  UpGeom g{a->B, a->C, a->Ti, a->Hi, a->Wi, a->To, a->Ho, a->Wo};
  const int64_t total = (int64_t)g.B * g.To * g.Ho * g.Wo * (g.C / VEC);
  hipLaunchKernelGGL(trilinear_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g, x, x_dt, addend, addend_dt, y, y_dt);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_trilinear_bwd(const csts_pool_geom* a, const void* dy, int dy_dt, void* dx, int dx_dt,
                                  hipStream_t stream) {
  CSTS_REQUIRE(a && dy && dx, "null pointer");
  CSTS_REQUIRE(a->To == a->Ti * a->st && a->Ho == a->Hi * a->sh && a->Wo == a->Wi * a->sw, "output grid must be input*scale");
  UpGeom g{a->B, a->C, a->Ti, a->Hi, a->Wi, a->To, a->Ho, a->Wo};
  const int64_t total = (int64_t)g.B * g.Ti * g.Hi * g.Wi * (g.C / VEC);
  hipLaunchKernelGGL(trilinear_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g, dy, dy_dt, dx, dx_dt);
  CSTS_LAUNCH_CHECK();
  return 0;
}
