// Layout, reduction and loss kernels around the GEMM/attention core of the CSTS path
// (K1 im2col, K9 token fold, K10 glue, K11 head + losses of SURVEY.md 2.3).  All HBM-bound, fp32 math.
#include "common.h"
#include <type_traits>



namespace {

int grid_for(int64_t total, int per_block = 256) { return (int)std::min<int64_t>(cdiv(total, per_block), 256 * 16); }

// ------------------------------------------------------------------ patch-embed im2col
// col[(b,to,ho,wo)][k], k = ((ci*KT+kt)*KH+kh)*KW+kw  == flattening of Conv3d weight (Cout, Cin, KT, KH, KW)
// (stem_helper.py:27-38: Conv3d k(3,7,7) s(2,4,4) p(1,3,3), tokens ordered t -> h -> w)
struct Im2colP {
  int B, Cin, T, H, W, KT, KH, KW, st, sh, sw, pt, ph, pw, To, Ho, Wo, K, Kpad;
};
__global__ __launch_bounds__(256) void im2col_kernel(Im2colP g, const void* __restrict__ x, int x_dt,
                                                     void* __restrict__ col, int c_dt) {
  const int64_t rows = (int64_t)g.B * g.To * g.Ho * g.Wo;
  const int64_t total = rows * g.Kpad;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.Kpad);
    int64_t r = idx / g.Kpad;
    float v = 0.f;
    if (k < g.K) {
      const int wo = (int)(r % g.Wo); r /= g.Wo;
      const int ho = (int)(r % g.Ho); r /= g.Ho;
      const int to = (int)(r % g.To);
      const int b = (int)(r / g.To);
      int kk = k;
      const int kw = kk % g.KW; kk /= g.KW;
      const int kh = kk % g.KH; kk /= g.KH;
      const int kt = kk % g.KT;
      const int ci = kk / g.KT;
      const int t = to * g.st - g.pt + kt, h = ho * g.sh - g.ph + kh, w = wo * g.sw - g.pw + kw;
      if (t >= 0 && t < g.T && h >= 0 && h < g.H && w >= 0 && w < g.W)
        v = ld_as_f32(x, x_dt, ((((int64_t)b * g.Cin + ci) * g.T + t) * g.H + h) * g.W + w);
    }
    st_from_f32(col, c_dt, idx, v);
  }
}

// Row-strip form: one workgroup per (b, to, ho).  The Cin*KT*KH input rows that strip of Wo patches touches are staged in
// LDS with coalesced 16-byte loads (rows outside the volume as zeros), then the Wo x Kpad slab of col is written with
// 16-byte stores, each thread gathering 8 consecutive k from LDS.  HBM sees every input row ~KH/sh times (L2 absorbs most
// of that) and every col byte once; the element-per-thread form above re-fetched the input ~13x (rocprofv3 FETCH_SIZE).
template <bool XF32, bool CF32>
__global__ __launch_bounds__(256) void im2col_strip_kernel(Im2colP g, const void* __restrict__ x, void* __restrict__ col) {
  // staged rows [Cin*KT*KH][W] in the OUTPUT dtype (the values are only copied: rounding them here or at the store is the
  // same), so that a bf16 strip is half the LDS and twice the workgroups per CU
  typedef typename std::conditional<CF32, float, bf16>::type E;
  extern __shared__ __attribute__((aligned(16))) char rows_raw[];
  E* rows = reinterpret_cast<E*>(rows_raw);
  const int nrows = g.Cin * g.KT * g.KH;
  const int tid = threadIdx.x;
  int blk = blockIdx.x;
  const int ho = blk % g.Ho; blk /= g.Ho;
  const int to = blk % g.To;
  const int b = blk / g.To;
  const int W4 = g.W / 4;
  auto stage = [&](int r, int w4, int kh, int kt, int ci) {
    const int t = to * g.st - g.pt + kt, h = ho * g.sh - g.ph + kh;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0 && t < g.T && h >= 0 && h < g.H) {
      const int64_t off = ((((int64_t)b * g.Cin + ci) * g.T + t) * g.H + h) * g.W + 4 * w4;
      if constexpr (XF32) v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + off);
      else {
        const bf16x4 q = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(x) + off);
        v = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
      }
    }
    if constexpr (CF32) *reinterpret_cast<float4*>(&rows[(int64_t)r * g.W + 4 * w4]) = v;
    else {
      const bf16x4 q = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
      *reinterpret_cast<bf16x4*>(&rows[(int64_t)r * g.W + 4 * w4]) = q;
    }
  };
  if (256 % W4 == 0) {
    // a thread keeps its column group and walks the rows: (kh, kt, ci) of its row by increments, no division per element
    const int rstep = 256 / W4, w4 = tid % W4;
    int r = tid / W4;
    int kh = r % g.KH, kt = (r / g.KH) % g.KT, ci = r / (g.KH * g.KT);
#pragma unroll 4
    for (; r < nrows; r += rstep) {
      stage(r, w4, kh, kt, ci);
      kh += rstep;
      while (kh >= g.KH) {
        kh -= g.KH;
        if (++kt == g.KT) { kt = 0; ++ci; }
      }
    }
  } else {
    for (int i = tid; i < nrows * W4; i += 256) {
      const int r = i / W4, w4 = i - r * W4;
      stage(r, w4, r % g.KH, (r / g.KH) % g.KT, r / (g.KH * g.KT));
    }
  }
  __syncthreads();
  const int K8 = g.Kpad / 8;
  const int64_t row0 = (((int64_t)b * g.To + to) * g.Ho + ho) * g.Wo;
  for (int i = tid; i < g.Wo * K8; i += 256) {
    const int wo = i / K8, k0 = (i - wo * K8) * 8;
    // (row, kw) of k0 by ONE division, then stepped: a division per element made this phase 8 x ~40 vector instructions per
    // 16 bytes stored and the kernel bound by instruction issue (profiles/r3_im2col_ab.txt)
    int r = k0 / g.KW, kw = k0 - r * g.KW;
    const int wbase = wo * g.sw - g.pw;
    E v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int w = wbase + kw;
      E val = (E)0.f;
      if (k0 + j < g.K && w >= 0 && w < g.W) val = rows[r * g.W + w];
      v[j] = val;
      if (++kw == g.KW) { kw = 0; ++r; }
    }
    if constexpr (CF32) {
      float* d = reinterpret_cast<float*>(col) + (row0 + wo) * g.Kpad + k0;
      reinterpret_cast<float4*>(d)[0] = make_float4(v[0], v[1], v[2], v[3]);
      reinterpret_cast<float4*>(d)[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      bf16x8 q;
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = v[j];
      *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(col) + (row0 + wo) * g.Kpad + k0) = q;
    }
  }
}

// pos[n][c] = spatial[n % HW][c] + temporal[n / HW][c]   (custom_multimodal_builder.py:362-366)
__global__ void posembed_kernel(const float* __restrict__ ps, const float* __restrict__ pt, float* __restrict__ pos,
                                int T, int HW, int C) {
  const int64_t total = (int64_t)T * HW * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const int64_t n = idx / C;
    pos[idx] = ps[(n % HW) * C + c] + pt[(n / HW) * C + c];
  }
}

// ------------------------------------------------------------------ batched transpose [batch][R][Cc] -> [batch][Cc][R]
// folds tokens for the (1,8,8) fusion convs: A'[(b,t)][ci*HW + hw] = x[b, t*HW + hw, ci]
// (custom_multimodal_builder.py:420,442,444: permute(0,4,1,2,3) before Conv3d)
__global__ __launch_bounds__(256) void transpose_kernel(const void* __restrict__ in, int in_dt, void* __restrict__ out,
                                                        int out_dt, int R, int Cc) {
  __shared__ float tile[32][33];
  const int64_t base = (int64_t)blockIdx.z * R * Cc;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < Cc) ? ld_as_f32(in, in_dt, base + (int64_t)r * Cc + c) : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < R && c < Cc) st_from_f32(out, out_dt, base + (int64_t)c * R + r, tile[tx][j]);
  }
}

// ------------------------------------------------------------------ column sums (bias / pos-embed / classifier grads)
// out[batch][n] = sum_m rw[m] * X[batch][m][n]    two-stage: ws[chunk][batch*N] then reduce_rows
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ X, int dt, const float* __restrict__ rw,
                                                     float* __restrict__ ws, int64_t M, int64_t N, int64_t chunk) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + tx;
  const int64_t batch = blockIdx.z;
  const int64_t mbeg = (int64_t)blockIdx.y * chunk, mend = min(M, mbeg + chunk);
  float s = 0.f;
  if (n < N) {
    for (int64_t m = mbeg + ty; m < mend; m += 4) {
      const float v = ld_as_f32(X, dt, (batch * M + m) * N + n);
      s += rw ? v * rw[batch * M + m] : v;
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N)
    ws[((int64_t)blockIdx.y * gridDim.z + batch) * N + n] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// narrow matrices (N % 4 == 0, N <= 256: the classifier's weight gradient over 262 k rows of 96 channels): N / 4 lanes read
// one row as 16-byte (fp32) / 8-byte (bf16) pieces, 256 / (N / 4) rows per pass -- whole rows are contiguous reads, where the
// 64-column blocks of colsum_kernel touch 256 of a row's 384 bytes per x-block with half of the second block's lanes idle
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const void* __restrict__ X, int dt, const float* __restrict__ rw,
                                                            float* __restrict__ ws, int64_t M, int N, int64_t chunk) {
  __shared__ float red[256][4];
  const int L4 = N >> 2, R = 256 / L4;
  const int tx = threadIdx.x % L4, ty = threadIdx.x / L4;
  const int64_t batch = blockIdx.z;
  const int64_t mbeg = (int64_t)blockIdx.y * chunk, mend = min(M, mbeg + chunk);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (ty < R) {
    for (int64_t m = mbeg + ty; m < mend; m += R) {
      const int64_t at = (batch * M + m) * N + 4 * tx;
      f32x4 v;
      if (dt == CSTS_F32) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(X) + at);
      else {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(X) + at);
        v = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      }
      const float wgt = rw ? rw[batch * M + m] : 1.f;
      s[0] += v[0] * wgt; s[1] += v[1] * wgt; s[2] += v[2] * wgt; s[3] += v[3] * wgt;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = s[j];
  __syncthreads();
  if (ty == 0) {
    for (int r = 1; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += red[r * L4 + tx][j];
    float* o = ws + ((int64_t)blockIdx.y * gridDim.z + batch) * N + 4 * tx;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
  }
}

// ------------------------------------------------------------------ row softmax with temperature (frame_softmax)
// slowfast/utils/utils.py:5-12 ; one workgroup per (b,t) row of H*W logits
template <bool BWD>
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ x, const float* __restrict__ dp,
                                                      float* __restrict__ out, int n, float inv_temp) {
  __shared__ float red[8];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * n;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  auto block_reduce = [&](float v, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
  };
  float mx = -INFINITY;
  for (int i = tid; i < n; i += 256) mx = fmaxf(mx, xr[i] * inv_temp);
  mx = block_reduce(mx, true);
  float s = 0.f;
  for (int i = tid; i < n; i += 256) s += __expf(xr[i] * inv_temp - mx);
  s = block_reduce(s, false);
  const float inv = 1.f / s;
  if (!BWD) {
    for (int i = tid; i < n; i += 256) out[row * n + i] = __expf(xr[i] * inv_temp - mx) * inv;
  } else {
    const float* dr = dp + row * n;
    float dot = 0.f;
    for (int i = tid; i < n; i += 256) dot += dr[i] * __expf(xr[i] * inv_temp - mx) * inv;
    dot = block_reduce(dot, false);
    for (int i = tid; i < n; i += 256) {
      const float p = __expf(xr[i] * inv_temp - mx) * inv;
      out[row * n + i] = p * (dr[i] - dot) * inv_temp;
    }
  }
}

// KLDiv (losses.py:59-82): row = sum_i p (log(p+eps) - log(q+eps))
__global__ __launch_bounds__(256) void kldiv_fwd_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                        float* __restrict__ rowloss, int n) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float pv = p[row * n + i];
    const float lq = q ? logf(q[row * n + i] + 1e-10f) : 0.f;
    s += pv * (logf(pv + 1e-10f) - lq);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) rowloss[row] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void kldiv_bwd_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                        const float* __restrict__ gout, float* __restrict__ dp,
                                                        int64_t total, float scale) {
  const float g = gout[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float pv = p[i];
    const float lq = q ? logf(q[i] + 1e-10f) : 0.f;
    dp[i] = g * (logf(pv + 1e-10f) + pv / (pv + 1e-10f) - lq);
  }
}

// ------------------------------------------------------------------ sim_matrix row normalisation (utils.py:15-24)
__global__ void rownorm_fwd_kernel(const float* __restrict__ a, float* __restrict__ an, float* __restrict__ nrm, int D,
                                   float eps) {
  const int64_t row = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 64) { const float v = a[row * D + i]; s += v * v; }
  const float n = sqrtf(wave_sum(s));
  const float d = fmaxf(n, eps);
  for (int i = threadIdx.x; i < D; i += 64) an[row * D + i] = a[row * D + i] / d;
  if (threadIdx.x == 0) nrm[row] = n;
}
__global__ void rownorm_bwd_kernel(const float* __restrict__ a, const float* __restrict__ nrm,
                                   const float* __restrict__ dan, float* __restrict__ da, int D, float eps) {
  const int64_t row = blockIdx.x;
  const float n = nrm[row];
  if (n > eps) {
    float dot = 0.f;
    for (int i = threadIdx.x; i < D; i += 64) dot += a[row * D + i] * dan[row * D + i];
    dot = wave_sum(dot);
    for (int i = threadIdx.x; i < D; i += 64) da[row * D + i] = dan[row * D + i] / n - a[row * D + i] * dot / (n * n * n);
  } else {
    for (int i = threadIdx.x; i < D; i += 64) da[row * D + i] = dan[row * D + i] / eps;
  }
}

// ------------------------------------------------------------------ EgoNCE (losses.py:157-170)
// loss = -(1/n) sum_i [ log softmax_row(x/tau)_ii + log softmax_col(x/tau)_ii ]
__global__ __launch_bounds__(256) void egonce_fwd_kernel(const float* __restrict__ x, float* __restrict__ loss,
                                                         float* __restrict__ lse_row, float* __restrict__ lse_col, int n,
                                                         float inv_tau) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float mr = -INFINITY, mc = -INFINITY;
    for (int j = 0; j < n; ++j) { mr = fmaxf(mr, x[i * n + j] * inv_tau); mc = fmaxf(mc, x[j * n + i] * inv_tau); }
    float sr = 0.f, sc = 0.f;
    for (int j = 0; j < n; ++j) { sr += expf(x[i * n + j] * inv_tau - mr); sc += expf(x[j * n + i] * inv_tau - mc); }
    const float lr = mr + logf(sr), lc = mc + logf(sc);
    lse_row[i] = lr; lse_col[i] = lc;
    const float d = x[i * n + i] * inv_tau;
    acc += (d - lr) + (d - lc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = -(red[0] + red[1] + red[2] + red[3]) / n;
}
__global__ void egonce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ lse_row,
                                  const float* __restrict__ lse_col, const float* __restrict__ gout,
                                  float* __restrict__ dx, int n, float inv_tau) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  const int i = idx / n, j = idx % n;
  const float z = x[idx] * inv_tau;
  const float dl = (i == j) ? 2.f : 0.f;
  dx[idx] = -gout[0] / n * inv_tau * (dl - expf(z - lse_row[i]) - expf(z - lse_col[j]));
}

// ------------------------------------------------------------------ glue (custom_multimodal_builder.py:454-461,493-494)
// y[b, t*HW+s, c] = x[b, t*HW+s, c] * w[b, t, c]
__global__ __launch_bounds__(256) void reweight_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, int64_t total, int HW, int C) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const int64_t bt = idx / C / HW;
    y[idx] = x[idx] * w[bt * C + c];
  }
}
// dw[b,t,c] = sum_s dy * x       (block = 64 channels x 4 lanes over the HW positions of one (b, t); fixed-order fold in LDS)
__global__ __launch_bounds__(256) void reweight_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dw, int64_t nbtc, int HW, int C) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int64_t bt = blockIdx.y;
  float s = 0.f;
  if (c < C)
    for (int i = ty; i < HW; i += 4) { const int64_t o = (bt * HW + i) * C + c; s += x[o] * dy[o]; }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) dw[bt * C + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
// out[b,c] = mean_n x[b,n,c]   ;  bwd: dx[b,n,c] = dout[b,c] / N      (block = 16 channels x 16 token lanes: B * C / 16
// workgroups -- 192 for the b=4 embeddings instead of the 48 of a 64 x 4 block, each lane with 4 independent partial sums)
__global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t nbc,
                                                         int N, int C) {
  __shared__ float red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx;
  const int64_t b = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* p = x + b * N * C + c;
    int n = ty;
    for (; n + 48 < N; n += 64) {
      s0 += p[(int64_t)n * C]; s1 += p[(int64_t)(n + 16) * C]; s2 += p[(int64_t)(n + 32) * C]; s3 += p[(int64_t)(n + 48) * C];
    }
    for (; n < N; n += 16) s0 += p[(int64_t)n * C];
  }
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += red[l][tx];
    out[b * C + c] = t / N;
  }
}
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx,
                                                             int64_t total, int N, int C) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const int64_t b = idx / C / N;
    dx[idx] = dout[b * C + c] / N;
  }
}
// out = alpha*a + beta*b (b optional), dtype-generic; also the cast kernel
__global__ __launch_bounds__(256) void axpby_kernel(const void* __restrict__ a, int a_dt, const void* __restrict__ b, int b_dt,
                                                    void* __restrict__ out, int o_dt, int64_t total, float alpha, float beta) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float v = alpha * ld_as_f32(a, a_dt, i);
    if (b) v += beta * ld_as_f32(b, b_dt, i);
    st_from_f32(out, o_dt, i, v);
  }
}
// the same, 8 elements per thread (n % 8 == 0, 16-byte aligned operands): the 16-bit gradient buckets cast 188 M elements per step
__global__ __launch_bounds__(256) void axpby8_kernel(const void* __restrict__ a, int a_dt, const void* __restrict__ b, int b_dt,
                                                     void* __restrict__ out, int o_dt, int64_t total8, float alpha, float beta) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8], w[8];
    ld8_as_f32(a, a_dt, i * 8, v);
    if (b) ld8_as_f32(b, b_dt, i * 8, w);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = b ? alpha * v[j] + beta * w[j] : alpha * v[j];
    st8_from_f32(out, o_dt, i * 8, v);
  }
}
// out = a + b in fp32 and, optionally, its bf16 copy (gradient of a residual-stream tensor with two consumers); 4 per thread
// copy_scale (optional): the bf16 copy (only the copy) is multiplied by copy_scale[i / elems_per_scale] -- the per-sample
// drop-path scale of the branch that consumes this gradient next (see csts_layernorm_bwd_ex); elems_per_scale % 4 == 0
__global__ __launch_bounds__(256) void add2_kernel(const void* __restrict__ a, int a_dt, const void* __restrict__ b, int b_dt,
                                                   float* __restrict__ out, bf16* __restrict__ out16, int64_t total,
                                                   const float* __restrict__ copy_scale, int64_t elems_per_scale) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * blockDim.x * 4) {
    float v[4];
    const float sc = copy_scale ? copy_scale[i / elems_per_scale] : 1.f;
    if (i + 4 <= total && a_dt == CSTS_F32 && b_dt == CSTS_F32) {
      const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a) + i);
      const float4 y = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(b) + i);
      v[0] = x.x + y.x; v[1] = x.y + y.y; v[2] = x.z + y.z; v[3] = x.w + y.w;
      *reinterpret_cast<float4*>(out + i) = make_float4(v[0], v[1], v[2], v[3]);
      if (out16) {
        const bf16x4 w = {(bf16)(v[0] * sc), (bf16)(v[1] * sc), (bf16)(v[2] * sc), (bf16)(v[3] * sc)};
        *reinterpret_cast<bf16x4*>(out16 + i) = w;
      }
    } else {
      for (int j = 0; j < 4 && i + j < total; ++j) {
        const float s_ = ld_as_f32(a, a_dt, i + j) + ld_as_f32(b, b_dt, i + j);
        out[i + j] = s_;
        if (out16) out16[i + j] = (bf16)(s_ * sc);
      }
    }
  }
}
// Many bf16 matrices transposed in ONE launch (the [K][N] twins of the Linear weights' bf16 shadows, refreshed once per
// training step): one 64 x 64 tile per workgroup, described by a device table.
__global__ __launch_bounds__(256) void transpose_multi_kernel(const csts_transpose_tile* __restrict__ tiles) {
  __shared__ bf16 t[64][64 + 8];
  const csts_transpose_tile d = tiles[blockIdx.x];
  const bf16* __restrict__ src = reinterpret_cast<const bf16*>(d.src);
  bf16* __restrict__ dst = reinterpret_cast<bf16*>(d.dst);
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {                       // 64 rows x 8 chunks of 8 elements
    const int c = tid + 256 * i, r = c >> 3, cc = (c & 7) * 8;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
    if (d.r0 + r < d.R && d.c0 + cc < d.C) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)(d.r0 + r) * d.C + d.c0 + cc);
#pragma unroll
    for (int j = 0; j < 8; ++j) t[r][cc + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {                       // output rows = source columns
    const int c = tid + 256 * i, r = c >> 3, cc = (c & 7) * 8;
    if (d.c0 + r < d.C && d.r0 + cc < d.R) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = t[cc + j][r];
      *reinterpret_cast<bf16x8*>(dst + (int64_t)(d.c0 + r) * d.R + d.r0 + cc) = v;
    }
  }
}
// ---- rows of a token grid that a strided 3x3x3 pool reads (csts_kv_rows_geom, include/csts_hip.h).
// compact cell ch of an axis <-> fine index ((ch + 1) / 3) * s + (ch + 1) % 3 - 1
__device__ __forceinline__ int kv_fine_index(int ch, int s) { const int q = (ch + 1) / 3; return q * s + (ch + 1 - 3 * q) - 1; }
// dst[b, t, ch, cw, :] = src[b, t, h(ch), w(cw), :]   (16 bytes per thread: 8 bf16 / 4 fp32 channels)
template <int EB>   // element bytes
__global__ __launch_bounds__(256) void rows_gather_kernel(csts_kv_rows_geom g, const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                          int64_t total, int cv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    int64_t r = i / cv;
    const int cw = (int)(r % g.Wc); r /= g.Wc;
    const int ch = (int)(r % g.Hc); r /= g.Hc;            // r = b * T + t
    const int64_t frow = (r * g.H + kv_fine_index(ch, g.sh)) * g.W + kv_fine_index(cw, g.sw);
    dst[i] = src[frow * cv + c];
  }
}
// dst[b, t, h(ch), w(cw), :] += src[b, t, ch, cw, :]   (every fine row is the image of at most one compact row: no races)
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(csts_kv_rows_geom g, const void* __restrict__ src, int s_dt,
                                                               void* __restrict__ dst, int d_dt, int64_t total, int c8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8);
    int64_t r = i / c8;
    const int64_t crow = r;
    const int cw = (int)(r % g.Wc); r /= g.Wc;
    const int ch = (int)(r % g.Hc); r /= g.Hc;
    const int64_t frow = (r * g.H + kv_fine_index(ch, g.sh)) * g.W + kv_fine_index(cw, g.sw);
    float a[8], b[8];
    ld8_as_f32(src, s_dt, (crow * c8 + c) * 8, a);
    ld8_as_f32(dst, d_dt, (frow * c8 + c) * 8, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] += a[j];
    st8_from_f32(dst, d_dt, (frow * c8 + c) * 8, b);
  }
}
// out[m,n] = x[m,n] * row_scale[m / rows_per_scale]   (backward of the drop-path scaling, common.py:46-59)
__global__ __launch_bounds__(256) void scale_rows_kernel(const void* __restrict__ x, int x_dt, const float* __restrict__ rs,
                                                         int64_t rows_per_scale, void* __restrict__ out, int o_dt,
                                                         int64_t total, int64_t N) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    st_from_f32(out, o_dt, i, ld_as_f32(x, x_dt, i) * rs[(i / N) / rows_per_scale]);
}
// out[m] = sum_c x[m,c] * w[c] + bias   (classifier Conv3d(96,1,1): custom_multimodal_builder.py:301,481); 32 lanes/row
__global__ __launch_bounds__(256) void rowdot_kernel(const void* __restrict__ x, int x_dt, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, int64_t M, int C) {
  const int sub = threadIdx.x & 31;
  const int64_t row0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 5;
  for (int64_t m = row0; m < M; m += stride) {
    float s = 0.f;
    for (int c = sub; c < C; c += 32) s += ld_as_f32(x, x_dt, m * C + c) * w[c];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0) out[m] = s + (bias ? bias[0] : 0.f);
  }
}
// dx[m,c] = dout[m] * w[c]
__global__ __launch_bounds__(256) void rowdot_dx_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                        void* __restrict__ dx, int dx_dt, int64_t total, int C) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    st_from_f32(dx, dx_dt, idx, dout[idx / C] * w[idx % C]);
}

// the same, 4 channels per thread (C % 4 == 0, fewer than 2^31 elements): 32-bit indices, 16-byte fp32 / 8-byte bf16 stores
__global__ __launch_bounds__(256) void rowdot_dx4_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                         void* __restrict__ dx, int dx_dt, int total4, int C4) {
  for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += gridDim.x * blockDim.x) {
    const int m = i4 / C4, c = (i4 - m * C4) * 4;
    const float d = dout[m];
    const float4 wv = *reinterpret_cast<const float4*>(w + c);
    if (dx_dt == CSTS_F32) {
      reinterpret_cast<float4*>(dx)[i4] = make_float4(d * wv.x, d * wv.y, d * wv.z, d * wv.w);
    } else {
      bf16x4 o;
      o[0] = (bf16)(d * wv.x); o[1] = (bf16)(d * wv.y); o[2] = (bf16)(d * wv.z); o[3] = (bf16)(d * wv.w);
      reinterpret_cast<bf16x4*>(dx)[i4] = o;
    }
  }
}

// c[m] = sum_n a[m,n] * b[m,n]   (gradient of a per-row weight: d w[m] = <dy[m,:], x[m,:]>); 32 lanes per row
__global__ __launch_bounds__(256) void rowdot2_kernel(const void* __restrict__ a, int a_dt, const void* __restrict__ b, int b_dt,
                                                      float* __restrict__ out, int64_t M, int C) {
  const int sub = threadIdx.x & 31;
  const int64_t row0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 5;
  for (int64_t m = row0; m < M; m += stride) {
    float s = 0.f;
    for (int c = sub; c < C; c += 32) s += ld_as_f32(a, a_dt, m * C + c) * ld_as_f32(b, b_dt, m * C + c);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0) out[m] = s;
  }
}

// ---- audio -> pixel attention map of the spatial fusion block (av_attention.py:356-370).
// Tokens of the block: T*HW video tokens (t-major) followed by T audio tokens; audio token t attends to the HW video
// tokens of frame t and to itself (the -1e8 mask of av_attention.py:336-346 leaves exactly these: exp underflows to 0
// for every other key).  Per head: p = softmax(scale * q_a . k_j) over those HW + 1 keys; the map is p restricted to
// the HW video keys, min-max rescaled: r = (p - min) / (max - min + 1e-8); custom_multimodal_builder.py:438 then
// averages r over the heads.  One 256-thread workgroup per (clip, frame); everything in fp32.
// block-wide reductions (all 256 threads call; `red` = 8 floats of LDS scratch)
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max256(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// probabilities of head h for (b, t) into p[0 .. HW] (p[HW] = the audio token itself); q (hd floats) is left in LDS
__device__ __forceinline__ void audio_attn_probs(const void* __restrict__ qkv, int dt, int64_t tok0, int t, int T, int HW, int C,
                                                 int h, int hd, float scale, float* q, float* p, float* red) {
  const int tid = threadIdx.x;
  const int64_t ld = 3 * (int64_t)C, arow = tok0 + (int64_t)T * HW + t;
  __syncthreads();                                         // previous head is done with q / p
  for (int d = tid; d < hd; d += 256) q[d] = ld_as_f32(qkv, dt, arow * ld + h * hd + d);
  __syncthreads();
  for (int j = tid; j <= HW; j += 256) {
    const int64_t row = j < HW ? tok0 + (int64_t)t * HW + j : arow;
    float acc = 0.f;
    for (int d = 0; d < hd; d += 8) {
      float kk[8];
      ld8_as_f32(qkv, dt, row * ld + C + h * hd + d, kk);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += q[d + e] * kk[e];
    }
    p[j] = acc * scale;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int j = tid; j <= HW; j += 256) m = fmaxf(m, p[j]);
  m = block_max256(m, red);
  float s = 0.f;
  for (int j = tid; j <= HW; j += 256) { const float e = __expf(p[j] - m); p[j] = e; s += e; }
  s = block_sum256(s, red);
  const float inv = 1.f / s;
  for (int j = tid; j <= HW; j += 256) p[j] *= inv;
  __syncthreads();
}

__global__ __launch_bounds__(256) void audio_attn_fwd_kernel(const void* __restrict__ qkv, int dt, float* __restrict__ aa,
                                                             float* __restrict__ wmap, int T, int HW, int C, int H, float scale) {
  extern __shared__ float sm[];
  const int hd = C / H, tid = threadIdx.x;
  float* q = sm; float* p = q + hd; float* acc = p + HW + 1; float* red = acc + HW;
  const int b = blockIdx.x / T, t = blockIdx.x % T;
  const int64_t tok0 = (int64_t)b * (T * HW + T);
  for (int j = tid; j < HW; j += 256) acc[j] = 0.f;
  for (int h = 0; h < H; ++h) {
    audio_attn_probs(qkv, dt, tok0, t, T, HW, C, h, hd, scale, q, p, red);
    float mx = -INFINITY, mn = INFINITY;
    for (int j = tid; j < HW; j += 256) { mx = fmaxf(mx, p[j]); mn = fminf(mn, p[j]); }
    mx = block_max256(mx, red);
    mn = -block_max256(-mn, red);
    const float invd = 1.f / (mx - mn + 1e-8f);
    for (int j = tid; j < HW; j += 256) {
      const float r = (p[j] - mn) * invd;
      if (aa != nullptr) aa[(((int64_t)b * H + h) * T + t) * HW + j] = r;
      acc[j] += r;
    }
  }
  for (int j = tid; j < HW; j += 256) wmap[((int64_t)b * T + t) * HW + j] = acc[j] / (float)H;
}

// backward: d(wmap) -> d(q of audio token t), d(k of the HW video tokens of frame t and of the audio token), written into
// a zero-initialised buffer shaped like qkv (no other workgroup touches these rows / slots).
__global__ __launch_bounds__(256) void audio_attn_bwd_kernel(const void* __restrict__ qkv, int dt, const float* __restrict__ dwmap,
                                                             void* __restrict__ dqkv, int T, int HW, int C, int H, float scale) {
  extern __shared__ float sm[];
  const int hd = C / H, tid = threadIdx.x;
  float* q = sm; float* p = q + hd; float* dl = p + HW + 1; float* red = dl + HW + 1;
  const int b = blockIdx.x / T, t = blockIdx.x % T;
  const int64_t tok0 = (int64_t)b * (T * HW + T), ld = 3 * (int64_t)C, arow = tok0 + (int64_t)T * HW + t;
  for (int h = 0; h < H; ++h) {
    audio_attn_probs(qkv, dt, tok0, t, T, HW, C, h, hd, scale, q, p, red);
    float mx = -INFINITY, mn = INFINITY;
    for (int j = tid; j < HW; j += 256) { mx = fmaxf(mx, p[j]); mn = fminf(mn, p[j]); }
    mx = block_max256(mx, red);
    mn = -block_max256(-mn, red);
    // first index that attains the extremum (torch.max / torch.min route the gradient to one index)
    float imx = 3.0e38f, imn = 3.0e38f;
    for (int j = tid; j < HW; j += 256) { if (p[j] == mx) imx = fminf(imx, (float)j); if (p[j] == mn) imn = fminf(imn, (float)j); }
    const int jmx = (int)(-block_max256(-imx, red)), jmn = (int)(-block_max256(-imn, red));
    const float invd = 1.f / (mx - mn + 1e-8f);
    float sg = 0.f, sgr = 0.f;
    for (int j = tid; j < HW; j += 256) {
      const float g = dwmap[((int64_t)b * T + t) * HW + j] / (float)H;
      sg += g; sgr += g * (p[j] - mn) * invd;
    }
    sg = block_sum256(sg, red);
    sgr = block_sum256(sgr, red);
    float dot = 0.f;                                        // sum_j p_j dp_j over the video keys (dp of the self key is 0)
    for (int j = tid; j < HW; j += 256) {
      float dp = dwmap[((int64_t)b * T + t) * HW + j] / (float)H * invd;
      if (j == jmn) dp += (sgr - sg) * invd;
      if (j == jmx) dp -= sgr * invd;
      dl[j] = dp;
      dot += p[j] * dp;
    }
    dot = block_sum256(dot, red);
    for (int j = tid; j <= HW; j += 256) dl[j] = p[j] * ((j < HW ? dl[j] : 0.f) - dot) * scale;   // d logits * scale
    __syncthreads();
    for (int d = tid; d < hd; d += 256) {                   // dq = sum_j dl_j k_j
      float acc = 0.f;
      for (int j = 0; j <= HW; ++j) {
        const int64_t row = j < HW ? tok0 + (int64_t)t * HW + j : arow;
        acc += dl[j] * ld_as_f32(qkv, dt, row * ld + C + h * hd + d);
      }
      st_from_f32(dqkv, dt, arow * ld + h * hd + d, acc);
    }
    for (int idx = tid; idx < (HW + 1) * hd; idx += 256) {  // dk_j = dl_j q
      const int j = idx / hd, d = idx - j * hd;
      const int64_t row = j < HW ? tok0 + (int64_t)t * HW + j : arow;
      st_from_f32(dqkv, dt, row * ld + C + h * hd + d, dl[j] * q[d]);
    }
  }
}

}  // namespace

extern "C" int csts_im2col(const csts_im2col_geom* g, const void* x, int x_dt, void* col, int col_dt, hipStream_t stream) {
  CSTS_REQUIRE(g && x && col, "null pointer");
  Im2colP p;
  p.B = g->B; p.Cin = g->Cin; p.T = g->T; p.H = g->H; p.W = g->W;
  p.KT = g->kernel[0]; p.KH = g->kernel[1]; p.KW = g->kernel[2];
  p.st = g->stride[0]; p.sh = g->stride[1]; p.sw = g->stride[2];
  p.pt = g->padding[0]; p.ph = g->padding[1]; p.pw = g->padding[2];
  CSTS_REQUIRE(p.B > 0 && p.Cin > 0 && p.st > 0 && p.sh > 0 && p.sw > 0, "bad geometry");
  p.To = (p.T + 2 * p.pt - p.KT) / p.st + 1;
  p.Ho = (p.H + 2 * p.ph - p.KH) / p.sh + 1;
  p.Wo = (p.W + 2 * p.pw - p.KW) / p.sw + 1;
  p.K = p.Cin * p.KT * p.KH * p.KW;
  p.Kpad = g->Kpad;
  CSTS_REQUIRE(p.Kpad >= p.K, "Kpad < K");
  CSTS_REQUIRE(p.To == g->To && p.Ho == g->Ho && p.Wo == g->Wo, "output grid mismatch");
  const size_t strip_lds = (size_t)p.Cin * p.KT * p.KH * p.W * (col_dt == CSTS_F32 ? 4 : 2);
  const int64_t strips = (int64_t)p.B * p.To * p.Ho;
  if (strip_lds <= 65536 && p.W % 4 == 0 && p.Kpad % 8 == 0 && strips < ((int64_t)1 << 31) && aligned16(x) &&
      aligned16(col)) {
    const bool xf = x_dt == CSTS_F32, cf = col_dt == CSTS_F32;
    const dim3 grid((unsigned)strips), block(256);
    if (xf && cf) hipLaunchKernelGGL((im2col_strip_kernel<true, true>), grid, block, strip_lds, stream, p, x, col);
    else if (xf) hipLaunchKernelGGL((im2col_strip_kernel<true, false>), grid, block, strip_lds, stream, p, x, col);
    else if (cf) hipLaunchKernelGGL((im2col_strip_kernel<false, true>), grid, block, strip_lds, stream, p, x, col);
    else hipLaunchKernelGGL((im2col_strip_kernel<false, false>), grid, block, strip_lds, stream, p, x, col);
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  const int64_t total = (int64_t)p.B * p.To * p.Ho * p.Wo * p.Kpad;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, stream, p, x, x_dt, col, col_dt);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_posembed_build(const float* spatial, const float* temporal, float* pos, int T, int HW, int C,
                                   hipStream_t stream) {
  CSTS_REQUIRE(spatial && temporal && pos && T > 0 && HW > 0 && C > 0, "bad args");
  hipLaunchKernelGGL(posembed_kernel, dim3(grid_for((int64_t)T * HW * C)), dim3(256), 0, stream, spatial, temporal, pos, T, HW, C);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_transpose_batched(const void* in, int in_dt, void* out, int out_dt, int64_t batch, int R, int Cc,
                                      hipStream_t stream) {
  CSTS_REQUIRE(in && out && batch > 0 && batch < 65536 && R > 0 && Cc > 0, "bad args");
  dim3 grid((unsigned)cdiv(Cc, 32), (unsigned)cdiv(R, 32), (unsigned)batch);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, in, in_dt, out, out_dt, R, Cc);
  CSTS_LAUNCH_CHECK();
  return 0;
}

static void colsum_plan(int64_t M, int64_t& chunk, int64_t& nchunk) {
  nchunk = std::max<int64_t>(1, std::min<int64_t>(256, M / 64));
  chunk = cdiv(M, nchunk);
  nchunk = cdiv(M, chunk);
}
extern "C" size_t csts_colsum_workspace(int64_t batch, int64_t M, int64_t N) {
  int64_t chunk, nchunk;
  colsum_plan(M, chunk, nchunk);
  return (size_t)nchunk * batch * N * sizeof(float);
}
extern "C" int csts_colsum(const void* X, int dt, const float* row_weight, float* out, int64_t batch, int64_t M, int64_t N,
                           void* workspace, size_t ws_bytes, hipStream_t stream) {
  CSTS_REQUIRE(X && out && workspace && batch > 0 && batch < 65536 && M > 0 && N > 0, "bad args");
  int64_t chunk, nchunk;
  colsum_plan(M, chunk, nchunk);
  CSTS_REQUIRE(ws_bytes >= (size_t)nchunk * batch * N * sizeof(float), "workspace too small");
  float* ws = reinterpret_cast<float*>(workspace);
  if (N % 4 == 0 && N >= 16 && N <= 256 && M >= 4096 && aligned16(X)) {
    hipLaunchKernelGGL(colsum_narrow_kernel, dim3(1, (unsigned)nchunk, (unsigned)batch), dim3(256), 0, stream, X, dt, row_weight, ws,
                       M, (int)N, chunk);
  } else {
    dim3 grid((unsigned)cdiv(N, 64), (unsigned)nchunk, (unsigned)batch);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, stream, X, dt, row_weight, ws, M, N, chunk);
  }
  CSTS_LAUNCH_CHECK();
  csts_reduce_rows_launch(ws, out, nchunk, batch * N, 1.f, stream);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_softmax_fwd(const float* x, float* p, int64_t rows, int n, float temperature, hipStream_t stream) {
  CSTS_REQUIRE(x && p && rows > 0 && n > 0 && temperature > 0.f, "bad args");
  hipLaunchKernelGGL(softmax_kernel<false>, dim3((unsigned)rows), dim3(256), 0, stream, x, nullptr, p, n, 1.f / temperature);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_softmax_bwd(const float* x, const float* dp, float* dx, int64_t rows, int n, float temperature,
                                hipStream_t stream) {
  CSTS_REQUIRE(x && dp && dx && rows > 0 && n > 0 && temperature > 0.f, "bad args");
  hipLaunchKernelGGL(softmax_kernel<true>, dim3((unsigned)rows), dim3(256), 0, stream, x, dp, dx, n, 1.f / temperature);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_kldiv_fwd(const float* p, const float* q, float* loss, float* rowloss_ws, int64_t rows, int n,
                              float scale, hipStream_t stream) {
  CSTS_REQUIRE(p && loss && rowloss_ws && rows > 0 && n > 0, "bad args");
  hipLaunchKernelGGL(kldiv_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, p, q, rowloss_ws, n);
  CSTS_LAUNCH_CHECK();
  csts_reduce_rows_launch(rowloss_ws, loss, rows, 1, scale, stream);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_kldiv_bwd(const float* p, const float* q, const float* grad_out, float* dp, int64_t rows, int n,
                              float scale, hipStream_t stream) {
  CSTS_REQUIRE(p && grad_out && dp && rows > 0 && n > 0, "bad args");
  hipLaunchKernelGGL(kldiv_bwd_kernel, dim3(grid_for(rows * n)), dim3(256), 0, stream, p, q, grad_out, dp, rows * n, scale);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_rownorm_fwd(const float* a, float* a_normed, float* norms, int64_t rows, int D, float eps,
                                hipStream_t stream) {
  CSTS_REQUIRE(a && a_normed && norms && rows > 0 && D > 0, "bad args");
  hipLaunchKernelGGL(rownorm_fwd_kernel, dim3((unsigned)rows), dim3(64), 0, stream, a, a_normed, norms, D, eps);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_rownorm_bwd(const float* a, const float* norms, const float* d_normed, float* da, int64_t rows, int D,
                                float eps, hipStream_t stream) {
  CSTS_REQUIRE(a && norms && d_normed && da && rows > 0 && D > 0, "bad args");
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)rows), dim3(64), 0, stream, a, norms, d_normed, da, D, eps);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_egonce_fwd(const float* sim, float* loss, float* lse_row, float* lse_col, int n, float temperature,
                               hipStream_t stream) {
  CSTS_REQUIRE(sim && loss && lse_row && lse_col && n > 0 && temperature > 0.f, "bad args");
  hipLaunchKernelGGL(egonce_fwd_kernel, dim3(1), dim3(256), 0, stream, sim, loss, lse_row, lse_col, n, 1.f / temperature);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_egonce_bwd(const float* sim, const float* lse_row, const float* lse_col, const float* grad_out,
                               float* dsim, int n, float temperature, hipStream_t stream) {
  CSTS_REQUIRE(sim && lse_row && lse_col && grad_out && dsim && n > 0, "bad args");
  hipLaunchKernelGGL(egonce_bwd_kernel, dim3((unsigned)cdiv((int64_t)n * n, 256)), dim3(256), 0, stream, sim, lse_row, lse_col,
                     grad_out, dsim, n, 1.f / temperature);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_reweight_fwd(const float* x, const float* w, float* y, int64_t BT, int HW, int C, hipStream_t stream) {
  CSTS_REQUIRE(x && w && y && BT > 0 && HW > 0 && C > 0, "bad args");
  const int64_t total = BT * HW * C;
  hipLaunchKernelGGL(reweight_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, w, y, total, HW, C);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_reweight_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, int64_t BT, int HW,
                                 int C, hipStream_t stream) {
  CSTS_REQUIRE(x && w && dy && BT > 0 && HW > 0 && C > 0, "bad args");
  const int64_t total = BT * HW * C;
  if (dx) hipLaunchKernelGGL(reweight_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, w, dx, total, HW, C);
  if (dw) hipLaunchKernelGGL(reweight_dw_kernel, dim3((unsigned)cdiv(C, 64), (unsigned)BT), dim3(256), 0, stream, x, dy, dw, BT * C, HW, C);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_token_mean_fwd(const float* x, float* out, int64_t B, int N, int C, hipStream_t stream) {
  CSTS_REQUIRE(x && out && B > 0 && N > 0 && C > 0, "bad args");
  hipLaunchKernelGGL(token_mean_kernel, dim3((unsigned)cdiv(C, 16), (unsigned)B), dim3(256), 0, stream, x, out, B * C, N, C);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_token_mean_bwd(const float* dout, float* dx, int64_t B, int N, int C, hipStream_t stream) {
  CSTS_REQUIRE(dout && dx && B > 0 && N > 0 && C > 0, "bad args");
  const int64_t total = B * N * C;
  hipLaunchKernelGGL(token_mean_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dout, dx, total, N, C);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_axpby(const void* a, int a_dt, const void* b, int b_dt, void* out, int out_dt, int64_t n, float alpha,
                          float beta, hipStream_t stream) {
  CSTS_REQUIRE(a && out && n > 0, "bad args");
  if (n % 8 == 0 && aligned16(a) && aligned16(out) && (b == nullptr || aligned16(b)))
    hipLaunchKernelGGL(axpby8_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, a, a_dt, b, b_dt, out, out_dt, n / 8, alpha, beta);
  else
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, a_dt, b, b_dt, out, out_dt, n, alpha, beta);
  CSTS_LAUNCH_CHECK();
  return 0;
}
static int kv_rows_check(const csts_kv_rows_geom* g) {
  CSTS_REQUIRE(g != nullptr && g->B > 0 && g->T > 0 && g->H > 0 && g->W > 0 && g->C > 0 && g->C % 8 == 0, "bad geometry (C % 8)");
  CSTS_REQUIRE(g->sh >= 3 && g->sw >= 3 && g->Hc > 0 && g->Wc > 0, "compaction needs pool strides >= 3");
  // every compact cell must be a row of the fine grid
  CSTS_REQUIRE(((g->Hc) / 3) * g->sh + (g->Hc) % 3 - 1 < g->H && ((g->Wc) / 3) * g->sw + (g->Wc) % 3 - 1 < g->W, "compact grid exceeds the fine grid");
  return 0;
}
extern "C" int csts_rows_gather(const csts_kv_rows_geom* g, const void* src, int dt, void* dst, hipStream_t stream) {
  if (int rc = kv_rows_check(g)) return rc;
  CSTS_REQUIRE(src && dst && aligned16(src) && aligned16(dst) && (dt == CSTS_F32 || dt == CSTS_BF16), "bad args");
  const int cv = dt == CSTS_F32 ? g->C / 4 : g->C / 8;
  const int64_t total = (int64_t)g->B * g->T * g->Hc * g->Wc * cv;
  hipLaunchKernelGGL(rows_gather_kernel<2>, dim3(grid_for(total)), dim3(256), 0, stream, *g, reinterpret_cast<const uint4*>(src),
                     reinterpret_cast<uint4*>(dst), total, cv);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_rows_scatter_add(const csts_kv_rows_geom* g, const void* src, int src_dt, void* dst, int dst_dt, hipStream_t stream) {
  if (int rc = kv_rows_check(g)) return rc;
  CSTS_REQUIRE(src && dst && aligned16(src) && aligned16(dst), "bad args");
  const int64_t total = (int64_t)g->B * g->T * g->Hc * g->Wc * (g->C / 8);
  hipLaunchKernelGGL(rows_scatter_add_kernel, dim3(grid_for(total)), dim3(256), 0, stream, *g, src, src_dt, dst, dst_dt, total, g->C / 8);
  CSTS_LAUNCH_CHECK();
  return 0;
}
namespace {
struct TokenSegs { csts_token_segment s[4]; int64_t chunks_end[4]; };     // chunks_end[i]: 16-byte chunks of segments 0 .. i
__global__ __launch_bounds__(256) void copy_token_segments_kernel(TokenSegs ts, int nseg, int64_t total, int row16, int elsz) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int k = 0;
    while (k + 1 < nseg && i >= ts.chunks_end[k]) ++k;               // (wave-uniform except at a segment seam)
    const int64_t j = i - (k ? ts.chunks_end[k - 1] : 0);
    const int64_t per_b = (int64_t)ts.s[k].n * row16;                // chunks of this segment per batch element
    const int64_t b = j / per_b, r = j - b * per_b;
    const char* sp = reinterpret_cast<const char*>(ts.s[k].src) + (b * ts.s[k].src_bs + ts.s[k].src_off) * elsz + r * 16;
    char* dp = reinterpret_cast<char*>(ts.s[k].dst) + (b * ts.s[k].dst_bs + ts.s[k].dst_off) * elsz + r * 16;
    *reinterpret_cast<uint4*>(dp) = *reinterpret_cast<const uint4*>(sp);
  }
}
}  // namespace
extern "C" int csts_copy_token_segments(const csts_token_segment* segs, int nseg, int B, int C, int dt, hipStream_t stream) {
  CSTS_REQUIRE(segs && nseg >= 1 && nseg <= 4 && B > 0 && C > 0 && (dt == CSTS_F32 || dt == CSTS_BF16), "bad args");
  const int elsz = dt == CSTS_F32 ? 4 : 2;
  CSTS_REQUIRE(((int64_t)C * elsz) % 16 == 0, "rows must be whole 16-byte chunks");
  TokenSegs ts;
  int64_t total = 0;
  for (int i = 0; i < nseg; ++i) {
    const csts_token_segment& s = segs[i];
    CSTS_REQUIRE(s.src && s.dst && s.n >= 0 && aligned16(s.src) && aligned16(s.dst), "bad segment");
    CSTS_REQUIRE((s.src_bs * elsz) % 16 == 0 && (s.dst_bs * elsz) % 16 == 0 && (s.src_off * elsz) % 16 == 0 && (s.dst_off * elsz) % 16 == 0,
                 "segment strides / offsets must keep 16-byte alignment");
    ts.s[i] = s;
    total += (int64_t)B * s.n * ((int64_t)C * elsz / 16);
    ts.chunks_end[i] = total;
  }
  for (int i = nseg; i < 4; ++i) { ts.s[i] = ts.s[0]; ts.chunks_end[i] = total; }
  if (total == 0) return 0;
  hipLaunchKernelGGL(copy_token_segments_kernel, dim3(grid_for(total)), dim3(256), 0, stream, ts, nseg, total, (int)((int64_t)C * elsz / 16), elsz);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_scale_rows(const void* x, int x_dt, const float* row_scale, int64_t rows_per_scale, void* out, int out_dt,
                               int64_t M, int64_t N, hipStream_t stream) {
  CSTS_REQUIRE(x && row_scale && out && rows_per_scale > 0 && M > 0 && N > 0, "bad args");
  hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(M * N)), dim3(256), 0, stream, x, x_dt, row_scale, rows_per_scale, out,
                     out_dt, M * N, N);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_rowdot_fwd(const void* x, int x_dt, const float* w, const float* bias, float* out, int64_t M, int C,
                               hipStream_t stream) {
  CSTS_REQUIRE(x && w && out && M > 0 && C > 0, "bad args");
  hipLaunchKernelGGL(rowdot_kernel, dim3(grid_for(M * 32)), dim3(256), 0, stream, x, x_dt, w, bias, out, M, C);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_rowdot_dx(const float* dout, const float* w, void* dx, int dx_dt, int64_t M, int C, hipStream_t stream) {
  CSTS_REQUIRE(dout && w && dx && M > 0 && C > 0, "bad args");
  if (C % 4 == 0 && M * C < ((int64_t)1 << 31) && aligned16(w) && aligned16(dx))
    hipLaunchKernelGGL(rowdot_dx4_kernel, dim3(grid_for(M * C / 4)), dim3(256), 0, stream, dout, w, dx, dx_dt, (int)(M * C / 4), C / 4);
  else
    hipLaunchKernelGGL(rowdot_dx_kernel, dim3(grid_for(M * C)), dim3(256), 0, stream, dout, w, dx, dx_dt, M * C, C);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_rowdot2(const void* a, int a_dt, const void* b, int b_dt, float* out, int64_t M, int C, hipStream_t stream) {
  CSTS_REQUIRE(a && b && out && M > 0 && C > 0, "bad args");
  hipLaunchKernelGGL(rowdot2_kernel, dim3(grid_for(M * 32)), dim3(256), 0, stream, a, a_dt, b, b_dt, out, M, C);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_audio_attn_fwd(const void* qkv, int dt, float* audio_attn, float* wmap, int B, int T, int HW, int C, int H,
                                   float scale, hipStream_t stream) {
  CSTS_REQUIRE(qkv && wmap && B > 0 && T > 0 && HW > 0 && H > 0 && C % H == 0 && (C / H) % 8 == 0, "bad args");
  const size_t smem = ((size_t)(C / H) + 2 * (HW + 1) + 8) * sizeof(float);
  CSTS_REQUIRE(smem <= 64 * 1024, "frame too large for the audio-attention kernel");
  hipLaunchKernelGGL(audio_attn_fwd_kernel, dim3(B * T), dim3(256), smem, stream, qkv, dt, audio_attn, wmap, T, HW, C, H, scale);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_audio_attn_bwd(const void* qkv, int dt, const float* d_wmap, void* dqkv, int B, int T, int HW, int C, int H,
                                   float scale, hipStream_t stream) {
  CSTS_REQUIRE(qkv && d_wmap && dqkv && B > 0 && T > 0 && HW > 0 && H > 0 && C % H == 0 && (C / H) % 8 == 0, "bad args");
  const size_t smem = ((size_t)(C / H) + 2 * (HW + 1) + 8) * sizeof(float);
  CSTS_REQUIRE(smem <= 64 * 1024, "frame too large for the audio-attention kernel");
  hipLaunchKernelGGL(audio_attn_bwd_kernel, dim3(B * T), dim3(256), smem, stream, qkv, dt, d_wmap, dqkv, T, HW, C, H, scale);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_add2(const void* a, int a_dt, const void* b, int b_dt, float* out, void* out_bf16, int64_t n, hipStream_t stream) {
  return csts_add2_scaled_copy(a, a_dt, b, b_dt, out, out_bf16, nullptr, 4, n, stream);
}
extern "C" int csts_add2_scaled_copy(const void* a, int a_dt, const void* b, int b_dt, float* out, void* out_bf16,
                                     const float* copy_scale, int64_t elems_per_scale, int64_t n, hipStream_t stream) {
  CSTS_REQUIRE(a && b && out && n > 0, "bad args");
  CSTS_REQUIRE((aligned16(a) && aligned16(b) && aligned16(out) && (!out_bf16 || (reinterpret_cast<uintptr_t>(out_bf16) & 7) == 0)),
               "operands must be 16-byte aligned");
  CSTS_REQUIRE(copy_scale == nullptr || (out_bf16 != nullptr && elems_per_scale > 0 && elems_per_scale % 4 == 0),
               "copy scale needs the bf16 copy and elems_per_scale % 4 == 0");
  hipLaunchKernelGGL(add2_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, stream, a, a_dt, b, b_dt, out,
                     reinterpret_cast<bf16*>(out_bf16), n, copy_scale, elems_per_scale > 0 ? elems_per_scale : 4);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_transpose_multi(const csts_transpose_tile* device_tiles, int ntiles, hipStream_t stream) {
  CSTS_REQUIRE(device_tiles != nullptr && ntiles > 0, "no tiles");
  hipLaunchKernelGGL(transpose_multi_kernel, dim3((unsigned)ntiles), dim3(256), 0, stream, device_tiles);
  CSTS_LAUNCH_CHECK();
  return 0;
}
