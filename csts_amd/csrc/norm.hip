// LayerNorm forward / backward (K2 of SURVEY.md 2.3) and the shared partial-sum reducer.
// Reference: block norms nn.LayerNorm(C, eps=1e-6) (attention.py:192,214) and the per-head
// nn.LayerNorm(hd, eps=1e-5) applied to pooled q/k/v (attention.py:108,112,116) -- same kernel, rows = B*N*h.
// One wave per row, row kept in registers (C <= 768 -> <= 12 values per lane), two-pass statistics in fp32.
// HBM-bound: algorithmic bytes = rows*C*(in + out) (+ 8 B/row of statistics).
#include "common.h"

namespace {

constexpr int MAXV = 12;  // C <= 768
constexpr int LN_BWD_WAVES = 8;  // 512-thread blocks: few, fat partial rows for the second-stage reduction

// compile-time dtypes: no branch sits between a load and the next one, so a wave keeps R rows x NV loads in flight
template <bool F32> __device__ __forceinline__ float ldt(const void* p, int64_t i) {
  if constexpr (F32) return reinterpret_cast<const float*>(p)[i];
  else return h16_lo((unsigned)reinterpret_cast<const unsigned short*>(p)[i]);
}
template <bool F32> __device__ __forceinline__ void stt(void* p, int64_t i, float v) {
  if constexpr (F32) reinterpret_cast<float*>(p)[i] = v;
  else reinterpret_cast<bf16*>(p)[i] = (bf16)v;
}

// one wave handles R rows per iteration (R*NV independent loads), two-pass statistics in registers
// optional input of the forward kernels: x + addend is what gets normalised, and the sum is written to `sum` (dtype of x):
// the decoder's `feat + en_feat` skip (custom_multimodal_builder.py:467-479) folded into the next block's norm1
struct FwdAdd {
  const void* addend = nullptr;
  void* sum = nullptr;
};
template <int NV, int R, bool XF32, bool YF32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                     int C, float eps, FwdAdd fa) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  float gm[NV], bt[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = lane + 64 * j;
    gm[j] = c < C ? gamma[c] : 0.f;
    bt[j] = c < C ? beta[c] : 0.f;
  }
  const float invC = 1.f / C;
  for (int64_t row0 = wave * R; row0 < rows; row0 += nwaves * R) {
    float v[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = min(row0 + r, rows - 1);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = lane + 64 * j;
        v[r][j] = ldt<XF32>(x, row * C + min(c, C - 1));
        if (fa.addend) {                                         // wave-uniform
          v[r][j] += ldt<XF32>(fa.addend, row * C + min(c, C - 1));
          if (c < C && row0 + r < rows) stt<XF32>(fa.sum, row * C + c, v[r][j]);
        }
      }
    }
    float mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) s += (lane + 64 * j < C) ? v[r][j] : 0.f;
      mu[r] = wave_sum(s) * invC;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float d = (lane + 64 * j < C) ? v[r][j] - mu[r] : 0.f;
        q += d * d;
      }
      rs[r] = rsqrtf(wave_sum(q) * invC + eps);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = row0 + r;
      if (row >= rows) break;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int c = lane + 64 * j;
        if (c < C) stt<YF32>(y, row * C + c, (v[r][j] - mu[r]) * rs[r] * gm[j] + bt[j]);
      }
      if (lane == 0) {
        if (mean) mean[row] = mu[r];
        if (rstd) rstd[row] = rs[r];
      }
    }
  }
}

// optional extras of the backward kernels: a bf16 copy of dx (+ per-sample scale applied to the copy only: row r uses
// scale[r / rps]) and a second incoming gradient (LayerNorm output with two consumers)
struct Copy16 {
  bf16* p = nullptr;
  const float* scale = nullptr;
  int64_t rps = 1;
  const void* dy2 = nullptr;     // a second incoming gradient of the same shape / dtype as dy: the kernel reads dy + dy2
};
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// per-wave partial dgamma/dbeta are summed through LDS and written to ws[block][2*C]
template <int NV, int R, bool DYF32, bool XF32>
__global__ __launch_bounds__(512) void ln_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, void* __restrict__ dx,
                                                     const void* __restrict__ addend, float* __restrict__ ws, int64_t rows,
                                                     int C, const float* __restrict__ gamma1, Copy16 dx16) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [LN_BWD_WAVES][2*C]
  // blockIdx.y = segment: a second tensor of the same shape stacked behind the first (rows each), with its own gamma and
  // its own partial rows (the k and v LayerNorms of one attention in one launch)
  const int64_t ro = (int64_t)blockIdx.y * rows;
  if (blockIdx.y == 1) gamma = gamma1;
  ws += (int64_t)blockIdx.y * gridDim.x * 2 * C;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * LN_BWD_WAVES + w;
  const int64_t nwaves = (int64_t)gridDim.x * LN_BWD_WAVES;
  float dg[NV], db[NV], gm[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    dg[j] = 0.f; db[j] = 0.f;
    const int c = lane + 64 * j;
    gm[j] = c < C ? gamma[c] : 0.f;
  }
  const float invC = 1.f / C;
  for (int64_t row0 = wave * R; row0 < rows; row0 += nwaves * R) {
    float d[R][NV], xv[R][NV], ad[R][NV], mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = ro + min(row0 + r, rows - 1);
      mu[r] = mean[row];
      rs[r] = rstd[row];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int64_t i = row * C + min(lane + 64 * j, C - 1);
        d[r][j] = ldt<DYF32>(dy, i);
        if (dx16.dy2) d[r][j] += ldt<DYF32>(dx16.dy2, i);    // wave-uniform test
        xv[r][j] = ldt<XF32>(x, i);
        ad[r][j] = addend ? ldt<XF32>(addend, i) : 0.f;     // residual-branch gradient folded into dx (wave-uniform test)
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool live = row0 + r < rows;
      float s1 = 0.f, s2 = 0.f, xh[NV], g[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const bool ok = live && (lane + 64 * j < C);
        const float dd = ok ? d[r][j] : 0.f;
        xh[j] = ok ? (xv[r][j] - mu[r]) * rs[r] : 0.f;
        g[j] = dd * gm[j];
        dg[j] += dd * xh[j];
        db[j] += dd;
        s1 += g[j];
        s2 += g[j] * xh[j];
      }
      s1 = wave_sum(s1) * invC;
      s2 = wave_sum(s2) * invC;
      if (live) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const int c = lane + 64 * j;
          if (c < C) {
            const float val = rs[r] * (g[j] - s1 - xh[j] * s2) + ad[r][j];
            stt<XF32>(dx, (ro + row0 + r) * C + c, val);
            // the copy the next GEMMs read (they round to bf16 anyway), times the per-sample drop-path scale of the
            // branch that consumes it when the caller knows it (saves that branch a scale_rows pass over dx)
            if (dx16.p) dx16.p[(ro + row0 + r) * C + c] = (bf16)(dx16.scale ? val * dx16.scale[(row0 + r) / dx16.rps] : val);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = lane + 64 * j;
    if (c < C) { red[w * 2 * C + c] = dg[j]; red[w * 2 * C + C + c] = db[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < LN_BWD_WAVES; ++k) t += red[k * 2 * C + i];
    ws[(int64_t)blockIdx.x * 2 * C + i] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Vectorised forms (C % 8 == 0, the only case the model produces: C in {96, 192, 384, 768}): a group of GL lanes owns a
// row, every lane NCH chunks of 8 consecutive channels -> 16-byte loads / stores instead of one element per lane per
// instruction, statistics by shuffles inside the group, 64 / GL rows per wave and R rows in flight per group.
// ---------------------------------------------------------------------------------------------------------------
template <bool F32> __device__ __forceinline__ void ld8t(const void* p, int64_t i, float (&o)[8]) {
  if constexpr (F32) {
    const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
    const float4 a = q[0], b = q[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else {
    const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p) + i);
    o[0] = h16_lo(r.x); o[1] = h16_hi(r.x);
    o[2] = h16_lo(r.y); o[3] = h16_hi(r.y);
    o[4] = h16_lo(r.z); o[5] = h16_hi(r.z);
    o[6] = h16_lo(r.w); o[7] = h16_hi(r.w);
  }
}
template <bool F32> __device__ __forceinline__ void st8t(void* p, int64_t i, const float (&o)[8]) {
  if constexpr (F32) {
    float4* q = reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i);
    q[0] = make_float4(o[0], o[1], o[2], o[3]);
    q[1] = make_float4(o[4], o[5], o[6], o[7]);
  } else {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)o[j];
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p) + i) = v;
  }
}
template <int GL> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = GL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Workgroups are dealt round-robin over the 8 XCDs; like the GEMMs (gemm.hip), give XCD x the CONTIGUOUS row range x of 8: a
// consumer then finds what the producer's workgroups of the same XCD have just written in that XCD's L2 instead of the memory-side
// cache (tools/launch_floor.hip, profiles/r4_stream_policy.txt part 5: 6 MB handed over in 1.65 us instead of 4.8 us; it holds up to
// ~1 MB per XCD).  Bijective for any grid size.  -DCSTS_LN_NO_XCD_MAP: the plain order (A/B builds).
__device__ __forceinline__ int64_t xcd_block(int64_t bid, int64_t nwg) {
#ifdef CSTS_LN_NO_XCD_MAP
  return bid;
#else
  const int64_t q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
#endif
}

template <int GL, int NCH, int R, bool XF32, bool YF32>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, void* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                         int C, float eps, FwdAdd fa) {
  const int sub = threadIdx.x % GL;
  const int64_t grp = (xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) / GL, ngrp = (int64_t)gridDim.x * 256 / GL;
  int c0[NCH];
  bool act[NCH];
  float gm[NCH][8], bt[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (sub + GL * i) * 8;
    act[i] = c < C;
    c0[i] = min(c, C - 8);
    ld8t<true>(gamma, c0[i], gm[i]);
    ld8t<true>(beta, c0[i], bt[i]);
  }
  const float invC = 1.f / C;
  for (int64_t row0 = grp * R; row0 < rows; row0 += ngrp * R) {
    float v[R][NCH][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = min(row0 + r, rows - 1);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        ld8t<XF32>(x, row * C + c0[i], v[r][i]);
        if (fa.addend) {
          float a2[8];
          ld8t<XF32>(fa.addend, row * C + c0[i], a2);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[r][i][j] += a2[j];
          if (act[i] && row0 + r < rows) st8t<XF32>(fa.sum, row * C + c0[i], v[r][i]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (act[i]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) s += v[r][i][j];
        }
      const float mu = group_sum<GL>(s) * invC;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (act[i]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float d = v[r][i][j] - mu; q += d * d; }
        }
      const float rs = rsqrtf(group_sum<GL>(q) * invC + eps);
      const int64_t row = row0 + r;
      if (row < rows) {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          if (act[i]) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[r][i][j] - mu) * rs * gm[i][j] + bt[i][j];
            st8t<YF32>(y, row * C + c0[i], o);
          }
        if (sub == 0) {
          if (mean) mean[row] = mu;
          if (rstd) rstd[row] = rs;
        }
      }
    }
  }
}

#ifndef CSTS_LN_BWD_R
#define CSTS_LN_BWD_R 1            // rows per lane group and pass of ln_bwd_vec_kernel at C <= 512.  Round 5: 1 instead of 2 (-0.1 ms per step, profiles/r5_stencil_ab.txt)
#endif
#ifndef CSTS_LN_BWD_MINW
#define CSTS_LN_BWD_MINW 1         // minimum waves per SIMD ln_bwd_vec_kernel is compiled for (A/B builds: 6 with R = 1)
#endif
template <int GL, int NCH, int R, bool DYF32, bool XF32>
__global__ __launch_bounds__(512, CSTS_LN_BWD_MINW) void ln_bwd_vec_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, void* __restrict__ dx,
                                                         const void* __restrict__ addend, float* __restrict__ ws, int64_t rows,
                                                         int C, const float* __restrict__ gamma1, Copy16 dx16) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [512 / GL groups][2*C]
  const int64_t ro = (int64_t)blockIdx.y * rows;               // segment (second stacked tensor), see ln_bwd_kernel
  if (blockIdx.y == 1) gamma = gamma1;
  ws += (int64_t)blockIdx.y * gridDim.x * 2 * C;
  const int sub = threadIdx.x % GL, gib = threadIdx.x / GL;    // group in block
  constexpr int GPB = 512 / GL;
  const int64_t grp = xcd_block(blockIdx.x, gridDim.x) * GPB + gib, ngrp = (int64_t)gridDim.x * GPB;
  int c0[NCH];
  bool act[NCH];
  float gm[NCH][8], dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (sub + GL * i) * 8;
    act[i] = c < C;
    c0[i] = min(c, C - 8);
    ld8t<true>(gamma, c0[i], gm[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }
  }
  const float invC = 1.f / C;
  for (int64_t row0 = grp * R; row0 < rows; row0 += ngrp * R) {
    float d[R][NCH][8], xv[R][NCH][8], ad[R][NCH][8], mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = ro + min(row0 + r, rows - 1);
      mu[r] = mean[row];
      rs[r] = rstd[row];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        ld8t<DYF32>(dy, row * C + c0[i], d[r][i]);
        if (dx16.dy2) {
          float d2[8];
          ld8t<DYF32>(dx16.dy2, row * C + c0[i], d2);
#pragma unroll
          for (int j = 0; j < 8; ++j) d[r][i][j] += d2[j];
        }
        ld8t<XF32>(x, row * C + c0[i], xv[r][i]);
        if (addend) ld8t<XF32>(addend, row * C + c0[i], ad[r][i]);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) ad[r][i][j] = 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool live = row0 + r < rows;
      float s1 = 0.f, s2 = 0.f;
      float xh[NCH][8], g[NCH][8];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const bool ok = live && act[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dd = ok ? d[r][i][j] : 0.f;
          xh[i][j] = ok ? (xv[r][i][j] - mu[r]) * rs[r] : 0.f;
          g[i][j] = dd * gm[i][j];
          dg[i][j] += dd * xh[i][j];
          db[i][j] += dd;
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
        }
      }
      s1 = group_sum<GL>(s1) * invC;
      s2 = group_sum<GL>(s2) * invC;
      if (live) {
        const int64_t row = ro + row0 + r;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          if (act[i]) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rs[r] * (g[i][j] - s1 - xh[i][j] * s2) + ad[r][i][j];
            st8t<XF32>(dx, row * C + c0[i], o);
            if (dx16.p) {                                        // the copy the next GEMMs read (see ln_bwd_kernel)
              if (dx16.scale) {
                const float sc = dx16.scale[(row0 + r) / dx16.rps];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] *= sc;
              }
              st8t<false>(dx16.p, row * C + c0[i], o);
            }
          }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (act[i]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[gib * 2 * C + c0[i] + j] = dg[i][j];
        red[gib * 2 * C + C + c0[i] + j] = db[i][j];
      }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float t = 0.f;
#pragma unroll 4
    for (int k = 0; k < GPB; ++k) t += red[k * 2 * C + i];
    ws[(int64_t)blockIdx.x * 2 * C + i] = t;
  }
}

// out[j] = sum_i ws[i][j]   (deterministic second stage of every cross-block reduction in the library)
// block = 32 columns x RL row-lanes; fixed summation order -> bitwise reproducible
template <int RL>
__global__ void reduce_rows_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t nrows, int64_t ncols,
                                   float scale) {
  __shared__ float red[RL][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t j = (int64_t)blockIdx.x * 32 + tx;
  float s = 0.f;
  if (j < ncols) {
    int64_t i = ty;
    for (; i + 3 * RL < nrows; i += 4 * RL) {
      const float a = ws[i * ncols + j], b = ws[(i + RL) * ncols + j], c = ws[(i + 2 * RL) * ncols + j],
                  d = ws[(i + 3 * RL) * ncols + j];
      s += a; s += b; s += c; s += d;
    }
    for (; i < nrows; i += RL) s += ws[i * ncols + j];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < ncols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < RL; ++r) t += red[r][tx];
    out[j] = t * scale;
  }
}

// the same second stage for MANY independent reductions in one launch: blockIdx.y picks the descriptor
__global__ __launch_bounds__(1024) void reduce_rows_batched_kernel(const csts_reduce_desc* __restrict__ descs) {
  __shared__ float red[32][33];
  const csts_reduce_desc d = descs[blockIdx.y];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t j = (int64_t)blockIdx.x * 32 + tx;
  if ((int64_t)blockIdx.x * 32 >= d.ncols) return;     // block-uniform
  float s = 0.f;
  if (j < d.ncols) {
    int64_t i = ty;
    for (; i + 3 * 32 < d.nrows; i += 4 * 32) {
      const float a = d.ws[i * d.ncols + j], b = d.ws[(i + 32) * d.ncols + j], c = d.ws[(i + 64) * d.ncols + j],
                  e = d.ws[(i + 96) * d.ncols + j];
      s += a; s += b; s += c; s += e;
    }
    for (; i < d.nrows; i += 32) s += d.ws[i * d.ncols + j];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < d.ncols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[r][tx];
    d.out[j] = t * d.scale;
  }
}

// wide, shallow reductions (split-K partial slabs of the grouped weight gradients: a handful of rows, up to millions of
// columns): one thread per 4 consecutive columns, rows summed in order -> 16-byte loads, no idle row lanes
__global__ __launch_bounds__(256) void reduce_rows_wide_kernel(const csts_reduce_desc* __restrict__ descs) {
  const csts_reduce_desc d = descs[blockIdx.y];
  const int64_t nvec = d.ncols >> 2;           // host guarantees ncols % 4 == 0 and 16-byte aligned pointers
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = reinterpret_cast<const float4*>(d.ws) + v;
    int64_t i = 0;
    for (; i + 3 < d.nrows; i += 4) {
      const float4 a = p[i * nvec], b = p[(i + 1) * nvec], c = p[(i + 2) * nvec], e = p[(i + 3) * nvec];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
      s.x += e.x; s.y += e.y; s.z += e.z; s.w += e.w;
    }
    for (; i < d.nrows; ++i) {
      const float4 a = p[i * nvec];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    reinterpret_cast<float4*>(d.out)[v] = make_float4(s.x * d.scale, s.y * d.scale, s.z * d.scale, s.w * d.scale);
  }
}

}  // namespace

extern "C" int csts_reduce_rows_wide(const csts_reduce_desc* device_descs, int n, int64_t max_ncols, hipStream_t stream) {
  CSTS_REQUIRE(device_descs != nullptr && n > 0 && n < 65536 && max_ncols > 0, "bad args");
  const unsigned gx = (unsigned)std::min<int64_t>(cdiv(max_ncols / 4, 256), 4096);
  hipLaunchKernelGGL(reduce_rows_wide_kernel, dim3(gx, (unsigned)n), dim3(256), 0, stream, device_descs);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_reduce_rows_batched(const csts_reduce_desc* device_descs, int n, int64_t max_ncols, hipStream_t stream) {
  CSTS_REQUIRE(device_descs != nullptr && n > 0 && n < 65536 && max_ncols > 0, "bad args");
  hipLaunchKernelGGL(reduce_rows_batched_kernel, dim3((unsigned)cdiv(max_ncols, 32), (unsigned)n), dim3(1024), 0, stream,
                     device_descs);
  CSTS_LAUNCH_CHECK();
  return 0;
}

int csts_reduce_rows_launch(const float* ws, float* out, int64_t nrows, int64_t ncols, float scale, hipStream_t s) {
  const dim3 grid((unsigned)cdiv(ncols, 32));
  if (nrows > 64) hipLaunchKernelGGL(reduce_rows_kernel<32>, grid, dim3(1024), 0, s, ws, out, nrows, ncols, scale);
  else if (nrows > 8) hipLaunchKernelGGL(reduce_rows_kernel<8>, grid, dim3(256), 0, s, ws, out, nrows, ncols, scale);
  else hipLaunchKernelGGL(reduce_rows_kernel<2>, grid, dim3(64), 0, s, ws, out, nrows, ncols, scale);
  return 0;
}

extern "C" int csts_reduce_rows(const float* ws, float* out, int64_t nrows, int64_t ncols, float scale,
                                hipStream_t stream) {
  CSTS_REQUIRE(ws && out && nrows > 0 && ncols > 0, "bad args");
  csts_reduce_rows_launch(ws, out, nrows, ncols, scale, stream);
  CSTS_LAUNCH_CHECK();
  return 0;
}

template <int NV, int R>
static void ln_fwd_launch(bool xf, bool yf, dim3 grid, hipStream_t st, const void* x, const float* g, const float* b, void* y,
                          float* mean, float* rstd, int64_t rows, int C, float eps, FwdAdd fa) {
  if (xf && yf) hipLaunchKernelGGL((ln_fwd_kernel<NV, R, true, true>), grid, dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps, fa);
  else if (xf) hipLaunchKernelGGL((ln_fwd_kernel<NV, R, true, false>), grid, dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps, fa);
  else if (yf) hipLaunchKernelGGL((ln_fwd_kernel<NV, R, false, true>), grid, dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps, fa);
  else hipLaunchKernelGGL((ln_fwd_kernel<NV, R, false, false>), grid, dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps, fa);
}
template <int NV, int R>
static void ln_bwd_launch(bool df, bool xf, dim3 grid, size_t sh, hipStream_t st, const void* dy, const void* x, const float* g,
                          const float* mean, const float* rstd, void* dx, const void* addend, float* ws, int64_t rows, int C,
                          const float* g1 = nullptr, Copy16 dx16 = Copy16{}) {
  if (df && xf) hipLaunchKernelGGL((ln_bwd_kernel<NV, R, true, true>), grid, dim3(64 * LN_BWD_WAVES), sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  else if (df) hipLaunchKernelGGL((ln_bwd_kernel<NV, R, true, false>), grid, dim3(64 * LN_BWD_WAVES), sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  else if (xf) hipLaunchKernelGGL((ln_bwd_kernel<NV, R, false, true>), grid, dim3(64 * LN_BWD_WAVES), sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  else hipLaunchKernelGGL((ln_bwd_kernel<NV, R, false, false>), grid, dim3(64 * LN_BWD_WAVES), sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
}
template <int GL, int NCH, int R>
static void ln_fwd_vec_launch(bool xf, bool yf, dim3 grid, hipStream_t st, const void* x, const float* g, const float* b, void* y,
                              float* mean, float* rstd, int64_t rows, int C, float eps, FwdAdd fa) {
  if (xf && yf) hipLaunchKernelGGL((ln_fwd_vec_kernel<GL, NCH, R, true, true>), grid, dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps, fa);
  else if (xf) hipLaunchKernelGGL((ln_fwd_vec_kernel<GL, NCH, R, true, false>), grid, dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps, fa);
  else if (yf) hipLaunchKernelGGL((ln_fwd_vec_kernel<GL, NCH, R, false, true>), grid, dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps, fa);
  else hipLaunchKernelGGL((ln_fwd_vec_kernel<GL, NCH, R, false, false>), grid, dim3(256), 0, st, x, g, b, y, mean, rstd, rows, C, eps, fa);
}
template <int GL, int NCH, int R>
static void ln_bwd_vec_launch(bool df, bool xf, dim3 grid, size_t sh, hipStream_t st, const void* dy, const void* x, const float* g,
                              const float* mean, const float* rstd, void* dx, const void* addend, float* ws, int64_t rows, int C,
                              const float* g1, Copy16 dx16) {
  if (df && xf) hipLaunchKernelGGL((ln_bwd_vec_kernel<GL, NCH, R, true, true>), grid, dim3(512), sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  else if (df) hipLaunchKernelGGL((ln_bwd_vec_kernel<GL, NCH, R, true, false>), grid, dim3(512), sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  else if (xf) hipLaunchKernelGGL((ln_bwd_vec_kernel<GL, NCH, R, false, true>), grid, dim3(512), sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  else hipLaunchKernelGGL((ln_bwd_vec_kernel<GL, NCH, R, false, false>), grid, dim3(512), sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
}
// vector kernels: C % 8 == 0 and every pointer 16-byte aligned
static bool ln_vec_ok(int C, std::initializer_list<const void*> ptrs) {
  if (C % 8 != 0 || C < 8 || C > 1024) return false;
  for (const void* q : ptrs)
    if (q != nullptr && !aligned16(q)) return false;
  return true;
}
static int ln_group_lanes(int C) { return C <= 128 ? 16 : (C <= 256 ? 32 : 64); }
#ifndef CSTS_LN_FWD_R
#define CSTS_LN_FWD_R 2            // rows per lane group and pass of ln_fwd_vec_kernel at C <= 512 (A/B build: 1)
#endif
static int64_t ln_fwd_vec_blocks(int64_t rows, int C) { return std::min<int64_t>(cdiv(rows * ln_group_lanes(C), 256 * (C <= 512 ? CSTS_LN_FWD_R : 2)), 4096); }
static bool ln_bwd_vec(bool df, bool xf, dim3 grid, hipStream_t st, const void* dy, const void* x, const float* g, const float* mean,
                       const float* rstd, void* dx, const void* addend, float* ws, int64_t rows, int C, const float* g1, Copy16 dx16) {
  const size_t sh = (size_t)(512 / ln_group_lanes(C)) * 2 * C * sizeof(float);
  if (sh > 65536) return false;
  if (C <= 128) ln_bwd_vec_launch<16, 1, CSTS_LN_BWD_R>(df, xf, grid, sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  else if (C <= 256) ln_bwd_vec_launch<32, 1, CSTS_LN_BWD_R>(df, xf, grid, sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  else if (C <= 512) ln_bwd_vec_launch<64, 1, CSTS_LN_BWD_R>(df, xf, grid, sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  else ln_bwd_vec_launch<64, 2, 1>(df, xf, grid, sh, st, dy, x, g, mean, rstd, dx, addend, ws, rows, C, g1, dx16);
  return true;
}
static int rows_per_wave(int C) { return C <= 192 ? 4 : (C <= 384 ? 2 : 1); }
static int64_t ln_fwd_blocks(int64_t rows, int C) { return std::min<int64_t>(cdiv(rows, 4 * rows_per_wave(C)), 4096); }
#ifndef CSTS_LN_BWD_CAP
#define CSTS_LN_BWD_CAP 512        // workgroups of a LayerNorm-backward launch (two 512-thread workgroups per CU; A/B builds: 768 with CSTS_LN_BWD_MINW=6)
#endif
static int64_t ln_bwd_blocks(int64_t rows, int C) { return std::min<int64_t>(cdiv(rows, LN_BWD_WAVES * rows_per_wave(C)), CSTS_LN_BWD_CAP); }

extern "C" int csts_layernorm_fwd(const void* x, int x_dt, const float* gamma, const float* beta, void* y, int y_dt,
                                  float* mean, float* rstd, int64_t rows, int C, float eps, hipStream_t stream) {
  return csts_layernorm_fwd_add(x, nullptr, nullptr, x_dt, gamma, beta, y, y_dt, mean, rstd, rows, C, eps, stream);
}

extern "C" int csts_layernorm_fwd_add(const void* x, const void* addend, void* sum_out, int x_dt, const float* gamma,
                                      const float* beta, void* y, int y_dt, float* mean, float* rstd, int64_t rows, int C,
                                      float eps, hipStream_t stream) {
  CSTS_REQUIRE(x && gamma && beta && y, "null pointer");
  CSTS_REQUIRE((addend == nullptr) == (sum_out == nullptr), "addend and sum_out: both or neither");
  CSTS_REQUIRE(rows > 0 && C > 0 && C <= 64 * MAXV, "C must be in (0, 768]");
  const bool xf = x_dt == CSTS_F32, yf = y_dt == CSTS_F32;
  FwdAdd fa;
  fa.addend = addend; fa.sum = sum_out;
  if (ln_vec_ok(C, {x, y, gamma, beta, addend, sum_out})) {
    const dim3 vgrid((unsigned)ln_fwd_vec_blocks(rows, C));
    if (C <= 128) ln_fwd_vec_launch<16, 1, CSTS_LN_FWD_R>(xf, yf, vgrid, stream, x, gamma, beta, y, mean, rstd, rows, C, eps, fa);
    else if (C <= 256) ln_fwd_vec_launch<32, 1, CSTS_LN_FWD_R>(xf, yf, vgrid, stream, x, gamma, beta, y, mean, rstd, rows, C, eps, fa);
    else if (C <= 512) ln_fwd_vec_launch<64, 1, CSTS_LN_FWD_R>(xf, yf, vgrid, stream, x, gamma, beta, y, mean, rstd, rows, C, eps, fa);
    else ln_fwd_vec_launch<64, 2, 2>(xf, yf, vgrid, stream, x, gamma, beta, y, mean, rstd, rows, C, eps, fa);
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  const dim3 grid((unsigned)ln_fwd_blocks(rows, C));
  if (C <= 128) ln_fwd_launch<2, 4>(xf, yf, grid, stream, x, gamma, beta, y, mean, rstd, rows, C, eps, fa);
  else if (C <= 192) ln_fwd_launch<3, 4>(xf, yf, grid, stream, x, gamma, beta, y, mean, rstd, rows, C, eps, fa);
  else if (C <= 384) ln_fwd_launch<6, 2>(xf, yf, grid, stream, x, gamma, beta, y, mean, rstd, rows, C, eps, fa);
  else ln_fwd_launch<12, 1>(xf, yf, grid, stream, x, gamma, beta, y, mean, rstd, rows, C, eps, fa);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t csts_layernorm_bwd_workspace(int64_t rows, int C) {
  return (size_t)ln_bwd_blocks(rows, C) * 2 * C * sizeof(float);
}

extern "C" int csts_layernorm_bwd(const void* dy, int dy_dt, const void* x, int x_dt, const float* gamma,
                                  const float* mean, const float* rstd, void* dx, int dx_dt, const void* addend,
                                  void* dx_bf16, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes,
                                  int64_t rows, int C, hipStream_t stream) {
  return csts_layernorm_bwd_ex(dy, nullptr, dy_dt, x, x_dt, gamma, mean, rstd, dx, dx_dt, addend, dx_bf16, nullptr, 1, dgamma,
                               dbeta, workspace, ws_bytes, rows, C, stream);
}

extern "C" int csts_layernorm_bwd_ex(const void* dy, const void* dy2, int dy_dt, const void* x, int x_dt, const float* gamma,
                                     const float* mean, const float* rstd, void* dx, int dx_dt, const void* addend,
                                     void* dx_bf16, const float* copy_row_scale, int64_t rows_per_scale,
                                     float* dgamma, float* dbeta, void* workspace, size_t ws_bytes, int64_t rows,
                                     int C, hipStream_t stream) {
  CSTS_REQUIRE(dy && x && gamma && mean && rstd && dx && workspace, "null pointer");
  CSTS_REQUIRE(copy_row_scale == nullptr || (dx_bf16 != nullptr && rows_per_scale > 0), "copy scale needs the bf16 copy and rows_per_scale > 0");
  Copy16 c16;
  c16.p = reinterpret_cast<bf16*>(dx_bf16); c16.scale = copy_row_scale; c16.rps = rows_per_scale > 0 ? rows_per_scale : 1;
  c16.dy2 = dy2;
  CSTS_REQUIRE(rows > 0 && C > 0 && C <= 64 * MAXV, "C must be in (0, 768]");
  CSTS_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "dgamma and dbeta: both or neither (neither = deferred second stage)");
  CSTS_REQUIRE(dgamma == nullptr || dbeta == dgamma + C, "dgamma/dbeta must be one contiguous [2*C] buffer");
  CSTS_REQUIRE(dx_dt == x_dt, "dx must have the dtype of x");
  const int64_t nb = ln_bwd_blocks(rows, C);
  CSTS_REQUIRE(ws_bytes >= (size_t)nb * 2 * C * sizeof(float), "workspace too small");
  const dim3 grid((unsigned)nb);
  const size_t sh = (size_t)LN_BWD_WAVES * 2 * C * sizeof(float);
  float* ws = reinterpret_cast<float*>(workspace);
  const bool df = dy_dt == CSTS_F32, xf = x_dt == CSTS_F32;
  if (ln_vec_ok(C, {dy, dy2, x, dx, addend, dx_bf16, gamma}) &&
      ln_bwd_vec(df, xf, grid, stream, dy, x, gamma, mean, rstd, dx, addend, ws, rows, C, nullptr, c16)) {
    CSTS_LAUNCH_CHECK();
    if (dgamma != nullptr) {
      csts_reduce_rows_launch(ws, dgamma, nb, 2 * C, 1.f, stream);
      CSTS_LAUNCH_CHECK();
    }
    return 0;
  }
  if (C <= 128) ln_bwd_launch<2, 4>(df, xf, grid, sh, stream, dy, x, gamma, mean, rstd, dx, addend, ws, rows, C, nullptr, c16);
  else if (C <= 192) ln_bwd_launch<3, 4>(df, xf, grid, sh, stream, dy, x, gamma, mean, rstd, dx, addend, ws, rows, C, nullptr, c16);
  else if (C <= 384) ln_bwd_launch<6, 2>(df, xf, grid, sh, stream, dy, x, gamma, mean, rstd, dx, addend, ws, rows, C, nullptr, c16);
  else ln_bwd_launch<12, 1>(df, xf, grid, sh, stream, dy, x, gamma, mean, rstd, dx, addend, ws, rows, C, nullptr, c16);
  CSTS_LAUNCH_CHECK();
  if (dgamma != nullptr) {
    csts_reduce_rows_launch(ws, dgamma, nb, 2 * C, 1.f, stream);
    CSTS_LAUNCH_CHECK();
  }
  return 0;
}

/* two stacked tensors (rows each) with separate gammas, one launch; ws = 2 x csts_layernorm_bwd_workspace(rows, C) */
extern "C" int csts_layernorm_bwd2(const void* dy, int dy_dt, const void* x, int x_dt, const float* gamma0,
                                   const float* gamma1, const float* mean, const float* rstd, void* dx, int dx_dt,
                                   float* dgb0, float* dgb1, void* workspace, size_t ws_bytes, int64_t rows, int C,
                                   hipStream_t stream) {
  CSTS_REQUIRE(dy && x && gamma0 && gamma1 && mean && rstd && dx && workspace, "null pointer");
  CSTS_REQUIRE(rows > 0 && C > 0 && C <= 64 * MAXV, "C must be in (0, 768]");
  CSTS_REQUIRE(dx_dt == x_dt, "dx must have the dtype of x");
  CSTS_REQUIRE((dgb0 == nullptr) == (dgb1 == nullptr), "dgb0/dgb1: both or neither (neither = deferred second stage)");
  const int64_t nb = ln_bwd_blocks(rows, C);
  CSTS_REQUIRE(ws_bytes >= (size_t)2 * nb * 2 * C * sizeof(float), "workspace too small");
  const dim3 grid((unsigned)nb, 2);
  const size_t sh = (size_t)LN_BWD_WAVES * 2 * C * sizeof(float);
  float* ws = reinterpret_cast<float*>(workspace);
  const bool df = dy_dt == CSTS_F32, xf = x_dt == CSTS_F32;
  if (ln_vec_ok(C, {dy, x, dx, gamma0, gamma1}) && ((rows * C) % 8 == 0) &&
      ln_bwd_vec(df, xf, grid, stream, dy, x, gamma0, mean, rstd, dx, nullptr, ws, rows, C, gamma1, Copy16{})) {
    CSTS_LAUNCH_CHECK();
    if (dgb0 != nullptr) {
      csts_reduce_rows_launch(ws, dgb0, nb, 2 * C, 1.f, stream);
      csts_reduce_rows_launch(ws + nb * 2 * C, dgb1, nb, 2 * C, 1.f, stream);
      CSTS_LAUNCH_CHECK();
    }
    return 0;
  }
  if (C <= 128) ln_bwd_launch<2, 4>(df, xf, grid, sh, stream, dy, x, gamma0, mean, rstd, dx, nullptr, ws, rows, C, gamma1);
  else if (C <= 192) ln_bwd_launch<3, 4>(df, xf, grid, sh, stream, dy, x, gamma0, mean, rstd, dx, nullptr, ws, rows, C, gamma1);
  else if (C <= 384) ln_bwd_launch<6, 2>(df, xf, grid, sh, stream, dy, x, gamma0, mean, rstd, dx, nullptr, ws, rows, C, gamma1);
  else ln_bwd_launch<12, 1>(df, xf, grid, sh, stream, dy, x, gamma0, mean, rstd, dx, nullptr, ws, rows, C, gamma1);
  CSTS_LAUNCH_CHECK();
  if (dgb0 != nullptr) {
    csts_reduce_rows_launch(ws, dgb0, nb, 2 * C, 1.f, stream);
    csts_reduce_rows_launch(ws + nb * 2 * C, dgb1, nb, 2 * C, 1.f, stream);
    CSTS_LAUNCH_CHECK();
  }
  return 0;
}
