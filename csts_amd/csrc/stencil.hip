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
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

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

// 8 consecutive channels (16-byte bf16 / 2 x 16-byte f32 accesses); offsets are 32-bit inside one batch element
__device__ __forceinline__ void ld8v(const void* p, int dt, int i, float (&o)[8]) {
  if (dt == CSTS_F32) {
    const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
    const float4 a = q[0], b = q[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p) + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
  }
}
__device__ __forceinline__ const void* batch_ptr(const void* p, int dt, int64_t off) {
  return dt == CSTS_F32 ? (const void*)(reinterpret_cast<const float*>(p) + off) : (const void*)(reinterpret_cast<const bf16*>(p) + off);
}
__device__ __forceinline__ void* batch_ptr(void* p, int dt, int64_t off) {
  return dt == CSTS_F32 ? (void*)(reinterpret_cast<float*>(p) + off) : (void*)(reinterpret_cast<bf16*>(p) + off);
}

// weights staged as wl[tap][HD] (fp32) so that lanes read consecutive channels
__device__ __forceinline__ void stage_weights(const float* __restrict__ w, float* wl, int HD) {
  const int nthr = blockDim.x * blockDim.y, tid = threadIdx.y * blockDim.x + threadIdx.x;
  for (int i = tid; i < HD * 27; i += nthr) {   // consecutive LDS addresses (conflict-free), strided L2-cached reads
    const int k = i / HD, c = i - k * HD;
    wl[i] = w[c * 27 + k];
  }
  __syncthreads();
}

// The depthwise kernels keep their tap loads free of control flow AND of dtype conversion: a batch of raw loads
// (clamped, always in-range addresses) is issued back to back so that the memory latencies overlap, and only then
// converted and accumulated.  Out-of-range taps read a ZERO weight row (strided / transposed) or are masked on the
// raw bits (wgrad).  Measured with rocprofv3 PMC on MI355X: the earlier "load, convert, continue" form spent 79 %
// of its wave cycles in s_waitcnt (one HBM/L2 round trip per tap).  Strides are powers of two (mask / shift).
// n / d for 0 <= n < 2^31 as one 64-bit multiply-add and one shift (m = ceil(2^(31+s) / d), s = ceil(log2 d): exact, see fast_div()):
// a division by a run-time integer is ~25 vector instructions, and every work item of the stencil kernels did five of them
// (token -> (b, t, h, w), channel chunk, head) for ~500 instructions of arithmetic (round 5; profiles/r5_stencil_wgrad_isa.txt).
struct FastDiv { unsigned m; int sh; int d; };
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) { return (int)(((unsigned long long)(unsigned)n * f.m) >> f.sh); }
struct RowGeom {
  Geom g;
  int lt, lh, lw;        // log2 strides
  FastDiv dNc, dHc, dWc; // coarse tokens per batch element, coarse H, W
  FastDiv dNf, dHf, dWf; // fine tokens per batch element, fine H, W
  FastDiv dN2, dH2, dW2; // the 2 x 2-block grid of dwconv_transposed_s22 (Tf * Hf/2 * Wf/2, Hf/2, Wf/2)
  FastDiv dC4, dC8, dHD, dHeads;   // C / 4, C / 8 channel chunks per token, head_dim, heads
};
__device__ __forceinline__ void decompf(int bt, const FastDiv& dN, const FastDiv& dH, const FastDiv& dW, int& b, int& t, int& h, int& w) {
  b = fdiv(bt, dN);
  int o = bt - b * dN.d;
  const int q = fdiv(o, dW);
  w = o - q * dW.d;
  t = fdiv(q, dH);
  h = q - t * dH.d;
}

// 32-bit decomposition of a flat token index (host guarantees B*ntok*C < 2^31)
__device__ __forceinline__ void decomp(int bt, int ntok, int H, int W, int& b, int& t, int& h, int& w) {
  b = bt / ntok;
  int o = bt - b * ntok;
  w = o % W; o /= W;
  h = o % H;
  t = o / H;
}

template <bool F32> struct Raw4;                       // 4 consecutive channels, unconverted
template <> struct Raw4<true> { float4 v; };
template <> struct Raw4<false> { uint2 v; };
template <bool F32> __device__ __forceinline__ Raw4<F32> raw4_load(const void* base, int i) {
  Raw4<F32> r;
  if constexpr (F32) r.v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + i);
  else r.v = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(base) + i);
  return r;
}
template <bool F32> __device__ __forceinline__ void raw4_cvt(const Raw4<F32>& r, float (&o)[4]) {
  if constexpr (F32) { o[0] = r.v.x; o[1] = r.v.y; o[2] = r.v.z; o[3] = r.v.w; }
  else {
    o[0] = h16_lo(r.v.x); o[1] = h16_hi(r.v.x);
    o[2] = h16_lo(r.v.y); o[3] = h16_hi(r.v.y);
  }
}
template <bool F32> __device__ __forceinline__ void st4t(void* base, int i, const float (&o)[4]) {
  if constexpr (F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + i) = make_float4(o[0], o[1], o[2], o[3]);
  else {
    bf16x4 a;
    a[0] = (bf16)o[0]; a[1] = (bf16)o[1]; a[2] = (bf16)o[2]; a[3] = (bf16)o[3];
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(base) + i) = a;
  }
}
template <bool F32> __device__ __forceinline__ const void* bptr(const void* p, int64_t off) {
  if constexpr (F32) return reinterpret_cast<const float*>(p) + off;
  else return reinterpret_cast<const bf16*>(p) + off;
}
template <bool F32> __device__ __forceinline__ void* bptr(void* p, int64_t off) {
  if constexpr (F32) return reinterpret_cast<float*>(p) + off;
  else return reinterpret_cast<bf16*>(p) + off;
}

template <bool F32> struct Raw8;
template <> struct Raw8<true> { float4 a, b; };
template <> struct Raw8<false> { uint4 a; };
template <bool F32> __device__ __forceinline__ Raw8<F32> raw8_load(const void* base, int i) {
  Raw8<F32> r;
  if constexpr (F32) {
    const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + i);
    r.a = q[0]; r.b = q[1];
  } else {
    r.a = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(base) + i);
  }
  return r;
}
template <bool F32> __device__ __forceinline__ void raw8_cvt(const Raw8<F32>& r, float (&o)[8]) {
  if constexpr (F32) {
    o[0] = r.a.x; o[1] = r.a.y; o[2] = r.a.z; o[3] = r.a.w; o[4] = r.b.x; o[5] = r.b.y; o[6] = r.b.z; o[7] = r.b.w;
  } else {
    o[0] = h16_lo(r.a.x); o[1] = h16_hi(r.a.x);
    o[2] = h16_lo(r.a.y); o[3] = h16_hi(r.a.y);
    o[4] = h16_lo(r.a.z); o[5] = h16_hi(r.a.z);
    o[6] = h16_lo(r.a.w); o[7] = h16_hi(r.a.w);
  }
}
template <bool F32> __device__ __forceinline__ void st8t(void* base, int64_t i, const float (&o)[8]) {
  if constexpr (F32) {
    float4* q = reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + i);
    q[0] = make_float4(o[0], o[1], o[2], o[3]);
    q[1] = make_float4(o[4], o[5], o[6], o[7]);
  } else {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)o[j];
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(base) + i) = v;
  }
}

// V consecutive channels per thread: 4 (fp32 tensors: 16-byte accesses) or 8 (bf16 tensors: 16-byte accesses; the 8-byte
// bf16x4 form kept the vector-memory path at about half rate -- 18..27 tap loads per output make these kernels bound by
// the L1/TA path, not by HBM)
template <int V, bool F32> struct RawV { typedef Raw4<F32> T; };
template <bool F32> struct RawV<8, F32> { typedef Raw8<F32> T; };
template <int V, bool F32> __device__ __forceinline__ typename RawV<V, F32>::T rawv_load(const void* base, int i) {
  if constexpr (V == 8) return raw8_load<F32>(base, i);
  else return raw4_load<F32>(base, i);
}
template <int V, bool F32> __device__ __forceinline__ void rawv_cvt(const typename RawV<V, F32>::T& r, float (&o)[V]) {
  if constexpr (V == 8) raw8_cvt<F32>(r, o);
  else raw4_cvt<F32>(r, o);
}
template <int V, bool F32> __device__ __forceinline__ void stvt(void* base, int i, const float (&o)[V]) {
  if constexpr (V == 8) st8t<F32>(base, i, o);
  else st4t<F32>(base, i, o);
}
// acc[j] += v[j] * wl[j]  for V channels (wl: 16-byte aligned row of the staged weights)
template <int V> __device__ __forceinline__ void fmav(float (&acc)[V], const float (&v)[V], const float* wrow) {
#pragma unroll
  for (int q = 0; q < V / 4; ++q) {
    const float4 ww = *reinterpret_cast<const float4*>(wrow + 4 * q);
    acc[4 * q] += v[4 * q] * ww.x; acc[4 * q + 1] += v[4 * q + 1] * ww.y;
    acc[4 * q + 2] += v[4 * q + 2] * ww.z; acc[4 * q + 3] += v[4 * q + 3] * ww.w;
  }
}

// weights staged as wl[28][HD + WPAD] fp32: 27 taps + one all-zero row (index 27) that invalid taps point to.
// WPAD: the source is [channel][tap], so the 27 consecutive threads that hold one channel's taps store to addresses one ROW
// apart -- with a row of HD = 96 or 192 floats (a multiple of the 32 banks) that is a 27-way bank conflict on every staging
// store (5.3 of the 20 us of the 384-channel k|v launches went into staging two tables: tools/pool_ln_bench.py with the
// staging compiled out); four floats of padding spread a channel's taps over 8 banks and keep the rows 16-byte aligned.
constexpr int WPAD = 4;
// Coalesced global reads, 12 per thread IN FLIGHT before the first LDS store (a plain load -> store loop waits for every
// load on its own: 11 round trips in a row for a 96-channel table on 256 threads, and this table is the first thing every
// workgroup of the stencil kernels needs); the (conflicting) transposition is paid in LDS.
// SPLIT8 (pool_ln_fwd_kernel: a lane owns 8 consecutive channels and reads them as two 16-byte pieces): inside a tap row the
// FIRST four channels of every 8-channel group are stored next to each other, then all the second fours -- channel 8 i + j sits
// at 4 i + j (j < 4) or HD / 2 + 4 i + j - 4 -- so that each of the two ds_read_b128 of a tap is one contiguous run over the
// lanes (the straight layout put consecutive lanes 32 bytes apart: two lanes per bank pair on every read, 3.6 conflict cycles
// per LDS instruction by SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS).
template <bool SPLIT8 = false>
__device__ __forceinline__ void stage_weight_rows(const float* __restrict__ w, float* wl, int HD, int nthr, int tid) {
  const int n = HD * 27;
  // (channel, tap) of element i = tid + u * nthr WITHOUT a division per element (the stencil kernels are bound by VALU
  // issue, and this table costs every workgroup ~22 elements per thread): one division for the first element, then
  // i += nthr moves the tap by nthr % 27 and the channel by nthr / 27 (+ 1 on wrap-around)
  const int dq = nthr / 27, dr = nthr - dq * 27;
  int c = tid / 27, k = tid - c * 27;
  for (int base = tid; base < n; base += nthr * 12) {
    float v[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int i = base + u * nthr;
      v[u] = i < n ? w[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if (base + u * nthr < n) wl[k * (HD + WPAD) + (SPLIT8 ? ((c & 4) ? (HD >> 1) + ((c >> 3) << 2) + (c & 3) : ((c >> 3) << 2) + (c & 3)) : c)] = v[u];
      k += dr; c += dq;
      if (k >= 27) { k -= 27; ++c; }
    }
  }
  for (int i = tid; i < HD; i += nthr) wl[27 * (HD + WPAD) + i] = 0.f;
}
__device__ __forceinline__ void stage_weights_z(const float* __restrict__ w, float* wl, int HD) {
  stage_weight_rows(w, wl, HD, blockDim.x * blockDim.y, threadIdx.y * blockDim.x + threadIdx.x);
  __syncthreads();
}

// coarse[b,o,c] = sum_k fine[b, o*s-1+k, c] * w[c%HD][k]          (V channels per thread)
#ifndef CSTS_DW_WAVES
#define CSTS_DW_WAVES 1       // minimum waves per SIMD the strided / transposed stencils are compiled for (A/B builds: 3, 4)
#endif
template <bool FF32, bool CF32, int V>
__global__ __launch_bounds__(256, CSTS_DW_WAVES) void dwconv_strided_kernel(RowGeom rg, const void* __restrict__ fine,
                                                             const float* __restrict__ w, void* __restrict__ coarse) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const Geom& g = rg.g;
  stage_weights_z(w, wl, g.HD);
  const int CQ = g.C / V;
  const int ntok = g.Tc * g.Hc * g.Wc;
  const int total = g.B * ntok * CQ;
  const int fts = (int)g.f_ts, cts = (int)g.c_ts;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int bt = fdiv(idx, V == 8 ? rg.dC8 : rg.dC4);
    const int c = (idx - bt * CQ) * V, cw = c - fdiv(c, rg.dHD) * g.HD;
    int b, ot, oh, ow;
    decompf(bt, rg.dNc, rg.dHc, rg.dWc, b, ot, oh, ow);
    int tof[3], hof[3], xof[3];
    bool tv[3], hv[3], xv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int t = ot * g.st - 1 + k, h = oh * g.sh - 1 + k, x = ow * g.sw - 1 + k;
      tv[k] = (unsigned)t < (unsigned)g.Tf; tof[k] = min(max(t, 0), g.Tf - 1) * g.Hf * g.Wf * fts;
      hv[k] = (unsigned)h < (unsigned)g.Hf; hof[k] = min(max(h, 0), g.Hf - 1) * g.Wf * fts;
      xv[k] = (unsigned)x < (unsigned)g.Wf; xof[k] = min(max(x, 0), g.Wf - 1) * fts + c;
    }
    const void* fb = bptr<FF32>(fine, (int64_t)b * g.f_bs);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      typename RawV<V, FF32>::T raw[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) raw[kh * 3 + kw] = rawv_load<V, FF32>(fb, tof[kt] + hof[kh] + xof[kw]);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = (tv[kt] && hv[kh] && xv[kw]) ? kt * 9 + kh * 3 + kw : 27;
          float v[V];
          rawv_cvt<V, FF32>(raw[kh * 3 + kw], v);
          fmav<V>(acc, v, &wl[tap * (g.HD + WPAD) + cw]);
        }
    }
    stvt<V, CF32>(bptr<CF32>(coarse, (int64_t)b * g.c_bs), (bt - b * ntok) * cts + c, acc);
  }
}

// Fused pooling: depthwise conv (as dwconv_strided) + LayerNorm(hd) of the result, for up to two tensors that share the
// geometry (k and v of one attention: attention.py:104-116 -> pool_k + norm_k, pool_v + norm_v), in ONE launch.
// A group of GL lanes owns one (batch, out-token, head, slot) item, 8 channels per lane (HD/8 lanes active), so the
// LayerNorm statistics are a shuffle reduction inside the group and the pooled row never leaves registers before it is
// normalised.  Writes the pre-LN row (saved for backward), the normalised row, mean and rstd.
#ifdef CSTS_POOLLN_SPLIT8
constexpr bool POOLLN_SPLIT8 = true;
#else
constexpr bool POOLLN_SPLIT8 = false;
#endif
#ifndef POOL_LN_WGS
#define POOL_LN_WGS 3      // workgroups per CU the register budget is set for.  Round 5: 3 (168 registers, ~no spills) instead of 4 (128 registers,
                           // 8-12 spilled): -0.19 ms per step once the index arithmetic had shrunk (profiles/r5_stencil_ab.txt); 2: -0.14 ms
#endif
struct PoolLnSlots {
  const void* fine[2];
  const float* w[2];
  const float* gamma[2];
  const float* beta[2];
  void* conv[2];
  void* y[2];
  float* mean[2];
  float* rstd[2];
};
template <int GL, bool F32>
__global__ __launch_bounds__(256, POOL_LN_WGS) void pool_ln_fwd_kernel(RowGeom rg, PoolLnSlots sl, int nslots, float eps) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [nslots][28][HD]
  const Geom& g = rg.g;
  const int HD = g.HD, H = g.C / HD;
  const int lane_in = threadIdx.x % GL;
  const bool active = lane_in * 8 < HD;
  const int c8 = min(lane_in * 8, HD - 8);                    // idle lanes shadow the last active one (loads stay in range)
  const int groups_per_block = blockDim.x / GL;
  const int ntok = g.Tc * g.Hc * g.Wc;
  const int per_slot = g.B * ntok * H;
  const int total = per_slot * nslots;
  const int fts = (int)g.f_ts, cts = (int)g.c_ts;
  const float invHD = 1.f / HD;
  // -DCSTS_POOLLN_EARLY_LOADS (round 5, measured and NOT kept): the tap loads of the workgroup's first items in flight while the weight
  // tables are staged (the loads do not depend on the weights).  With the staging inside the item loop the kernel spills 30-36 registers
  // instead of 8-12 at its 128-register budget and runs 25-35 % SLOWER per launch, +0.37 ms per step (profiles/r5_stencil_ab.txt).
  bool staged = false;
  auto stage = [&]() {   // both slots' weights (tap-major, one zero row each)
    // POOLLN_SPLIT8 (round 4, -DCSTS_POOLLN_SPLIT8): the conflict-free row layout took SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS from 3.56
    // to 0.53 and made the kernel 8-10 % SLOWER (tools/pool_ln_bench.py: 17.4 -> 19.1 us at the 384-channel stage): the kernel is
    // bound by its dependent global-load rounds, and the permuted staging costs every workgroup more than the conflicts did.  Off.
    for (int s2 = 0; s2 < nslots; ++s2) stage_weight_rows<POOLLN_SPLIT8>(sl.w[s2], wl + s2 * 28 * (HD + WPAD), HD, blockDim.x, threadIdx.x);
    __syncthreads();
    staged = true;
  };
#ifndef CSTS_POOLLN_EARLY_LOADS               // default: tables staged in front of the first tap loads
  stage();
#endif
  const int item0 = blockIdx.x * groups_per_block + threadIdx.x / GL;
  // every thread runs the first pass (an item past the end is clamped and only not stored): the barrier inside stage() is then reached
  // by the whole workgroup at the same place
  for (int item_raw = item0, pass = 0; pass == 0 || item_raw < total; item_raw += gridDim.x * groups_per_block, ++pass) {
    const bool live = item_raw < total;
    const int item = min(item_raw, total - 1);
    const int slot = item >= per_slot ? 1 : 0;            // (nslots <= 2)
    int r = item - slot * per_slot;
    const int bt = fdiv(r, rg.dHeads);
    const int head = r - bt * H;
    int b, ot, oh, ow;
    decompf(bt, rg.dNc, rg.dHc, rg.dWc, b, ot, oh, ow);
    const int c = head * HD + c8;
    int hof[3], xof[3];
    bool hv[3], xv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#ifdef CSTS_POOLLN_SHIFT
      const int h = (oh << rg.lh) - 1 + k, x = (ow << rg.lw) - 1 + k;
#else
      const int h = oh * g.sh - 1 + k, x = ow * g.sw - 1 + k;
#endif
      hv[k] = (unsigned)h < (unsigned)g.Hf; hof[k] = min(max(h, 0), g.Hf - 1) * g.Wf * fts;
      xv[k] = (unsigned)x < (unsigned)g.Wf; xof[k] = min(max(x, 0), g.Wf - 1) * fts + c;
    }
    const void* fb = bptr<F32>(sl.fine[slot], (int64_t)b * g.f_bs);
    const float* wls = wl + slot * 28 * (HD + WPAD) + (POOLLN_SPLIT8 ? (c8 >> 1) : c8);      // SPLIT8 layout: first fours at 4 i, second fours at HD / 2 + 4 i
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // 9 taps (one temporal slice) in flight at a time: a 128-register budget, FOUR workgroups per CU, so the 1024-workgroup
    // launches of the 384-channel stages are resident at once.  (All 27 taps at once -- one memory round trip per item
    // instead of three -- needs 176 registers = two workgroups per CU = two rounds.  Measured alike: 27.9 vs 28.2 us for
    // those launches, 10-30 % faster for the smaller ones; rocprofv3 kernel trace, profiles/r2_pool_ln_ab.txt.)
#pragma unroll 1
    for (int kt = 0; kt < 3; ++kt) {           // a real loop: unrolled, the scheduler hoists all 27 loads again and spills
#ifdef CSTS_POOLLN_SHIFT
      const int t = (ot << rg.lt) - 1 + kt;
#else
      const int t = ot * g.st - 1 + kt;
#endif
      const bool tvk = (unsigned)t < (unsigned)g.Tf;
      const int tofk = min(max(t, 0), g.Tf - 1) * g.Hf * g.Wf * fts;
      Raw8<F32> raw[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) raw[kh * 3 + kw] = raw8_load<F32>(fb, tofk + hof[kh] + xof[kw]);
      if (!staged) stage();                    // first pass, first round: the tables arrive under these loads
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = (tvk && hv[kh] && xv[kw]) ? kt * 9 + kh * 3 + kw : 27;
          const float4 w0 = *reinterpret_cast<const float4*>(&wls[tap * (HD + WPAD)]);
          const float4 w1 = *reinterpret_cast<const float4*>(&wls[tap * (HD + WPAD) + (POOLLN_SPLIT8 ? (HD >> 1) : 4)]);
          float v[8];
          raw8_cvt<F32>(raw[kh * 3 + kw], v);
          acc[0] += v[0] * w0.x; acc[1] += v[1] * w0.y; acc[2] += v[2] * w0.z; acc[3] += v[3] * w0.w;
          acc[4] += v[4] * w1.x; acc[5] += v[5] * w1.y; acc[6] += v[6] * w1.z; acc[7] += v[7] * w1.w;
        }
    }
    // the pre-LN row is stored in the activation dtype and the statistics are taken from the STORED values, exactly
    // like the two-kernel path (conv -> round -> LayerNorm) and the reference (attention.py:37-47)
    if constexpr (!F32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = (float)(bf16)acc[j];
    }
    float s1 = 0.f;
    if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s1 += acc[j];
    }
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    const float mu = s1 * invHD;
    float s2 = 0.f;
    if (active) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = acc[j] - mu; s2 += d * d; }
    }
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rs = rsqrtf(s2 * invHD + eps);
    if (active && live) {
      const int64_t off = (int64_t)b * g.c_bs + (int64_t)(bt - b * ntok) * cts + c;
      st8t<F32>(sl.conv[slot], off, acc);
      const float4 g0 = *reinterpret_cast<const float4*>(sl.gamma[slot] + c8), g1 = *reinterpret_cast<const float4*>(sl.gamma[slot] + c8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(sl.beta[slot] + c8), b1 = *reinterpret_cast<const float4*>(sl.beta[slot] + c8 + 4);
      float y[8];
      y[0] = (acc[0] - mu) * rs * g0.x + b0.x; y[1] = (acc[1] - mu) * rs * g0.y + b0.y;
      y[2] = (acc[2] - mu) * rs * g0.z + b0.z; y[3] = (acc[3] - mu) * rs * g0.w + b0.w;
      y[4] = (acc[4] - mu) * rs * g1.x + b1.x; y[5] = (acc[5] - mu) * rs * g1.y + b1.y;
      y[6] = (acc[6] - mu) * rs * g1.z + b1.z; y[7] = (acc[7] - mu) * rs * g1.w + b1.w;
      st8t<F32>(sl.y[slot], off, y);
      if (lane_in == 0) {
        const int64_t row = (int64_t)bt * H + head;       // row index of the (B*N, H) x HD LayerNorm
        sl.mean[slot][row] = mu;
        sl.rstd[slot][row] = rs;
      }
    }
  }
}

// Round 5: the same kernel with TWELVE channels per lane (head dims 96 / 192: 8 / 16 lanes per item, every lane active -- with 8 channels per lane
// a 96-wide head keeps 12 of its 16 lanes busy -- and 32 / 16 items per workgroup instead of 16 / 8: the 384-channel stage's launches become 512
// workgroups, ONE round at three workgroups per CU instead of 1024 = 1.33 rounds).  Same arithmetic per channel, same order of taps: bit-identical
// conv rows; the LayerNorm sums run over a different lane tree (8 / 16 lanes of 12 instead of 16 / 32 of 8): equal to rounding.
template <bool F32> struct Raw12;
template <> struct Raw12<true> { float4 a, b, c; };
template <> struct Raw12<false> { uint4 a; uint2 b; };
template <bool F32> __device__ __forceinline__ Raw12<F32> raw12_load(const void* base, int i) {
  Raw12<F32> r;
  if constexpr (F32) {
    const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + i);
    r.a = q[0]; r.b = q[1]; r.c = q[2];
  } else {
    const char* q = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(base) + i);      // 24 k bytes: 8-byte aligned
    uint2 lo = *reinterpret_cast<const uint2*>(q), mid = *reinterpret_cast<const uint2*>(q + 8);
    r.a = make_uint4(lo.x, lo.y, mid.x, mid.y);
    r.b = *reinterpret_cast<const uint2*>(q + 16);
  }
  return r;
}
template <bool F32> __device__ __forceinline__ void raw12_cvt(const Raw12<F32>& r, float (&o)[12]) {
  if constexpr (F32) {
    o[0] = r.a.x; o[1] = r.a.y; o[2] = r.a.z; o[3] = r.a.w; o[4] = r.b.x; o[5] = r.b.y; o[6] = r.b.z; o[7] = r.b.w;
    o[8] = r.c.x; o[9] = r.c.y; o[10] = r.c.z; o[11] = r.c.w;
  } else {
    o[0] = h16_lo(r.a.x); o[1] = h16_hi(r.a.x); o[2] = h16_lo(r.a.y); o[3] = h16_hi(r.a.y);
    o[4] = h16_lo(r.a.z); o[5] = h16_hi(r.a.z); o[6] = h16_lo(r.a.w); o[7] = h16_hi(r.a.w);
    o[8] = h16_lo(r.b.x); o[9] = h16_hi(r.b.x); o[10] = h16_lo(r.b.y); o[11] = h16_hi(r.b.y);
  }
}
template <bool F32> __device__ __forceinline__ void st12t(void* base, int64_t i, const float (&o)[12]) {
  if constexpr (F32) {
    float4* q = reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + i);
    q[0] = make_float4(o[0], o[1], o[2], o[3]);
    q[1] = make_float4(o[4], o[5], o[6], o[7]);
    q[2] = make_float4(o[8], o[9], o[10], o[11]);
  } else {
    bf16x4 v[3];
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j >> 2][j & 3] = (bf16)o[j];
    uint2* q = reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(base) + i);                        // 8-byte aligned
    q[0] = __builtin_bit_cast(uint2, v[0]); q[1] = __builtin_bit_cast(uint2, v[1]); q[2] = __builtin_bit_cast(uint2, v[2]);
  }
}

// ALL27: all 27 taps of an item in flight at once (one memory round trip per item instead of three) at TWO workgroups per CU -- every launch of
// this form is at most 512 workgroups, one round either way.
template <int GL, bool F32, bool ALL27>
__global__ __launch_bounds__(256, (ALL27 ? 2 : POOL_LN_WGS)) void pool_ln_fwd12_kernel(RowGeom rg, PoolLnSlots sl, int nslots, float eps) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [nslots][28][HD + WPAD]
  const Geom& g = rg.g;
  const int HD = g.HD, H = g.C / HD;                          // HD == 12 GL (host-checked)
  const int lane_in = threadIdx.x % GL;
  const int c12 = lane_in * 12;
  constexpr int groups_per_block = 256 / GL;
  const int ntok = g.Tc * g.Hc * g.Wc;
  const int per_slot = g.B * ntok * H;
  const int total = per_slot * nslots;
  const int fts = (int)g.f_ts, cts = (int)g.c_ts;
  const float invHD = 1.f / HD;
  for (int s2 = 0; s2 < nslots; ++s2) stage_weight_rows<false>(sl.w[s2], wl + s2 * 28 * (HD + WPAD), HD, blockDim.x, threadIdx.x);
  __syncthreads();
  const int item0 = blockIdx.x * groups_per_block + threadIdx.x / GL;
  for (int item = item0; item < total; item += gridDim.x * groups_per_block) {
    const int slot = item >= per_slot ? 1 : 0;            // (nslots <= 2)
    int r = item - slot * per_slot;
    const int bt = fdiv(r, rg.dHeads);
    const int head = r - bt * H;
    int b, ot, oh, ow;
    decompf(bt, rg.dNc, rg.dHc, rg.dWc, b, ot, oh, ow);
    const int c = head * HD + c12;
    int hof[3], xof[3];
    bool hv[3], xv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int h = oh * g.sh - 1 + k, x = ow * g.sw - 1 + k;
      hv[k] = (unsigned)h < (unsigned)g.Hf; hof[k] = min(max(h, 0), g.Hf - 1) * g.Wf * fts;
      xv[k] = (unsigned)x < (unsigned)g.Wf; xof[k] = min(max(x, 0), g.Wf - 1) * fts + c;
    }
    const void* fb = bptr<F32>(sl.fine[slot], (int64_t)b * g.f_bs);
    const float* wls = wl + slot * 28 * (HD + WPAD) + c12;
    float acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = 0.f;
    auto slice_load = [&](int kt, Raw12<F32> (&raw)[9], bool& tvk) {
      const int t = ot * g.st - 1 + kt;
      tvk = (unsigned)t < (unsigned)g.Tf;
      const int tofk = min(max(t, 0), g.Tf - 1) * g.Hf * g.Wf * fts;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) raw[kh * 3 + kw] = raw12_load<F32>(fb, tofk + hof[kh] + xof[kw]);
    };
    auto slice_mac = [&](int kt, const Raw12<F32> (&raw)[9], bool tvk) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = (tvk && hv[kh] && xv[kw]) ? kt * 9 + kh * 3 + kw : 27;
          const float* wr = &wls[tap * (HD + WPAD)];
          const float4 w0 = *reinterpret_cast<const float4*>(wr), w1 = *reinterpret_cast<const float4*>(wr + 4), w2 = *reinterpret_cast<const float4*>(wr + 8);
          float v[12];
          raw12_cvt<F32>(raw[kh * 3 + kw], v);
          acc[0] += v[0] * w0.x; acc[1] += v[1] * w0.y; acc[2] += v[2] * w0.z; acc[3] += v[3] * w0.w;
          acc[4] += v[4] * w1.x; acc[5] += v[5] * w1.y; acc[6] += v[6] * w1.z; acc[7] += v[7] * w1.w;
          acc[8] += v[8] * w2.x; acc[9] += v[9] * w2.y; acc[10] += v[10] * w2.z; acc[11] += v[11] * w2.w;
        }
    };
    if constexpr (ALL27) {
      Raw12<F32> r0[9], r1[9], r2[9];
      bool t0, t1, t2;
      slice_load(0, r0, t0); slice_load(1, r1, t1); slice_load(2, r2, t2);
      slice_mac(0, r0, t0); slice_mac(1, r1, t1); slice_mac(2, r2, t2);
    } else {
#pragma unroll 1
      for (int kt = 0; kt < 3; ++kt) {           // a real loop: one temporal slice (9 taps) in flight at a time, as in pool_ln_fwd_kernel
        Raw12<F32> raw[9];
        bool tvk;
        slice_load(kt, raw, tvk);
        slice_mac(kt, raw, tvk);
      }
    }
    if constexpr (!F32) {
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[j] = (float)(bf16)acc[j];
    }
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) s1 += acc[j];
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
    const float mu = s1 * invHD;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) { const float d = acc[j] - mu; s2 += d * d; }
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rs = rsqrtf(s2 * invHD + eps);
    const int64_t off = (int64_t)b * g.c_bs + (int64_t)(bt - b * ntok) * cts + c;
    st12t<F32>(sl.conv[slot], off, acc);
    const float* gp = sl.gamma[slot] + c12;
    const float* bp = sl.beta[slot] + c12;
    float y[12];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float4 gq = *reinterpret_cast<const float4*>(gp + 4 * q), bq = *reinterpret_cast<const float4*>(bp + 4 * q);
      y[4 * q + 0] = (acc[4 * q + 0] - mu) * rs * gq.x + bq.x; y[4 * q + 1] = (acc[4 * q + 1] - mu) * rs * gq.y + bq.y;
      y[4 * q + 2] = (acc[4 * q + 2] - mu) * rs * gq.z + bq.z; y[4 * q + 3] = (acc[4 * q + 3] - mu) * rs * gq.w + bq.w;
    }
    st12t<F32>(sl.y[slot], off, y);
    if (lane_in == 0) {
      const int64_t row = (int64_t)bt * H + head;       // row index of the (B*N, H) x HD LayerNorm
      sl.mean[slot][row] = mu;
      sl.rstd[slot][row] = rs;
    }
  }
}

// candidate taps of one axis of the transposed form: k with (f + 1 - k) % s == 0, o = (f + 1 - k) / s.
// stride 1: three candidates, stride 2: two (parity), stride >= 4: at most one.
template <int N>
__device__ __forceinline__ void axis_cand(int f, int s, int ls, int nc, int (&k)[N], int (&o)[N], bool (&ok)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int kk, n;
    if (N == 3) { kk = i; n = f + 1 - i; }
    else if (N == 2) { kk = ((f + 1) & 1) + 2 * i; n = f + 1 - kk; }
    else if (ls >= 0) { kk = (f + 1) & (s - 1); n = f + 1 - kk; }
    else { kk = (f + 1) % s; n = f + 1 - kk; }                       // stride 3: the compact K/V grids (csts_kv_rows_geom)
    const int oo = ls >= 0 ? (n >> ls) : (n >= 0 ? n / s : -1);
    ok[i] = kk <= 2 && n >= 0 && oo < nc;
    k[i] = min(kk, 2);
    o[i] = min(max(oo, 0), nc - 1);
  }
}

// fine[b,f,c] = sum_{k : (f+1-k) % s == 0} coarse[b, (f+1-k)/s, c] * w[c%HD][k]      (4 channels per thread)
// up to two tensors that share one geometry (the k and v pools of one attention): blockIdx.y picks the slot
struct Slots2 {
  const void* src[2];
  void* dst[2];
  const float* w[2];
};

template <int NT, int NH, int NW, bool CF32, bool FF32, int V>
__global__ __launch_bounds__(256, CSTS_DW_WAVES) void dwconv_transposed_kernel(RowGeom rg, Slots2 sl) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const void* __restrict__ coarse = sl.src[blockIdx.y];
  const float* __restrict__ w = sl.w[blockIdx.y];
  void* __restrict__ fine = sl.dst[blockIdx.y];
  const Geom& g = rg.g;
  stage_weights_z(w, wl, g.HD);
  const int CQ = g.C / V;
  const int ntok = g.Tf * g.Hf * g.Wf;
  const int total = g.B * ntok * CQ;
  const int fts = (int)g.f_ts, cts = (int)g.c_ts;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int bt = fdiv(idx, V == 8 ? rg.dC8 : rg.dC4);
    const int c = (idx - bt * CQ) * V, cw = c - fdiv(c, rg.dHD) * g.HD;
    int b, ft, fh, fw;
    decompf(bt, rg.dNf, rg.dHf, rg.dWf, b, ft, fh, fw);
    int kt[NT], ot[NT], kh[NH], oh[NH], kw[NW], ow[NW];
    bool vt[NT], vh[NH], vw[NW];
    axis_cand<NT>(ft, g.st, rg.lt, g.Tc, kt, ot, vt);
    axis_cand<NH>(fh, g.sh, rg.lh, g.Hc, kh, oh, vh);
    axis_cand<NW>(fw, g.sw, rg.lw, g.Wc, kw, ow, vw);
#pragma unroll
    for (int i = 0; i < NT; ++i) ot[i] *= g.Hc * g.Wc * cts;
#pragma unroll
    for (int i = 0; i < NH; ++i) oh[i] *= g.Wc * cts;
#pragma unroll
    for (int i = 0; i < NW; ++i) ow[i] = ow[i] * cts + c;
    const void* cb = bptr<CF32>(coarse, (int64_t)b * g.c_bs);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      typename RawV<V, CF32>::T raw[NH * NW];
#pragma unroll
      for (int e = 0; e < NH; ++e)
#pragma unroll
        for (int f = 0; f < NW; ++f) raw[e * NW + f] = rawv_load<V, CF32>(cb, ot[a] + oh[e] + ow[f]);
#pragma unroll
      for (int e = 0; e < NH; ++e)
#pragma unroll
        for (int f = 0; f < NW; ++f) {
          const int tap = (vt[a] && vh[e] && vw[f]) ? kt[a] * 9 + kh[e] * 3 + kw[f] : 27;
          float v[V];
          rawv_cvt<V, CF32>(raw[e * NW + f], v);
          fmav<V>(acc, v, &wl[tap * (g.HD + WPAD) + cw]);
        }
    }
    stvt<V, FF32>(bptr<FF32>(fine, (int64_t)b * g.f_bs), (bt - b * ntok) * fts + c, acc);
  }
}

// Transposed convolution with stride 2 along H and W (the (1,2,2) pools of the 384 / 768-channel stages and the decoder's
// upsampling), bf16, fine H and W even: one thread produces the 2 x 2 block of fine positions (2 mh + {0,1}, 2 mw + {0,1})
// for 8 channels.  With kernel 3 / padding 1 an even fine index 2m has the single tap (o = m, k = 1) per axis and an odd
// index 2m + 1 the taps (o = m, k = 2) and (o = m + 1, k = 0): the whole block reads the 2 x 2 coarse neighbourhood
// {mh, mh+1} x {mw, mw+1} once per temporal tap -- 4 loads per 4 outputs where the generic kernel issues 4 per output
// (two of them on average for taps that do not exist at that parity).  The tap loads are what bounds these kernels.
template <int NT>
__global__ __launch_bounds__(256, CSTS_DW_WAVES) void dwconv_transposed_s22_kernel(RowGeom rg, Slots2 sl) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const bf16* __restrict__ coarse = reinterpret_cast<const bf16*>(sl.src[blockIdx.y]);
  bf16* __restrict__ fine = reinterpret_cast<bf16*>(sl.dst[blockIdx.y]);
  const Geom& g = rg.g;
  stage_weights_z(sl.w[blockIdx.y], wl, g.HD);
  const int CQ = g.C / 8;
  const int H2 = g.Hf >> 1, W2 = g.Wf >> 1;
  const int nblk = g.Tf * H2 * W2;
  const int total = g.B * nblk * CQ;
  const int fts = (int)g.f_ts, cts = (int)g.c_ts;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int bt = fdiv(idx, rg.dC8);
    const int c = (idx - bt * CQ) * 8, cw = c - fdiv(c, rg.dHD) * g.HD;
    int b, ft, mh, mw;
    decompf(bt, rg.dN2, rg.dH2, rg.dW2, b, ft, mh, mw);
    int kt[NT], ot[NT];
    bool vt[NT];
    axis_cand<NT>(ft, g.st, rg.lt, g.Tc, kt, ot, vt);
    const bool vh1 = mh + 1 < g.Hc, vw1 = mw + 1 < g.Wc;
    const int oh0 = mh * g.Wc * cts, oh1 = min(mh + 1, g.Hc - 1) * g.Wc * cts;
    const int ow0 = mw * cts + c, ow1 = min(mw + 1, g.Wc - 1) * cts + c;
    const bf16* cb = coarse + (int64_t)b * g.c_bs;
    float o00[8], o01[8], o10[8], o11[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { o00[j] = 0.f; o01[j] = 0.f; o10[j] = 0.f; o11[j] = 0.f; }
    Raw8<false> raw[NT][4];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int tb = ot[a] * g.Hc * g.Wc * cts;
      raw[a][0] = raw8_load<false>(cb, tb + oh0 + ow0);
      raw[a][1] = raw8_load<false>(cb, tb + oh0 + ow1);
      raw[a][2] = raw8_load<false>(cb, tb + oh1 + ow0);
      raw[a][3] = raw8_load<false>(cb, tb + oh1 + ow1);
    }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      float i00[8], i01[8], i10[8], i11[8];
      raw8_cvt<false>(raw[a][0], i00);
      raw8_cvt<false>(raw[a][1], i01);
      raw8_cvt<false>(raw[a][2], i10);
      raw8_cvt<false>(raw[a][3], i11);
      // weight row of tap (kt, kh, kw); taps that do not exist read the all-zero row 27
      const int t9 = kt[a] * 9;
      auto wrow = [&](bool ok, int kh, int kw) { return &wl[(ok ? t9 + kh * 3 + kw : 27) * (g.HD + WPAD) + cw]; };
      const bool v = vt[a];
      fmav<8>(o00, i00, wrow(v, 1, 1));
      fmav<8>(o01, i00, wrow(v, 1, 2));
      fmav<8>(o01, i01, wrow(v && vw1, 1, 0));
      fmav<8>(o10, i00, wrow(v, 2, 1));
      fmav<8>(o10, i10, wrow(v && vh1, 0, 1));
      fmav<8>(o11, i00, wrow(v, 2, 2));
      fmav<8>(o11, i01, wrow(v && vw1, 2, 0));
      fmav<8>(o11, i10, wrow(v && vh1, 0, 2));
      fmav<8>(o11, i11, wrow(v && vh1 && vw1, 0, 0));
    }
    bf16* fb = fine + (int64_t)b * g.f_bs;
    const int f00 = ((ft * g.Hf + 2 * mh) * g.Wf + 2 * mw) * fts + c;
    st8t<false>(fb, f00, o00);
    st8t<false>(fb, f00 + fts, o01);
    st8t<false>(fb, f00 + g.Wf * fts, o10);
    st8t<false>(fb, f00 + g.Wf * fts + fts, o11);
  }
}

// partial dW: block = (slab/2 channel pairs) x (token lanes), chunk of coarse tokens -> ws[part][HD*27]
// token lanes and heads of a slab are folded in LDS in a fixed order (bitwise reproducible).
struct WgSlots2 {
  const void* fine[2];
  const void* coarse[2];
  float* ws[2];
};
// One workgroup's share: channel slab bx of nbx, coarse-token chunk by; thread (tx, ty) = (channel pair, token lane) of
// (slab / 2) x lanes.  Threads with ty >= lanes (the grouped launch's fixed-size workgroups) only take part in the barriers.
template <bool FF32, bool CF32>
__device__ __forceinline__ void dwconv_wgrad_body(const Geom& g, const void* __restrict__ fine, const void* __restrict__ coarse,
                                                  float* __restrict__ ws, int slab, int chunk, int bx, int by, int nbx, int tx, int ty,
                                                  int lanes, int nthr, float* __restrict__ red) {
  const int cl = 2 * tx;
  const int c = bx * slab + cl;
  const int ntok = g.Tc * g.Hc * g.Wc;
  const int total = g.B * ntok;
  const int beg = by * chunk, end = ty < lanes ? min(total, beg + chunk) : 0;
  const int fts = (int)g.f_ts, cts = (int)g.c_ts;
  float a0[27], a1[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) { a0[k] = 0.f; a1[k] = 0.f; }
  // (b, ot, oh, ow) of this lane's token: ONE decomposition, then stepped with the token index (three integer divisions per
  // token were a quarter of this loop's vector instructions)
  int b = 0, ot = 0, oh = 0, ow = 0;
  if (beg + ty < end) decomp(beg + ty, ntok, g.Hc, g.Wc, b, ot, oh, ow);
  const int tstep = lanes;
  // Round 5: every tap address is ONE 32-bit add on top of a wave-uniform base (scalar base + 32-bit lane offset form of global_load:
  // the host guarantees B * tokens * C < 2^31 elements).  The per-lane 64-bit base of rounds 1-4 (fine + b * f_bs as a pointer) made the
  // compiler carry every one of the 27 addresses as a 64-bit quantity: 43 v_lshl_add_u64 + 17 v_mul_lo + sign extensions per token, a
  // third of the loop's 430 vector instructions in a kernel that is bound by vector issue (profiles/r5_stencil_wgrad_isa.txt).
  const char* fbase = reinterpret_cast<const char*>(fine);
  const char* cbase = reinterpret_cast<const char*>(coarse);
  constexpr unsigned FES = FF32 ? 4u : 2u, CES = CF32 ? 4u : 2u;
  for (int bt = beg + ty; bt < end; bt += tstep) {
    unsigned tof[3], hx[9];
    bool tv[3], hxv[9];
    {
      unsigned hof[3], xof[3];
      bool hv[3], xv[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int t = ot * g.st - 1 + k, h = oh * g.sh - 1 + k, x = ow * g.sw - 1 + k;
        tv[k] = (unsigned)t < (unsigned)g.Tf; tof[k] = (unsigned)(min(max(t, 0), g.Tf - 1) * g.Hf * g.Wf * fts);
        hv[k] = (unsigned)h < (unsigned)g.Hf; hof[k] = (unsigned)(min(max(h, 0), g.Hf - 1) * g.Wf * fts);
        xv[k] = (unsigned)x < (unsigned)g.Wf; xof[k] = (unsigned)(min(max(x, 0), g.Wf - 1) * fts + c);
      }
      const unsigned boff = (unsigned)((int)g.f_bs * b);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) { hx[kh * 3 + kw] = boff + hof[kh] + xof[kw]; hxv[kh * 3 + kw] = hv[kh] && xv[kw]; }
    }
    float c0, c1;
    {
      const unsigned coff = ((unsigned)((int)g.c_bs * b) + (unsigned)((bt - b * ntok) * cts + c)) * CES;
      if constexpr (CF32) {
        const float2 v = *reinterpret_cast<const float2*>(cbase + coff);
        c0 = v.x; c1 = v.y;
      } else {
        const unsigned v = *reinterpret_cast<const unsigned*>(cbase + coff);
        c0 = h16_lo(v); c1 = h16_hi(v);
      }
    }
    if constexpr (FF32) {
      float2 raw[27];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int j = 0; j < 9; ++j) raw[kt * 9 + j] = *reinterpret_cast<const float2*>(fbase + (tof[kt] + hx[j]) * FES);
#pragma unroll
      for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          const int tap = kt * 9 + j;
          const bool ok = tv[kt] && hxv[j];
          a0[tap] += (ok ? raw[tap].x : 0.f) * c0;
          a1[tap] += (ok ? raw[tap].y : 0.f) * c1;
        }
    } else {
      unsigned raw[27];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int j = 0; j < 9; ++j) raw[kt * 9 + j] = *reinterpret_cast<const unsigned*>(fbase + (tof[kt] + hx[j]) * FES);
#pragma unroll
      for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          const int tap = kt * 9 + j;
          const unsigned r = (tv[kt] && hxv[j]) ? raw[tap] : 0u;
#ifdef CSTS_SWG_FMA            // A/B build: fused multiply-add (one rounding less per tap; NOT bit-identical to the shipped form)
          a0[tap] = __builtin_fmaf(h16_lo(r), c0, a0[tap]);
          a1[tap] = __builtin_fmaf(h16_hi(r), c1, a1[tap]);
#else
          a0[tap] += h16_lo(r) * c0;
          a1[tap] += h16_hi(r) * c1;
#endif
        }
    }
    ow += tstep;
    while (ow >= g.Wc) {
      ow -= g.Wc;
      if (++oh == g.Hc) {
        oh = 0;
        if (++ot == g.Tc) { ot = 0; ++b; }
      }
    }
  }
  // fold the token lanes in a fixed order (lane 0 stores, lanes 1.. add in turn, as straight-line read / add / write batches:
  // a per-element "first lane?" test inside the unrolled loops compiled to one branch per element)
  if (ty == 0) {
#pragma unroll
    for (int k = 0; k < 27; ++k) { red[cl * 27 + k] = a0[k]; red[(cl + 1) * 27 + k] = a1[k]; }
  }
  __syncthreads();
  for (int l = 1; l < lanes; ++l) {
    if (ty == l) {
      float t0[27], t1[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) { t0[k] = red[cl * 27 + k]; t1[k] = red[(cl + 1) * 27 + k]; }
#pragma unroll
      for (int k = 0; k < 27; ++k) { red[cl * 27 + k] = t0[k] + a0[k]; red[(cl + 1) * 27 + k] = t1[k] + a1[k]; }
    }
    __syncthreads();
  }
  const int heads_in_slab = slab / g.HD;
  float* out = ws + ((int64_t)by * nbx + bx) * g.HD * 27;
  const int tid = ty * (slab / 2) + tx;
  for (int i = tid; i < g.HD * 27; i += nthr) {
    const int cc = i / 27, k = i - cc * 27;
    float s2 = 0.f;
    for (int hh = 0; hh < heads_in_slab; ++hh) s2 += red[(cc + hh * g.HD) * 27 + k];
    out[i] = s2;
  }
}

template <bool FF32, bool CF32>
__global__ void dwconv_wgrad_kernel(RowGeom rg, WgSlots2 sl, int slab, int chunk) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [slab][27]
  dwconv_wgrad_body<FF32, CF32>(rg.g, sl.fine[blockIdx.z], sl.coarse[blockIdx.z], sl.ws[blockIdx.z], slab, chunk, blockIdx.x, blockIdx.y,
                                gridDim.x, threadIdx.x, threadIdx.y, blockDim.y, blockDim.x * blockDim.y, red);
}

// Grouped form (csts_dwconv_wgrad_grouped): every stencil weight gradient of a backward pass in ONE launch.  The item table
// lives in device memory (written by csts_dwconv_wgrad_grouped_plan on the host, uploaded by the caller); workgroup -> item by
// the items' first-block numbers (items sorted longest workgroups first), fixed 512-thread workgroups cut into
// (slab / 2) x lanes like the single launch's.
// (pointers that are LOADED from memory are generic to the compiler and become flat_load / flat_store; declared in the global
// address space here, the casts at the call site let it prove the accesses global again)
typedef __attribute__((address_space(1))) const void* gcvoid_p;
typedef __attribute__((address_space(1))) float* gfloat_p;
struct WgItem {
  Geom g;
  gcvoid_p fine; gcvoid_p coarse; gfloat_p ws;
  int slab, chunk, nslab, lanes;
  int block_begin, nblocks;
};
constexpr int WG_GROUP_THREADS = 512;
// Register budget (round 5): two 512-thread workgroups per CU = four waves per SIMD = 128 registers.  The bf16 build landed on exactly
// 128 by itself, the IEEE-half build on 130 -- one allocation granule more, ONE workgroup per CU, and the launch took 1.09 ms instead of
// 0.65: that launch alone was the whole "fp16 mode is 4 % slower than bf16" of round 4 (per-entry diff of the two bench lines on one
// box, profiles/r5_fp16_vs_bf16.txt).  (The second __launch_bounds__ argument is waves per SIMD.)
template <bool F32>
__global__ __launch_bounds__(WG_GROUP_THREADS, (F32 ? 2 : 4)) void dwconv_wgrad_grouped_kernel(const WgItem* __restrict__ items, int nitems) {
  __shared__ __attribute__((aligned(16))) float red[192 * 27];
  int it = 0;
  while (it + 1 < nitems && (int)blockIdx.x >= items[it + 1].block_begin) ++it;
  const WgItem& w = items[it];
  const int local = blockIdx.x - w.block_begin;
  const int half = w.slab / 2;
  const int ty = threadIdx.x / half, tx = threadIdx.x - ty * half;
  dwconv_wgrad_body<F32, F32>(w.g, (const void*)w.fine, (const void*)w.coarse, (float*)w.ws, w.slab, w.chunk, local % w.nslab, local / w.nslab,
                              w.nslab, tx, ty, w.lanes, WG_GROUP_THREADS, red);
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

// ------------------------------------------------------------------ specialised resampling kernels (scale factors 1 and 2)
// The generic kernels above recover (b, t, h, w, channel group) from a flat 64-bit index with runtime divisions and walk
// runtime-sized candidate windows, testing each candidate with `%` / `/` (max-pool) or re-deriving its interpolation weights
// from torch's source-index formula (trilinear backward: 108 candidates for 16 contributing taps at scale (1, 2, 2)) -- they
// are bound by vector-instruction issue (~1200 instructions per 16 output bytes), not by the tensors they move.  CSTS only
// ever resamples by 1 or 2 per axis: here the contributing taps and their weights are closed forms, a workgroup row is one
// (b, t, h) line of the tensor being written (its decomposition is wave-uniform), and (w, channel group) comes from one
// reciprocal multiply.  Same taps, same weights (exact binary fractions), same order of accumulation as the generic forms:
// bit-identical results (tests/test_gpu_ops.py::test_resampling_specialisations_match_generic).
struct FastRow { int wq; int cq; float inv_cq; };      // wq = (row length in tokens) * cq work items per row

// taps of the trilinear ADJOINT along one axis: coarse index i of n receives output o with weight w (ascending o)
template <int R> __device__ __forceinline__ int up_adj_taps(int i, int n, int (&o)[4], float (&w)[4]) {
  if constexpr (R == 1) { o[0] = i; w[0] = 1.f; return 1; }
  else {
    int k = 0;
    if (i >= 1) { o[k] = 2 * i - 1; w[k] = 0.25f; ++k; }
    o[k] = 2 * i; w[k] = i >= 1 ? 0.75f : 1.f; ++k;
    o[k] = 2 * i + 1; w[k] = i == n - 1 ? 1.f : 0.75f; ++k;       // the last cell: l0 + l1 of the clamped pair
    if (i <= n - 2) { o[k] = 2 * i + 2; w[k] = 0.25f; ++k; }
    return k;
  }
}

template <int RT, int RH, int RW>
__global__ __launch_bounds__(256) void trilinear_bwd_fast_kernel(UpGeom g, FastRow fr, const void* __restrict__ dy, int dy_dt,
                                                                 void* __restrict__ dx, int dx_dt) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= fr.wq) return;
  const int w = (int)(((float)idx + 0.5f) * fr.inv_cq), c = (idx - w * fr.cq) * VEC;
  const int row = blockIdx.y;                                     // (b * Ti + t) * Hi + h, wave-uniform
  const int h = row % g.Hi, bt = row / g.Hi, t = bt % g.Ti, b = bt / g.Ti;
  int ot[4], oh[4], ow[4];
  float wt[4], wh[4], ww[4];
  const int nt = up_adj_taps<RT>(t, g.Ti, ot, wt), nh = up_adj_taps<RH>(h, g.Hi, oh, wh), nw = up_adj_taps<RW>(w, g.Wi, ow, ww);
  const int64_t obase = (int64_t)b * g.To * g.Ho * g.Wo;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < (RT == 1 ? 1 : 4); ++a) {
    if (a >= nt) break;
#pragma unroll
    for (int e = 0; e < (RH == 1 ? 1 : 4); ++e) {
      if (e >= nh) break;
#pragma unroll
      for (int f = 0; f < (RW == 1 ? 1 : 4); ++f) {
        if (f >= nw) break;
        const float wgt = wt[a] * wh[e] * ww[f];
        float v[4];
        ld4(dy, dy_dt, (obase + (int64_t)(ot[a] * g.Ho + oh[e]) * g.Wo + ow[f]) * g.C + c, v);
        s[0] += wgt * v[0]; s[1] += wgt * v[1]; s[2] += wgt * v[2]; s[3] += wgt * v[3];
      }
    }
  }
  st4(dx, dx_dt, ((int64_t)row * g.Wi + w) * g.C + c, s);
}

// MaxPool3d kernel (1, 3, 3), stride (1, 2, 2), padding (0, 1, 1): the only geometry pool_skip takes in CSTS
__global__ __launch_bounds__(256) void maxpool133_fwd_kernel(PoolGeom g, FastRow fr, const void* __restrict__ x, int dt,
                                                             void* __restrict__ y, uint8_t* __restrict__ arg) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= fr.wq) return;
  const int ow = (int)(((float)idx + 0.5f) * fr.inv_cq), c = (idx - ow * fr.cq) * VEC;
  const int row = blockIdx.y;                                     // (b * To + ot) * Ho + oh  (To == Ti)
  const int oh = row % g.Ho, bt = row / g.Ho;                     // bt = b * Ti + t
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0, 0, 0, 0};
  bool first = true;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int h = 2 * oh - 1 + e;
    if (h < 0 || h >= g.Hi) continue;                             // wave-uniform
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const int w = 2 * ow - 1 + f;
      if (w < 0 || w >= g.Wi) continue;
      float v[4];
      ld4(x, dt, (((int64_t)bt * g.Hi + h) * g.Wi + w) * g.C + c, v);
      const int tap = e * 3 + f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (first || v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = tap; }   // first max wins; NaN propagates
      first = false;
    }
  }
  const int64_t o = ((int64_t)row * g.Wo + ow) * g.C + c;
  st4(y, dt, o, best);
  if (arg) *reinterpret_cast<uchar4*>(arg + o) = make_uchar4((uint8_t)bi[0], (uint8_t)bi[1], (uint8_t)bi[2], (uint8_t)bi[3]);
}

__global__ __launch_bounds__(256) void maxpool133_bwd_kernel(PoolGeom g, FastRow fr, const void* __restrict__ dy, int dt,
                                                             const uint8_t* __restrict__ arg, void* __restrict__ dx) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= fr.wq) return;
  const int w = (int)(((float)idx + 0.5f) * fr.inv_cq), c = (idx - w * fr.cq) * VEC;
  const int row = blockIdx.y;                                     // (b * Ti + t) * Hi + h
  const int h = row % g.Hi, bt = row / g.Hi;
  // windows that contain (h, w), in the generic kernel's order (tap index ascending): h even -> oh = h / 2 through tap row 1;
  // h odd -> oh = (h + 1) / 2 through tap row 0, then oh = (h - 1) / 2 through tap row 2
  int oh[2], eh[2], ow[2], fw[2];
  int nh = 0, nw = 0;
  if ((h & 1) == 0) { oh[0] = h >> 1; eh[0] = 1; nh = 1; }
  else {
    if (((h + 1) >> 1) < g.Ho) { oh[nh] = (h + 1) >> 1; eh[nh] = 0; ++nh; }
    oh[nh] = (h - 1) >> 1; eh[nh] = 2; ++nh;
  }
  if ((w & 1) == 0) { ow[0] = w >> 1; fw[0] = 1; nw = 1; }
  else {
    if (((w + 1) >> 1) < g.Wo) { ow[nw] = (w + 1) >> 1; fw[nw] = 0; ++nw; }
    ow[nw] = (w - 1) >> 1; fw[nw] = 2; ++nw;
  }
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if (e >= nh) break;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (f >= nw) break;
      const int64_t o = (((int64_t)bt * g.Ho + oh[e]) * g.Wo + ow[f]) * g.C + c;
      const uchar4 am = *reinterpret_cast<const uchar4*>(arg + o);
      const uint8_t tap = (uint8_t)(eh[e] * 3 + fw[f]);
      float v[4];
      ld4(dy, dt, o, v);
      if (am.x == tap) s[0] += v[0];
      if (am.y == tap) s[1] += v[1];
      if (am.z == tap) s[2] += v[2];
      if (am.w == tap) s[3] += v[3];
    }
  }
  st4(dx, dt, ((int64_t)row * g.Wi + w) * g.C + c, s);
}

// CSTS_RESAMPLE_FAST=0 (A/B switch) sends everything through the generic kernels
bool resample_fast_on() {
  static const bool on = [] { const char* e = getenv("CSTS_RESAMPLE_FAST"); return !(e && e[0] == '0'); }();
  return on;
}
bool fast_row(int row_tokens, int C, int64_t rows, FastRow& fr) {
  fr.cq = C / VEC;
  fr.wq = row_tokens * fr.cq;
  fr.inv_cq = 1.f / (float)fr.cq;
  // the reciprocal-multiply division is exact while the error of (i + 0.5) / cq stays below 0.5 / cq
  return resample_fast_on() && C % VEC == 0 && fr.cq <= 256 && fr.wq <= 16384 && rows <= 65535 * (int64_t)1 && rows > 0;
}

int grid_for(int64_t total) { return (int)std::min<int64_t>(cdiv(total, 256), 256 * 32); }
// kernels that stage the 27 x HD weight table per workgroup.  Whole-step sweeps on MI355X, same box (elementwise cap /
// staged cap -> ms per step): 2048 / 2048 -> 25.04, 2048 / 1024 -> 24.97, 8192 / 1024 -> 24.82..24.96, 8192 / 768 -> 24.85,
// 2048 / 4096 -> 25.27, 1024 / 2048 -> 25.27: 4 staged workgroups per CU, the element-wise kernels effectively uncapped.
int grid_for_staged(int64_t total) { return (int)std::min<int64_t>(cdiv(total, 256), 256 * 4); }

int ilog2(int v) { if (v <= 0 || (v & (v - 1)) != 0) return -1; int l = 0; while ((1 << l) < v) ++l; return l; }   // -1: not a power of two
bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// FastDiv of d >= 1 for dividends below 2^31: with s = ceil(log2 d) and m = ceil(2^(31+s) / d) (< 2^32 because d > 2^(s-1)),
// n m / 2^(31+s) = n / d + n e / 2^(31+s) with 0 <= e < 1, and n / 2^(31+s) < 1 / 2^s <= 1 / d: the floor is that of n / d.
FastDiv fast_div(int d) {
  FastDiv f; f.d = d;
  if (d <= 1) { f.m = 1; f.sh = 0; return f; }
  int sft = 0;
  while ((1LL << sft) < d) ++sft;
  const unsigned long long num = 1ULL << (31 + sft);
  f.m = (unsigned)((num + d - 1) / d);
  f.sh = 31 + sft;
  return f;
}
int fill_geom(const csts_dwconv_geom* a, RowGeom& rg) {
  Geom& g = rg.g;
  g.B = a->B; g.C = a->C; g.HD = a->HD;
  g.Tf = a->Tf; g.Hf = a->Hf; g.Wf = a->Wf; g.Tc = a->Tc; g.Hc = a->Hc; g.Wc = a->Wc;
  g.st = a->st; g.sh = a->sh; g.sw = a->sw;
  g.f_bs = a->fine_batch_stride; g.f_ts = a->fine_token_stride;
  g.c_bs = a->coarse_batch_stride; g.c_ts = a->coarse_token_stride;
  rg.lt = ilog2(a->st); rg.lh = ilog2(a->sh); rg.lw = ilog2(a->sw);
  rg.dNc = fast_div(a->Tc * a->Hc * a->Wc); rg.dHc = fast_div(a->Hc); rg.dWc = fast_div(a->Wc);
  rg.dNf = fast_div(a->Tf * a->Hf * a->Wf); rg.dHf = fast_div(a->Hf); rg.dWf = fast_div(a->Wf);
  rg.dN2 = fast_div(std::max(1, a->Tf * (a->Hf / 2) * (a->Wf / 2))); rg.dH2 = fast_div(std::max(1, a->Hf / 2)); rg.dW2 = fast_div(std::max(1, a->Wf / 2));
  rg.dC4 = fast_div(std::max(1, a->C / 4)); rg.dC8 = fast_div(std::max(1, a->C / 8)); rg.dHD = fast_div(a->HD); rg.dHeads = fast_div(a->C / a->HD);
  return 0;
}

}  // namespace

#define CHECK_GEOM(a)                                                                                         \
  CSTS_REQUIRE((a) != nullptr, "null geometry");                                                              \
  CSTS_REQUIRE((a)->B > 0 && (a)->C > 0 && (a)->HD > 0 && (a)->C % (a)->HD == 0 && (a)->HD % 8 == 0, "bad channels (head_dim % 8)"); \
  CSTS_REQUIRE((a)->HD <= 192 && (a)->C <= 1024, "head_dim > 192 / C > 1024 unsupported");                   \
  CSTS_REQUIRE((a)->st >= 1 && (a)->sh >= 1 && (a)->sw >= 1 && (pow2((a)->st) || (a)->st >= 3) &&                \
               (pow2((a)->sh) || (a)->sh >= 3) && (pow2((a)->sw) || (a)->sw >= 3), "bad strides");               \
  CSTS_REQUIRE((a)->Tc == ((a)->Tf - 1) / (a)->st + 1 && (a)->Hc == ((a)->Hf - 1) / (a)->sh + 1 &&          \
                   (a)->Wc == ((a)->Wf - 1) / (a)->sw + 1,                                                   \
               "coarse grid must equal floor((fine-1)/stride)+1");                                           \
  CSTS_REQUIRE((a)->fine_token_stride % 8 == 0 && (a)->coarse_token_stride % 8 == 0 &&                       \
                   (a)->fine_batch_stride % 8 == 0 && (a)->coarse_batch_stride % 8 == 0,                     \
               "strides must be multiples of 8 elements");                                                   \
  CSTS_REQUIRE((int64_t)(a)->Tf * (a)->Hf * (a)->Wf * (a)->fine_token_stride < ((int64_t)1 << 31) &&          \
                   (int64_t)(a)->Tc * (a)->Hc * (a)->Wc * (a)->coarse_token_stride < ((int64_t)1 << 31),      \
               "one batch element must span < 2^31 elements");                                               \
  CSTS_REQUIRE((int64_t)(a)->B * (a)->Tf * (a)->Hf * (a)->Wf * (a)->C < ((int64_t)1 << 31), "B*N*C must be < 2^31")

extern "C" int csts_dwconv_strided(const csts_dwconv_geom* a, const void* fine, int fine_dt, const float* weight,
                                   void* coarse, int coarse_dt, hipStream_t stream) {
  CHECK_GEOM(a);
  CSTS_REQUIRE(fine && weight && coarse, "null pointer");
  CSTS_REQUIRE(((uintptr_t)fine & 15) == 0 && ((uintptr_t)coarse & 15) == 0, "tensors must be 16-byte aligned");
  RowGeom rg; fill_geom(a, rg);
  const bool ff = fine_dt == CSTS_F32, cf = coarse_dt == CSTS_F32;
  const int vec = (!ff && !cf) ? 8 : VEC;     // bf16 on both sides: 8 channels (16 bytes) per thread
  const int64_t total = (int64_t)a->B * a->Tc * a->Hc * a->Wc * (a->C / vec);
  const dim3 grid(grid_for_staged(total)), block(256);
  const size_t sm = (size_t)(a->HD + WPAD) * 28 * 4;
  if (ff && cf) hipLaunchKernelGGL((dwconv_strided_kernel<true, true, 4>), grid, block, sm, stream, rg, fine, weight, coarse);
  else if (!ff && !cf) hipLaunchKernelGGL((dwconv_strided_kernel<false, false, 8>), grid, block, sm, stream, rg, fine, weight, coarse);
  else if (ff) hipLaunchKernelGGL((dwconv_strided_kernel<true, false, 4>), grid, block, sm, stream, rg, fine, weight, coarse);
  else hipLaunchKernelGGL((dwconv_strided_kernel<false, true, 4>), grid, block, sm, stream, rg, fine, weight, coarse);
  CSTS_LAUNCH_CHECK();
  return 0;
}

static int transposed_launch(const csts_dwconv_geom* a, int nslots, const void* const* coarse, int coarse_dt,
                             const float* const* weight, void* const* fine, int fine_dt, hipStream_t stream) {
  CHECK_GEOM(a);
  CSTS_REQUIRE(nslots == 1 || nslots == 2, "nslots must be 1 or 2");
  Slots2 sl{};
  for (int i = 0; i < nslots; ++i) {
    CSTS_REQUIRE(fine[i] && weight[i] && coarse[i], "null pointer");
    CSTS_REQUIRE(((uintptr_t)fine[i] & 15) == 0 && ((uintptr_t)coarse[i] & 15) == 0, "tensors must be 16-byte aligned");
    sl.src[i] = coarse[i]; sl.dst[i] = fine[i]; sl.w[i] = weight[i];
  }
  RowGeom rg; fill_geom(a, rg);
  CSTS_REQUIRE(coarse_dt == fine_dt, "transposed stencil: both tensors must have the same dtype");
  const bool f32 = fine_dt == CSTS_F32;
  const int64_t total = (int64_t)a->B * a->Tf * a->Hf * a->Wf * (a->C / (f32 ? VEC : 8));   // bf16: 8 channels per thread
  const dim3 grid(grid_for_staged(total), nslots), block(256);
  const size_t sm = (size_t)(a->HD + WPAD) * 28 * 4;
  auto nc = [](int st) { return st == 1 ? 3 : (st == 2 ? 2 : 1); };
  if (!f32 && a->sh == 2 && a->sw == 2 && a->Hf % 2 == 0 && a->Wf % 2 == 0) {   // 2 x 2 output blocks (see the kernel)
    const int64_t items = (int64_t)a->B * a->Tf * (a->Hf / 2) * (a->Wf / 2) * (a->C / 8);
    const dim3 grid2(grid_for_staged(items), nslots);
    if (nc(a->st) == 3) hipLaunchKernelGGL((dwconv_transposed_s22_kernel<3>), grid2, block, sm, stream, rg, sl);
    else if (nc(a->st) == 2) hipLaunchKernelGGL((dwconv_transposed_s22_kernel<2>), grid2, block, sm, stream, rg, sl);
    else hipLaunchKernelGGL((dwconv_transposed_s22_kernel<1>), grid2, block, sm, stream, rg, sl);
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  const int key = nc(a->st) * 100 + nc(a->sh) * 10 + nc(a->sw);
#define TR_CASE(NT, NH, NW)                                                                                          \
  case NT * 100 + NH * 10 + NW:                                                                                      \
    if (f32) hipLaunchKernelGGL((dwconv_transposed_kernel<NT, NH, NW, true, true, 4>), grid, block, sm, stream, rg, sl); \
    else hipLaunchKernelGGL((dwconv_transposed_kernel<NT, NH, NW, false, false, 8>), grid, block, sm, stream, rg, sl);   \
    break;
  switch (key) {
    TR_CASE(3, 3, 3) TR_CASE(3, 3, 2) TR_CASE(3, 3, 1) TR_CASE(3, 2, 3) TR_CASE(3, 2, 2) TR_CASE(3, 2, 1)
    TR_CASE(3, 1, 3) TR_CASE(3, 1, 2) TR_CASE(3, 1, 1) TR_CASE(2, 3, 3) TR_CASE(2, 3, 2) TR_CASE(2, 3, 1)
    TR_CASE(2, 2, 3) TR_CASE(2, 2, 2) TR_CASE(2, 2, 1) TR_CASE(2, 1, 3) TR_CASE(2, 1, 2) TR_CASE(2, 1, 1)
    TR_CASE(1, 3, 3) TR_CASE(1, 3, 2) TR_CASE(1, 3, 1) TR_CASE(1, 2, 3) TR_CASE(1, 2, 2) TR_CASE(1, 2, 1)
    TR_CASE(1, 1, 3) TR_CASE(1, 1, 2) TR_CASE(1, 1, 1)
    default: CSTS_FAIL("unsupported stride combination");
  }
#undef TR_CASE
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_dwconv_transposed(const csts_dwconv_geom* a, const void* coarse, int coarse_dt, const float* weight,
                                      void* fine, int fine_dt, hipStream_t stream) {
  return transposed_launch(a, 1, &coarse, coarse_dt, &weight, &fine, fine_dt, stream);
}

extern "C" int csts_dwconv_transposed2(const csts_dwconv_geom* a, const void* const coarse[2], int coarse_dt,
                                       const float* const weight[2], void* const fine[2], int fine_dt, hipStream_t stream) {
  CSTS_REQUIRE(coarse && weight && fine, "null pointer");
  return transposed_launch(a, 2, coarse, coarse_dt, weight, fine, fine_dt, stream);
}

static void wgrad_plan(const csts_dwconv_geom* a, int& slab, int& nslab, int64_t& chunk, int64_t& nchunk) {
  int k = std::max(1, 192 / a->HD);    // a slab holds k whole heads, <= 192 channels
  while (k > 1 && a->C % (a->HD * k) != 0) --k;
  slab = a->HD * k;
  nslab = a->C / slab;
  const int64_t total = (int64_t)a->B * a->Tc * a->Hc * a->Wc;
  // <= 512 workgroups per tensor (2 per CU), >= 16 coarse tokens each.  Sweep on MI355X (tools/dwconv_bench.py, wgrad calls of
  // one step): caps of 2048 / 1024 / 512 / 256 / 128 -> 1.39 / 1.34 / 1.31 / 1.40 / 2.3 ms; 32 / 64 tokens minimum: worse.
  nchunk = std::max<int64_t>(1, std::min<int64_t>(512 / nslab, cdiv(total, 16)));
  chunk = cdiv(total, nchunk);
  nchunk = cdiv(total, chunk);
}

extern "C" size_t csts_dwconv_wgrad_workspace(const csts_dwconv_geom* a) {
  if (!a || a->C <= 0 || a->HD <= 0) return 0;
  int slab, nslab; int64_t chunk, nchunk;
  wgrad_plan(a, slab, nslab, chunk, nchunk);
  return (size_t)nchunk * nslab * a->HD * 27 * sizeof(float);
}

static int wgrad_launch(const csts_dwconv_geom* a, int nslots, const void* const* fine, int fine_dt, const void* const* coarse,
                        int coarse_dt, float* const* dweight, void* workspace, size_t ws_bytes, hipStream_t stream) {
  CHECK_GEOM(a);
  CSTS_REQUIRE(nslots == 1 || nslots == 2, "nslots must be 1 or 2");
  CSTS_REQUIRE(workspace != nullptr, "null workspace");
  int slab, nslab; int64_t chunk, nchunk;
  wgrad_plan(a, slab, nslab, chunk, nchunk);
  CSTS_REQUIRE(a->C % slab == 0 && slab % a->HD == 0 && slab % 2 == 0, "channel slab must hold whole heads");
  const size_t per_slot = (size_t)nchunk * nslab * a->HD * 27;
  CSTS_REQUIRE(ws_bytes >= per_slot * nslots * sizeof(float), "workspace too small");
  RowGeom rg; fill_geom(a, rg);
  WgSlots2 sl{};
  for (int i = 0; i < nslots; ++i) {
    CSTS_REQUIRE(fine[i] && coarse[i], "null pointer");
    sl.fine[i] = fine[i]; sl.coarse[i] = coarse[i];
    sl.ws[i] = reinterpret_cast<float*>(workspace) + per_slot * i;
  }
  CSTS_REQUIRE(coarse_dt == fine_dt, "stencil wgrad: both tensors must have the same dtype");
  const int lanes = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, 512 / (slab / 2)), chunk / 2));
  const dim3 wg(nslab, (unsigned)nchunk, nslots), wb(slab / 2, lanes);
  // (a 4-channel-per-thread form with 8-byte loads was measured slower: 230 VGPRs halve the resident waves and the tap
  // loads of this kernel are latency-bound)
  if (fine_dt == CSTS_F32) hipLaunchKernelGGL((dwconv_wgrad_kernel<true, true>), wg, wb, (size_t)slab * 27 * 4, stream, rg, sl, slab, (int)chunk);
  else hipLaunchKernelGGL((dwconv_wgrad_kernel<false, false>), wg, wb, (size_t)slab * 27 * 4, stream, rg, sl, slab, (int)chunk);
  CSTS_LAUNCH_CHECK();
  for (int i = 0; i < nslots; ++i) {
    if (dweight != nullptr && dweight[i] != nullptr) {     // NULL: second stage deferred to the caller
      csts_reduce_rows_launch(sl.ws[i], dweight[i], nchunk * nslab, (int64_t)a->HD * 27, 1.f, stream);
      CSTS_LAUNCH_CHECK();
    }
  }
  return 0;
}

extern "C" int csts_dwconv_wgrad(const csts_dwconv_geom* a, const void* fine, int fine_dt, const void* coarse,
                                 int coarse_dt, float* dweight, void* workspace, size_t ws_bytes, hipStream_t stream) {
  return wgrad_launch(a, 1, &fine, fine_dt, &coarse, coarse_dt, &dweight, workspace, ws_bytes, stream);
}

extern "C" int csts_dwconv_wgrad2(const csts_dwconv_geom* a, const void* const fine[2], int fine_dt, const void* const coarse[2],
                                  int coarse_dt, float* const dweight[2], void* workspace, size_t ws_bytes, hipStream_t stream) {
  CSTS_REQUIRE(fine && coarse, "null pointer");
  return wgrad_launch(a, 2, fine, fine_dt, coarse, coarse_dt, dweight, workspace, ws_bytes, stream);
}

static_assert(sizeof(WgItem) == CSTS_DWCONV_WGRAD_TABLE_ENTRY, "device table entry size is part of the C-ABI");

// Chunking of the GROUPED launch: the single launches above cut a tensor into up to 512 workgroups of >= 16 coarse tokens because
// each of them has the chip to itself for one latency-bound round; grouped, the problems fill the chip together, and longer
// chunks amortise the fixed part of a workgroup (LDS fold over the token lanes, partial row, launch) and leave fewer partial rows
// to the second stage.  CSTS_SWG_TOKENS (environment, read once) sets the minimum tokens per workgroup for sweeps.
static int grouped_min_tokens() {
  static const int v = [] { const char* e = getenv("CSTS_SWG_TOKENS"); const int x = e ? atoi(e) : 0; return x >= 8 ? x : 64; }();
  return v;
}
static void wgrad_plan_grouped(const csts_dwconv_geom* a, int& slab, int& nslab, int64_t& chunk, int64_t& nchunk) {
  wgrad_plan(a, slab, nslab, chunk, nchunk);
  const int64_t total = (int64_t)a->B * a->Tc * a->Hc * a->Wc;
  nchunk = std::max<int64_t>(1, std::min<int64_t>(nchunk, cdiv(total, grouped_min_tokens())));
  chunk = cdiv(total, nchunk);
  nchunk = cdiv(total, chunk);
}
extern "C" size_t csts_dwconv_wgrad_grouped_workspace(const csts_dwconv_geom* a) {
  if (!a || a->C <= 0 || a->HD <= 0) return 0;
  int slab, nslab; int64_t chunk, nchunk;
  wgrad_plan_grouped(a, slab, nslab, chunk, nchunk);
  return (size_t)nchunk * nslab * a->HD * 27 * sizeof(float);
}

extern "C" int csts_dwconv_wgrad_grouped_plan(const csts_dwconv_wgrad_item* items, int nitems, void* table_host, size_t table_bytes,
                                              int* nblocks) {
  CSTS_REQUIRE(items != nullptr && table_host != nullptr && nblocks != nullptr && nitems > 0, "null argument");
  CSTS_REQUIRE(table_bytes >= (size_t)nitems * sizeof(WgItem), "table too small");
  std::vector<WgItem> tab((size_t)nitems);
  for (int i = 0; i < nitems; ++i) {
    const csts_dwconv_geom* a = &items[i].geom;
    CHECK_GEOM(a);
    CSTS_REQUIRE(items[i].fine && items[i].coarse && items[i].workspace, "null pointer");
    int slab, nslab; int64_t chunk, nchunk;
    wgrad_plan_grouped(a, slab, nslab, chunk, nchunk);
    CSTS_REQUIRE(a->C % slab == 0 && slab % a->HD == 0 && slab % 2 == 0 && slab <= 192, "channel slab must hold whole heads");
    RowGeom rg; fill_geom(a, rg);
    WgItem& w = tab[(size_t)i];
    w.g = rg.g;
    w.fine = (gcvoid_p)items[i].fine; w.coarse = (gcvoid_p)items[i].coarse; w.ws = (gfloat_p)reinterpret_cast<float*>(items[i].workspace);
    w.slab = slab; w.chunk = (int)chunk; w.nslab = nslab;
    w.lanes = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, WG_GROUP_THREADS / (slab / 2)), chunk / 2));
    w.nblocks = (int)(nchunk * nslab);
    w.block_begin = 0;
  }
  // longest workgroups first (tokens per lane): the short ones fill the tail of the launch
  std::stable_sort(tab.begin(), tab.end(), [](const WgItem& x, const WgItem& y) { return cdiv(x.chunk, x.lanes) > cdiv(y.chunk, y.lanes); });
  int64_t at = 0;
  for (auto& w : tab) { w.block_begin = (int)at; at += w.nblocks; }
  CSTS_REQUIRE(at < ((int64_t)1 << 31), "too many workgroups");
  std::memcpy(table_host, tab.data(), (size_t)nitems * sizeof(WgItem));
  *nblocks = (int)at;
  return 0;
}

extern "C" int csts_dwconv_wgrad_grouped(const void* table_dev, int nitems, int nblocks, int dt, hipStream_t stream) {
  CSTS_REQUIRE(table_dev != nullptr && nitems > 0 && nblocks > 0, "null / empty table");
  CSTS_REQUIRE(dt == CSTS_F32 || dt == CSTS_BF16, "bad dtype");
  const WgItem* items = reinterpret_cast<const WgItem*>(table_dev);
  if (dt == CSTS_F32) hipLaunchKernelGGL(dwconv_wgrad_grouped_kernel<true>, dim3((unsigned)nblocks), dim3(WG_GROUP_THREADS), 0, stream, items, nitems);
  else hipLaunchKernelGGL(dwconv_wgrad_grouped_kernel<false>, dim3((unsigned)nblocks), dim3(WG_GROUP_THREADS), 0, stream, items, nitems);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_pool_ln_fwd(const csts_pool_ln_args* a, hipStream_t stream) {
  CSTS_REQUIRE(a != nullptr, "null args");
  const csts_dwconv_geom* gm = &a->geom;
  CHECK_GEOM(gm);
  CSTS_REQUIRE(a->nslots == 1 || a->nslots == 2, "nslots must be 1 or 2");
  CSTS_REQUIRE(a->dt == CSTS_F32 || a->dt == CSTS_BF16, "bad dtype");
  PoolLnSlots sl{};
  for (int i = 0; i < a->nslots; ++i) {
    CSTS_REQUIRE(a->fine[i] && a->weight[i] && a->gamma[i] && a->beta[i] && a->conv_out[i] && a->y[i] && a->mean[i] && a->rstd[i],
                 "null pointer");
    CSTS_REQUIRE(((uintptr_t)a->fine[i] & 15) == 0 && ((uintptr_t)a->conv_out[i] & 15) == 0 && ((uintptr_t)a->y[i] & 15) == 0 &&
                     ((uintptr_t)a->gamma[i] & 15) == 0 && ((uintptr_t)a->beta[i] & 15) == 0,
                 "tensors must be 16-byte aligned");
    sl.fine[i] = a->fine[i]; sl.w[i] = a->weight[i]; sl.gamma[i] = a->gamma[i]; sl.beta[i] = a->beta[i];
    sl.conv[i] = a->conv_out[i]; sl.y[i] = a->y[i]; sl.mean[i] = a->mean[i]; sl.rstd[i] = a->rstd[i];
  }
  RowGeom rg; fill_geom(gm, rg);
  const int64_t items = (int64_t)gm->B * gm->Tc * gm->Hc * gm->Wc * (gm->C / gm->HD) * a->nslots;
  // twelve channels per lane for the 96- / 192-wide heads (CSTS_POOLLN_CPL12=0: the 8-channel form for every head dim; 1, default: nine taps
  // in flight; 2: all 27).  Whole step, same box, 3 rounds: 19.92 -> 19.75 (1) / 19.77 (2) ms; another box 19.54 -> 19.46 (gpurun_out/r5ao, r5ap)
  static const int cpl12 = [] { const char* e = getenv("CSTS_POOLLN_CPL12"); return e ? atoi(e) : 1; }();
  if (cpl12 > 0 && (gm->HD == 96 || gm->HD == 192) && (((uintptr_t)a->fine[0] | (uintptr_t)a->conv_out[0] | (uintptr_t)a->y[0]) & 7) == 0) {
    const int gl12 = gm->HD / 12, gpb12 = 256 / gl12;
    const dim3 grid12((unsigned)std::min<int64_t>(cdiv(items, gpb12), grid_for_staged((int64_t)1 << 40))), block12(256);
    const size_t sm12 = (size_t)a->nslots * (gm->HD + WPAD) * 28 * 4;
    const bool f32_12 = a->dt == CSTS_F32;
#define CSTS_PL12(GLv, F32v, ALLv) hipLaunchKernelGGL((pool_ln_fwd12_kernel<GLv, F32v, ALLv>), grid12, block12, sm12, stream, rg, sl, a->nslots, a->eps)
    const bool all27 = cpl12 >= 2 && grid12.x <= 512 && !f32_12;       // (fp32 operands: 27 x 12 registers of taps do not fit)
    if (gl12 == 8) {
      if (f32_12) { if (all27) CSTS_PL12(8, true, true); else CSTS_PL12(8, true, false); }
      else { if (all27) CSTS_PL12(8, false, true); else CSTS_PL12(8, false, false); }
    } else {
      if (f32_12) { if (all27) CSTS_PL12(16, true, true); else CSTS_PL12(16, true, false); }
      else { if (all27) CSTS_PL12(16, false, true); else CSTS_PL12(16, false, false); }
    }
#undef CSTS_PL12
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  const int gl = gm->HD <= 128 ? 16 : 32;
  const int gpb = 256 / gl;
  const dim3 grid((unsigned)std::min<int64_t>(cdiv(items, gpb), grid_for_staged((int64_t)1 << 40))), block(256);
  const size_t sm = (size_t)a->nslots * (gm->HD + WPAD) * 28 * 4;
  const bool f32 = a->dt == CSTS_F32;
  if (gl == 16) {
    if (f32) hipLaunchKernelGGL((pool_ln_fwd_kernel<16, true>), grid, block, sm, stream, rg, sl, a->nslots, a->eps);
    else hipLaunchKernelGGL((pool_ln_fwd_kernel<16, false>), grid, block, sm, stream, rg, sl, a->nslots, a->eps);
  } else {
    if (f32) hipLaunchKernelGGL((pool_ln_fwd_kernel<32, true>), grid, block, sm, stream, rg, sl, a->nslots, a->eps);
    else hipLaunchKernelGGL((pool_ln_fwd_kernel<32, false>), grid, block, sm, stream, rg, sl, a->nslots, a->eps);
  }
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
  FastRow fr;
  const bool k133 = a->st == 1 && a->sh == 2 && a->sw == 2;      // kernel (1, 3, 3), padding (0, 1, 1)
  if (k133 && fast_row(g.Wo, g.C, (int64_t)g.B * g.To * g.Ho, fr)) {
    hipLaunchKernelGGL(maxpool133_fwd_kernel, dim3((unsigned)cdiv(fr.wq, 256), (unsigned)(g.B * g.To * g.Ho)), dim3(256), 0, stream,
                       g, fr, x, dt, y, argmax);
    CSTS_LAUNCH_CHECK();
    return 0;
  }
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
  FastRow fr;
  const bool k133 = a->st == 1 && a->sh == 2 && a->sw == 2;
  if (k133 && fast_row(g.Wi, g.C, (int64_t)g.B * g.Ti * g.Hi, fr)) {
    hipLaunchKernelGGL(maxpool133_bwd_kernel, dim3((unsigned)cdiv(fr.wq, 256), (unsigned)(g.B * g.Ti * g.Hi)), dim3(256), 0, stream,
                       g, fr, dy, dt, argmax, dx);
    CSTS_LAUNCH_CHECK();
    return 0;
  }
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
  FastRow fr;
  const int rt = g.To / g.Ti, rh = g.Ho / g.Hi, rw = g.Wo / g.Wi;
  if (rt <= 2 && rh <= 2 && rw <= 2 && fast_row(g.Wi, g.C, (int64_t)g.B * g.Ti * g.Hi, fr)) {
    const dim3 gridf((unsigned)cdiv(fr.wq, 256), (unsigned)(g.B * g.Ti * g.Hi));
#define TRI_CASE(RT, RH, RW)                                                                                              \
    if (rt == RT && rh == RH && rw == RW)                                                                                   \
      hipLaunchKernelGGL((trilinear_bwd_fast_kernel<RT, RH, RW>), gridf, dim3(256), 0, stream, g, fr, dy, dy_dt, dx, dx_dt);
    TRI_CASE(1, 1, 1) TRI_CASE(1, 1, 2) TRI_CASE(1, 2, 1) TRI_CASE(1, 2, 2) TRI_CASE(2, 1, 1) TRI_CASE(2, 1, 2) TRI_CASE(2, 2, 1) TRI_CASE(2, 2, 2)
#undef TRI_CASE
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(trilinear_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g, dy, dy_dt, dx, dx_dt);
  CSTS_LAUNCH_CHECK();
  return 0;
}
