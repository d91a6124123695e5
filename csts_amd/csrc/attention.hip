// Fused pooled multi-head attention for the CSTS path (K6 of SURVEY.md 2.3).
// Reference: softmax(q k^T * hd^-0.5) v, attention.py:154-158, :384-388 (decoder), av_attention.py:137-141
// (temporal) and :334-350 (spatial, same-frame block mask).  The N_q x N_kv matrix is never materialised.
//
// All three kernels keep the "long" index on the MFMA lane and the "short" (reduced) index in the
// accumulator registers, so that the score tile is directly the B operand of the second product
// (no LDS round trip, no cross-lane shuffles except one xor-32 max):
//   fwd  : S^T = K Q^T  (keys in regs, queries on lanes) -> online softmax -> O^T += V^T P^T
//   dQ   : S^T, dP^T = V dO^T, dS = P (dP - delta)       -> dQ^T += K^T dS^T
//   dKdV : S = Q K^T (queries in regs, keys on lanes), dP = dO V^T -> dV^T += dO^T P, dK^T += Q^T dS
// K/V (or Q/dO) tiles are staged in LDS once per workgroup and shared by its 4 waves; the transposed
// operand comes from ds_read_b64_tr_b16.  bf16 mode uses v_mfma_f32_32x32x16_bf16, f32 (parity) mode
// v_mfma_f32_32x32x2_f32 with the accumulator registers consumed directly as the next B operand.
// Softmax statistics, LSE and delta are fp32.  dK/dV are reduced over query splits by a second,
// deterministic pass (no atomics).
#include <type_traits>
#include <utility>
#include "common.h"

// This source is compiled TWICE (round 5): part 0 (attention.hip itself, the library's default flags) holds every kernel and entry
// point except the FAST dK/dV kernel; part 1 (attention_dkv.hip: `#define CSTS_ATTN_PART 1` + `#include "attention.hip"`, built with
// -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize) holds only attn_dkv_kernel<96, false, false> and its launcher.  Why per kernel:
// with the compiler's default AGPR form that kernel copied every S / dP accumulator out through v_accvgpr_read behind an s_nop
// stall (192 copies per 96 MFMAs), with the VGPR form it does not (-9 ... -15 % per launch) -- while the one-pass decoder backward
// and the hd-192 backward kernels measured 4-6 % SLOWER in the VGPR form, and the forward's softmax loses without SLP packing
// (profiles/r5_attn_stream_ab.txt).
#ifndef CSTS_ATTN_PART
#define CSTS_ATTN_PART 0
#endif
#ifndef CSTS_DKV_DEPTH
#define CSTS_DKV_DEPTH 4          // LDS operands requested this many MFMA slots ahead in the dK/dV stream
#endif
#ifndef CSTS_DKV_SWZ
// FAST dK/dV kernel: 1 = 256-byte LDS rows with XOR-swizzled 16-byte chunks (conflict-free transposed reads), 0 = 208-byte padded
// rows (transposed reads 2-way conflicted).  Measured alike per launch (profiles/r5_attn_stream_ab.txt: the kernel is not held by
// its LDS reads at one wave per SIMD, and the swizzled image costs two more LDS-DMA instructions per wave and tile): padded rows
// ship; the swizzled image is what an eight-wave form of this kernel would need (four waves already use the whole LDS port of
// the conflicted transposed reads).
#define CSTS_DKV_SWZ 0
#endif

// Diagnostics build (-DCSTS_ATTN_STAMPS, `make stamps`, tools/attn_stamps.py): thread 0 of workgroup (0,0,0) of the dK/dV
// kernel records shader-clock stamps around the phases of every query tile.  No stamp code exists in the library build.
#ifdef CSTS_ATTN_STAMPS
#if CSTS_ATTN_PART == 1
__device__ unsigned long long g_attn_stamps[4096];
#else
static __device__ unsigned long long g_attn_stamps[4096];     // never filled: the stamped kernel lives in part 1
#endif
#define AT_STAMP() do { if (stamp_on && nstamp < 4000) g_attn_stamps[nstamp++] = (unsigned long long)__builtin_amdgcn_s_memtime(); } while (0)
#if CSTS_ATTN_PART == 1
extern "C" int csts_debug_attn_stamps(unsigned long long* dst_host) {
  return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * 4096);
}
#endif
#define AT_STAMP_P() do { if (stamp_on && *nstamp_p < 4000) g_attn_stamps[(*nstamp_p)++] = (unsigned long long)__builtin_amdgcn_s_memtime(); } while (0)
#define AT_STAMP_PARAMS , bool stamp_on = false, int* nstamp_p = nullptr
#define AT_STAMP_ARGS , stamp_on, &nstamp
#else
#define AT_STAMP() do { } while (0)
#define AT_STAMP_P() do { } while (0)
#define AT_STAMP_PARAMS
#define AT_STAMP_ARGS
#endif

namespace csts_attn {
struct AttnP {
  const void* Q; const void* K; const void* V; void* O; float* LSE;
  const void* dO; float* delta; void* dQ; void* dK; void* dV; float* ws;
  int dt, B, H, Nq, Nk;
  int64_t q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs, o_bs, o_ts, o_hs;
  int64_t do_bs, do_ts, do_hs, dq_bs, dq_ts, dq_hs, dk_bs, dk_ts, dk_hs, dv_bs, dv_ts, dv_hs;
  float scale, scale_log2;
  int mask_mode, mask_T, mask_HW;
  float mask_inv_hw;
  int q_chunk, nsplit;
};
}  // namespace csts_attn
// the FAST dK/dV kernel's launcher (part 1); false: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed
bool csts_attn_dkv_fast_launch(const csts_attn::AttnP& p, dim3 grid, hipStream_t stream);

namespace {
using csts_attn::AttnP;

__device__ __forceinline__ int rowoff(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// frame index of a token of the spatial-fusion sequence [T*HW visual tokens, T audio tokens] (av_attention.py:337-346).
// Branch-free on purpose; the quotient comes from a float multiply, exact while T*HW < 2^20 (checked on the host).
__device__ __forceinline__ int frame_of(const AttnP& p, int x) {
  const int thw = p.mask_T * p.mask_HW;
  const int f = (int)(((float)x + 0.5f) * p.mask_inv_hw);
  return x < thw ? f : x - thw;
}
__device__ __forceinline__ bool is_masked(const AttnP& p, int q, int k) {
  return (p.mask_mode != 0) & (frame_of(p, q) != frame_of(p, k));   // -1e8 outside the same-frame blocks
}

template <int HD, bool F32> struct RowFrag;   // B-operand fragments of one [HD]-row per lane (lane&31 = row)
template <int HD> struct RowFrag<HD, false> {
  bf16x8 b[HD / 16];
  __device__ __forceinline__ void load(const void* base, int64_t off, int h) {
    const bf16* p = reinterpret_cast<const bf16*>(base) + off;
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) b[s] = *reinterpret_cast<const bf16x8*>(p + 16 * s + 8 * h);
  }
};
template <int HD> struct RowFrag<HD, true> {
  float f[HD / 2];
  __device__ __forceinline__ void load(const void* base, int64_t off, int h) {
    const float* p = reinterpret_cast<const float*>(base) + off;
    // plain form: f[s] = row[2s + h]
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) f[s] = p[2 * s + h];
  }
};

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

template <bool F32> struct El;
template <> struct El<false> { typedef bf16 T; };
template <> struct El<true> { typedef float T; };

// cooperative tile load: ROWS x HD elements -> LDS (row stride LD), rows >= lim zero-filled
template <int HD, int ROWS, int LD, bool F32>
__device__ __forceinline__ void load_tile(typename El<F32>::T* S, const void* src, int64_t base, int64_t ts, int row0,
                                          int lim, int tid) {
  constexpr int CPR = HD / 8, CHUNKS = ROWS * CPR;
#pragma unroll
  for (int c = tid; c < CHUNKS; c += 256) {
    const int row = c / CPR, col = (c - row * CPR) * 8;
    if (F32) {
      float v[8];
      if (row0 + row < lim) ld8_as_f32(src, CSTS_F32, base + (int64_t)(row0 + row) * ts + col, v);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      float* d = reinterpret_cast<float*>(S) + row * LD + col;
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = v[j];
    } else {
      bf16x8 v;
      if (row0 + row < lim) v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(src) + base + (int64_t)(row0 + row) * ts + col);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(S) + row * LD + col) = v;
    }
  }
}

// acc(32x32) += X[32 rows][HD] * frag^T   (rows of X -> accumulator rows/registers, frag rows -> lanes)
template <int HD, int LD, bool F32>
__device__ __forceinline__ void score(f32x16& acc, const typename El<F32>::T* X, const RowFrag<HD, F32>& fr, int lane) {
  const int h = lane >> 5;
  if constexpr (F32) {
    const float* Xf = X + (lane & 31) * LD + h;
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xf[2 * s], fr.f[s], acc, 0, 0, 0);
  } else {
    const bf16* Xb = X + (lane & 31) * LD + 8 * h;
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Xb + 16 * s);
      acc = CSTS_MFMA16(a, fr.b[s], acc, 0, 0, 0);
    }
  }
}

// out[dtile](32 x 32, rows = columns dtile*32.. of Y) += Y[32 rows][HD]^T * P   where P (32x32 accumulator
// layout: row index in registers, lane = column) is consumed as the B operand without leaving registers.
template <int HD, int LD, bool F32>
__device__ __forceinline__ void pv(f32x16 (&out)[HD / 32], const typename El<F32>::T* Y, const f32x16& P, int lane) {
  const int h = lane >> 5;
  if constexpr (F32) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float* Yr = Y + ((t & 3) + 8 * (t >> 2) + 4 * h) * LD + (lane & 31);
#pragma unroll
      for (int d = 0; d < HD / 32; ++d) out[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(Yr[32 * d], P[t], out[d], 0, 0, 0);
    }
  } else {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 bp;
#pragma unroll
      for (int j = 0; j < 8; ++j) bp[j] = (bf16)P[8 * s + j];
      // k order inside the step: element j of lane-half h <-> row 16s + 8(j>>2) + 4h + (j&3)
      const bf16* Yb = Y + (16 * s + 4 * h + qq) * LD + 16 * (g & 1) + 4 * pp;
#pragma unroll
      for (int d = 0; d < HD / 32; ++d) {
        const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb + 32 * d));
        const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb + 32 * d + 8 * LD));
        const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0), b1 = __builtin_bit_cast(bf16x4, t1);
        bf16x8 a;
        a[0] = b0[0]; a[1] = b0[1]; a[2] = b0[2]; a[3] = b0[3];
        a[4] = b1[0]; a[5] = b1[1]; a[6] = b1[2]; a[7] = b1[3];
        out[d] = CSTS_MFMA16(a, bp, out[d], 0, 0, 0);
      }
    }
  }
}

// The same two products with their LDS operands read into registers AHEAD of the MFMAs (bf16).  A workgroup of the
// backward kernels is one wave per SIMD: a ds_read issued right in front of the MFMA that consumes it (what the compiler
// schedules by itself) exposes the whole LDS latency once per MFMA, and the matrix pipe sat idle ~80 % of the time.
template <int HD> struct ScoreOps {          // A operands of score(): 32 rows x HD of one staged tile
  bf16x8 a[HD / 16];
  __device__ __forceinline__ void read(const bf16* X, int LD, int lane) {
    const bf16* Xb = X + (lane & 31) * LD + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) a[s] = *reinterpret_cast<const bf16x8*>(Xb + 16 * s);
  }
};
template <int HD>
__device__ __forceinline__ void score_regs(f32x16& acc, const ScoreOps<HD>& x, const RowFrag<HD, false>& fr) {
#pragma unroll
  for (int s = 0; s < HD / 16; ++s) acc = CSTS_MFMA16(x.a[s], fr.b[s], acc, 0, 0, 0);
}
template <int HD> struct PvOps {             // transposed A operands of pv(): the same 32 rows, read column-wise
  s16x4 t[2][HD / 32][2];
  __device__ __forceinline__ void read(const bf16* Y, int LD, int lane) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int h = lane >> 5, g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16* Yb = Y + (16 * s + 4 * h + qq) * LD + 16 * (g & 1) + 4 * pp;
#pragma unroll
      for (int d = 0; d < HD / 32; ++d) {
        t[s][d][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb + 32 * d));
        t[s][d][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb + 32 * d + 8 * LD));
      }
    }
  }
};
template <int HD>
__device__ __forceinline__ void pv_regs(f32x16 (&out)[HD / 32], const PvOps<HD>& y, const f32x16& P) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 bp;
#pragma unroll
    for (int j = 0; j < 8; ++j) bp[j] = (bf16)P[8 * s + j];
#pragma unroll
    for (int d = 0; d < HD / 32; ++d) {
      const bf16x4 b0 = __builtin_bit_cast(bf16x4, y.t[s][d][0]), b1 = __builtin_bit_cast(bf16x4, y.t[s][d][1]);
      bf16x8 a;
      a[0] = b0[0]; a[1] = b0[1]; a[2] = b0[2]; a[3] = b0[3];
      a[4] = b1[0]; a[5] = b1[1]; a[6] = b1[2]; a[7] = b1[3];
      out[d] = CSTS_MFMA16(a, bp, out[d], 0, 0, 0);
    }
  }
}

// store a transposed accumulator: lane owns row `tok`, 4 consecutive columns per register quad
template <int HD>
__device__ __forceinline__ void store_rows(void* dst, int dt, int64_t off, const f32x16 (&acc)[HD / 32], float mul, int h) {
#pragma unroll
  for (int d = 0; d < HD / 32; ++d) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int col = d * 32 + 8 * rq + 4 * h;
      if (dt == CSTS_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + off + col) =
            make_float4(acc[d][4 * rq] * mul, acc[d][4 * rq + 1] * mul, acc[d][4 * rq + 2] * mul, acc[d][4 * rq + 3] * mul);
      } else {
        bf16x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (bf16)(acc[d][4 * rq + j] * mul);
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(dst) + off + col) = v;
      }
    }
  }
}

template <int HD, bool F32> struct Cfg {
  static constexpr int KVBLK = (HD == 96) ? 64 : 32;            // rows per LDS tile
  static constexpr int LD_ROW = F32 ? HD + 1 : HD + 8;           // tiles read row-wise (and possibly transposed)
  static constexpr int LD_TR = F32 ? HD + 4 : (HD == 96 ? 96 : 224);  // tiles only read transposed
};

// raw register copy of this thread's share of one ROWS x HD bf16 tile (16-byte chunks)
template <int HD, int ROWS> struct TileStage {
  static constexpr int CPR = HD / 8, NCH = ROWS * CPR / 256;
  uint4 r[NCH];
  __device__ __forceinline__ void gload(const void* src, int64_t base, int64_t ts, int row0, int lim, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i, row = c / CPR, col = (c - row * CPR) * 8;
      r[i] = (row0 + row < lim)
                 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(src) + base + (int64_t)(row0 + row) * ts + col)
                 : make_uint4(0, 0, 0, 0);
    }
  }
  __device__ __forceinline__ void lstore(bf16* S, int LD, int tid) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i, row = c / CPR, col = (c - row * CPR) * 8;
      *reinterpret_cast<uint4*>(S + row * LD + col) = r[i];
    }
  }
  // the same load with this thread's element offsets (row * ts + col: tile-invariant) computed once by offsets(): inside a
  // tile loop the 64-bit address arithmetic of gload() was ~100 cycles per chunk (in-kernel stamps, tools/attn_stamps.py)
  int off[NCH];
  __device__ __forceinline__ void offsets(int64_t ts, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i, row = c / CPR, col = (c - row * CPR) * 8;
      off[i] = (int)(row * ts) + col;
    }
  }
  __device__ __forceinline__ void gload_at(const bf16* tile0, int row0, int lim, int tid) {   // tile0 = src + base + row0 * ts
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = (tid + 256 * i) / CPR;
      r[i] = (row0 + row < lim) ? *reinterpret_cast<const uint4*>(tile0 + off[i]) : make_uint4(0, 0, 0, 0);
    }
  }
};

// Iterate over ROWS-row tiles [row_beg, row_end) of two tensors X, Y staged in LDS and call body(Xs, Ys, row0).
// bf16: the next tile is prefetched into registers while the current one is consumed, LDS is double-buffered
// (one barrier per tile, the HBM round trip hides under the MFMAs).  f32 (parity mode): synchronous, single buffer.
// STATS: two per-row fp32 vectors (LSE and delta of the tile's rows) travel with the tiles: 2 * ROWS threads prefetch one
// float each and the body finds them in LDS (s1 = stats, s2 = stats + ROWS; rows past row_end repeat the last row).  The
// consumer runs one wave per SIMD, so a global load issued inside the body is an exposed L2 round trip per tile.
template <int HD, int ROWS, int LDX, int LDY, bool F32, bool STATS = false, typename Body>
__device__ __forceinline__ void tile_loop(typename El<F32>::T* smem, const void* X, int64_t xbase, int64_t xts,
                                          const void* Y, int64_t ybase, int64_t yts, int row_beg, int row_end, int tid,
                                          Body&& body, const float* S1 = nullptr, const float* S2 = nullptr AT_STAMP_PARAMS) {
  typedef typename El<F32>::T T;
  [[maybe_unused]] float snext = 0.f;
  auto sload = [&](int r0) {
    if constexpr (STATS) {
      if (tid < 2 * ROWS) {
        const float* sp = tid < ROWS ? S1 : S2;
        snext = sp[min(r0 + (tid < ROWS ? tid : tid - ROWS), row_end - 1)];
      }
    }
  };
  if constexpr (F32) {
    float* stats = reinterpret_cast<float*>(smem + ROWS * (LDX + LDY));
    for (int row0 = row_beg; row0 < row_end; row0 += ROWS) {
      __syncthreads();
      sload(row0);
      load_tile<HD, ROWS, LDX, true>(smem, X, xbase, xts, row0, row_end, tid);
      load_tile<HD, ROWS, LDY, true>(smem + ROWS * LDX, Y, ybase, yts, row0, row_end, tid);
      if constexpr (STATS) { if (tid < 2 * ROWS) stats[tid] = snext; }
      __syncthreads();
      if constexpr (STATS) body(smem, smem + ROWS * LDX, row0, stats);
      else body(smem, smem + ROWS * LDX, row0);
    }
  } else {
    constexpr int TILE = ROWS * (LDX + LDY);
    float* stats = reinterpret_cast<float*>(smem + 2 * TILE);      // [2 buffers][2][ROWS]
    TileStage<HD, ROWS> sx, sy;
    sx.offsets(xts, tid);
    sy.offsets(yts, tid);
    sx.gload(X, xbase, xts, row_beg, row_end, tid);
    sy.gload(Y, ybase, yts, row_beg, row_end, tid);
    sload(row_beg);
    sx.lstore(smem, LDX, tid);
    sy.lstore(smem + ROWS * LDX, LDY, tid);
    if constexpr (STATS) { if (tid < 2 * ROWS) stats[tid] = snext; }
    if (row_beg + ROWS < row_end) {
      sx.gload(X, xbase, xts, row_beg + ROWS, row_end, tid);
      sy.gload(Y, ybase, yts, row_beg + ROWS, row_end, tid);
      sload(row_beg + ROWS);
    }
    __syncthreads();
    int it = 0;
    for (int row0 = row_beg; row0 < row_end; row0 += ROWS, ++it) {
      T* cur = smem + (it & 1) * TILE;
      T* nxt = smem + ((it + 1) & 1) * TILE;
      if constexpr (STATS) body(cur, cur + ROWS * LDX, row0, stats + (it & 1) * 2 * ROWS);
      else body(cur, cur + ROWS * LDX, row0);
      if (row0 + ROWS < row_end) {
        sx.lstore(nxt, LDX, tid);
        sy.lstore(nxt + ROWS * LDX, LDY, tid);
        if constexpr (STATS) { if (tid < 2 * ROWS) stats[((it + 1) & 1) * 2 * ROWS + tid] = snext; }
        AT_STAMP_P();                          // prefetched tile has arrived and is on its way into LDS
        if (row0 + 2 * ROWS < row_end) {
          sx.gload_at(reinterpret_cast<const bf16*>(X) + xbase + (int64_t)(row0 + 2 * ROWS) * xts, row0 + 2 * ROWS, row_end, tid);
          sy.gload_at(reinterpret_cast<const bf16*>(Y) + ybase + (int64_t)(row0 + 2 * ROWS) * yts, row0 + 2 * ROWS, row_end, tid);
          sload(row0 + 2 * ROWS);
        }
        AT_STAMP_P();                          // next loads issued
      }
      __syncthreads();
    }
  }
}

// The same loop with the tiles (and the two statistics vectors) moved global -> LDS by LDS-DMA (global_load_lds): no staging
// registers, no ds_write phase, and the next tile is in flight for the whole body of the current one.  A wave instruction
// writes 64 consecutive 16-byte slots of the LDS image, the image has LD / 8 = 13 slots per row (12 chunks + the 16 bytes of
// row padding that keep the fragment reads conflict-free), so slot -> (row, chunk) is a division by 13 done once per thread;
// the lane that owns a padding slot fetches the row's last chunk again (never read).  bf16, whole tiles only (every row of
// every tile exists: no zero fill), ROWS = 64 or 128.
// a wave-uniform pointer pinned into scalar registers (otherwise loop strength reduction folds the tile base into per-lane
// 64-bit induction variables: 16 more VGPRs in a kernel that has none to spare)
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));      // opaque to the optimiser
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
typedef __attribute__((address_space(1))) const void* attn_gptr_t;
typedef __attribute__((address_space(3))) void* attn_lptr_t;
// one LDS-DMA wave instruction, 16 bytes per lane: global (scalar base + 32-bit lane offset) -> LDS (wave-uniform base in M0 +
// 16 * lane).  Written out because the builtin, inside the tile loop, was given full 64-bit lane addresses (zero-extended
// offsets kept as register pairs: 16 VGPRs, spilled).  The compiler does not count these loads: callers wait with vmcnt(0).
__device__ __forceinline__ void dma16(const char* base, unsigned lane_off, const void* lds_dst) {
  const uint32_t l = (uint32_t)(uintptr_t)(attn_lptr_t)lds_dst;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(lane_off), "s"(base) : "memory", "m0");
}
// SWZ: 256-byte LDS rows (LD = 128) whose sixteen 16-byte chunks are XOR-swizzled by the row, chunk c of row r at position
// c ^ swz16(r): conflict-free for BOTH access patterns of the backward kernels -- ds_read_b128 down a column of chunks (16 lanes =
// 16 rows: 16 different positions) and ds_read_b64_tr_b16 (32 lanes = 4 rows x 64 bytes: the four rows land in different bank
// quarters).  With the padded 208-byte rows the transposed reads were 2-way bank-conflicted (rows r and r + 1 overlap in 4 of 16
// banks): 128 B/clk/CU, exactly what four waves ask for at one transposed operand per 32-cycle MFMA, so the dV / dK blocks ran at
// ~70 cycles per MFMA (in-kernel stamps, profiles/r5_attn_stream_ab.txt).  LDS-DMA writes a wave's 64 chunks to consecutive
// positions, so the swizzle is applied on the SOURCE side: the lane that fills position c' of row r fetches chunk c' ^ swz16(r).
__device__ __forceinline__ constexpr int swz16(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
template <int HD, int ROWS, int LD, bool STATS = true, int LDY = LD, bool SPREAD = false, bool SWZ = false, typename Body>
__device__ __forceinline__ void tile_loop_dma(bf16* smem, const void* X, int64_t xbase, int64_t xts, const void* Y,
                                              int64_t ybase, int64_t yts, int row_beg, int row_end, int tid, Body&& body,
                                              const float* S1 = nullptr, const float* S2 = nullptr AT_STAMP_PARAMS) {
  constexpr int CPR = HD / 8, SPRX = LD / 8, SPRY = LDY / 8, NIX = ROWS * SPRX / 64, NIY = ROWS * SPRY / 64;
  constexpr int MAXIX = (NIX + 3) / 4, MAXIY = (NIY + 3) / 4, TILE = ROWS * (LD + LDY);
  static_assert(LD % 8 == 0 && LDY % 8 == 0 && (ROWS * SPRX) % 64 == 0 && (ROWS * SPRY) % 64 == 0 && SPRX >= CPR && SPRY >= CPR &&
                    (ROWS == 64 || ROWS == 128), "LDS images must be whole wave instructions");
  static_assert(!SWZ || (LD == 128 && LDY == 128), "the swizzled image has 256-byte rows");
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned xo[MAXIX], yo[MAXIY];               // byte offsets of this lane's slot in instruction w + 4 i of a tile (tile-invariant)
  auto src_chunk = [](int row, int pos) {      // which 16-byte chunk of the row goes to position pos of its LDS row
    if constexpr (SWZ) { const int c = pos ^ swz16(row); return c < CPR ? c : (c & 7); }   // positions >= CPR: padding, never read
    else return min(pos, CPR - 1);
  };
#pragma unroll
  for (int i = 0; i < MAXIX; ++i) {
    const int slot = (w + 4 * i) * 64 + lane, row = slot / SPRX, c = src_chunk(row, slot - row * SPRX);
    xo[i] = (unsigned)(row * (int)xts + c * 8) * 2u;
  }
#pragma unroll
  for (int i = 0; i < MAXIY; ++i) {
    const int slot = (w + 4 * i) * 64 + lane, row = slot / SPRY, c = src_chunk(row, slot - row * SPRY);
    yo[i] = (unsigned)(row * (int)yts + c * 8) * 2u;
  }
  float* stats = reinterpret_cast<float*>(smem + 2 * TILE);      // [2 buffers][2][ROWS]
  constexpr int WPS = ROWS / 64;
  // instruction i of this wave's share of a tile: i < MAXIX: X image, < MAXIX + MAXIY: Y image, == MAXIX + MAXIY: statistics
  // (scalar tile base + 32-bit lane offset: the saddr form of the instruction)
  auto tile_x = [&](int row0) { return uniform_ptr(reinterpret_cast<const bf16*>(X) + xbase + (int64_t)row0 * xts); };
  auto tile_y = [&](int row0) { return uniform_ptr(reinterpret_cast<const bf16*>(Y) + ybase + (int64_t)row0 * yts); };
  auto issue_one = [&](auto ic, const char* xs, const char* ys, int row0, int buf) {
    constexpr int i = decltype(ic)::value;
    char* dx = reinterpret_cast<char*>(smem + buf * TILE);
    if constexpr (i < MAXIX) {
      const int t = w + 4 * i;                 // wave-uniform
      if (t < NIX) dma16(xs, xo[i], dx + t * 1024);
    } else if constexpr (i < MAXIX + MAXIY) {
      const int t = w + 4 * (i - MAXIX);
      if (t < NIY) dma16(ys, yo[i - MAXIX], dx + ROWS * LD * 2 + t * 1024);
    } else {
      // LSE and delta of the tile's rows, one dword per lane: 64 rows -> waves 0 / 1; 128 rows -> waves 0, 1 / 2, 3
      if (STATS && w < 2 * WPS)
        __builtin_amdgcn_global_load_lds((attn_gptr_t)((w < WPS ? S1 : S2) + row0 + (w % WPS) * 64 + lane),
                                         (attn_lptr_t)(stats + buf * 2 * ROWS + (w / WPS) * ROWS + (w % WPS) * 64), 4, 0, 0);
    }
  };
  auto issue = [&](int row0, int buf) {
    const char* xs = tile_x(row0);
    const char* ys = tile_y(row0);
    static_for<MAXIX + MAXIY + 1>([&](auto ic) { issue_one(ic, xs, ys, row0, buf); });
  };
  issue(row_beg, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int it = 0;
  for (int row0 = row_beg; row0 < row_end; row0 += ROWS, ++it) {
    bf16* cur = smem + (it & 1) * TILE;
    const bool more = row0 + ROWS < row_end;   // wave-uniform.  (The other buffer was last read before the previous barrier.)
    if constexpr (SPREAD) {
      // the body issues the next tile's instructions itself, one per call of dma(integral_constant<i>), i < NDMA, between its MFMAs:
      // issued in one burst in front of the body they were ~1000 cycles per tile in which the wave (alone on its SIMD) did nothing else
      const char* xs = tile_x(more ? row0 + ROWS : row0);
      const char* ys = tile_y(more ? row0 + ROWS : row0);
      auto dma = [&](auto ic) { if (more) issue_one(ic, xs, ys, row0 + ROWS, (it + 1) & 1); };
      AT_STAMP_P();
      body(cur, cur + ROWS * LD, row0, stats + (it & 1) * 2 * ROWS, dma);
    } else {
      if (more) issue(row0 + ROWS, (it + 1) & 1);
      AT_STAMP_P();
      if constexpr (STATS) body(cur, cur + ROWS * LD, row0, stats + (it & 1) * 2 * ROWS);
      else body(cur, cur + ROWS * LD, row0);
    }
    AT_STAMP_P();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}
// number of dma(i) calls a SPREAD body owes per tile
template <int ROWS, int LD, int LDY = LD> constexpr int tile_dma_count() {
  return (ROWS * (LD / 8) / 64 + 3) / 4 + (ROWS * (LDY / 8) / 64 + 3) / 4 + 1;
}

// Row-per-lane operands (the 128 query rows of a workgroup, one per lane) are moved through LDS: the tile is loaded /
// stored by the whole workgroup in 16-byte chunks that run along each 2*HD-byte row (consecutive lanes -> consecutive
// chunks), and only the LDS side is accessed row-per-lane.  A direct row-per-lane global access makes every wave
// instruction a 64-row gather / scatter (64 different 128-byte lines per instruction).
template <int HD>
__device__ __forceinline__ void stage_rows_in(bf16* S, const void* src, int64_t base, int64_t ts, int row0, int nrows_valid,
                                              int tid) {
  constexpr int CPR = HD / 8, LDS_LD = HD + 8;
#pragma unroll 2
  for (int c = tid; c < 128 * CPR; c += 256) {
    const int row = c / CPR, col = (c - row * CPR) * 8;
    const int r = min(row0 + row, nrows_valid - 1);
    *reinterpret_cast<uint4*>(S + row * LDS_LD + col) =
        *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(src) + base + (int64_t)r * ts + col);
  }
}
template <int HD>
__device__ __forceinline__ void frag_from_lds(RowFrag<HD, false>& f, const bf16* S, int row, int h) {
  constexpr int LDS_LD = HD + 8;
#pragma unroll
  for (int s = 0; s < HD / 16; ++s) f.b[s] = *reinterpret_cast<const bf16x8*>(S + row * LDS_LD + 16 * s + 8 * h);
}
// accumulator rows (lane = row) -> LDS [128][HD+8] bf16 -> coalesced 16-byte global stores
template <int HD>
__device__ __forceinline__ void stage_rows_out(bf16* S, void* dst, int64_t base, int64_t ts, int row0, int nrows_valid,
                                               const f32x16 (&acc)[HD / 32], float mul, int row_in_tile, int h, int tid) {
  constexpr int CPR = HD / 8, LDS_LD = HD + 8;
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      bf16x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (bf16)(acc[d][4 * rq + j] * mul);
      *reinterpret_cast<bf16x4*>(S + row_in_tile * LDS_LD + d * 32 + 8 * rq + 4 * h) = v;
    }
  __syncthreads();
#pragma unroll 2
  for (int c = tid; c < 128 * CPR; c += 256) {
    const int row = c / CPR, col = (c - row * CPR) * 8;
    if (row0 + row < nrows_valid)
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(dst) + base + (int64_t)(row0 + row) * ts + col) =
          *reinterpret_cast<const uint4*>(S + row * LDS_LD + col);
  }
}

// same, for ONE wave's 32 rows (rows 32 w .. 32 w + 31 of the staging tile are written and read back by wave w only: LDS
// executes a wave's instructions in order, so no barrier is needed)
template <int HD>
__device__ __forceinline__ void stage_rows_out_wave(bf16* S, void* dst, int64_t base, int64_t ts, int row0, int nrows_valid,
                                                    const f32x16 (&acc)[HD / 32], float mul, int w, int lane) {
  constexpr int CPR = HD / 8, LDS_LD = HD + 8;
  const int h = lane >> 5, row_in_tile = w * 32 + (lane & 31);
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      bf16x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (bf16)(acc[d][4 * rq + j] * mul);
      *reinterpret_cast<bf16x4*>(S + row_in_tile * LDS_LD + d * 32 + 8 * rq + 4 * h) = v;
    }
#pragma unroll
  for (int i = 0; i < 32 * CPR / 64; ++i) {
    const int c = lane + 64 * i, r = c / CPR, col = (c - r * CPR) * 8, row = w * 32 + r;
    if (row0 + row < nrows_valid)
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(dst) + base + (int64_t)(row0 + row) * ts + col) =
          *reinterpret_cast<const uint4*>(S + row * LDS_LD + col);
  }
}

// ---------------------------------------------------------------------------------------------- forward
// hd 96 / bf16: a 256-register budget = two workgroups (2 waves per SIMD) per CU; unconstrained the compiler took 272
// registers and the kernel ran one wave per SIMD (63 -> 44 us on the 32 k-query x 512-key block, 108 -> 64 us on the decoder)
// Workgroups are dealt round-robin over the 8 XCDs in linear-id order (x fastest), so the query tiles (forward, dQ) or key blocks and
// splits (dK/dV) of ONE (clip, head) land on all eight XCDs and each XCD's L2 fetches that head's K / V (or Q / dO) again: PMC traffic
// 1.8 x (forward) and 2.9 x (backward) the algorithmic bytes (profiles/r4_v6_mfma_util.txt).  The remap gives XCD x the contiguous
// eighth x of the linear ids -- all workgroups of a (clip, head) on one XCD.  Bijective for any grid.  -DCSTS_ATTN_NO_XCD_MAP: plain order.
__device__ __forceinline__ void xcd_remap3(int& bx, int& by, int& bz) {
#ifdef CSTS_ATTN_NO_XCD_MAP
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#else
  const int gx = gridDim.x, gy = gridDim.y, n = gx * gy * (int)gridDim.z;
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int q = n >> 3, r = n & 7, xcd = lin & 7, slot = lin >> 3;
  const int l2 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  bx = l2 % gx;
  by = (l2 / gx) % gy;
  bz = l2 / (gx * gy);
#endif
}
template <int HD, bool F32>
__global__ __launch_bounds__(256, ((HD == 96 && !F32) ? 2 : 1)) void attn_fwd_kernel(AttnP p) {
  typedef typename El<F32>::T T;
  typedef Cfg<HD, F32> C;
  constexpr int KT = C::KVBLK / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  int BX, BY, BZ;
  xcd_remap3(BX, BY, BZ);                // XCD-contiguous workgroup order: see xcd_remap3
  const int b = BZ, head = BY;
  const int qraw = BX * 128 + w * 32 + (lane & 31);
  const bool qvalid = qraw < p.Nq;
  const int qi = qvalid ? qraw : p.Nq - 1;

  RowFrag<HD, F32> qf;
  if constexpr (F32) {
    qf.load(p.Q, (int64_t)b * p.q_bs + (int64_t)qi * p.q_ts + (int64_t)head * p.q_hs, h);
  } else {
    stage_rows_in<HD>(smem, p.Q, (int64_t)b * p.q_bs + (int64_t)head * p.q_hs, p.q_ts, BX * 128, p.Nq, tid);
    __syncthreads();
    frag_from_lds<HD>(qf, smem, w * 32 + (lane & 31), h);
    __syncthreads();          // the K/V tile loop reuses this LDS
  }
  f32x16 O[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
  float m = -1e30f, lsum = 0.f;
  const int fq = frame_of(p, qi);
  const int64_t kbase = (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
  const int64_t vbase = (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;

  tile_loop<HD, C::KVBLK, C::LD_ROW, C::LD_TR, F32>(smem, p.K, kbase, p.k_ts, p.V, vbase, p.v_ts, 0, p.Nk, tid,
                                                    [&](const T* Ks, const T* Vs, int k0) {
    f32x16 S[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
      score<HD, C::LD_ROW, F32>(S[kt], Ks + kt * 32 * C::LD_ROW, qf, lane);
    }
    // tiles that need no masking (all but the ragged last one, and every tile of unmasked attention) skip the
    // per-element key tests; the softmax scale rides on the exp2 argument's fma (S stays unscaled)
    if ((k0 + C::KVBLK > p.Nk) || p.mask_mode != 0) {
      // dead keys sit at -2^16 in the exp2 domain: far below any live score, yet small enough that the fma's
      // rounding error (2^16 * 2^-24) cannot blow up exp2 while a row has seen dead keys only (that state is
      // wiped by alpha = 0 at the row's first live key)
      const float dead_raw = -65536.f / p.scale_log2;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kt * 32 + rowoff(r, h);
          const bool dead = (key >= p.Nk) | ((p.mask_mode != 0) & (frame_of(p, key) != fq));
          S[kt][r] = dead ? dead_raw : S[kt][r];
        }
    }
    float mx = S[0][0];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float ms = mx * p.scale_log2;
    // lazy running max: rescale only when some row's max grew by more than 2^8 (p <= 256 keeps full fp32/bf16
    // precision and the final 1/lsum normalisation makes the result independent of the reference point)
    const bool grow = ms > m + 8.f;
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
      const float mn = grow ? ms : m;
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      m = mn;
      lsum *= alpha;
#pragma unroll
      for (int d = 0; d < HD / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
    }
    const float negm = -m;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][r], p.scale_log2, negm));
        S[kt][r] = e;
        lsum += e;
      }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) pv<HD, C::LD_TR, F32>(O, Vs + kt * 32 * C::LD_TR, S[kt], lane);
  });
  lsum += __shfl_xor(lsum, 32, 64);
  if constexpr (F32) {
    if (qvalid) store_rows<HD>(p.O, p.dt, (int64_t)b * p.o_bs + (int64_t)qi * p.o_ts + (int64_t)head * p.o_hs, O, 1.f / lsum, h);
  } else {
    stage_rows_out<HD>(smem, p.O, (int64_t)b * p.o_bs + (int64_t)head * p.o_hs, p.o_ts, BX * 128, p.Nq, O, 1.f / lsum,
                       w * 32 + (lane & 31), h, tid);
  }
  if (qvalid && h == 0 && p.LSE) p.LSE[((int64_t)b * p.H + head) * p.Nq + qi] = m + __builtin_amdgcn_logf(lsum);  // log2 domain
}

#if CSTS_ATTN_PART == 0     // (non-template kernels are emitted by every translation unit that sees them)
// forward, FAST form (hd 96, bf16, no mask, every 64-key tile whole): K / V tiles by LDS-DMA (K rows padded to 104 elements,
// V -- read transposed only -- unpadded) and the MFMA slot stream of the backward kernels: 12 slots S = Q K^T (two 32-key
// units), the online softmax of the tile (it needs the maximum over both units, so it cannot ride on its own block), 12 slots
// O += P V; the operand of slot g + DEPTH is requested before the MFMA of slot g, across the softmax as well.
#ifndef CSTS_ATTN_FWD_WAVES
#define CSTS_ATTN_FWD_WAVES 3     // waves per SIMD attn_fwd_fast_kernel is compiled for (A/B builds: 2, 4)
#endif
__global__ __launch_bounds__(256, CSTS_ATTN_FWD_WAVES) void attn_fwd_fast_kernel(AttnP p) {
  constexpr int HD = 96;
  typedef Cfg<HD, false> C;
  constexpr int KBLK = 64, LDK = C::LD_ROW, LDV = C::LD_TR;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  int BX, BY, BZ;
  xcd_remap3(BX, BY, BZ);                // XCD-contiguous workgroup order: see xcd_remap3
  const int b = BZ, head = BY;
  const int qraw = BX * 128 + w * 32 + (lane & 31);
  const bool qvalid = qraw < p.Nq;
  const int qi = qvalid ? qraw : p.Nq - 1;

  RowFrag<HD, false> qf;
  stage_rows_in<HD>(smem, p.Q, (int64_t)b * p.q_bs + (int64_t)head * p.q_hs, p.q_ts, BX * 128, p.Nq, tid);
  __syncthreads();
  frag_from_lds<HD>(qf, smem, w * 32 + (lane & 31), h);
  __syncthreads();          // the K/V tile loop reuses this LDS
  f32x16 O[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
  float m = -1e30f, lsum = 0.f;
  const int64_t kbase = (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
  const int64_t vbase = (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;

  tile_loop_dma<HD, KBLK, LDK, false, LDV>(smem, p.K, kbase, p.k_ts, p.V, vbase, p.v_ts, 0, p.Nk, tid,
                                           [&](const bf16* Ks, const bf16* Vs, int) {
    constexpr int NS = HD / 16, ND = HD / 32, SCB = 2 * NS, DEPTH = 4, TOTAL = 2 * SCB;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int g1 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
    const bf16* rowK = Ks + (lane & 31) * LDK + 8 * h;
    const bf16* trV = Vs + (4 * h + qq) * LDV + 16 * g1 + 4 * pp;
    bf16x8 ring[DEPTH + 1];
    auto request = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      bf16x8& dst = ring[g % (DEPTH + 1)];
      if constexpr (g < SCB) {
        constexpr int u = g / NS, k = g % NS;
        dst = *reinterpret_cast<const bf16x8*>(rowK + u * 32 * LDK + 16 * k);
      } else {
        constexpr int j = g - SCB, u = j / (2 * ND), s2 = (j % (2 * ND)) / ND, d = j % ND;
        const bf16* Yb = trV + (u * 32 + 16 * s2) * LDV + 32 * d;
        const bf16x4 b0 = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb)));
        const bf16x4 b1 = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb + 8 * LDV)));
        dst[0] = b0[0]; dst[1] = b0[1]; dst[2] = b0[2]; dst[3] = b0[3];
        dst[4] = b1[0]; dst[5] = b1[1]; dst[6] = b1[2]; dst[7] = b1[3];
      }
    };
    static_for<DEPTH>([&](auto gc) { request(gc); });
    f32x16 S[2];
    bf16x8 bp[2][2];
    static_for<TOTAL>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g == 0 || g == NS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[g / NS][r] = 0.f;
      }
      if constexpr (g == SCB) {
        // online softmax of the tile (same arithmetic, in the same order, as attn_fwd_kernel): lazy running maximum -- rescale
        // only when some row's maximum grew by more than 2^8 -- the scale rides on the exp2 argument's fma
        float mx = S[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float ms = mx * p.scale_log2;
        const bool grow = ms > m + 8.f;
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {
          const float mn = grow ? ms : m;
          const float alpha = __builtin_amdgcn_exp2f(m - mn);
          m = mn;
          lsum *= alpha;
#pragma unroll
          for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
        }
        const float negm = -m;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][r], p.scale_log2, negm));
            lsum += e;
            bp[kt][r / 8][r % 8] = (bf16)e;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (g + DEPTH < TOTAL) request(std::integral_constant<int, g + DEPTH>{});
      const bf16x8 a = ring[g % (DEPTH + 1)];
      if constexpr (g < SCB) {
        S[g / NS] = CSTS_MFMA16(a, qf.b[g % NS], S[g / NS], 0, 0, 0);
      } else {
        constexpr int j = g - SCB, u = j / (2 * ND), s2 = (j % (2 * ND)) / ND, d = j % ND;
        O[d] = CSTS_MFMA16(a, bp[u][s2], O[d], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  lsum += __shfl_xor(lsum, 32, 64);
  stage_rows_out<HD>(smem, p.O, (int64_t)b * p.o_bs + (int64_t)head * p.o_hs, p.o_ts, BX * 128, p.Nq, O, 1.f / lsum,
                     w * 32 + (lane & 31), h, tid);
  if (qvalid && h == 0 && p.LSE) p.LSE[((int64_t)b * p.H + head) * p.Nq + qi] = m + __builtin_amdgcn_logf(lsum);  // log2 domain
}

#endif
// ---------------------------------------------------------------------------------------------- dQ
template <int HD, bool F32>
__global__ __launch_bounds__(256, ((HD == 96 && !F32) ? 2 : 1)) void attn_dq_kernel(AttnP p) {
  typedef typename El<F32>::T T;
  typedef Cfg<HD, F32> C;
  constexpr int KT = C::KVBLK / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  int BX, BY, BZ;
  xcd_remap3(BX, BY, BZ);                // XCD-contiguous workgroup order: see xcd_remap3
  const int b = BZ, head = BY;
  const int qraw = BX * 128 + w * 32 + (lane & 31);
  const bool qvalid = qraw < p.Nq;
  const int qi = qvalid ? qraw : p.Nq - 1;

  RowFrag<HD, F32> qf, dof;
  [[maybe_unused]] RowFrag<HD, F32> of_pre;       // hd 96 / bf16: O row fragments, staged together with Q and dO
  constexpr bool PRE3 = !F32 && HD == 96;          // two 128-row staging regions fit the kernel's LDS only at hd 96
  if constexpr (F32) {
    qf.load(p.Q, (int64_t)b * p.q_bs + (int64_t)qi * p.q_ts + (int64_t)head * p.q_hs, h);
    dof.load(p.dO, (int64_t)b * p.do_bs + (int64_t)qi * p.do_ts + (int64_t)head * p.do_hs, h);
  } else if constexpr (PRE3) {
    // all three row tiles are requested at once (ONE memory round trip instead of three in a row), then pass through the
    // two LDS staging regions
    constexpr int LDS_LD = HD + 8;
    TileStage<HD, 128> tq, tdo, to;
    tq.gload(p.Q, (int64_t)b * p.q_bs + (int64_t)head * p.q_hs, p.q_ts, BX * 128, p.Nq, tid);
    tdo.gload(p.dO, (int64_t)b * p.do_bs + (int64_t)head * p.do_hs, p.do_ts, BX * 128, p.Nq, tid);
    to.gload(p.O, (int64_t)b * p.o_bs + (int64_t)head * p.o_hs, p.o_ts, BX * 128, p.Nq, tid);
    tq.lstore(smem, LDS_LD, tid);
    tdo.lstore(smem + 128 * LDS_LD, LDS_LD, tid);
    __syncthreads();
    frag_from_lds<HD>(qf, smem, w * 32 + (lane & 31), h);
    frag_from_lds<HD>(dof, smem + 128 * LDS_LD, w * 32 + (lane & 31), h);
    __syncthreads();
    to.lstore(smem, LDS_LD, tid);
    __syncthreads();
    frag_from_lds<HD>(of_pre, smem, w * 32 + (lane & 31), h);
    __syncthreads();
  } else {
    stage_rows_in<HD>(smem, p.Q, (int64_t)b * p.q_bs + (int64_t)head * p.q_hs, p.q_ts, BX * 128, p.Nq, tid);
    __syncthreads();
    frag_from_lds<HD>(qf, smem, w * 32 + (lane & 31), h);
    __syncthreads();
    stage_rows_in<HD>(smem, p.dO, (int64_t)b * p.do_bs + (int64_t)head * p.do_hs, p.do_ts, BX * 128, p.Nq, tid);
    __syncthreads();
    frag_from_lds<HD>(dof, smem, w * 32 + (lane & 31), h);
    __syncthreads();
  }
  const float L = p.LSE[((int64_t)b * p.H + head) * p.Nq + qi];
  // delta = rowsum(dO * O): a dot product of two row fragments of this lane's query (the half-waves hold alternate
  // pieces of the row); kept for the dK/dV kernel, which runs after this one
  float dl = 0.f;
  {
    RowFrag<HD, F32> of;
    if constexpr (F32) {
      of.load(p.O, (int64_t)b * p.o_bs + (int64_t)qi * p.o_ts + (int64_t)head * p.o_hs, h);
#pragma unroll
      for (int s2 = 0; s2 < HD / 2; ++s2) dl += dof.f[s2] * of.f[s2];
    } else {
      if constexpr (PRE3) {
        of = of_pre;
      } else {
        stage_rows_in<HD>(smem, p.O, (int64_t)b * p.o_bs + (int64_t)head * p.o_hs, p.o_ts, BX * 128, p.Nq, tid);
        __syncthreads();
        frag_from_lds<HD>(of, smem, w * 32 + (lane & 31), h);
        __syncthreads();
      }
#pragma unroll
      for (int s2 = 0; s2 < HD / 16; ++s2)
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) dl += (float)dof.b[s2][j2] * (float)of.b[s2][j2];
    }
    dl += __shfl_xor(dl, 32, 64);
    if (h == 0 && qvalid) p.delta[((int64_t)b * p.H + head) * p.Nq + qi] = dl;
  }
  const float negL = -L;
  const int fq = frame_of(p, qi);
  f32x16 acc[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  const int64_t kbase = (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
  const int64_t vbase = (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;

  tile_loop<HD, C::KVBLK, C::LD_ROW, C::LD_ROW, F32>(smem, p.K, kbase, p.k_ts, p.V, vbase, p.v_ts, 0, p.Nk, tid,
                                                     [&](const T* Ks, const T* Vs, int k0) {
    const bool slow = (k0 + C::KVBLK > p.Nk) || p.mask_mode != 0;   // wave-uniform
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      f32x16 S, dP;
#pragma unroll
      for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
      score<HD, C::LD_ROW, F32>(S, Ks + kt * 32 * C::LD_ROW, qf, lane);
      score<HD, C::LD_ROW, F32>(dP, Vs + kt * 32 * C::LD_ROW, dof, lane);
      if (slow) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kt * 32 + rowoff(r, h);
          const bool dead = (key >= p.Nk) | ((p.mask_mode != 0) & (frame_of(p, key) != fq));
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], p.scale_log2, negL));
          const float pr = dead ? 0.f : e;
          S[r] = pr * (dP[r] - dl);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], p.scale_log2, negL)) * (dP[r] - dl);
      }
      pv<HD, C::LD_ROW, F32>(acc, Ks + kt * 32 * C::LD_ROW, S, lane);
    }
  });
  if constexpr (F32) {
    if (qvalid)
      store_rows<HD>(p.dQ, p.dt, (int64_t)b * p.dq_bs + (int64_t)qi * p.dq_ts + (int64_t)head * p.dq_hs, acc, p.scale, h);
  } else {
    stage_rows_out<HD>(smem, p.dQ, (int64_t)b * p.dq_bs + (int64_t)head * p.dq_hs, p.dq_ts, BX * 128, p.Nq, acc, p.scale,
                       w * 32 + (lane & 31), h, tid);
  }
}

// dQ, FAST form (hd 96, bf16, no mask, every KT * 32-key tile whole): K / V tiles by LDS-DMA and the MFMA slot stream of
// attn_dkv_kernel (operands requested DEPTH slots ahead, scheduling fence per slot).  Per 32-key unit u: SC(u) = S and dP
// (12 slots, operands = rows of K / V, the query fragments in registers), PV(u) = dQ += dS K (6 slots, transposed reads of
// K).  Block order SC(0) SC(1) PV(0) SC(2) PV(1) ... SC(KT-1) PV(KT-2) PV(KT-1); dS(u) = P (dP - delta) rides in two-register
// pieces on the block after SC(u): dS(0) on SC(1) (one piece per slot), dS(u >= 1) on PV(u - 1) (8 pieces on 6 slots).
// KT = 2: 64-key tiles, two workgroups per CU.  KT = 4: 128-key tiles, one workgroup per CU (106 KB of LDS).
template <int KT>
#ifndef CSTS_ATTN_DQ_WAVES
#define CSTS_ATTN_DQ_WAVES 2      // waves per SIMD attn_dq_fast_kernel<2> is compiled for (A/B build: 3)
#endif
__global__ __launch_bounds__(256, (KT == 2 ? CSTS_ATTN_DQ_WAVES : 1)) void attn_dq_fast_kernel(AttnP p) {
  constexpr int HD = 96;
  typedef Cfg<HD, false> C;
  constexpr int KBLK = 32 * KT, LD = C::LD_ROW;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  bf16* smem = reinterpret_cast<bf16*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  int BX, BY, BZ;
  xcd_remap3(BX, BY, BZ);                // XCD-contiguous workgroup order: see xcd_remap3
  const int b = BZ, head = BY;
  const int qraw = BX * 128 + w * 32 + (lane & 31);
  const bool qvalid = qraw < p.Nq;
  const int qi = qvalid ? qraw : p.Nq - 1;

  RowFrag<HD, false> qf, dof;
  float dl = 0.f;
  {
    // all three row tiles are requested at once (ONE memory round trip instead of three in a row), then pass through the
    // two LDS staging regions
    RowFrag<HD, false> of;
    TileStage<HD, 128> tq, tdo, to;
    tq.gload(p.Q, (int64_t)b * p.q_bs + (int64_t)head * p.q_hs, p.q_ts, BX * 128, p.Nq, tid);
    tdo.gload(p.dO, (int64_t)b * p.do_bs + (int64_t)head * p.do_hs, p.do_ts, BX * 128, p.Nq, tid);
    to.gload(p.O, (int64_t)b * p.o_bs + (int64_t)head * p.o_hs, p.o_ts, BX * 128, p.Nq, tid);
    tq.lstore(smem, LD, tid);
    tdo.lstore(smem + 128 * LD, LD, tid);
    __syncthreads();
    frag_from_lds<HD>(qf, smem, w * 32 + (lane & 31), h);
    frag_from_lds<HD>(dof, smem + 128 * LD, w * 32 + (lane & 31), h);
    __syncthreads();
    to.lstore(smem, LD, tid);
    __syncthreads();
    frag_from_lds<HD>(of, smem, w * 32 + (lane & 31), h);
    __syncthreads();
    // delta = rowsum(dO * O): kept for the dK/dV kernel, which runs after this one
#pragma unroll
    for (int s2 = 0; s2 < HD / 16; ++s2)
#pragma unroll
      for (int j2 = 0; j2 < 8; ++j2) dl += (float)dof.b[s2][j2] * (float)of.b[s2][j2];
    dl += __shfl_xor(dl, 32, 64);
    if (h == 0 && qvalid) p.delta[((int64_t)b * p.H + head) * p.Nq + qi] = dl;
  }
  const float negL = -p.LSE[((int64_t)b * p.H + head) * p.Nq + qi];
  f32x16 acc[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  const int64_t kbase = (int64_t)b * p.k_bs + (int64_t)head * p.k_hs;
  const int64_t vbase = (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;

  tile_loop_dma<HD, KBLK, LD, false>(smem, p.K, kbase, p.k_ts, p.V, vbase, p.v_ts, 0, p.Nk, tid,
                                     [&](const bf16* Ks, const bf16* Vs, int) {
    constexpr int NS = HD / 16, ND = HD / 32, SCB = 2 * NS, PVB = 2 * ND, DEPTH = 4, NBLK = 2 * KT, TOTAL = KT * (SCB + PVB);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int g1 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
    const bf16* rowK = Ks + (lane & 31) * LD + 8 * h;
    const bf16* rowV = Vs + (lane & 31) * LD + 8 * h;
    const bf16* trK = Ks + (4 * h + qq) * LD + 16 * g1 + 4 * pp;
    // block bi of the order above: is it a PV block, which unit, how many slots, its first slot
    auto blk_pv = [](int bi) constexpr { return bi == NBLK - 1 || (bi >= 2 && (bi & 1) == 0); };
    auto blk_unit = [](int bi) constexpr { return bi == 0 ? 0 : bi == NBLK - 1 ? KT - 1 : (bi & 1) ? (bi + 1) / 2 : bi / 2 - 1; };
    auto blk_of = [=](int g) constexpr { int bi = 0, s0 = 0; for (;; ++bi) { const int n = blk_pv(bi) ? PVB : SCB; if (g < s0 + n) return bi; s0 += n; } };
    auto blk_start = [=](int bi) constexpr { int s0 = 0; for (int i = 0; i < bi; ++i) s0 += blk_pv(i) ? PVB : SCB; return s0; };
    bf16x8 ring[DEPTH + 1];
    auto request = [&](auto gc) {
      constexpr int g = decltype(gc)::value, bi = blk_of(g), u = blk_unit(bi), k = g - blk_start(bi);
      bf16x8& dst = ring[g % (DEPTH + 1)];
      if constexpr (!blk_pv(bi)) {
        dst = *reinterpret_cast<const bf16x8*>((k < NS ? rowK : rowV) + u * 32 * LD + 16 * (k % NS));
      } else {
        constexpr int s2 = k / ND, d = k % ND;
        const bf16* Yb = trK + (u * 32 + 16 * s2) * LD + 32 * d;
        const bf16x4 b0 = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb)));
        const bf16x4 b1 = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb + 8 * LD)));
        dst[0] = b0[0]; dst[1] = b0[1]; dst[2] = b0[2]; dst[3] = b0[3];
        dst[4] = b1[0]; dst[5] = b1[1]; dst[6] = b1[2]; dst[7] = b1[3];
      }
    };
    static_for<DEPTH>([&](auto gc) { request(gc); });
    f32x16 S[2], dP[2];
    bf16x8 bp[2][2];
    // dS of registers 2 q, 2 q + 1 of unit su (S / dP / bp double-buffered by unit parity)
    auto piece = [&](auto suc, auto qc) {
      constexpr int sp = decltype(suc)::value & 1, q = decltype(qc)::value;
#pragma unroll
      for (int r = 2 * q; r < 2 * q + 2; ++r) {
        const float ds = __builtin_amdgcn_exp2f(__builtin_fmaf(S[sp][r], p.scale_log2, negL)) * (dP[sp][r] - dl);
        bp[sp][r / 8][r % 8] = (bf16)ds;
      }
    };
    static_for<TOTAL>([&](auto gc) {
      constexpr int g = decltype(gc)::value, bi = blk_of(g), u = blk_unit(bi), k = g - blk_start(bi), par = u & 1;
      constexpr bool is_pv = blk_pv(bi);
      if constexpr (!is_pv && k == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[par][r] = 0.f; dP[par][r] = 0.f; }
      }
      if constexpr (g + DEPTH < TOTAL) request(std::integral_constant<int, g + DEPTH>{});
      const bf16x8 a = ring[g % (DEPTH + 1)];
      if constexpr (!is_pv) {
        if constexpr (k < NS) S[par] = CSTS_MFMA16(a, qf.b[k], S[par], 0, 0, 0);
        else dP[par] = CSTS_MFMA16(a, dof.b[k - NS], dP[par], 0, 0, 0);
      } else {
        constexpr int s2 = k / ND, d = k % ND;
        acc[d] = CSTS_MFMA16(a, bp[par][s2], acc[d], 0, 0, 0);
      }
      // the dS pieces of the unit whose SC block came just before this block
      if constexpr (bi == 1) {                                       // dS(0) under SC(1): one piece per slot
        if constexpr (k < 8) piece(std::integral_constant<int, 0>{}, std::integral_constant<int, k>{});
      } else if constexpr (bi >= 2 && (bi & 1) == 0) {               // dS(bi / 2) under PV(bi / 2 - 1): 8 pieces on 6 slots
        constexpr int su = bi / 2;
        if constexpr (k < 4) piece(std::integral_constant<int, su>{}, std::integral_constant<int, k>{});
        else {
          piece(std::integral_constant<int, su>{}, std::integral_constant<int, 4 + 2 * (k - 4)>{});
          piece(std::integral_constant<int, su>{}, std::integral_constant<int, 5 + 2 * (k - 4)>{});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  stage_rows_out<HD>(smem, p.dQ, (int64_t)b * p.dq_bs + (int64_t)head * p.dq_hs, p.dq_ts, BX * 128, p.Nq, acc, p.scale,
                     w * 32 + (lane & 31), h, tid);
}

// ---------------------------------------------------------------------------------------------- dK, dV
// (a 256-register budget for two workgroups per CU, as attn_fwd / attn_dq have, was measured 2-4 % slower here: this
// kernel carries dK, dV, the key fragments and the score tiles and spills)
// SLOW: frame mask and / or a ragged last query tile (N_q not a multiple of the tile), as per-element tests.  A kernel-level
// switch on purpose: with both forms of the tile body inside one loop the register allocator split the dK / dV accumulators
// at the join and copied 96 registers between the VGPR and AGPR halves around the MFMAs of every tile.
// FAST (hd 96, bf16, no mask, every 128-query tile whole): tiles by LDS-DMA and the software-pipelined MFMA stream below.
// (Two 32-key groups per wave -- every LDS operand feeding two MFMAs -- was built and measured: the second group's
// accumulators do not fit the 256 AGPRs beside S / dP and were copied to and from VGPRs around every product; no gain.)
template <int HD, bool F32, bool SLOW>
__global__ __launch_bounds__(256) void attn_dkv_kernel(AttnP p) {
  typedef typename El<F32>::T T;
  typedef Cfg<HD, F32> C;
  constexpr bool FAST = !F32 && HD == 96 && !SLOW;
  constexpr bool SWZ = FAST && (CSTS_DKV_SWZ != 0);                  // swizzled 256-byte LDS rows (tile_loop_dma)
  constexpr int LDF = SWZ ? 128 : C::LD_ROW;                         // LDS row stride of the staged Q / dO tiles
  constexpr int QBLK = FAST ? 128 : C::KVBLK, QT = QBLK / 32;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  int BX, BY, BZ;
  xcd_remap3(BX, BY, BZ);                // XCD-contiguous workgroup order: see xcd_remap3
  const int b = BZ / p.H, head = BZ % p.H;
  const int split = BY;
  const int kraw = BX * 128 + w * 32 + (lane & 31);
  const bool kvalid = kraw < p.Nk;
  const int ki = kvalid ? kraw : p.Nk - 1;

  RowFrag<HD, F32> kf, vf;
  kf.load(p.K, (int64_t)b * p.k_bs + (int64_t)ki * p.k_ts + (int64_t)head * p.k_hs, h);
  vf.load(p.V, (int64_t)b * p.v_bs + (int64_t)ki * p.v_ts + (int64_t)head * p.v_hs, h);
  f32x16 dK[HD / 32], dV[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dK[d][r] = 0.f; dV[d][r] = 0.f; }
  const int64_t qbase = (int64_t)b * p.q_bs + (int64_t)head * p.q_hs;
  const int64_t dobase = (int64_t)b * p.do_bs + (int64_t)head * p.do_hs;
  const int qbeg = split * p.q_chunk, qend = min(p.Nq, qbeg + p.q_chunk);
  const float* Lrow = p.LSE + ((int64_t)b * p.H + head) * p.Nq;     // per-query statistics: staged with the tiles
  const float* Drow = p.delta + ((int64_t)b * p.H + head) * p.Nq;
  const int fk = frame_of(p, ki);
#ifdef CSTS_ATTN_STAMPS
  const bool stamp_on = tid == 0 && BX == 0 && BY == 0 && BZ == 0;
  int nstamp = 1;
  const unsigned long long rt0 = (unsigned long long)__builtin_amdgcn_s_memrealtime();
#endif
  AT_STAMP();

  auto tile_body = [&](const T* Qs, const T* dOs, int q0, const float* Ls, auto&& dma) {
    const float* Ds = Ls + QBLK;
    AT_STAMP();                                // tile body begins
    // P = exp2(S * scale - LSE), dS = P * (dP - delta) for registers r0 .. r0 + n - 1 (n <= 4, inside one quad: the 4
    // registers of a quad are 4 consecutive queries -> one 16-byte broadcast LDS read of LSE / delta)
    auto softmax_bwd = [&](f32x16& S, f32x16& dP, int qt, int r0, int n) {
      const int j = r0 >> 2, qo = qt * 32 + 8 * j + 4 * h;
      const float4 L4 = *reinterpret_cast<const float4*>(Ls + qo), D4 = *reinterpret_cast<const float4*>(Ds + qo);
      const float l[4] = {L4.x, L4.y, L4.z, L4.w}, dd[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
      for (int r = r0; r < r0 + n; ++r) {
        const int i = r & 3;
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], p.scale_log2, -l[i]));
        float pr = e;
        if constexpr (SLOW) {
          // clamped (invalid) key lanes are never stored, so they need no masking
          const int q = q0 + qo + i;
          const bool dead = (q >= qend) | ((p.mask_mode != 0) & (frame_of(p, q) != fk));
          pr = dead ? 0.f : e;
        }
        S[r] = pr;
        dP[r] = pr * (dP[r] - dd[i]);
      }
    };
    if constexpr (F32 || HD != 96) {          // (hd 192: dK, dV and the key fragments alone are 288 registers)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x16 S, dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
        score<HD, C::LD_ROW, F32>(S, Qs + qt * 32 * C::LD_ROW, kf, lane);
        score<HD, C::LD_ROW, F32>(dP, dOs + qt * 32 * C::LD_ROW, vf, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) softmax_bwd(S, dP, qt, 4 * j, 4);
        pv<HD, C::LD_ROW, F32>(dV, dOs + qt * 32 * C::LD_ROW, S, lane);
        pv<HD, C::LD_ROW, F32>(dK, Qs + qt * 32 * C::LD_ROW, dP, lane);
      }
    } else {
      // hd 96 / bf16: the tile's MFMAs as ONE stream of slots, each with one LDS operand (16 bytes per lane: a row chunk
      // for the S / dP products, two transposed 8-byte reads for the dV / dK products).  The workgroup is one wave per
      // SIMD, so nothing hides a latency unless the instruction stream does:
      //  * the operand of slot g + DEPTH is requested before the MFMA of slot g is issued and a scheduling fence after
      //    every slot keeps it that way (with the read next to its MFMA -- what the compiler schedules on its own -- every
      //    MFMA waited a whole LDS round trip and the matrix pipe was busy 18 % of the time);
      //  * blocks of 12 slots are ordered SC(0) SC(1) PV(0) SC(2) PV(1) ... SC(QT-1) PV(QT-2) PV(QT-1)  (SC(u): S and dP of
      //    the 32-query unit u, PV(u): its dV and dK products).  S / dP and their bf16 forms are double-buffered by unit parity;
      //  * round 5: the softmax backward of unit u is 16 HALF-pieces of two registers -- (a) P = exp2(S c - LSE), (b) dS = P (dP -
      //    delta) and the two bf16 packs -- spread evenly over every slot between the end of SC(u) and the start of PV(u): the
      //    block after SC(u) for the first and the last unit, two blocks (PV(u-1), SC(u+1)) for the others.  A half-piece is
      //    ~25 cycles of vector issue and hides under one 32-cycle MFMA; the 8 two-register pieces on 8 of 12 slots of round 3
      //    were ~60 cycles each, and their LSE / delta came from LDS in the slot that consumed them (one exposed LDS round
      //    trip per piece: the in-loop rate was 70 cycles per MFMA, not the 35 the round-3 notes claim -- their count of
      //    MFMAs per tile was off by two).  LSE / delta of a unit are now read into registers under its own SC block;
      //  * the next tile's LDS-DMA instructions are issued one per slot under SC(0), which carries no softmax.
      constexpr int LD = LDF, NS = HD / 16, ND = HD / 32, BLK = 2 * NS, DEPTH = CSTS_DKV_DEPTH, NBLK = 2 * QT, TOTAL = BLK * NBLK;
      static_assert(BLK == 4 * ND, "S/dP and dV/dK blocks have the same number of slots");
      static_assert(QT >= 2, "the block order needs two units");
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const int g1 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
      constexpr int DOFF = QBLK * LD;          // the dO image follows the Q image
      // element offsets of this lane's operands inside a 32-query unit of the Q image (the dO image: + DOFF; unit u: + u * 32 * LD;
      // second half of a transposed step: + 16 * LD).  Padded rows: one base + compile-time chunk offsets.  Swizzled rows: the
      // chunk position depends on the row, so every (chunk | d, read) has its own offset (12 registers instead of 4).
      int rowo[NS], tro[ND][2];
#pragma unroll
      for (int k = 0; k < NS; ++k)
        rowo[k] = (lane & 31) * LD + (SWZ ? (((2 * k + h) ^ swz16(lane & 15)) * 8) : (16 * k + 8 * h));
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const int row = 4 * h + qq + 8 * rd;
          tro[d][rd] = row * LD + (SWZ ? ((((4 * d + 2 * g1 + (pp >> 1)) ^ swz16(row)) * 8) + 4 * (pp & 1)) : (32 * d + 16 * g1 + 4 * pp));
        }
      bf16x8 ring[DEPTH + 1];
      // block bi -> (is it a dV/dK block, which unit)
      auto blk_pv = [](int bi) constexpr { return bi == NBLK - 1 || (bi >= 2 && (bi & 1) == 0); };
      auto blk_unit = [](int bi) constexpr { return bi == 0 ? 0 : bi == NBLK - 1 ? QT - 1 : (bi & 1) ? (bi + 1) / 2 : bi / 2 - 1; };
      // slot of half-piece j (0 .. 15) of unit u: its window starts one slot after SC(u) ends (the first read of S does not wait
      // for the accumulator of the MFMA issued just before it) and ends where PV(u) starts
      auto hp_slot = [](int u, int j) constexpr {
        const int w0 = (u == 0 ? BLK : u == QT - 1 ? BLK * (NBLK - 2) : BLK * 2 * u) + 1;
        const int n = ((u == 0 || u == QT - 1) ? BLK : 2 * BLK) - 1;
        return w0 + j * n / 16;
      };
      auto request = [&](auto gc) {
        constexpr int g = decltype(gc)::value, bi = g / BLK, k = g % BLK;
        constexpr bool is_pv = blk_pv(bi);
        constexpr int u = blk_unit(bi);
        bf16x8& dst = ring[g % (DEPTH + 1)];
        if constexpr (!is_pv) {
          dst = *reinterpret_cast<const bf16x8*>(Qs + rowo[k % NS] + (k < NS ? 0 : DOFF) + u * 32 * LD);
        } else {
          constexpr int idx = k % (2 * ND), s2 = idx / ND, d = idx % ND;
          const bf16* Yb = Qs + (k < 2 * ND ? DOFF : 0) + (u * 32 + 16 * s2) * LD;
          const bf16x4 b0 = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb + tro[d][0])));
          const bf16x4 b1 = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb + tro[d][1])));
          dst[0] = b0[0]; dst[1] = b0[1]; dst[2] = b0[2]; dst[3] = b0[3];
          dst[4] = b1[0]; dst[5] = b1[1]; dst[6] = b1[2]; dst[7] = b1[3];
        }
      };
      static_for<DEPTH>([&](auto gc) { request(gc); });
      f32x16 S[2], dP[2];
      bf16x8 bpS[2][2], bpD[2][2];
      float4 Lq[2][4], Dq[2][4];                // LSE / delta of the unit's 4 register quads (4 consecutive queries each)
      auto comp = [](const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; };
      auto half_piece = [&](auto uc, auto jc) {
        constexpr int u = decltype(uc)::value, j = decltype(jc)::value, sp = u & 1, r0 = 2 * (j / 2), quad = r0 >> 2;
        if constexpr ((j & 1) == 0) {
#pragma unroll
          for (int r = r0; r < r0 + 2; ++r) {
            float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[sp][r], p.scale_log2, -comp(Lq[sp][quad], r & 3)));
            if constexpr (SLOW) {
              // clamped (invalid) key lanes are never stored, so they need no masking
              const int q = q0 + u * 32 + 8 * quad + 4 * h + (r & 3);
              const bool dead = (q >= qend) | ((p.mask_mode != 0) & (frame_of(p, q) != fk));
              e = dead ? 0.f : e;
            }
            S[sp][r] = e;
          }
        } else {
#pragma unroll
          for (int r = r0; r < r0 + 2; ++r) dP[sp][r] = S[sp][r] * (dP[sp][r] - comp(Dq[sp][quad], r & 3));
          constexpr int s2 = r0 / 8, e0 = r0 % 8;
          bpS[sp][s2][e0] = (bf16)S[sp][r0]; bpS[sp][s2][e0 + 1] = (bf16)S[sp][r0 + 1];
          bpD[sp][s2][e0] = (bf16)dP[sp][r0]; bpD[sp][s2][e0 + 1] = (bf16)dP[sp][r0 + 1];
        }
      };
      constexpr int NDMA = tile_dma_count<QBLK, LD>();
      static_for<TOTAL>([&](auto gc) {
        constexpr int g = decltype(gc)::value, bi = g / BLK, k = g % BLK;
        constexpr bool is_pv = blk_pv(bi);
        constexpr int u = blk_unit(bi), par = u & 1;
        if constexpr (k == 0) { AT_STAMP(); }   // diagnostics build only: one stamp per 12-slot block
        if constexpr (!is_pv && k == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { S[par][r] = 0.f; dP[par][r] = 0.f; }
        }
        if constexpr (!is_pv && k < 8) {        // this unit's statistics: one 16-byte broadcast read per slot
          const int qo = u * 32 + 8 * (k & 3) + 4 * h;
          if constexpr (k < 4) Lq[par][k] = *reinterpret_cast<const float4*>(Ls + qo);
          else Dq[par][k - 4] = *reinterpret_cast<const float4*>(Ds + qo);
        }
#ifndef CSTS_DKV_DIAG_NOREAD                // (timing diagnostics only, wrong results: the stream without its LDS operand reads)
        if constexpr (g + DEPTH < TOTAL) request(std::integral_constant<int, g + DEPTH>{});
#endif
        const bf16x8 a = ring[g % (DEPTH + 1)];
        if constexpr (!is_pv) {
          if constexpr (k < NS) S[par] = CSTS_MFMA16(a, kf.b[k], S[par], 0, 0, 0);
          else dP[par] = CSTS_MFMA16(a, vf.b[k - NS], dP[par], 0, 0, 0);
        } else {
          constexpr int idx = k % (2 * ND), s2 = idx / ND, d = idx % ND;
#if defined(CSTS_DKV_DIAG_BFRAG) || defined(CSTS_DKV_DIAG_BZERO)
          asm volatile("" ::"v"(bpS[par][s2]), "v"(bpD[par][s2]));     // keeps the softmax backward alive without consuming it
#endif
#ifdef CSTS_DKV_DIAG_BFRAG                  // (timing diagnostics only: real data in B without any dependence on the softmax)
          if constexpr (k < 2 * ND) dV[d] = CSTS_MFMA16(a, kf.b[s2 + 2 * par], dV[d], 0, 0, 0);
          else dK[d] = CSTS_MFMA16(a, vf.b[s2 + 2 * par], dK[d], 0, 0, 0);
#elif defined(CSTS_DKV_DIAG_BZERO)          // (timing diagnostics only: zeros in B)
          bf16x8 zz;
#pragma unroll
          for (int e = 0; e < 8; ++e) zz[e] = (bf16)0.f;
          asm volatile("" : "+v"(zz));
          if constexpr (k < 2 * ND) dV[d] = CSTS_MFMA16(a, zz, dV[d], 0, 0, 0);
          else dK[d] = CSTS_MFMA16(a, zz, dK[d], 0, 0, 0);
#else
          if constexpr (k < 2 * ND) dV[d] = CSTS_MFMA16(a, bpS[par][s2], dV[d], 0, 0, 0);
          else dK[d] = CSTS_MFMA16(a, bpD[par][s2], dK[d], 0, 0, 0);
#endif
        }
#ifdef CSTS_DKV_DIAG_NOSM
        if constexpr (!is_pv && k == BLK - 1) asm volatile("" ::"v"(S[par]), "v"(dP[par]));     // S / dP stay alive (their MFMAs are not dead code)
#endif
#ifndef CSTS_DKV_DIAG_NOSM                  // (timing diagnostics only, wrong results: the stream without its softmax backward)
        // the half-pieces of the softmax backward that live on this slot
        static_for<QT * 16>([&](auto ic) {
          constexpr int i = decltype(ic)::value, su = i / 16, j = i % 16;
          if constexpr (hp_slot(su, j) == g) half_piece(std::integral_constant<int, su>{}, std::integral_constant<int, j>{});
        });
#endif
#ifndef CSTS_DKV_DIAG_NODMA                 // (timing diagnostics only: the next tile is not fetched)
        // the next tile's LDS-DMA instructions: under SC(0) (and the first slots of SC(1) when there are more than 12)
        if constexpr (g < NDMA) dma(std::integral_constant<int, g>{});
#endif
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    AT_STAMP();                                // tile body done
  };
  if constexpr (FAST)
    tile_loop_dma<HD, QBLK, LDF, true, LDF, true, SWZ>(smem, p.Q, qbase, p.q_ts, p.dO, dobase, p.do_ts, qbeg, qend, tid, tile_body,
                                                       Lrow, Drow AT_STAMP_ARGS);
  else
    tile_loop<HD, QBLK, C::LD_ROW, C::LD_ROW, F32, true>(smem, p.Q, qbase, p.q_ts, p.dO, dobase, p.do_ts, qbeg, qend, tid,
                                                         [&](const T* Qs, const T* dOs, int q0, const float* Ls) {
                                                           tile_body(Qs, dOs, q0, Ls, [](auto) {});
                                                         }, Lrow, Drow AT_STAMP_ARGS);
  AT_STAMP();
#ifdef CSTS_ATTN_STAMPS
  if (stamp_on) {
    g_attn_stamps[0] = nstamp;
    g_attn_stamps[4095] = (unsigned long long)__builtin_amdgcn_s_memrealtime() - rt0;     // 100 MHz ticks over the kernel body
  }
#endif
  if (!kvalid) return;
  if (p.nsplit == 1) {
    store_rows<HD>(p.dK, p.dt, (int64_t)b * p.dk_bs + (int64_t)ki * p.dk_ts + (int64_t)head * p.dk_hs, dK, p.scale, h);
    store_rows<HD>(p.dV, p.dt, (int64_t)b * p.dv_bs + (int64_t)ki * p.dv_ts + (int64_t)head * p.dv_hs, dV, 1.f, h);
  } else {
    // ws layout: [2 (dK,dV)][split][B][H][Nk][HD] fp32
    const int64_t per = (int64_t)p.B * p.H * p.Nk * HD;
    const int64_t off = (((int64_t)split * p.B + b) * p.H + head) * (int64_t)p.Nk * HD + (int64_t)ki * HD;
    store_rows<HD>(p.ws, CSTS_F32, off, dK, p.scale, h);
    store_rows<HD>(p.ws, CSTS_F32, (int64_t)p.nsplit * per + off, dV, 1.f, h);
  }
}

// ---------------------------------------------------------------------------------------------- dQ, dK, dV in ONE pass
// Short key sequences (N_k <= 128: the decoder blocks, whose K/V are pooled down to 128 tokens while the queries are the
// 32 k .. 64 k upsampled tokens).  The two-kernel form streams Q and dO from HBM twice (once per kernel) for a handful
// of MFMAs per byte; here K and V live in LDS for the whole workgroup, every 128-query tile of Q / dO is staged ONCE and
// used for both products:
//   part A (= attn_dkv_kernel's body): wave w owns keys 32 w .. 32 w + 31 as row fragments and walks the tile's four
//           32-query units: S, dP with the queries in the accumulator registers -> dV += P^T dO, dK += dS^T Q;
//   part B (= attn_dq_kernel's body):  wave w owns queries 32 w .. 32 w + 31 of the tile as row fragments (read back
//           from the staged tile) and walks the four 32-key units of K / V in LDS -> dQ = dS K, staged out through this
//           wave's rows of the O tile as coalesced 16-byte stores BEFORE part A (they drain under its MFMAs).  The O tile is staged beside dO, so delta = rowsum(dO * O)
//           is a dot product of two row fragments here (no separate delta launch, no second read of dO); LSE and
//           delta of the tile sit in LDS for part A.
// S / dP are computed in both orientations (MFMA time is free here: the kernel is bound by the Q / dO / dQ streams).
// dK / dV partials of the query splits go to the same workspace layout attn_dkv_reduce_kernel sums.  bf16, no mask.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_fused_kernel(AttnP p) {
  typedef Cfg<HD, false> C;
  constexpr int LD = C::LD_ROW, TILE = 128 * LD;
  __shared__ __attribute__((aligned(16))) bf16 smem[5 * TILE + 512];
  bf16* Ks = smem;
  bf16* Vs = smem + TILE;
  bf16* Qs = smem + 2 * TILE;
  bf16* dOs = smem + 3 * TILE;
  bf16* Os = smem + 4 * TILE;
  float* Ls = reinterpret_cast<float*>(smem + 5 * TILE);   // LSE and delta of the tile's 128 queries
  float* Ds = Ls + 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5;
  int BX, BY, BZ;
  xcd_remap3(BX, BY, BZ);                // XCD-contiguous workgroup order: see xcd_remap3
  const int b = BY / p.H, head = BY % p.H;
  const int split = BX;
  const int qbeg = split * p.q_chunk, qend = min(p.Nq, qbeg + p.q_chunk);
  const int64_t kbase = (int64_t)b * p.k_bs + (int64_t)head * p.k_hs, vbase = (int64_t)b * p.v_bs + (int64_t)head * p.v_hs;
  const int64_t qbase = (int64_t)b * p.q_bs + (int64_t)head * p.q_hs, dobase = (int64_t)b * p.do_bs + (int64_t)head * p.do_hs;
  const int64_t obase = (int64_t)b * p.o_bs + (int64_t)head * p.o_hs;
  {                                         // K and V: all chunks requested at once (rows >= Nk are zero)
    TileStage<HD, 128> sk, sv;
    sk.gload(p.K, kbase, p.k_ts, 0, p.Nk, tid);
    sv.gload(p.V, vbase, p.v_ts, 0, p.Nk, tid);
    sk.lstore(Ks, LD, tid);
    sv.lstore(Vs, LD, tid);
  }

  const int kraw = w * 32 + (lane & 31);
  const bool kvalid = kraw < p.Nk;
  const int ki = kvalid ? kraw : p.Nk - 1;
  RowFrag<HD, false> kf, vf;
  kf.load(p.K, kbase + (int64_t)ki * p.k_ts, h);
  vf.load(p.V, vbase + (int64_t)ki * p.v_ts, h);
  f32x16 dK[HD / 32], dV[HD / 32];
#pragma unroll
  for (int d = 0; d < HD / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dK[d][r] = 0.f; dV[d][r] = 0.f; }
  const float* Lrow = p.LSE + ((int64_t)b * p.H + head) * p.Nq;

  TileStage<HD, 128> sq, sdo, so;
  float lnext = 0.f;
  sq.gload(p.Q, qbase, p.q_ts, qbeg, qend, tid);
  sdo.gload(p.dO, dobase, p.do_ts, qbeg, qend, tid);
  so.gload(p.O, obase, p.o_ts, qbeg, qend, tid);
  if (tid < 128) lnext = Lrow[min(qbeg + tid, qend - 1)];
  for (int q0 = qbeg; q0 < qend; q0 += 128) {
    __syncthreads();                       // everyone is done with the previous tile (Q, dO, stats; its dQ has left the O buffer)
    sq.lstore(Qs, LD, tid);
    sdo.lstore(dOs, LD, tid);
    so.lstore(Os, LD, tid);
    if (tid < 128) Ls[tid] = lnext;
    if (q0 + 128 < qend) {                 // the next tile's HBM round trip runs under this tile's MFMAs
      sq.gload(p.Q, qbase, p.q_ts, q0 + 128, qend, tid);
      sdo.gload(p.dO, dobase, p.do_ts, q0 + 128, qend, tid);
      so.gload(p.O, obase, p.o_ts, q0 + 128, qend, tid);
      if (tid < 128) lnext = Lrow[min(q0 + 128 + tid, qend - 1)];
    }
    __syncthreads();
    // ---- part B: this wave's 32 queries (row fragments) against all keys -> dQ; delta = rowsum(dO * O) on the way
    f32x16 acc[HD / 32];
#pragma unroll
    for (int d = 0; d < HD / 32; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    {
      const int qrow = w * 32 + (lane & 31);
      RowFrag<HD, false> qf, dof;
      frag_from_lds<HD>(qf, Qs, qrow, h);
      frag_from_lds<HD>(dof, dOs, qrow, h);
      float dl = 0.f;
      {
        RowFrag<HD, false> of;
        frag_from_lds<HD>(of, Os, qrow, h);
#pragma unroll
        for (int s2 = 0; s2 < HD / 16; ++s2)
#pragma unroll
          for (int j2 = 0; j2 < 8; ++j2) dl += (float)dof.b[s2][j2] * (float)of.b[s2][j2];
      }
      dl += __shfl_xor(dl, 32, 64);          // the two half-waves hold the two halves of every 16-wide chunk of the row
      if (h == 0) Ds[qrow] = dl;
      const float negL = -Ls[qrow];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        if (kt * 32 >= p.Nk) break;          // wave-uniform
        f32x16 S, dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
        score<HD, LD, false>(S, Ks + kt * 32 * LD, qf, lane);
        score<HD, LD, false>(dP, Vs + kt * 32 * LD, dof, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], p.scale_log2, negL));
          const float pr = (kt * 32 + rowoff(r, h) >= p.Nk) ? 0.f : e;
          S[r] = pr * (dP[r] - dl);
        }
        pv<HD, LD, false>(acc, Ks + kt * 32 * LD, S, lane);
      }
    }
    // dQ leaves NOW, through this wave's rows of the O tile (read for delta above and not needed again): the stores then
    // drain under part A.  Issued at the end of the tile they were the youngest entries of the in-order memory counter
    // and the wait for the prefetched next tile had to sit through their whole latency.
    stage_rows_out_wave<HD>(Os, p.dQ, (int64_t)b * p.dq_bs + (int64_t)head * p.dq_hs, p.dq_ts, q0, qend, acc, p.scale, w, lane);
    __syncthreads();                       // every wave's delta is in LDS
    // ---- part A: this wave's 32 keys (row fragments) against the tile's 128 queries -> dK, dV
    const bool ragged = q0 + 128 > qend;   // wave-uniform
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x16 S, dP;
#pragma unroll
      for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
      score<HD, LD, false>(S, Qs + qt * 32 * LD, kf, lane);
      score<HD, LD, false>(dP, dOs + qt * 32 * LD, vf, lane);
      // the 4 registers of a quad are 4 consecutive queries: one 16-byte LDS read of LSE / delta each
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qo = qt * 32 + 8 * j + 4 * h;
        const float4 L4 = *reinterpret_cast<const float4*>(Ls + qo), D4 = *reinterpret_cast<const float4*>(Ds + qo);
        const float l[4] = {L4.x, L4.y, L4.z, L4.w}, dd[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[4 * j + i], p.scale_log2, -l[i]));
          const float pr = (ragged && q0 + qo + i >= qend) ? 0.f : e;
          S[4 * j + i] = pr;
          dP[4 * j + i] = pr * (dP[4 * j + i] - dd[i]);
        }
      }
      pv<HD, LD, false>(dV, dOs + qt * 32 * LD, S, lane);
      pv<HD, LD, false>(dK, Qs + qt * 32 * LD, dP, lane);
    }
  }
  if (!kvalid) return;
  if (p.nsplit == 1) {
    store_rows<HD>(p.dK, p.dt, (int64_t)b * p.dk_bs + (int64_t)ki * p.dk_ts + (int64_t)head * p.dk_hs, dK, p.scale, h);
    store_rows<HD>(p.dV, p.dt, (int64_t)b * p.dv_bs + (int64_t)ki * p.dv_ts + (int64_t)head * p.dv_hs, dV, 1.f, h);
  } else {
    const int64_t per = (int64_t)p.B * p.H * p.Nk * HD;
    const int64_t off = (((int64_t)split * p.B + b) * p.H + head) * (int64_t)p.Nk * HD + (int64_t)ki * HD;
    store_rows<HD>(p.ws, CSTS_F32, off, dK, p.scale, h);
    store_rows<HD>(p.ws, CSTS_F32, (int64_t)p.nsplit * per + off, dV, 1.f, h);
  }
}

#if CSTS_ATTN_PART == 0
// second pass: dK/dV[b, k, head, :] = sum_split ws[...]; a thread owns 8 consecutive channels of one key row (two 16-byte
// loads per split, one 16-byte bf16 store; the element-per-thread form spent its time in 64-bit index divisions)
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(AttnP p, int HD) {
  const int64_t per = (int64_t)p.B * p.H * p.Nk * HD;
  const int H8 = HD / 8;
  const int64_t total8 = 2 * per / 8;
  for (int64_t i8 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i8 < total8; i8 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = i8 * 8;
    const int which = idx >= per;
    int64_t e = idx - which * per;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < p.nsplit; ++sp) {            // splits in order: the sum does not depend on the launch geometry
      const float4* q = reinterpret_cast<const float4*>(p.ws + ((int64_t)which * p.nsplit + sp) * per + e);
      const float4 a = q[0], b4 = q[1];
      s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w; s[4] += b4.x; s[5] += b4.y; s[6] += b4.z; s[7] += b4.w;
    }
    int64_t r = e / 8;
    const int d = (int)(r % H8) * 8; r /= H8;
    const int k = (int)(r % p.Nk); r /= p.Nk;
    const int head = (int)(r % p.H);
    const int b = (int)(r / p.H);
    if (which == 0) st8_from_f32(p.dK, p.dt, (int64_t)b * p.dk_bs + (int64_t)k * p.dk_ts + (int64_t)head * p.dk_hs + d, s);
    else st8_from_f32(p.dV, p.dt, (int64_t)b * p.dv_bs + (int64_t)k * p.dv_ts + (int64_t)head * p.dv_hs + d, s);
  }
}

__global__ __launch_bounds__(256) void attn_probs_kernel(AttnP p, int HD, float* probs) {
  const int64_t total = (int64_t)p.B * p.H * p.Nq * p.Nk;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % p.Nk);
    int64_t e = idx / p.Nk;
    const int q = (int)(e % p.Nq); e /= p.Nq;
    const int head = (int)(e % p.H);
    const int b = (int)(e / p.H);
    float s = 0.f;
    const int64_t qo = (int64_t)b * p.q_bs + (int64_t)q * p.q_ts + (int64_t)head * p.q_hs;
    const int64_t ko = (int64_t)b * p.k_bs + (int64_t)k * p.k_ts + (int64_t)head * p.k_hs;
    for (int d = 0; d < HD; ++d) s += ld_as_f32(p.Q, p.dt, qo + d) * ld_as_f32(p.K, p.dt, ko + d);
    const float L = p.LSE[((int64_t)b * p.H + head) * p.Nq + q];
    probs[idx] = is_masked(p, q, k) ? 0.f : exp2f(s * p.scale_log2 - L);
  }
}

#endif
#if CSTS_ATTN_PART == 0
template <int HD, bool F32> size_t smem_fwd() {
  typedef Cfg<HD, F32> C;
  return (size_t)C::KVBLK * (C::LD_ROW + C::LD_TR) * (F32 ? 4 : 2 * 2);   // bf16: double-buffered
}
template <int HD, bool F32> size_t smem_bwd() {
  typedef Cfg<HD, F32> C;
  return (size_t)C::KVBLK * 2 * C::LD_ROW * (F32 ? 4 : 2 * 2);
}

int fill(const csts_attn_args* a, AttnP& p) {
  p.Q = a->Q; p.K = a->K; p.V = a->V; p.O = a->O; p.LSE = a->LSE; p.dO = a->dO; p.delta = a->delta;
  p.dQ = a->dQ; p.dK = a->dK; p.dV = a->dV; p.ws = nullptr;
  p.dt = a->dtype; p.B = a->B; p.H = a->H; p.Nq = a->Nq; p.Nk = a->Nk;
  p.q_bs = a->q_strides[0]; p.q_ts = a->q_strides[1]; p.q_hs = a->q_strides[2];
  p.k_bs = a->k_strides[0]; p.k_ts = a->k_strides[1]; p.k_hs = a->k_strides[2];
  p.v_bs = a->v_strides[0]; p.v_ts = a->v_strides[1]; p.v_hs = a->v_strides[2];
  p.o_bs = a->o_strides[0]; p.o_ts = a->o_strides[1]; p.o_hs = a->o_strides[2];
  p.do_bs = a->do_strides[0]; p.do_ts = a->do_strides[1]; p.do_hs = a->do_strides[2];
  p.dq_bs = a->dq_strides[0]; p.dq_ts = a->dq_strides[1]; p.dq_hs = a->dq_strides[2];
  p.dk_bs = a->dk_strides[0]; p.dk_ts = a->dk_strides[1]; p.dk_hs = a->dk_strides[2];
  p.dv_bs = a->dv_strides[0]; p.dv_ts = a->dv_strides[1]; p.dv_hs = a->dv_strides[2];
  p.scale = a->scale;
  p.scale_log2 = a->scale * 1.4426950408889634f;
  p.mask_mode = a->mask_mode; p.mask_T = a->mask_T; p.mask_HW = a->mask_HW;
  p.mask_inv_hw = (a->mask_mode != 0 && a->mask_HW > 0) ? 1.f / (float)a->mask_HW : 0.f;
  p.q_chunk = 0; p.nsplit = 1;
  return 0;
}

bool strides_ok(const int64_t* s, int dt) {
  const int a = dt == CSTS_F32 ? 4 : 8;
  return s[0] % a == 0 && s[1] % a == 0 && s[2] % a == 0;
}

// one-pass backward (attn_bwd_fused_kernel): bf16, head_dim 96, all keys in one LDS tile, no mask, enough queries to split
bool fused_bwd_ok(const csts_attn_args* a) {
  static const bool off = [] { const char* e = getenv("CSTS_ATTN_FUSED_BWD"); return e && atoi(e) == 0; }();
  return !off && a->dtype == CSTS_BF16 && a->head_dim == 96 && a->Nk <= 128 && a->mask_mode == 0 && a->Nq >= 1024;
}

// attn_dkv_kernel's FAST form: hd 96, bf16, no mask, every query tile whole (128 queries)
bool dkv_fast(int dt, int hd, int mask_mode, int Nq) {
  static const bool off = [] { const char* e = getenv("CSTS_ATTN_DKV_FAST"); return e && atoi(e) == 0; }();
  return !off && dt == CSTS_BF16 && hd == 96 && mask_mode == 0 && Nq % 128 == 0;
}

// attn_dq_fast_kernel: hd 96, bf16, no mask, every key tile whole.  Returns the 32-key units per tile (0: generic kernel).
// 2 (64-key tiles, two workgroups per CU) is the library's choice: 128-key tiles at one workgroup per CU measured 3-17 %
// slower per call (profiles/r3_attn_backward_ab.txt).  CSTS_ATTN_DQ_FAST = 0 / 4 forces the generic kernel / the 128-key form.
int dq_fast(int dt, int hd, int mask_mode, int Nk) {
  static const int force = [] { const char* e = getenv("CSTS_ATTN_DQ_FAST"); return e ? atoi(e) : -1; }();
  if (force == 0 || dt != CSTS_BF16 || hd != 96 || mask_mode != 0 || Nk % 64 != 0) return 0;
  return (force == 4 && Nk % 128 == 0) ? 4 : 2;
}

void dkv_plan(const csts_attn_args* a, int& nsplit, int& q_chunk) {
  if (fused_bwd_ok(a)) {      // one workgroup per CU (its LDS holds K, V and the Q / dO tile): aim at two rounds of the chip
    const int64_t want = std::max<int64_t>(1, std::min<int64_t>(256 / ((int64_t)a->B * a->H), cdiv(a->Nq, 256)));
    q_chunk = (int)(cdiv(cdiv(a->Nq, want), 128) * 128);
    nsplit = (int)cdiv(a->Nq, q_chunk);
    return;
  }
  const int qblk = dkv_fast(a->dtype, a->head_dim, a->mask_mode, a->Nq) ? 128 : a->head_dim == 96 ? 64 : 32;
  const int64_t base = cdiv(a->Nk, 128) * a->B * a->H;
  int64_t want = std::max<int64_t>(1, 256 / base);   // one workgroup per CU: the kernel runs one wave per SIMD (measured: 512 -> 76 us, 256 -> 66 us, 128 -> 82 us on the 2048 x 512 block)   // one workgroup per CU (the kernel runs one wave per SIMD)
  want = std::min<int64_t>(want, cdiv(a->Nq, 4 * qblk));
  want = std::max<int64_t>(want, 1);
  q_chunk = (int)(cdiv(cdiv(a->Nq, want), qblk) * qblk);
  nsplit = (int)cdiv(a->Nq, q_chunk);
}

}  // namespace

#define ATTN_COMMON_CHECKS(a)                                                                           \
  CSTS_REQUIRE((a) != nullptr, "null args");                                                            \
  CSTS_REQUIRE((a)->head_dim == 96 || (a)->head_dim == 192, "head_dim must be 96 or 192");              \
  CSTS_REQUIRE((a)->dtype == CSTS_F32 || (a)->dtype == CSTS_BF16, "bad dtype");                         \
  CSTS_REQUIRE((a)->B > 0 && (a)->H > 0 && (a)->Nq > 0 && (a)->Nk > 0, "empty problem");                \
  CSTS_REQUIRE((a)->Q && (a)->K && (a)->V, "null q/k/v");                                               \
  CSTS_REQUIRE(aligned16((a)->Q) && aligned16((a)->K) && aligned16((a)->V), "q/k/v must be 16-byte aligned"); \
  CSTS_REQUIRE(strides_ok((a)->q_strides, (a)->dtype) && strides_ok((a)->k_strides, (a)->dtype) &&      \
                   strides_ok((a)->v_strides, (a)->dtype),                                             \
               "strides must keep 16-byte row alignment");                                              \
  if ((a)->mask_mode == CSTS_MASK_SPATIAL)                                                              \
  CSTS_REQUIRE((a)->mask_HW > 0 && (a)->Nq < (1 << 20) && (a)->Nq == (a)->mask_T * (a)->mask_HW + (a)->mask_T && (a)->Nk == (a)->Nq, "spatial mask needs Nq == Nk == T*HW + T < 2^20")

enum { K_FWD = 0, K_DQ = 1, K_DKV = 2 };
template <int HD, bool F32>
static bool attn_launch(int which, const AttnP& p, dim3 grid, hipStream_t stream) {
  if (which == K_FWD) {
    const size_t sm = smem_fwd<HD, F32>();
    static const bool fast_off = [] { const char* e = getenv("CSTS_ATTN_FWD_FAST"); return e && atoi(e) == 0; }();
    if (!fast_off && !F32 && HD == 96 && p.mask_mode == 0 && p.Nk % 64 == 0)
      hipLaunchKernelGGL(attn_fwd_fast_kernel, grid, dim3(256), sm, stream, p);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<HD, F32>), grid, dim3(256), sm, stream, p);
  } else if (which == K_DQ) {
    const size_t sm = smem_bwd<HD, F32>();
    if (const int kt = dq_fast(p.dt, HD, p.mask_mode, p.Nk); kt == 4) {
      constexpr size_t sm4 = (size_t)2 * 128 * 2 * Cfg<96, false>::LD_ROW * 2;       // 128-key K and V tiles, double-buffered
      if (!csts_dyn_lds_optin(reinterpret_cast<const void*>(&attn_dq_fast_kernel<4>), (int)sm4)) return false;
      hipLaunchKernelGGL(attn_dq_fast_kernel<4>, grid, dim3(256), sm4, stream, p);
    } else if (kt == 2) hipLaunchKernelGGL(attn_dq_fast_kernel<2>, grid, dim3(256), sm, stream, p);
    else hipLaunchKernelGGL((attn_dq_kernel<HD, F32>), grid, dim3(256), sm, stream, p);
  } else {
    const size_t sm = smem_bwd<HD, F32>() + (size_t)Cfg<HD, F32>::KVBLK * 4 * sizeof(float);   // + LSE / delta of the tile(s)
    if (!dkv_fast(p.dt, HD, p.mask_mode, p.Nq)) {
      // masked / ragged shapes (and every fp32 call) take the per-element tests.  <96, false, false> IS the FAST form (128-query
      // tiles, its own LDS size): every other hd 96 / bf16 call takes the SLOW instantiation, whatever its shape
      const bool slow = F32 || (HD == 96) || p.mask_mode != 0 || p.Nq % Cfg<HD, F32>::KVBLK != 0;   // (q_chunk is a multiple of the tile)
      if (slow) hipLaunchKernelGGL((attn_dkv_kernel<HD, F32, true>), grid, dim3(256), sm, stream, p);
      else hipLaunchKernelGGL((attn_dkv_kernel<HD, F32, F32 || HD == 96>), grid, dim3(256), sm, stream, p);
    } else if constexpr (!F32 && HD == 96) {
      if (!csts_attn_dkv_fast_launch(p, grid, stream)) return false;      // part 1 of this source (see the top of the file)
    }
  }
  return true;
}
static bool attn_dispatch(int which, const csts_attn_args* a, const AttnP& p, dim3 grid, hipStream_t stream) {
  const bool f32 = a->dtype == CSTS_F32;
  if (a->head_dim == 96) return f32 ? attn_launch<96, true>(which, p, grid, stream) : attn_launch<96, false>(which, p, grid, stream);
  return f32 ? attn_launch<192, true>(which, p, grid, stream) : attn_launch<192, false>(which, p, grid, stream);
}

extern "C" int csts_attn_fwd(const csts_attn_args* a, hipStream_t stream) {
  ATTN_COMMON_CHECKS(a);
  CSTS_REQUIRE(a->O && a->LSE && aligned16(a->O) && strides_ok(a->o_strides, a->dtype), "bad output");
  AttnP p; fill(a, p);
  dim3 grid((unsigned)cdiv(a->Nq, 128), a->H, a->B);
  CSTS_REQUIRE(attn_dispatch(K_FWD, a, p, grid, stream), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on this device");
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t csts_attn_bwd_workspace(const csts_attn_args* a) {
  if (!a) return 0;
  int nsplit, q_chunk;
  dkv_plan(a, nsplit, q_chunk);
  return nsplit > 1 ? (size_t)2 * nsplit * a->B * a->H * a->Nk * a->head_dim * sizeof(float) : 16;
}

extern "C" int csts_attn_bwd(const csts_attn_args* a, void* workspace, size_t ws_bytes, hipStream_t stream) {
  ATTN_COMMON_CHECKS(a);
  CSTS_REQUIRE(a->O && a->LSE && a->dO && a->delta && a->dQ && a->dK && a->dV, "null backward operand");
  CSTS_REQUIRE(aligned16(a->dO) && aligned16(a->dQ) && aligned16(a->dK) && aligned16(a->dV), "alignment");
  CSTS_REQUIRE(strides_ok(a->do_strides, a->dtype) && strides_ok(a->dq_strides, a->dtype) &&
                   strides_ok(a->dk_strides, a->dtype) && strides_ok(a->dv_strides, a->dtype) &&
                   strides_ok(a->o_strides, a->dtype), "strides must keep 16-byte row alignment");
  AttnP p; fill(a, p);
  if (fused_bwd_ok(a)) {   // delta, dQ, dK, dV in one pass over Q / dO / O
    dkv_plan(a, p.nsplit, p.q_chunk);
    if (p.nsplit > 1) {
      CSTS_REQUIRE(workspace != nullptr && ws_bytes >= csts_attn_bwd_workspace(a), "workspace too small");
      p.ws = reinterpret_cast<float*>(workspace);
    }
    hipLaunchKernelGGL((attn_bwd_fused_kernel<96>), dim3((unsigned)p.nsplit, (unsigned)(a->B * a->H)), dim3(256), 0, stream, p);
    CSTS_LAUNCH_CHECK();
    if (p.nsplit > 1) {
      const int64_t total = (int64_t)2 * a->B * a->H * a->Nk * a->head_dim / 8;
      hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0,
                         stream, p, a->head_dim);
      CSTS_LAUNCH_CHECK();
    }
    return 0;
  }
  // 1 + 2. delta = rowsum(dO * O) and dQ (the dQ kernel computes delta from its staged O rows and leaves it for step 3)
  {
    dim3 grid((unsigned)cdiv(a->Nq, 128), a->H, a->B);
    CSTS_REQUIRE(attn_dispatch(K_DQ, a, p, grid, stream), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on this device");
    CSTS_LAUNCH_CHECK();
  }
  // 3. dK, dV (query-split, deterministic reduce)
  {
    dkv_plan(a, p.nsplit, p.q_chunk);
    if (p.nsplit > 1) {
      CSTS_REQUIRE(workspace != nullptr && ws_bytes >= csts_attn_bwd_workspace(a), "workspace too small");
      p.ws = reinterpret_cast<float*>(workspace);
    }
    dim3 grid((unsigned)cdiv(a->Nk, 128), p.nsplit, a->B * a->H);
    CSTS_REQUIRE(attn_dispatch(K_DKV, a, p, grid, stream), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on this device");
    CSTS_LAUNCH_CHECK();
    if (p.nsplit > 1) {
      const int64_t total = (int64_t)2 * a->B * a->H * a->Nk * a->head_dim / 8;
      hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0,
                         stream, p, a->head_dim);
      CSTS_LAUNCH_CHECK();
    }
  }
  return 0;
}

extern "C" int csts_attn_probs(const csts_attn_args* a, float* probs, hipStream_t stream) {
  ATTN_COMMON_CHECKS(a);
  CSTS_REQUIRE(a->LSE && probs, "null pointer");
  AttnP p; fill(a, p);
  const int64_t total = (int64_t)a->B * a->H * a->Nq * a->Nk;
  hipLaunchKernelGGL(attn_probs_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0, stream, p,
                     a->head_dim, probs);
  CSTS_LAUNCH_CHECK();
  return 0;
}
#else   // CSTS_ATTN_PART == 1: the FAST dK/dV kernel alone
}  // namespace
bool csts_attn_dkv_fast_launch(const csts_attn::AttnP& p, dim3 grid, hipStream_t stream) {
  // 128-query tiles, double-buffered, + statistics: 108 KB of dynamic LDS (one workgroup per CU either way)
  constexpr size_t smf = (size_t)2 * 128 * 2 * (CSTS_DKV_SWZ ? 128 : Cfg<96, false>::LD_ROW) * 2 + 2 * 2 * 128 * sizeof(float);
  if (!csts_dyn_lds_optin(reinterpret_cast<const void*>(&attn_dkv_kernel<96, false, false>), (int)smf)) return false;
  hipLaunchKernelGGL((attn_dkv_kernel<96, false, false>), grid, dim3(256), smf, stream, p);
  return true;
}
#endif
