// Shared between gemm.hip (dispatcher, register-staged and LDS-DMA ring kernels) and gemm4.hip (8-wave LDS-DMA ring kernel).
#pragma once
#include "common.h"

struct csts_gemm_params {
  const void* A; const void* B; void* C;
  const float* bias; void* aux; const void* residual; const float* row_scale; float* ws;
  float* colsum; float* colsum_ws;   // TN only: colsum[m] = sum_k A[k][m] (bias gradient), fused into the v2 kernel
  int64_t lda, ldb, ldc, ldaux, ldr;
  int64_t M, N, K, res_row_mod, rows_per_scale, k_chunk;
  int a_dt, b_dt, c_dt, aux_dt, r_dt, epilogue, split_k, a_vec, b_vec, ntiles_n;
  int64_t ntiles;   // gemm3 (persistent): output tiles of the whole problem
  unsigned long long* stamps;   // gemm3 built with -DCSTS_GEMM3_STAMPS: cycle stamps of two workgroups (diagnostics)
  int res_lines;                // gemm2: fp32 residual-stream epilogue in whole 128-byte lines (A/B switch CSTS_GEMM_RES_LINES=0)
  int store_aware;              // gemm4: k-tile waits that leave the previous tile's epilogue stores in flight (see kstep)
  // trilinear-upsampled residual (csts_gemm_args.res_up): coarse / fine grids, log2 of the (power-of-two) fine sizes
  int ru_Ti, ru_Hi, ru_Wi, ru_To, ru_Ho, ru_Wo, ru_lw, ru_lh, ru_lt;
};

namespace {

constexpr int BK2 = 64;                    // k-tile of the bf16 kernels

typedef csts_gemm_params Params;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it leaves LDS-DMA loads in flight
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// one LDS-DMA wave instruction, 16 bytes per lane: global (wave-uniform base + 32-bit lane byte offset: the saddr form) ->
// LDS (wave-uniform address in M0 + 16 * lane).  Written out because the builtin is handed full 64-bit lane addresses; the
// compiler does not count these loads (callers wait with explicit vmcnt, as they do for the builtin's).
__device__ __forceinline__ void lds_dma16(const char* base, unsigned lane_off, const void* lds_dst) {
  const uint32_t l = (uint32_t)(uintptr_t)(lptr_t)lds_dst;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(lane_off), "s"(base) : "memory", "m0");
}

// 4 consecutive elements -> 4 floats
__device__ __forceinline__ f32x4 ld4_as_f32(const void* p, int dt, int64_t i) {
  if (dt == CSTS_F32) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + i);
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p) + i);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
// Store a lane's four 4-column runs o[q] (columns c0 + 8 q .. + 3 of one row; c0 already includes this half-wave's
// +4 * hi) at element offset `at` (= row * ld + c0).  fp32: one 16-byte store per run.  bf16: lanes l and l + 32 hold
// adjacent runs of the same row, so v_permlane32_swap pairs them into 8 contiguous columns per lane -> one 16-byte
// store per two runs (lower half-wave: columns 16 p .. + 7, upper: 16 p + 8 .. + 15, relative to the unit's first column).
// Must be called with all lanes active (the swap is a cross-lane exchange); `rowok` / N only mask the stores.
// (st4x4_bf16_whole below: the same stores for a tile that lies wholly inside the matrix, no tests)
__device__ __forceinline__ void st4x4(void* base, int dt, int64_t at, const f32x4 (&o)[4], bool rowok, int64_t c0, int64_t N, int hi) {
  if (dt == CSTS_F32) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (rowok && c0 + 8 * q < N) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + at + 8 * q) = o[q];
  } else {
    uint2 pk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16x4 b = {(bf16)o[q][0], (bf16)o[q][1], (bf16)o[q][2], (bf16)o[q][3]};
      pk[q] = __builtin_bit_cast(uint2, b);
    }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      uint2 a = pk[2 * pr], b = pk[2 * pr + 1];
      const auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
      const auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
      a.x = rx[0]; b.x = rx[1]; a.y = ry[0]; b.y = ry[1];
      // lower half-wave now holds columns 16 pr .. 16 pr + 7 of the unit, the upper half-wave 16 pr + 8 .. 16 pr + 15
      const int64_t cfirst = c0 - 4 * hi + 16 * pr + 8 * hi;
      if (rowok && cfirst < N)
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(base) + at - 4 * hi + 16 * pr + 8 * hi) = make_uint4(a.x, a.y, b.x, b.y);
    }
  }
}

// ---- residual = nn.Upsample(trilinear, align_corners=False) of a coarse token grid, evaluated in the epilogue.
// The arithmetic is csts_trilinear_fwd's (stencil.hip), term for term: per axis the source cell and weights of torch's
// area_pixel_compute_source_index, products tl * hl * wl, accumulation over (t, h, w) taps in that order from 0.f,
// zero-weight taps skipped -- so a GEMM with res_up set equals trilinear -> GEMM with a plain residual bit for bit.
struct ResUpRow {
  int64_t base;          // first coarse row of the batch element
  int ti[2], hi[2], wi[2];
  float tl[2], hl[2], wl[2];
};
__device__ __forceinline__ void res_up_axis(int o, int in, int out, int& i0, int& i1, float& l0, float& l1) {
  const float scale = (float)in / (float)out;
  float s = ((float)o + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}
__device__ __forceinline__ ResUpRow res_up_row(const Params& p, int64_t m) {
  ResUpRow r;
  const int ow = (int)(m & (p.ru_Wo - 1));
  const int oh = (int)((m >> p.ru_lw) & (p.ru_Ho - 1));
  const int ot = (int)((m >> (p.ru_lw + p.ru_lh)) & (p.ru_To - 1));
  const int64_t b = m >> (p.ru_lw + p.ru_lh + p.ru_lt);
  r.base = b * ((int64_t)p.ru_Ti * p.ru_Hi * p.ru_Wi);
  res_up_axis(ot, p.ru_Ti, p.ru_To, r.ti[0], r.ti[1], r.tl[0], r.tl[1]);
  res_up_axis(oh, p.ru_Hi, p.ru_Ho, r.hi[0], r.hi[1], r.hl[0], r.hl[1]);
  res_up_axis(ow, p.ru_Wi, p.ru_Wo, r.wi[0], r.wi[1], r.wl[0], r.wl[1]);
  return r;
}
// NV consecutive columns n .. n + NV - 1 of the upsampled residual of the row described by r
template <int NV>
__device__ __forceinline__ void res_up_load(const Params& p, const ResUpRow& r, int64_t n, float (&acc)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    if (r.tl[a] == 0.f) continue;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (r.hl[e] == 0.f) continue;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        if (r.wl[f] == 0.f) continue;
        const float wgt = r.tl[a] * r.hl[e] * r.wl[f];
        const int64_t at = (r.base + (int64_t)(r.ti[a] * p.ru_Hi + r.hi[e]) * p.ru_Wi + r.wi[f]) * p.ldr + n;
        if constexpr (NV == 8) {
          float v[8];
          ld8_as_f32(p.residual, p.r_dt, at, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += wgt * v[j];
        } else {
#pragma unroll
          for (int j = 0; j < NV; ++j) acc[j] += wgt * ld_as_f32(p.residual, p.r_dt, at + j);
        }
      }
    }
  }
}

// st4x4 for bf16 and a tile wholly inside the matrix: two unconditional 16-byte stores per lane (straight-line code: the
// epilogue's vector-memory waits can then be counted ones)
__device__ __forceinline__ void st4x4_bf16_whole(bf16* base, int64_t at, const f32x4 (&o)[4], int hi) {
  uint2 pk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bf16x4 b = {(bf16)o[q][0], (bf16)o[q][1], (bf16)o[q][2], (bf16)o[q][3]};
    pk[q] = __builtin_bit_cast(uint2, b);
  }
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    uint2 a = pk[2 * pr], b = pk[2 * pr + 1];
    const auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    a.x = rx[0]; b.x = rx[1]; a.y = ry[0]; b.y = ry[1];
    *reinterpret_cast<uint4*>(base + at - 4 * hi + 16 * pr + 8 * hi) = make_uint4(a.x, a.y, b.x, b.y);
  }
}

// one pair of runs (columns 16 pr .. 16 pr + 15 of a 32-column unit): the lower half-wave ends up with columns 16 pr .. + 7
// of its row, the upper one with 16 pr + 8 .. + 15 -> ONE 16-byte store per lane.  `at` = row * ld + unit column 0 + 4 * hi.
__device__ __forceinline__ void st4x2_bf16_whole(bf16* base, int64_t at, const f32x4& o0, const f32x4& o1, int pr, int hi) {
  const bf16x4 b0 = {(bf16)o0[0], (bf16)o0[1], (bf16)o0[2], (bf16)o0[3]};
  const bf16x4 b1 = {(bf16)o1[0], (bf16)o1[1], (bf16)o1[2], (bf16)o1[3]};
  uint2 a = __builtin_bit_cast(uint2, b0), b = __builtin_bit_cast(uint2, b1);
  const auto rx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
  const auto ry = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
  a.x = rx[0]; b.x = rx[1]; a.y = ry[0]; b.y = ry[1];
  *reinterpret_cast<uint4*>(base + at - 4 * hi + 16 * pr + 8 * hi) = make_uint4(a.x, a.y, b.x, b.y);
}

}  // namespace

// gemm4.hip: persistent NT kernel, 256-row (8-wave) and 128-row (4- or 8-wave) tiles, LDS-DMA ring.
// variant = 10 * shape + stages (see gemm4.hip); returns false when the variant does not exist.
bool csts_gemm4_launch(const csts_gemm_params& p, const csts_gemm_args* a, int variant, int wpc, hipStream_t s);
bool csts_gemm4_name(const csts_gemm_params& p, int variant, char* buf, int buflen);

// gemm5.hip: streaming NT kernel for the thin problems (K = 96 / 192, N % 96 == 0, large M): weights resident in LDS, one stream per wave.
bool csts_gemm5_ok(const csts_gemm_args* a, int split);
bool csts_gemm5_launch(const csts_gemm_params& p, const csts_gemm_args* a, hipStream_t s);
bool csts_gemm5_name(const csts_gemm_args* a, char* buf, int buflen);
