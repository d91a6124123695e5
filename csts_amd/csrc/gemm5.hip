// gemm5: STREAMING NT kernel for the thin problems of the 96- / 192-channel stages (K = 96 or 192, N a multiple of 96, M = 32 k .. 262 k
// token rows): activations x bf16 shadow weights, both k-contiguous.
//
// Why (profiles/r5_final_gemm_replay_shapes.txt, r4_traffic_mix.txt): these problems are pure streams -- 192 .. 384 B of A per token row
// against 192 .. 1536 B of outputs / residual, 1.5 .. 3 k-tiles of work per output tile -- and the tiled kernels ran them at 1.8 - 3.5 TB/s
// (the dominant kernel of the step: 27 launches, 0.28 of HBM) where the same bytes with no arithmetic move at 5 - 6 TB/s.  A tile kernel is
// all prologue and epilogue here: load A and B tile, barrier, 6 - 12 MFMAs per wave, barrier, stage C through LDS, barrier, store; and the
// weights (18 - 74 KB) are re-staged by every one of thousands of tiles.
//
// Structure: ONE persistent 8-wave workgroup per CU.  The workgroup's slice of the weights (96 or 192 output features x K) is staged into
// LDS ONCE (padded rows: conflict-free 16-byte fragment reads); after that single barrier the waves never synchronise again.  Every wave
// is its own stream: it owns 32-row units u = wave id, wave id + #waves, ...; a unit's A rows (32 x K, 6 KB per 96 k) come in by LDS-DMA
// (global_load_lds_dwordx4, whole 192-byte row runs) into a wave-private ring one unit ahead; the wave multiplies the unit against the
// resident weights (MFMA 32x32x16, activations as the row operand, k ascending: the arithmetic order of gemm2_kernel), and writes the
// 32 x N outputs straight from the accumulators.  Weight rows are dealt to MFMA columns so that a lane owns PAIRS of adjacent output
// features (columns 2c, 2c + 1 of a 64-column group sit in the two MFMA blocks of the pair at lane c): a half-wave then stores 32 x 2
// adjacent values of ONE token row per instruction -- 128 B (bf16) or 256 B (fp32) contiguous, whole lines -- and reads the fp32 residual /
// the saved pre-activations the same way.  (An odd third block, N = 96, is written as 32 single values: 64 / 128 B runs.)
//
// LDS image of an A chunk (32 rows x 192 B, written linearly by the DMA): row stride 192 B would put rows r, r + 4, r + 8, ... on the same
// banks, so the 16-byte k-chunks of row r are rotated by (r >> 2) & 3 slots -- applied to the per-lane SOURCE address of the DMA and to
// the fragment read: the 16 rows a ds_read_b128 pass touches then land on 16 distinct 16-byte bank groups.
//
// FORM (compile time, host-selected): 0 bf16 C = acc + bias; 1 + GELU (pre-activation to aux); 2 x gelu'(aux); 3 fp32 C = (acc + bias) *
// row_scale + fp32 residual (the residual stream: proj / fc2); 4 fp32 C = acc + bias.  Same epilogue arithmetic, term for term, as gemm2_kernel.
#include "gemm_shared.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace {

template <int KCH, int NB, int W> struct G5 {
  static constexpr int K = 96 * KCH, NW = 32 * NB, NTHR = 64 * W;
  static constexpr int CHUNK = 32 * 192;                   // bytes of one A chunk: 32 rows x 96 k
  static constexpr int RING = 2 * CHUNK;                   // a wave's ring: two chunk slots (K = 96: this unit and the next; K = 192: a unit's two chunks, refilled as they are consumed)
  static constexpr int BL = 2 * K + 16;                    // weight row stride in LDS (bytes): (BL / 16) odd -> 16 consecutive rows hit 16 distinct bank groups
  static constexpr int B_BYTES = NW * BL, A_BYTES = W * RING, LDS = B_BYTES + A_BYTES;
  static constexpr int NPAIR = NB / 2, NSINGLE = NB % 2;   // column pairs / odd last block
  static constexpr int NDMA = 6 * KCH;                     // LDS-DMA wave instructions per unit
  static_assert(LDS <= 160 * 1024, "weights + rings exceed the CU's LDS");
  static_assert((BL / 16) % 2 == 1, "weight row stride must be an odd number of 16-byte groups");
};

template <int N> struct MinC { static constexpr int v = N < 63 ? N : 63; };

template <int KCH, int NB, int W, int FORM>
__global__ __launch_bounds__((G5<KCH, NB, W>::NTHR), 2) void gemm5_kernel(Params p, int nt_n, int units) {
  typedef G5<KCH, NB, W> G;
  constexpr int K = G::K, BL = G::BL, CHUNK = G::CHUNK, NPAIR = G::NPAIR, NSINGLE = G::NSINGLE, NDMA = G::NDMA;
  constexpr bool C32 = FORM >= 3, RES = FORM == 3, GELU = FORM == 1, DGELU = FORM == 2;
  static_assert(KCH == 1 || KCH == 2 || (KCH == 4 && !DGELU), "K = 96, 192, or 384 without the x gelu' form");
  __shared__ __attribute__((aligned(1024))) char smem[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* const Bs = smem;
  char* const As = smem + G::B_BYTES + wave * G::RING;

  // workgroups b and b + 8 sit on the same XCD: they take different column tiles of the SAME units, so the second read of a unit's A rows
  // comes from that XCD's L2
  const int grp = blockIdx.x >> 3, ntile = grp % nt_n;
  const int wg = (grp / nt_n) * 8 + (blockIdx.x & 7), nwg = gridDim.x / nt_n;
  const int gw = wg * W + wave, nw = nwg * W;
  const int64_t n0 = (int64_t)ntile * G::NW;

  // ---- the weights of this column tile, once: LDS row q of a pair (blocks 2j, 2j + 1) holds feature 64 j + 2 (q & 31) + (block & 1)
  {
    constexpr int CPR = K / 8;
    const bf16* __restrict__ Bg = reinterpret_cast<const bf16*>(p.B);
    for (int idx = tid; idx < G::NW * CPR; idx += G::NTHR) {
      const int prow = idx / CPR, ch = idx % CPR, blk = prow >> 5, c = prow & 31;
      const int n = (blk < 2 * NPAIR) ? (blk >> 1) * 64 + 2 * c + (blk & 1) : blk * 32 + c;
      const uint4 v = *reinterpret_cast<const uint4*>(Bg + (n0 + n) * p.ldb + ch * 8);
      *reinterpret_cast<uint4*>(Bs + prow * BL + ch * 16) = v;
    }
  }

  // ---- producer: per-lane source byte offsets of the six 1-KiB pieces of a chunk image (slot s = 64 i + lane: row s / 12, rotated chunk s % 12)
  unsigned soff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int s = 64 * i + lane, row = s / 12, slot = s % 12;
    int c = slot - ((row >> 2) & 3);
    if (c < 0) c += 12;
    soff[i] = (unsigned)(row * (int)p.lda * 2 + c * 16);
  }
  const char* const Ag = reinterpret_cast<const char*>(p.A);
  // chunk j of the unit with sequence number `it` lives in ring slot (K = 96) it & 1, (K = 192) j, (K = 384) j & 1
  auto issue_chunk = [&](int u, int it, int j) {
    const char* base = Ag + (int64_t)u * 32 * p.lda * 2 + j * 192;
    char* dst = As + (KCH == 1 ? (it & 1) : (j & 1)) * CHUNK;
#pragma unroll
    for (int i = 0; i < 6; ++i) lds_dma16(base, soff[i], dst + i * 1024);
  };

  // ---- consumer: fragment addresses
  int aoff[6];                                      // byte offset of this lane's 16-byte k-chunk of sub-step ks inside a chunk image
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) {
    int slot = 2 * ks + hi + ((r32 >> 2) & 3);
    if (slot >= 12) slot -= 12;
    aoff[ks] = r32 * 192 + slot * 16;
  }
  const char* const bfrag = Bs + r32 * BL + hi * 16;

  // ---- epilogue: this lane's columns and bias
  // pair j: columns cp = 64 j + 2 r32, cp + 1; odd block: column cs = 64 NPAIR + r32 (relative to n0)
  float bias_p[NPAIR > 0 ? NPAIR : 1][2], bias_s = 0.f;
#pragma unroll
  for (int j = 0; j < NPAIR; ++j) {
    bias_p[j][0] = 0.f; bias_p[j][1] = 0.f;
    if (p.bias != nullptr) {
      const float2 b = *reinterpret_cast<const float2*>(p.bias + n0 + 64 * j + 2 * r32);
      bias_p[j][0] = b.x; bias_p[j][1] = b.y;
    }
  }
  if (NSINGLE && p.bias != nullptr) bias_s = p.bias[n0 + 64 * NPAIR + r32];

  int u = gw;
  if (u < units) {
#pragma unroll
    for (int j = 0; j < (KCH < 2 ? KCH : 2); ++j) issue_chunk(u, 0, j);
  }
  __syncthreads();                                  // weights staged; no barrier after this one

  // Epilogue operands (the fp32 residual rows / the saved pre-activations) travel ONE UNIT AHEAD, like the A rows: requested after the
  // previous unit's MFMAs and BEFORE its stores.  Vector memory retires in order on one counter: a wait for this unit's operands is then a
  // wait for "at most the previous unit's NST stores outstanding" -- the stores themselves stay in flight (requested after them, every
  // unit would start by draining its predecessor's stores).
  constexpr int NRL = (RES || DGELU) ? 16 * (NPAIR + NSINGLE) : 0;
  constexpr int NST = 16 * (NPAIR + NSINGLE);        // lower bound of a unit's store instructions (GELU with a kept pre-activation: twice that)
  float2 rp[RES ? 16 : 1][NPAIR > 0 ? NPAIR : 1], rp_n[RES ? 16 : 1][NPAIR > 0 ? NPAIR : 1];
  float rs1[RES && NSINGLE ? 16 : 1], rs1_n[RES && NSINGLE ? 16 : 1];
  bf16x2 hp[DGELU ? 16 : 1][NPAIR > 0 ? NPAIR : 1], hp_n[DGELU ? 16 : 1][NPAIR > 0 ? NPAIR : 1];
  bf16 hs1[DGELU && NSINGLE ? 16 : 1], hs1_n[DGELU && NSINGLE ? 16 : 1];
  float rsc = 1.f, rsc_n = 1.f;
  auto request = [&](int un) {                      // epilogue operands of unit un into the *_n registers
    const int64_t mn = (int64_t)un * 32;
    if constexpr (RES) {
      const float* __restrict__ res = reinterpret_cast<const float*>(p.residual) + (mn + 4 * hi) * p.ldr + n0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* row = res + ((r & 3) + 8 * (r >> 2)) * p.ldr;
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) rp_n[r][j] = *reinterpret_cast<const float2*>(row + 64 * j + 2 * r32);
        if constexpr (NSINGLE) rs1_n[r] = row[64 * NPAIR + r32];
      }
      if (p.row_scale != nullptr) rsc_n = p.row_scale[mn / p.rows_per_scale];     // rows_per_scale % 32 == 0 (host-checked): one scale per unit
    }
    if constexpr (DGELU) {
      const bf16* __restrict__ hx = reinterpret_cast<const bf16*>(p.aux) + (mn + 4 * hi) * p.ldaux + n0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bf16* row = hx + ((r & 3) + 8 * (r >> 2)) * p.ldaux;
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) hp_n[r][j] = *reinterpret_cast<const bf16x2*>(row + 64 * j + 2 * r32);
        if constexpr (NSINGLE) hs1_n[r] = row[64 * NPAIR + r32];
      }
    }
  };
  // Measured inside the replayed step (gpurun_out r5t / r5v): ahead wins 3-4 % for the fp32 residual rows (12 KB per unit) and LOSES 5-9 %
  // for the bf16 pre-activations (x gelu': requested at the top of their own unit instead, and waited for after the MFMAs).
  constexpr bool AHEAD = RES;
  if (AHEAD && u < units) request(u);

  for (int it = 0; u < units; u += nw, ++it) {
    const int64_t m0 = (int64_t)u * 32;
    // row of accumulator register r: m0 + (r & 3) + 8 (r >> 2) + 4 hi
    if constexpr (NRL > 0 && !AHEAD) {
      // ---- (1) this unit's epilogue operands; (2) its A rows have landed: everything older than the operands just requested is complete
      request(u);
      wait_vm<MinC<NRL>::v>();
    } else if constexpr (KCH <= 2) {
      // ---- (1, 2) this unit's A rows (and epilogue operands) have landed: they were requested before the previous unit's stores
      if (it == 0) wait_vm<0>();
      else wait_vm<MinC<NST>::v>();
    }
    // (the LDS-DMA writes of THIS wave are visible to its own ds_reads once vmcnt says so: no other wave touches this ring)
    if constexpr (AHEAD && KCH <= 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) {
          if constexpr (RES) rp[r][j] = rp_n[r][j];
          if constexpr (DGELU) hp[r][j] = hp_n[r][j];
        }
        if constexpr (RES && NSINGLE) rs1[r] = rs1_n[r];
        if constexpr (DGELU && NSINGLE) hs1[r] = hs1_n[r];
      }
      rsc = rsc_n;
    }

    // ---- (3) multiply
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const bool more = u + nw < units;
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
      const char* a_img = As + (KCH == 1 ? (it & 1) : (j & 1)) * CHUNK;
      if constexpr (KCH == 4) {
        // K = 384: four chunks through TWO slots, each refilled two chunks ahead.  Chunk j has landed once at most the wave instructions issued
        // AFTER its DMA are outstanding (in-order retirement; lower bounds): j = 0, 1 were requested in the previous unit before its epilogue
        // (operand requests NRL_A, stores NST) with one younger DMA (6); j = 2, 3 in this unit with one younger DMA
        constexpr int NRL_A = AHEAD ? NRL : 0;
        if (j < 2) { if (it == 0) wait_vm<MinC<6 + NRL_A>::v>(); else wait_vm<MinC<6 + NRL_A + NST>::v>(); }
        else if (j == 2) wait_vm<6>();
        else { if (more) wait_vm<6>(); else wait_vm<0>(); }
      }
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_img + aoff[ks]);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(bfrag + b * 32 * BL + (j * 12 + 2 * ks) * 16);
          acc[b] = CSTS_MFMA16(a, w, acc[b], 0, 0, 0);
        }
      }
      // ---- (4) the same chunk of the next unit: K = 96 into the other slot (consumed one unit ago); K = 192 into the slot whose fragments
      // have just been read (every ds_read above has returned: its MFMA has issued)
      if constexpr (KCH == 4) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j < 2) issue_chunk(u, it, j + 2);
        else if (more) issue_chunk(u + nw, it + 1, j - 2);
      } else if (more) {
        if (KCH > 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_chunk(u + nw, it + 1, j);
      }
    }
    if constexpr (AHEAD && KCH == 4) {
      // this unit's epilogue operands (requested before the previous unit's stores) have landed: behind them sit those stores and four DMAs
      if (it == 0) { if (more) wait_vm<24>(); else wait_vm<12>(); }
      else { if (more) wait_vm<MinC<24 + NST>::v>(); else wait_vm<MinC<12 + NST>::v>(); }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) rp[r][j] = rp_n[r][j];
        if constexpr (NSINGLE) rs1[r] = rs1_n[r];
      }
      rsc = rsc_n;
    }
    // ---- (5) the next unit's epilogue operands / this unit's have landed (only the DMA just issued may still be in flight)
    if (AHEAD && more) request(u + nw);
    if constexpr (NRL > 0 && !AHEAD) {
      if (more) wait_vm<MinC<NDMA>::v>();
      else wait_vm<0>();
    }

    // ---- (6) epilogue, straight from the accumulators
    if constexpr (C32) {
      float* __restrict__ Cf = reinterpret_cast<float*>(p.C) + (m0 + 4 * hi) * p.ldc + n0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* row = Cf + ((r & 3) + 8 * (r >> 2)) * p.ldc;
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) {
          float v0 = acc[2 * j][r] + bias_p[j][0], v1 = acc[2 * j + 1][r] + bias_p[j][1];
          if constexpr (RES) {
            if (p.row_scale != nullptr) { v0 *= rsc; v1 *= rsc; }
            v0 += rp[r][j].x; v1 += rp[r][j].y;
          }
          *reinterpret_cast<float2*>(row + 64 * j + 2 * r32) = make_float2(v0, v1);
        }
        if constexpr (NSINGLE) {
          float v = acc[NB - 1][r] + bias_s;
          if constexpr (RES) {
            if (p.row_scale != nullptr) v *= rsc;
            v += rs1[r];
          }
          row[64 * NPAIR + r32] = v;
        }
      }
    } else {
      bf16* __restrict__ Cb = reinterpret_cast<bf16*>(p.C) + (m0 + 4 * hi) * p.ldc + n0;
      bf16* __restrict__ Hb = GELU && p.aux != nullptr ? reinterpret_cast<bf16*>(p.aux) + (m0 + 4 * hi) * p.ldaux + n0 : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) {
          float v0 = acc[2 * j][r] + bias_p[j][0], v1 = acc[2 * j + 1][r] + bias_p[j][1];
          if constexpr (GELU) {
            if (Hb != nullptr) {
              const bf16x2 h = {(bf16)v0, (bf16)v1};
              *reinterpret_cast<bf16x2*>(Hb + ro * p.ldaux + 64 * j + 2 * r32) = h;
            }
            v0 = gelu_fast(v0); v1 = gelu_fast(v1);
          }
          if constexpr (DGELU) { v0 *= dgelu_fast((float)hp_n[r][j][0]); v1 *= dgelu_fast((float)hp_n[r][j][1]); }
          const bf16x2 o = {(bf16)v0, (bf16)v1};
          *reinterpret_cast<bf16x2*>(Cb + ro * p.ldc + 64 * j + 2 * r32) = o;
        }
        if constexpr (NSINGLE) {
          float v = acc[NB - 1][r] + bias_s;
          if constexpr (GELU) {
            if (Hb != nullptr) Hb[ro * p.ldaux + 64 * NPAIR + r32] = (bf16)v;
            v = gelu_fast(v);
          }
          if constexpr (DGELU) v *= dgelu_fast((float)hs1_n[r]);
          Cb[ro * p.ldc + 64 * NPAIR + r32] = (bf16)v;
        }
      }
    }
  }
  wait_vm<0>();
}

// epilogue form of a problem (-1: not one this kernel carries)
int gemm5_form(const csts_gemm_args* a) {
  if (a->residual != nullptr) {
    if (a->c_dt == CSTS_F32 && a->r_dt == CSTS_F32 && a->res_row_mod == 0 && a->res_up[3] == 0 && a->epilogue == CSTS_EPI_NONE &&
        (a->row_scale == nullptr || (a->rows_per_scale > 0 && a->rows_per_scale % 32 == 0)))
      return 3;
    return -1;
  }
  if (a->row_scale != nullptr) return -1;
  if (a->c_dt == CSTS_F32) return a->epilogue == CSTS_EPI_NONE ? 4 : -1;
  if (a->c_dt != CSTS_BF16) return -1;
  if (a->epilogue == CSTS_EPI_NONE) return 0;
  if (a->epilogue == CSTS_EPI_GELU && a->aux_dt == CSTS_BF16) return 1;
  if (a->epilogue == CSTS_EPI_DGELU && a->aux != nullptr && a->aux_dt == CSTS_BF16) return 2;
  return -1;
}

template <int KCH, int NB, int W = 8>
bool launch5(const Params& p, int form, hipStream_t s) {
  typedef G5<KCH, NB, W> G;
  const int nt_n = (int)(p.N / G::NW), units = (int)(p.M / 32);
  const int g = (256 / (8 * nt_n)) * 8 * nt_n;                 // one workgroup per CU, whole groups of (8 XCDs x column tiles)
  const dim3 grid((unsigned)g), block(G::NTHR);
  switch (form) {
    case 0: hipLaunchKernelGGL((gemm5_kernel<KCH, NB, W, 0>), grid, block, 0, s, p, nt_n, units); return true;
    case 1: hipLaunchKernelGGL((gemm5_kernel<KCH, NB, W, 1>), grid, block, 0, s, p, nt_n, units); return true;
    case 2:
      if constexpr (KCH != 4) { hipLaunchKernelGGL((gemm5_kernel<KCH, NB, W, 2>), grid, block, 0, s, p, nt_n, units); return true; }
      return false;
    case 3:                                                    // the residual stream runs on 96-column tiles only (gemm5_nb): two register sets of residual rows
      if constexpr (NB == 3) { hipLaunchKernelGGL((gemm5_kernel<KCH, NB, W, 3>), grid, block, 0, s, p, nt_n, units); return true; }
      return false;
    case 4: hipLaunchKernelGGL((gemm5_kernel<KCH, NB, W, 4>), grid, block, 0, s, p, nt_n, units); return true;
  }
  return false;
}

// column blocks per workgroup: 6 (192 features) where that divides N, else 3.  K = 192 on 192 features (77 KB of weights) leaves room for
// SIX wave rings: gemm5_kernel<2, 6, 6, .> -- the A rows are then fetched N / 192 times instead of N / 96 times.
int gemm5_nb(const csts_gemm_args* a) {
  // OFF by default: inside the replayed step the wide form wins on two shapes (262144 x 384 x 192 + GELU 165 -> 156 us, 131072 x 384 x 192 plain
  // 45 -> 39 us) and loses on the others (x gelu' 72 -> 87 us, M = 32768 34 -> 44 us): +0.08 ms per step (CSTS_GEMM5_WIDE192=1 / algo 506)
  static const int wide192 = [] { const char* e = getenv("CSTS_GEMM5_WIDE192"); return e ? atoi(e) : 0; }();
  if (gemm5_form(a) == 3) return 3;                            // two register sets of residual rows: 96 columns
  if (a->K == 96) return a->N % 192 == 0 ? 6 : 3;
  if (a->K == 384) return 3;
  return (a->N % 192 == 0 && (a->algo % 1000 == 506 || (a->algo % 1000 != 503 && wide192))) ? 6 : 3;
}

}  // namespace

// Can this kernel run the problem at all (forced algo 500)?
bool csts_gemm5_ok(const csts_gemm_args* a, int split) {
  if (a->layout != CSTS_GEMM_NT || a->compute != CSTS_BF16 || a->a_dt != CSTS_BF16 || a->b_dt != CSTS_BF16 || split != 1) return false;
  if (a->colsum != nullptr || (a->K != 96 && a->K != 192 && a->K != 384) || a->N % 96 != 0 || a->N > 768 || a->M % 32 != 0 || a->M < 32) return false;
  if (gemm5_form(a) < 0) return false;
  if (a->K == 384 && (gemm5_form(a) == 2 || a->N > 384)) return false;          // four-chunk ring: no x gelu' form; 75 KB of weights per 96 columns
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al(a->A) || !al(a->B) || !al(a->C) || a->lda % 8 != 0 || a->ldb % 8 != 0 || a->ldc % 2 != 0) return false;
  if (a->aux != nullptr && (!al(a->aux) || a->ldaux % 2 != 0)) return false;
  if (a->residual != nullptr && (!al(a->residual) || a->ldr % 2 != 0)) return false;
  if (a->bias != nullptr && !al(a->bias)) return false;
  if (32 * a->lda * 2 >= (int64_t(1) << 31) || 32 * std::max(a->ldc, std::max(a->ldaux, a->ldr)) * 4 >= (int64_t(1) << 31)) return false;
  return true;
}

bool csts_gemm5_launch(const csts_gemm_params& p, const csts_gemm_args* a, hipStream_t s) {
  const int form = gemm5_form(a), nb = gemm5_nb(a);
  if (a->K == 96) return nb == 6 ? launch5<1, 6>(p, form, s) : launch5<1, 3>(p, form, s);
  if (a->K == 384) return launch5<4, 3, 6>(p, form, s);
  return nb == 6 ? launch5<2, 6, 6>(p, form, s) : launch5<2, 3>(p, form, s);
}

// the kernel csts_gemm5_launch starts, as rocprofv3 prints it
bool csts_gemm5_name(const csts_gemm_args* a, char* buf, int buflen) {
  const int nb = gemm5_nb(a);
  snprintf(buf, buflen, "gemm5_kernel<%d, %d, %d, %d>", (int)(a->K / 96), nb, ((a->K == 192 && nb == 6) || a->K == 384) ? 6 : 8, gemm5_form(a));
  return true;
}
