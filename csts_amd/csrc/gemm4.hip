// gemm4: PERSISTENT NT kernel (activations x bf16 shadow weights, both k-contiguous) on 8-wave workgroups.
//
// Why: a 128 x 128 x 64 k-step moves 32 KiB through the CU's vector-memory path for 16 MFMAs per wave -- the two take
// the same ~512 cycles, so 4-wave 128-row tiles sit ON the ridge between the L2 -> LDS path and the matrix pipe and every
// variant of them measured alike ("What bounds the activation GEMMs").  Here a workgroup is 8 waves (two per
// SIMD: one wave's fragment reads / LDS-DMA issue / waits run under its partner's MFMAs) on a 256-row tile, 128 / 192 /
// 256 columns wide: 85 / 110 / 128 FLOP per staged byte instead of 64, i.e. 1.33 - 2 x fewer bytes through L2 -> LDS per
// FLOP, and the row panel of A is read by half as many column tiles.
//
// Structure (as gemm3, gemm.hip): a workgroup owns a list of output tiles (an XCD works on a contiguous tile range, n
// fastest: the column tiles of one A row panel follow each other on one L2) and runs ONE flattened stream of 64-deep
// k-tiles over them; operand tiles go L2/HBM -> LDS by global_load_lds_dwordx4 into a ring of S stages that stays in
// flight across the single raw s_barrier of a k-step (counted s_waitcnt vmcnt) and across tile boundaries; the LDS image
// is lane-linear per wave-instruction (8 rows x 128 B) with the XOR swizzle on the per-lane SOURCE address and on the
// fragment read; MFMA operands are swapped (weights as the row operand) so that a lane owns 4-column runs of one output
// row and the epilogue goes straight from the accumulators to 16-byte stores (v_permlane32_swap pairs for bf16).
// Wave grid WM x 2, wave tile (32 MT) x (32 NT): tile (32 MT WM) x (64 NT).
// Preconditions (host-checked): bf16 A and B, 16-byte aligned rows, K % 16 == 0, split_k == 1, gridDim.x % 8 == 0.
#include "gemm_shared.h"
#include <algorithm>
#include <cstdio>
#include <type_traits>

// Diagnostics build (-DCSTS_GEMM4_STAMPS, `make stamps`, tools/gemm4_stamps.py): wave 0 of workgroups 0 and 137 records shader-clock
// stamps around the phases of every k-step and of the epilogue.  No stamp code exists in the library build.
#ifdef CSTS_GEMM4_STAMPS
#define G4_STAMP() do { if (stamp_on && nstamp < 500) stamp_buf[nstamp++] = (unsigned long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define G4_STAMP() do { } while (0)
#endif

namespace {

// NP > 0: wave specialisation.  NP extra PRODUCER waves issue every LDS-DMA instruction and do nothing else; the 2 WM consumer
// waves multiply and store and never issue or wait for a load.  Vector memory retires in order on one counter per wave, so a
// wave that both stores a tile and stages the next one cannot see its loads land before its stores have drained: on the
// short-K, large-M shapes every workgroup alternated between "storing" and "loading + multiplying" and the chip averaged 3 TB/s
// (profiles/r3_shortk_gemm_stamps.txt).  A producer's counter never holds a store.
template <int WM, int MT, int NT, int S, int NP_ = 0> struct G4 {
  static constexpr int NP = NP_;
  static constexpr int DB = (MT * NT >= 6) ? 1 : 2;      // fragment register sets: the 96 / 128-accumulator tiles have no room for two
  static constexpr int NWAVES = WM * 2, NTHR = 64 * (NWAVES + NP);
  static constexpr int DW = NP > 0 ? NP : NWAVES;        // waves that issue LDS-DMA
  static constexpr int BM = WM * 32 * MT, BN = 64 * NT;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int NA = BM / 8 / DW, NB = BN / 8 / DW, LPT = NA + NB;   // LDS-DMA instructions per (issuing) thread per k-tile
  static constexpr int FIT = (160 * 1024) / (S * STAGE);
  static constexpr int WG = NP > 0 ? 1 : (FIT < 1 ? 1 : (FIT > 2 ? 2 : FIT));        // workgroups per CU the register budget is set for
  static constexpr int WPS = (WG * (NWAVES + NP) + 3) / 4;                            // waves per SIMD
  static_assert(BM % (8 * DW) == 0 && BN % (8 * DW) == 0, "pieces must divide over the issuing waves");
  static_assert(S * STAGE <= 160 * 1024, "ring exceeds the CU's LDS");
  static_assert(2 * LPT <= 63, "vmcnt immediate");
};

// FORM: the epilogue this instantiation carries (host-selected, gemm4_form): 0 = generic (any epilogue, ragged tiles);
// 1 .. 4 = ONE straight-line epilogue for problems made of whole tiles with a bf16 C and no residual / row scale:
// 1 bias, 2 bias + GELU (pre-activation to aux), 3 plain, 4 DGELU (times gelu'(aux)).  One epilogue per kernel keeps the
// register allocation of the k-loop and of that epilogue inside the 128-register budget of two workgroups per CU (all forms
// in one kernel: 59 spilled registers).
template <int WM, int MT, int NT, int S, int FORM, int NP = 0>
__global__ __launch_bounds__((G4<WM, MT, NT, S, NP>::NTHR), (G4<WM, MT, NT, S, NP>::WPS)) void gemm4_kernel(Params p) {
  typedef G4<WM, MT, NT, S, NP> G;
  constexpr bool KTAIL = false;        // K % 64 != 0 is not instantiated (the library never picks this kernel for it)
  // experiment of round 4 (profiles/r4_gemm4_store_aware_ab.txt): compiled in only with -DCSTS_GEMM4_STORE_AWARE_BUILD -- even
  // switched off at run time, the extra wave-uniform branch in every k-step cost the whole family ~0.2 ms per step
#ifdef CSTS_GEMM4_STORE_AWARE_BUILD
  const bool STORE_AWARE = p.store_aware != 0;
#else
  constexpr bool STORE_AWARE = false;
#endif
  constexpr int BM = G::BM, BN = G::BN, A_BYTES = G::A_BYTES, STAGE = G::STAGE, NA = G::NA, NB = G::NB, LPT = G::LPT;
  __shared__ __attribute__((aligned(1024))) char smem_raw[S * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keep it (and all it feeds) scalar
  const int wm = wave >> 1, wn = wave & 1;
  const bool producer = NP > 0 && wave >= G::NWAVES;              // wave-uniform
  const bool issues = NP == 0 || producer;
  const int dw = NP > 0 ? wave - G::NWAVES : wave;                // index among the issuing waves
#ifdef CSTS_GEMM4_STAMPS
  const bool stamp_on = p.stamps != nullptr && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 137);
  unsigned long long* stamp_buf = p.stamps + (blockIdx.x == 0 ? 0 : 512);
  int nstamp = 1;
  if (stamp_on) stamp_buf[500] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
#endif
  G4_STAMP();
  const int nk = (int)((p.K + BK2 - 1) / BK2), nk_full = (int)(p.K / BK2), k_tail = KTAIL ? (int)((p.K % BK2) >> 4) : 0;

  // this workgroup's tiles: XCD x (= blockIdx.x % 8: workgroups are dealt round-robin over the XCDs) owns a contiguous
  // range of tiles; its workgroups walk the range with stride = workgroups on the XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, stride = gridDim.x >> 3;
  const int64_t tq = p.ntiles >> 3, tr = p.ntiles & 7;
  const int64_t t_beg = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq), t_end = t_beg + tq + (xcd < tr ? 1 : 0);
  const int64_t t_first = t_beg + slot;
  const int my_tiles = t_first < t_end ? (int)((t_end - t_first + stride - 1) / stride) : 0;
  const int total_k = my_tiles * nk;

  // ---- producer side: per-lane source offsets (bytes, relative to the tile's first row) of the pieces this wave stages
  // (8 rows x 128 B per wave-instruction); chunk slot (lane & 7) of row r holds k-chunk (lane & 7) ^ ((r >> 1) & 7)
  unsigned aoff[NA], boff[NB];
  const char* abase = nullptr;                       // wave-uniform: first row of the current producer tile
  const char* bbase = nullptr;
  auto chunk = [&](int piece) { return ((lane & 7) ^ (((piece * 8 + (lane >> 3)) >> 1) & 7)) * 16; };   // bytes
  auto set_tile = [&](int64_t t) {
    const int64_t m0 = (t / p.ntiles_n) * BM, n0 = (t % p.ntiles_n) * BN;
    abase = reinterpret_cast<const char*>(p.A) + m0 * p.lda * 2;
    bbase = reinterpret_cast<const char*>(p.B) + n0 * p.ldb * 2;
    const int ra = (int)min((int64_t)BM, p.M - m0) - 1, rb = (int)min((int64_t)BN, p.N - n0) - 1;   // last valid row of the tile
#pragma unroll
    for (int i = 0; i < NA; ++i) aoff[i] = (unsigned)(min((dw * NA + i) * 8 + (lane >> 3), ra) * (int)p.lda * 2 + chunk(dw * NA + i));
#pragma unroll
    for (int i = 0; i < NB; ++i) boff[i] = (unsigned)(min((dw * NB + i) * 8 + (lane >> 3), rb) * (int)p.ldb * 2 + chunk(dw * NB + i));
  };
  int64_t pt = t_first;                              // producer position: tile, k-tile inside it, flattened index
  int pk = 0, pidx = 0, pstage = 0;
  auto issue = [&](int i, char* st, int64_t k0, bool full) {      // LDS-DMA instruction i of this thread's share of a k-tile
    const int piece = (i < NA) ? dw * NA + i : dw * NB + (i - NA);
    unsigned off = (i < NA) ? aoff[i] : boff[i - NA];
    if (KTAIL && !full && k0 + chunk(piece) / 2 >= p.K) off -= chunk(piece);   // chunks past K are fetched from chunk 0 (never used)
    const char* src = ((i < NA) ? abase : bbase) + k0 * 2 + off;
    char* dst = st + ((i < NA) ? 0 : A_BYTES) + piece * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
  };
  auto advance = [&]() {
    ++pidx;
    pstage = (pstage + 1 == S) ? 0 : pstage + 1;
    if (++pk == nk) {
      pk = 0;
      pt += stride;
      if (pidx < total_k) set_tile(pt);
    }
  };
  if (my_tiles > 0 && issues) set_tile(pt);
#pragma unroll
  for (int s = 0; s < S - 1; ++s) {
    if (issues && pidx < total_k) {
      char* st = smem_raw + pstage * STAGE;
      const int64_t k0 = (int64_t)pk * BK2;
      const bool full = !KTAIL || k0 + BK2 <= p.K;
#pragma unroll
      for (int i = 0; i < LPT; ++i) issue(i, st, k0, full);
      advance();
    }
  }

  G4_STAMP();      // prologue issued
  if constexpr (NP > 0) {
    if (producer) {
      // ---- producer waves: one barrier per k-tile, exactly as the consumers' loop below.  k-tile c has landed once at most
      // `ahead` younger k-tiles of this wave's own pieces are outstanding (nothing else is ever on this wave's counter).
      for (int c = 0; c < total_k; ++c) {
        const int ahead = min(S - 2, total_k - 1 - c);
        if (S >= 4 && ahead == 2) wait_vm<2 * LPT>();
        else if (S >= 3 && ahead >= 1) wait_vm<LPT>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                // this k-tile is complete; everyone is done with the stage refilled next
        if (pidx < total_k) {
          char* st = smem_raw + pstage * STAGE;
          const int64_t k0 = (int64_t)pk * BK2;
#pragma unroll
          for (int i = 0; i < LPT; ++i) issue(i, st, k0, true);
          advance();
        }
      }
      return;
    }
  }
  // ---- consumer side.  Fragment row = obase + (lane & 31) with obase % 32 == 0: the swizzle key is a per-lane constant
  const int key = (lane >> 1) & 7, hi = lane >> 5;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = (lane & 31) * 128 + (((ks * 2 + hi) ^ key) << 4);
  int cidx = 0, cstage = 0;

  for (int64_t t = t_first; t < t_end; t += stride) {
    // The MFMA operands are SWAPPED (weights as the row operand): acc[mi][ni][r] is C[m][n] with m = the lane's row
    // (lane & 31) of 32-row unit mi and n = 32 * ni + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3) -- every lane owns runs of
    // 4 consecutive columns of ONE output row, so the epilogue needs no LDS transpose and no barrier.
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int64_t m0 = (t / p.ntiles_n) * BM + wm * 32 * MT + (lane & 31), n0 = (t % p.ntiles_n) * BN + wn * 32 * NT + 4 * hi;

    auto kstep = [&](auto nks_tag, int kt) {
      constexpr int NKS = decltype(nks_tag)::value;
      // k-tile cidx has landed once at most `ahead` younger k-tiles (LPT instructions each) are outstanding -- PLUS, for the
      // first S - 1 k-tiles of a tile (their loads were issued before the previous tile's epilogue), that epilogue's stores:
      // vector memory retires in order on one counter, so a count without them drained every store of the previous tile
      // before the first MFMA of the next one (EPI_ST is exact for the straight-line forms; 0 = over-wait for the generic one)
      if constexpr (NP == 0) {
        constexpr int EPI_ST = (FORM == 0 || FORM == 5) ? 0 : MT * NT * 2;      // a lower bound (FORM 2 stores twice as many when it keeps the pre-activation): under-counting only over-waits
        static_assert(2 * LPT + EPI_ST <= 63, "vmcnt immediate");
        const int ahead = min(S - 2, total_k - 1 - cidx);
        if (EPI_ST > 0 && STORE_AWARE && t != t_first && kt < S - 1) {
          if (S >= 4 && ahead == 2) wait_vm<2 * LPT + EPI_ST>();
          else if (S >= 3 && ahead >= 1) wait_vm<LPT + EPI_ST>();
          else wait_vm<EPI_ST>();
        } else {
          if (S >= 4 && ahead == 2) wait_vm<2 * LPT>();
          else if (S >= 3 && ahead >= 1) wait_vm<LPT>();
          else wait_vm<0>();
        }
      }
      G4_STAMP();      // own share landed
      __builtin_amdgcn_s_barrier();                  // every wave's share landed; everyone is done with the previous stage
      G4_STAMP();      // barrier passed
      const char* As = smem_raw + cstage * STAGE;
      const char* Bs = As + A_BYTES;
      cstage = (cstage + 1 == S) ? 0 : cstage + 1;
      ++cidx;
      const bool refill = NP == 0 && pidx < total_k; // wave-uniform
      char* st = smem_raw + pstage * STAGE;
      const int64_t k0 = (int64_t)pk * BK2;
      const bool full = !KTAIL || k0 + BK2 <= p.K;
      constexpr int DB = G::DB;
      bf16x8 fa[DB][MT], fb[DB][NT];
      auto frags = [&](int ks, int buf) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          fb[buf][j] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 32 * NT + j * 32) * 128 + foff[ks]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
          fa[buf][i] = *reinterpret_cast<const bf16x8*>(As + (wm * 32 * MT + i * 32) * 128 + foff[ks]);
      };
      frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (DB == 2 && ks + 1 < NKS) frags(ks + 1, (ks + 1) & 1);
        if (refill) {
#pragma unroll
          for (int i = 0; i < LPT; ++i) {
            // (front-loading the refill at ks == 0 measured 4-8 % slower on the 2-stage variants: the issue cost of 5-8
            // LDS-DMA instructions in a row is not hidden even with a partner wave)
            const bool mine = (NKS == 4) ? (i * 4 / LPT == ks) : (ks == 0);
            if (mine) issue(i, st, k0, full);
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if constexpr (FORM == 5) acc[i][j] = CSTS_MFMA16(fa[ks & (DB - 1)][i], fb[ks & (DB - 1)][j], acc[i][j], 0, 0, 0);   // rows = tokens: see the FORM 5 epilogue
            else acc[i][j] = CSTS_MFMA16(fb[ks & (DB - 1)][j], fa[ks & (DB - 1)][i], acc[i][j], 0, 0, 0);
        if (DB == 1 && ks + 1 < NKS) frags(ks + 1, 0);
      }
      if (refill) advance();
      G4_STAMP();      // fragment reads + MFMAs + refill issue of this k-step
    };
    for (int kt = 0; kt < nk_full; ++kt) kstep(std::integral_constant<int, 4>(), kt);
    if constexpr (KTAIL) {                                           // K % 64 = 16, 32 or 48 (K % 16 == 0)
      if (k_tail == 1) kstep(std::integral_constant<int, 1>(), nk_full);
      else if (k_tail == 2) kstep(std::integral_constant<int, 2>(), nk_full);
      else if (k_tail == 3) kstep(std::integral_constant<int, 3>(), nk_full);
    }

    G4_STAMP();        // epilogue begins
    // FAST PATH (whole tiles, bf16 C, no residual / row scale: every library-picked launch except ragged edge tiles).
    // The generic code below tests bounds and epilogue kinds per 4-column run; the compiler turns that into one branch and
    // one `s_waitcnt vmcnt(0)` per bias / aux load -- twelve dependent round trips per tile, each of which also drains every
    // store issued before it (vector memory retires in order).  In-kernel stamps (tools/gemm4_stamps.py, 8192 x 1536 x 384):
    // 8.5 k cycles of epilogue against 5.9 k of MFMA work per tile, 19.8 k with GELU.  Here the epilogue kind is a compile-time
    // case, the code is straight-line, and the loads of column unit ni + 1 are issued BEFORE the stores of unit ni, so the wait
    // for them is a counted one that leaves the stores in flight.  Same arithmetic, same order: bit-identical results.
    if constexpr (FORM == 5) {
      // fp32 residual-stream form (proj / fc2 forward: C = (acc + bias) * row_scale + residual, all fp32, whole tiles).  The MFMA
      // operands are NOT swapped here: accumulator r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 of its
      // 32 x 32 block, so every load / store instruction of a half-wave covers one whole 128-byte run of a row -- the register
      // epilogue of the swapped layout reached fp32 rows in 32-byte segments, which is why this family lost to gemm2's LDS-staged
      // epilogue on these shapes (profiles/r2_gemm4_instep_ab.txt).  The residual runs of column block ni + 1 are requested before
      // the stores of block ni (vector memory retires in order).  Same arithmetic order as gemm2: bit-identical.
      float* __restrict__ Cf = reinterpret_cast<float*>(p.C);
      const float* __restrict__ Rf = reinterpret_cast<const float*>(p.residual);
      const int64_t mb = m0 - (lane & 31) + 4 * hi, nb = n0 - 4 * hi + (lane & 31);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int64_t mrow = mb + mi * 32;
        float sc[16];
        if (p.row_scale != nullptr) {
          const int64_t s_lo = (mrow - 4 * hi) / p.rows_per_scale, s_hi = (mrow - 4 * hi + 31) / p.rows_per_scale;      // wave-uniform
          if (s_lo == s_hi) {
            const float v = p.row_scale[s_lo];
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = v;
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = p.row_scale[(mrow + (r & 3) + 8 * (r >> 2)) / p.rows_per_scale];
          }
        }
        float rr[2][16];
        const int64_t nwave = n0 - 4 * hi;                      // wave-uniform: first column of this wave's 32-column blocks
        auto loads = [&](int ni, int buf) {                      // (N % 32 == 0: a block is wholly inside the matrix or wholly outside)
          if (nwave + ni * 32 >= p.N) return;
#pragma unroll
          for (int r = 0; r < 16; ++r) rr[buf][r] = Rf[(mrow + (r & 3) + 8 * (r >> 2)) * p.ldr + nb + ni * 32];
        };
        loads(0, 0);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const int cur = ni & 1;
          if (nwave + ni * 32 >= p.N) break;
          const float bq = p.bias != nullptr ? p.bias[nb + ni * 32] : 0.f;
          if (ni + 1 < NT) loads(ni + 1, cur ^ 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[mi][ni][r] + bq;
            if (p.row_scale != nullptr) v *= sc[r];
            Cf[(mrow + (r & 3) + 8 * (r >> 2)) * p.ldc + nb + ni * 32] = v + rr[cur][r];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if constexpr (FORM != 0) {
      bf16* __restrict__ Cb = reinterpret_cast<bf16*>(p.C);
      bf16* __restrict__ Xb = reinterpret_cast<bf16*>(p.aux);
      const int hi16 = hi << 4;
      // forward form: bias (+ GELU with the pre-activation written to aux).  A lane needs bias[c .. c + 3] for the sixteen
      // 4-column runs it owns -- 48 floats per tile; instead of 12 float4 loads per lane, lane l fetches bias[unit column
      // (l & 31)] once per 32-column unit and the values travel by ds_bpermute (LDS crossbar, not vector memory)
      auto fwd = [&](auto gelu_tag) {
        constexpr bool GELU = decltype(gelu_tag)::value;
        const float* __restrict__ bias = p.bias;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int64_t m = m0 + mi * 32;
          float bu[NT];
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) bu[ni] = bias[n0 - 4 * hi + ni * 32 + (lane & 31)];
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {          // a pair of runs = one 16-byte store: eight values live at a time
              f32x4 o[2];
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const int q = 2 * pr + qq;
                  const float bq = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(hi16 + 4 * (8 * q + j), __builtin_bit_cast(int, bu[ni])));
                  o[qq][j] = bq + acc[mi][ni][4 * q + j];
                }
              if constexpr (GELU) {
                if (Xb != nullptr) st4x2_bf16_whole(Xb, m * p.ldaux + n0 + ni * 32, o[0], o[1], pr, hi);
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                  for (int j = 0; j < 4; ++j) o[qq][j] = gelu_fast(o[qq][j]);
              }
              st4x2_bf16_whole(Cb, m * p.ldc + n0 + ni * 32, o[0], o[1], pr, hi);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      };
      // data-gradient form: no bias; DGELU multiplies by gelu'(aux): the aux runs of unit ni + 1 are requested before the
      // stores of unit ni
      auto bwd = [&](auto dgelu_tag) {
        constexpr bool DGELU = decltype(dgelu_tag)::value;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int64_t m = m0 + mi * 32;
          bf16x4 hv[2][4];
          auto loads = [&](int ni, int buf) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[buf][q] = *reinterpret_cast<const bf16x4*>(Xb + m * p.ldaux + n0 + ni * 32 + 8 * q);
          };
          if constexpr (DGELU) loads(0, 0);
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) {
            const int cur = ni & 1;
            if constexpr (DGELU) { if (ni + 1 < NT) loads(ni + 1, cur ^ 1); }
            f32x4 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float v = 0.f + acc[mi][ni][4 * q + j];
                if constexpr (DGELU) v *= dgelu_fast((float)hv[cur][q][j]);
                o[q][j] = v;
              }
              if constexpr (DGELU) __builtin_amdgcn_sched_barrier(0);
            }
            st4x4_bf16_whole(Cb, m * p.ldc + n0 + ni * 32, o, hi);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      if constexpr (FORM == 2) fwd(std::true_type());
      else if constexpr (FORM == 1) fwd(std::false_type());
      else if constexpr (FORM == 4) bwd(std::true_type());
      else bwd(std::false_type());
    } else {
    // ---------------- epilogue straight from the accumulators (same arithmetic order as gemm2_kernel / gemm3_kernel)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int64_t m = m0 + mi * 32;
      const bool mok = m < p.M;
      float rsc = 1.f;
      if (p.row_scale != nullptr && mok) rsc = p.row_scale[m / p.rows_per_scale];
      const int64_t rm = (p.residual != nullptr && p.res_row_mod > 0) ? (m % p.res_row_mod) : m;
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        f32x4 o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t c = n0 + ni * 32 + 8 * q;
          const bool ok = mok && c < p.N;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (p.bias != nullptr && c < p.N) v = *reinterpret_cast<const f32x4*>(p.bias + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += acc[mi][ni][4 * q + j];
          if (p.epilogue == CSTS_EPI_DGELU) {
            f32x4 h = {0.f, 0.f, 0.f, 0.f};
            if (ok) h = ld4_as_f32(p.aux, p.aux_dt, m * p.ldaux + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= (p.aux_dt == CSTS_BF16) ? dgelu_fast(h[j]) : dgelu_f(h[j]);
          }
          o[q] = v;
        }
        if (p.epilogue == CSTS_EPI_GELU) {
          if (p.aux != nullptr) st4x4(p.aux, p.aux_dt, m * p.ldaux + n0 + ni * 32, o, mok, n0 + ni * 32, p.N, hi);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[q][j] = (p.aux_dt == CSTS_BF16) ? gelu_fast(o[q][j]) : gelu_f(o[q][j]);
        }
        if (p.row_scale != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] *= rsc;
        }
        if (p.residual != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t c = n0 + ni * 32 + 8 * q;
            if (mok && c < p.N) o[q] += ld4_as_f32(p.residual, p.r_dt, rm * p.ldr + c);
          }
        }
        st4x4(p.C, p.c_dt, m * p.ldc + n0 + ni * 32, o, mok, n0 + ni * 32, p.N, hi);
      }
    }
    }
    G4_STAMP();        // epilogue issued (stores in flight)
  }
  wait_vm<0>();
  G4_STAMP();          // everything drained
#ifdef CSTS_GEMM4_STAMPS
  if (stamp_on) { stamp_buf[0] = (unsigned long long)nstamp; stamp_buf[501] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); }
#endif
}

// which epilogue form (see gemm4_kernel) a problem takes on BM x BN tiles
int gemm4_form(const Params& p, int BM, int BN) {
  if (p.M % BM == 0 && p.N % 32 == 0 && p.c_dt == CSTS_F32 && p.residual != nullptr && p.r_dt == CSTS_F32 && p.res_row_mod == 0 && p.ru_To == 0 &&
      p.epilogue == CSTS_EPI_NONE && p.split_k <= 1)
    return 5;
  if (p.M % BM != 0 || p.N % BN != 0 || p.c_dt != CSTS_BF16 || p.residual != nullptr || p.row_scale != nullptr) return 0;
  if (p.bias != nullptr) {
    if (p.epilogue == CSTS_EPI_NONE) return 1;
    if (p.epilogue == CSTS_EPI_GELU && (p.aux == nullptr || p.aux_dt == CSTS_BF16)) return 2;     // aux == NULL: inference, the pre-activation is not kept
    return 0;
  }
  if (p.epilogue == CSTS_EPI_NONE) return 3;
  if (p.epilogue == CSTS_EPI_DGELU && p.aux != nullptr && p.aux_dt == CSTS_BF16) return 4;
  return 0;
}

template <int WM, int MT, int NT, int S, bool FORMS, int NP = 0>
bool launch4(Params p, int wpc, hipStream_t s) {
  typedef G4<WM, MT, NT, S, NP> G;
  if (p.K % BK2 != 0) return false;
  const int64_t ntiles = cdiv(p.M, G::BM) * cdiv(p.N, G::BN);
  p.ntiles = ntiles;
  p.ntiles_n = (int)cdiv(p.N, G::BN);
  if (wpc <= 0 || wpc > G::WG) wpc = G::WG;
  const int64_t g = std::min<int64_t>(cdiv(ntiles, 8) * 8, (int64_t)256 * wpc);
  const dim3 grid((unsigned)g), block(G::NTHR);
  const int form = FORMS ? gemm4_form(p, G::BM, G::BN) : 0;
  if constexpr (FORMS) {
    if (form == 1) { hipLaunchKernelGGL((gemm4_kernel<WM, MT, NT, S, 1, NP>), grid, block, 0, s, p); return true; }
    if (form == 2) { hipLaunchKernelGGL((gemm4_kernel<WM, MT, NT, S, 2, NP>), grid, block, 0, s, p); return true; }
    if (form == 3) { hipLaunchKernelGGL((gemm4_kernel<WM, MT, NT, S, 3, NP>), grid, block, 0, s, p); return true; }
    if (form == 4) { hipLaunchKernelGGL((gemm4_kernel<WM, MT, NT, S, 4, NP>), grid, block, 0, s, p); return true; }
    if (form == 5) { hipLaunchKernelGGL((gemm4_kernel<WM, MT, NT, S, 5, NP>), grid, block, 0, s, p); return true; }
  }
  hipLaunchKernelGGL((gemm4_kernel<WM, MT, NT, S, 0, NP>), grid, block, 0, s, p);
  return true;
}

struct Variant { int code, wm, mt, nt, s; };
// shape ids: 0 = 256x128, 1 = 256x192, 2 = 256x256 (8 waves, wave tile 64 x 64/96/128);
//            3 = 128x128, 6 = 128x192, 4 = 128x256 (8 waves, wave tile 32 x 64/96/128); 5 = 128x128 on 4 waves (wave tile 64 x 64);
//            7 = 64x192 on 4 waves (wave tile 32 x 96; round 5: M = 8192 problems with 128 or fewer 128 x 192 tiles -- half the CUs idle);
// (4-wave workgroups with 64 x 96 / 64 x 128 / 128 x 64 wave tiles -- fewer LDS fragment bytes per MFMA -- measured 10-90 %
//  slower than the 8-wave 128 x 192 variant on every CSTS shape and were removed: profiles/r2_gemm4_lab2.txt)
// last column: the specialised epilogue forms exist (the variants the library picks by itself)
#define G4_VARIANTS(X) \
  X(2, 4, 2, 2, 2, false) X(3, 4, 2, 2, 3, false) X(12, 4, 2, 3, 2, false) X(22, 4, 2, 4, 2, false) \
  X(32, 4, 1, 2, 2, true) X(33, 4, 1, 2, 3, true) X(34, 4, 1, 2, 4, true) X(62, 4, 1, 3, 2, true) X(63, 4, 1, 3, 3, true) X(64, 4, 1, 3, 4, true) X(42, 4, 1, 4, 2, false) \
  X(52, 2, 2, 2, 2, false) X(72, 2, 1, 3, 2, true) X(73, 2, 1, 3, 3, true) X(74, 2, 1, 3, 4, true)

}  // namespace

// wave-specialised variants (4 producer waves, 3-stage ring, one workgroup per CU): 83 = 128 x 192, 84 = 128 x 128
bool csts_gemm4_launch(const csts_gemm_params& p0, const csts_gemm_args* a, int variant, int wpc, hipStream_t s) {
  csts_gemm_params p = p0;
  // OFF by default: measured neutral on the 2-stage ring (21.63 vs 21.63 ms per step, profiles/r4_gemm4_store_aware_ab.txt) and
  // mixed with the 4-stage short-K ring (CSTS_GEMM4_SHORTK=64: -16 .. -18 % on two shapes, +6 .. +10 % on the DGELU forms, the
  // largest fc1 + GELU unchanged at 2.96 TB/s) -- the drain of the previous tile's stores is NOT what holds these GEMMs at 3 TB/s
  static const int store_aware = [] { const char* e = getenv("CSTS_GEMM4_STORE_AWARE"); return (e && e[0] == '1') ? 1 : 0; }();
  p.store_aware = store_aware;
  p.stamps = (a != nullptr && a->workspace != nullptr && a->ws_bytes >= 8192) ? reinterpret_cast<unsigned long long*>(a->workspace) : nullptr;
  if (variant == 83) return launch4<4, 1, 3, 3, true, 4>(p, 1, s);
  if (variant == 84) return launch4<4, 1, 2, 3, true, 4>(p, 1, s);
#define X(code, wm, mt, nt, st, forms) if (variant == code) return launch4<wm, mt, nt, st, forms>(p, wpc, s);
  G4_VARIANTS(X)
#undef X
  return false;
}

// the kernel csts_gemm4_launch starts for these parameters, as rocprofv3 prints it
bool csts_gemm4_name(const csts_gemm_params& p, int variant, char* buf, int buflen) {
  if (variant == 83 || variant == 84) {
    const int nt = variant == 83 ? 3 : 2;
    snprintf(buf, buflen, "gemm4_kernel<4, 1, %d, 3, %d, 4>", nt, gemm4_form(p, 128, 64 * nt));
    return true;
  }
#define X(code, wm, mt, nt, st, forms) if (variant == code) { \
    snprintf(buf, buflen, "gemm4_kernel<%d, %d, %d, %d, %d, 0>", wm, mt, nt, st, forms ? gemm4_form(p, wm * 32 * mt, 64 * nt) : 0); return true; }
  G4_VARIANTS(X)
#undef X
  return false;
}
