// Grouped weight gradients on 8-wave workgroups: dW[M,N] tiles of 192 x 384 (M = the Linear's output features, N = its
// input features), one (tile, token-chunk) work item per workgroup, same item table as wgrad_grouped_kernel (gemm.hip).
//
// Why this shape and this structure (profiles/r2_wgrad8_ab.txt, r2_wgrad_ladder.txt): the 128 x 128 items of
// wgrad_grouped_kernel run at ~750 TF/s with 35-50 % of their L2 requests hitting, i.e. ~6.3 TB/s of L2-miss traffic -- they
// sit on the fabric / HBM bandwidth at 64 FLOP per staged byte.  A 192 x 384 tile has 128 FLOP per staged byte and needs no
// luck in L2: the reuse is inside the workgroup.  A first version of this kernel staged the k-tile through registers
// (global_load -> ds_write_b128 between two barriers) and was exactly as fast as the kernel it replaced; the TN k-loop
// calibrated in isolation runs at 1010 TF/s from L2 when the k-tile comes in by LDS-DMA instead.  So: LDS-DMA
// (global_load_lds_dwordx4) into a 2-stage ring, one raw barrier per 64-token k-tile, no ds_write at all.
//
// TN form, bf16 x bf16: both operands are token-major, a k-tile is kept [token][feature], UNPADDED (the DMA image is
// lane-linear), and fed to v_mfma_f32_32x32x16_bf16 through ds_read_b64_tr_b16.  Bank conflicts of the transposing reads are
// removed by rotating the 16-byte chunks of a token row -- by 4 chunks x ((row >> 1) & 1) in the 192-wide image (row stride
// 384 B), by 4 x (row & 3) in the 384-wide one (768 B) -- applied to the per-lane SOURCE address of the DMA and, identically,
// to the fragment read (checked exhaustively: every 32-lane read group touches 64 distinct banks).
// Wave grid 2 x 4, wave tile 96 x 96 (9 accumulators).  The fp32 tile goes straight from the accumulators to memory as
// 16-byte stores (MFMA operands swapped so that a lane owns runs of 4 consecutive columns).  All Linear layers of the 384- and 768-channel stages
// have output features % 192 == 0 and input features % 384 == 0; token chunks must be multiples of 64 (host-checked).
#include "common.h"
#include <algorithm>

// Diagnostics build (-DCSTS_WGRAD8_STAMPS, `make stamps`, tools/wgrad8_stamps.py): thread 0 of workgroup 0 records shader-clock
// stamps around the phases of every k-tile.  No stamp code exists in the library build.
#ifdef CSTS_WGRAD8_STAMPS
__device__ unsigned long long g_w8_stamps[4096];
#define W8_STAMP() do { if (stamp_on && nstamp < 4000) g_w8_stamps[nstamp++] = (unsigned long long)__builtin_amdgcn_s_memtime(); } while (0)
extern "C" int csts_debug_wgrad8_stamps(unsigned long long* dst_host) {
  return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_w8_stamps), sizeof(unsigned long long) * 4096);
}
#else
#define W8_STAMP() do { } while (0)
#endif

namespace {

constexpr int W8_TM = 192, W8_TN = 384, W8_THR = 512;
constexpr int W8_AROW = W8_TM * 2, W8_BROW = W8_TN * 2;                  // bytes per token row: 384, 768
// ring geometry: BK tokens per stage, S stages.  <64, 2> (default): 72 KiB stages, one k-tile in flight beside the one being
// consumed; <32, 4>: 36 KiB stages, three in flight (108 KiB) -- measured slower, see csts_wgrad_grouped8.
template <int BK, int S> struct W8 {
  static constexpr int A_BYTES = BK * W8_AROW, B_BYTES = BK * W8_BROW, STAGE = A_BYTES + B_BYTES;
  static constexpr int NPIECE = STAGE / 1024;                              // 1-KiB LDS-DMA pieces per stage: 72 / 36
  static constexpr int NPW = (NPIECE + 7) / 8;                             // pieces per wave (the last one only for waves < NPIECE % 8)
  static constexpr int REM = NPIECE % 8;                                   // 0: every wave issues NPW; else waves >= REM issue NPW - 1
  static_assert(S * STAGE <= 160 * 1024, "ring exceeds the CU's LDS");
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// MFMA 32x32x16 fragment of output rows obase .. obase + 31 of an operand kept [token][feature] (rotated chunks, see above)
template <bool IS_A>
__device__ __forceinline__ bf16x8 frag_rot(const char* S, int obase, int ks, int lane) {
  constexpr int ROWB = IS_A ? W8_AROW : W8_BROW, NCH = ROWB / 16;
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int row = ks * 16 + 8 * (g >> 1) + q, col = obase + 16 * (g & 1) + 4 * pp;
  int ch = (col >> 3) + (IS_A ? 4 * ((row >> 1) & 1) : 4 * (row & 3));        // row + 4 has the same rotation
  if (ch >= NCH) ch -= NCH;
  const char* a = S + row * ROWB + ch * 16 + (col & 7) * 2;
  const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * ROWB));
  const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0), b1 = __builtin_bit_cast(bf16x4, t1);
  bf16x8 r;
  r[0] = b0[0]; r[1] = b0[1]; r[2] = b0[2]; r[3] = b0[3];
  r[4] = b1[0]; r[5] = b1[1]; r[6] = b1[2]; r[7] = b1[3];
  return r;
}

template <int N> __device__ __forceinline__ void w8_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BK, int S>
__global__ __launch_bounds__(W8_THR, 2) void wgrad8_kernel(const csts_wgrad_item* __restrict__ items, int nitems) {
  typedef W8<BK, S> G;
  constexpr int A_BYTES = G::A_BYTES, STAGE = G::STAGE, NPW = G::NPW, REM = G::REM;
  __shared__ __attribute__((aligned(1024))) char smem[S * STAGE];
  // Round 5: a workgroup walks items blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x a multiple of 8, so an item keeps the XCD its
  // position in the table was chosen for).  With gridDim.x == nitems this is the one-item-per-workgroup launch of rounds 2-4; with fewer
  // workgroups the launch occupies only that many CUs (its 144 KB of LDS leave room for nothing else on a CU) and can run BESIDE the
  // memory-bound end of the backward chain instead of after it (csts_wgrad_grouped8_limited, ops.flush_wgrads(early192=True)).
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
  if (item != (int)blockIdx.x) __syncthreads();   // everyone is done with the previous item's LDS (ring and column-sum scratch)
  const csts_wgrad_item it = items[item];
  if (it.A == nullptr) continue;        // padding slot (the host equalises the per-XCD lists); block-uniform
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t m0 = it.m0, n0 = it.n0, kbeg = it.kbeg;
  const int nk = (int)((it.kend - kbeg) / BK);             // whole k-tiles only (host-checked: token ranges are multiples of 64)
  const bool short_wave = REM != 0 && wave >= REM;         // wave-uniform: this wave issues NPW - 1 pieces per stage

  // ---- producer: piece p = wave + 8 i of the stage image (A rows first, then B rows) belongs to this wave; this lane's
  // source offset (elements, relative to the k-tile's first token row) of each.  LDS byte `off` of an image holds token row
  // off / ROWB, rotated chunk (off % ROWB) / 16; the lane fetches the chunk that belongs there.
  unsigned soff[NPW];                                         // BYTE offsets
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = min(wave + 8 * i, G::NPIECE - 1);
    if (p * 1024 < A_BYTES) {
      const int off = p * 1024 + lane * 16, row = off / W8_AROW, chp = (off % W8_AROW) / 16;
      int c = chp - 4 * ((row >> 1) & 1);
      if (c < 0) c += W8_AROW / 16;
      soff[i] = (unsigned)(row * (int)it.lda + (int)m0 + c * 8) * 2u;
    } else {
      const int off = p * 1024 - A_BYTES + lane * 16, row = off / W8_BROW, chp = (off % W8_BROW) / 16;
      int c = chp - 4 * (row & 3);
      if (c < 0) c += W8_BROW / 16;
      soff[i] = (unsigned)(row * (int)it.ldb + (int)n0 + c * 8) * 2u;
    }
  }
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(it.A);
  const bf16* __restrict__ B = reinterpret_cast<const bf16*>(it.B);
  // one LDS-DMA wave instruction as scalar base + 32-bit lane offset (the saddr form), M0 = the piece's LDS address.  Written
  // out: through the builtin the compiler formed a 64-bit lane address per piece (two v_lshl_add_u64 + readfirstlane per
  // instruction) and the nine instructions of a refill took ~700 ticks (in-kernel stamps, tools/wgrad8_stamps.py).
  auto issue = [&](char* st, int64_t k0) {
    const char* a = reinterpret_cast<const char*>(A + k0 * it.lda);
    const char* b = reinterpret_cast<const char*>(B + k0 * it.ldb);
    const uint32_t l0 = (uint32_t)(uintptr_t)(lptr_t)st;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      if (i == NPW - 1 && short_wave) break;
      const int p = wave + 8 * i;                              // scalar
      const char* src = (p * 1024 < A_BYTES) ? a : b;
      const uint32_t l = l0 + (uint32_t)p * 1024u;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(soff[i]), "s"(src) : "memory", "m0");
    }
  };
  // k-tile kt has landed once at most `ahead` younger k-tiles of this wave's pieces are outstanding
  auto wait_tile = [&](int ahead) {
    if (S >= 4 && ahead >= 2) { if (short_wave) w8_wait<2 * (NPW - 1)>(); else w8_wait<2 * NPW>(); }
    else if (S >= 3 && ahead == 1) { if (short_wave) w8_wait<NPW - 1>(); else w8_wait<NPW>(); }
    else w8_wait<0>();
  };

  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fused bias gradient (n0 == 0 tiles): 384 threads own a column pair of the dY tile and a quarter of a stage's token rows
  const bool do_colsum = it.colsum != nullptr && n0 == 0 && tid < 384;
  const int cp = tid % 96, csl = tid / 96;
  float cs0 = 0.f, cs1 = 0.f;

  // prologue: S - 1 stages in flight
#pragma unroll
  for (int s2 = 0; s2 < S - 1; ++s2)
    if (s2 < nk) issue(smem + s2 * STAGE, kbeg + (int64_t)s2 * BK);
  int cs = 0, ps = (S - 1) % S;                               // consumer stage, producer stage
#ifdef CSTS_WGRAD8_STAMPS
  const bool stamp_on = tid == 0 && item == 0;
  int nstamp = 1;
#endif
  W8_STAMP();
  for (int kt = 0; kt < nk; ++kt) {
    wait_tile(min(S - 2, nk - 1 - kt));                      // this wave's pieces of k-tile kt have landed
    W8_STAMP();
    __builtin_amdgcn_s_barrier();                            // ... everyone's have, and everyone is done with the stage refilled next
    W8_STAMP();
    const char* As = smem + cs * STAGE;
    const char* Bs = As + A_BYTES;
    if (kt + S - 1 < nk) issue(smem + ps * STAGE, kbeg + (int64_t)(kt + S - 1) * BK);
    W8_STAMP();
    cs = (cs + 1 == S) ? 0 : cs + 1;
    ps = (ps + 1 == S) ? 0 : ps + 1;
    if (do_colsum) {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        const int row = csl * (BK / 4) + kk;
        int ch = ((2 * cp) >> 3) + 4 * ((row >> 1) & 1);
        if (ch >= W8_AROW / 16) ch -= W8_AROW / 16;
        const bf16x2 t = *reinterpret_cast<const bf16x2*>(As + row * W8_AROW + ch * 16 + ((2 * cp) & 7) * 2);
        cs0 += (float)t[0];
        cs1 += (float)t[1];
      }
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[3], b[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = frag_rot<true>(As, wm * 96 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = frag_rot<false>(Bs, wn * 96 + j * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = CSTS_MFMA16(b[j], a[i], acc[i][j], 0, 0, 0);   // operands swapped: see the epilogue
    }
    W8_STAMP();
  }
#ifdef CSTS_WGRAD8_STAMPS
  if (stamp_on) g_w8_stamps[0] = nstamp;
#endif

  if (it.colsum != nullptr && n0 == 0) {   // block-uniform: fold the four token slices of every column (fixed order)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);               // [4][192]
    if (tid < 384) { red[csl * W8_TM + 2 * cp] = cs0; red[csl * W8_TM + 2 * cp + 1] = cs1; }
    __syncthreads();
    if (tid < W8_TM && m0 + tid < it.M) it.colsum[m0 + tid] = (red[tid] + red[W8_TM + tid]) + (red[2 * W8_TM + tid] + red[3 * W8_TM + tid]);
  }

  // ---------------- epilogue.  The MFMA operands are swapped (X as the row operand), so acc[i][j][r] is dW[m][n] with m = the
  // lane's row (lane & 31) of unit i and n = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of unit j: four consecutive columns per
  // register quad -> 16-byte stores (36 per thread instead of 144 dword stores; a 295 KB tile per item is a sixth of a
  // 2048-token item's life otherwise)
  const int hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int64_t m = m0 + wm * 96 + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int64_t n = n0 + wn * 96 + j * 32 + 8 * t + 4 * hi;
        if (m < it.M && n < it.N)
          *reinterpret_cast<f32x4*>(it.C + m * it.ldc + n) = f32x4{acc[i][j][4 * t], acc[i][j][4 * t + 1], acc[i][j][4 * t + 2], acc[i][j][4 * t + 3]};
      }
  }
  }   // item loop
}

}  // namespace

static int wgrad8_launch(const csts_wgrad_item* device_items, int nitems, int max_wgs, hipStream_t stream) {
  CSTS_REQUIRE(device_items != nullptr && nitems > 0, "no items");
  // CSTS_WGRAD8_RING=4: four stages of 32 tokens (three in flight: 108 KiB instead of 72) -- measured 20 % SLOWER on MI355X
  // (tools/wgrad8_bench.py: 862-877 us against 720-733 us for the 44 Linear layers of the 384-channel stage; a barrier every 18
  // MFMAs per wave costs more than the extra bytes in flight return).  Default: two stages of 64 tokens.
  static const bool ring4 = [] { const char* e = getenv("CSTS_WGRAD8_RING"); return e && e[0] == '4'; }();
  int grid = nitems;
  if (max_wgs > 0 && max_wgs < nitems) grid = std::max(8, max_wgs / 8 * 8);      // a multiple of 8: items keep their XCD
  if (ring4) hipLaunchKernelGGL((wgrad8_kernel<32, 4>), dim3((unsigned)grid), dim3(W8_THR), 0, stream, device_items, nitems);
  else hipLaunchKernelGGL((wgrad8_kernel<64, 2>), dim3((unsigned)grid), dim3(W8_THR), 0, stream, device_items, nitems);
  CSTS_LAUNCH_CHECK();
  return 0;
}
// tile 192 x 384, bf16 dY and X; every item is a whole tile (M % 192 == 0, N % 384 == 0), its token range a multiple of 64,
// rows 16-byte aligned
extern "C" int csts_wgrad_grouped8(const csts_wgrad_item* device_items, int nitems, hipStream_t stream) {
  return wgrad8_launch(device_items, nitems, 0, stream);
}
// the same on at most max_wgs workgroups (rounded down to a multiple of 8, at least 8), each walking several items
extern "C" int csts_wgrad_grouped8_limited(const csts_wgrad_item* device_items, int nitems, int max_wgs, hipStream_t stream) {
  return wgrad8_launch(device_items, nitems, max_wgs, stream);
}
