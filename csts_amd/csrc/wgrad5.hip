// Grouped weight gradients of the THIN layers (96- / 192-channel stages: dW[M,N] with M and N multiples of 96 that the 192 x 384 class
// of wgrad8.hip does not take), one (96 x 96 tile, token chunk) work item per WAVE.  Same item table as wgrad_grouped_kernel (gemm.hip).
//
// Why (profiles/r5_final_step_inventory.txt, r5_final_bench.json): the 128-wide classes of wgrad_grouped_kernel ran these layers at 0.29 of
// HBM -- 1.07 ms per step for 2.6 GB that must move -- with a register-staged k-tile between two workgroup barriers per 64 tokens and
// 1.5 x the algorithmic bytes fetched (a 96-feature layer on 128-wide tiles; the X panel re-read per tile row from another CU).  These
// problems are streams: 64 k .. 262 k tokens reduced into a 36 KB tile.  So the same structure as gemm5.hip: a wave is its own stream --
// its 96 x 96 fp32 tile lives in 144 accumulator registers, the token rows of dY[:, m0 : m0 + 96] and X[:, n0 : n0 + 96] (192 B runs)
// come in by LDS-DMA into a wave-private ring of 16-token stages, are read back as MFMA fragments through ds_read_b64_tr_b16 (both operands
// are token-major; a 192-byte row stride needs no swizzle: the 4 rows x 64 B a 32-lane read group touches cover the 64 banks exactly
// once) and multiplied by 9 MFMAs per stage.  No workgroup barrier anywhere: waves start, stream and exit on their own.  The tiles of
// one (layer, token chunk) sit in consecutive slots of ONE XCD's list (ops._wg_plan), i.e. in the waves of one workgroup: the X rows
// four n-tiles of an fc1 share are fetched from HBM once and served to the other three by that CU's L1 / the XCD's L2.
// LDS: 8 waves x 2 stages x 6 KB = 96 KB -- the grouped stencil weight gradients (20 KB) still fit beside it (ops.flush_wgrads).
#include "common.h"
#include <algorithm>

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void* lptr5_t;

constexpr int W5_T = 96, W5_ROWB = W5_T * 2, W5_TOK = 16, W5_W = 8;
constexpr int W5_HALF = W5_TOK * W5_ROWB;                 // one operand of a stage: 16 token rows x 192 B
constexpr int W5_STAGE = 2 * W5_HALF;                     // dY rows, then X rows

template <int N> __device__ __forceinline__ void w5_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MFMA 32x32x16 fragment (8 consecutive tokens of one feature per lane) of features obase .. obase + 31 from a [16 tokens][96 features] image
__device__ __forceinline__ bf16x8 w5_frag(const char* S, int obase, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const char* a = S + (8 * (g >> 1) + q) * W5_ROWB + (obase + 16 * (g & 1) + 4 * pp) * 2;
  const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * W5_ROWB));
  const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0), b1 = __builtin_bit_cast(bf16x4, t1);
  bf16x8 r;
  r[0] = b0[0]; r[1] = b0[1]; r[2] = b0[2]; r[3] = b0[3];
  r[4] = b1[0]; r[5] = b1[1]; r[6] = b1[2]; r[7] = b1[3];
  return r;
}

template <int S>
__global__ __launch_bounds__(64 * W5_W, 2) void wgrad5_kernel(const csts_wgrad_item* __restrict__ items, int nitems) {
  __shared__ __attribute__((aligned(1024))) char smem[W5_W * S * W5_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave w of workgroup b takes slot (b / 8) * 8 + w of XCD (b % 8)'s list (the table interleaves the eight lists: position slot * 8 + xcd)
  const int idx = (((int)blockIdx.x >> 3) * W5_W + wave) * 8 + ((int)blockIdx.x & 7);
  if (idx >= nitems) return;                                  // no barrier in this kernel: a wave may leave on its own
  const csts_wgrad_item it = items[idx];
  if (it.A == nullptr) return;                                // padding slot
  char* const ring = smem + wave * (S * W5_STAGE);
  // token stages of this item: kbeg, kbeg + 16 step, ... < kend (whole stages: host-checked).  it.M = step > 1: the items of one tile INTERLEAVE
  // their stages (item c of nch: stages c, c + nch, ...), so that the waves working on a layer read one moving window of its rows
  const int64_t kbeg = it.kbeg;
  const int step = it.M > 1 ? it.M : 1;
  const int64_t hop = (int64_t)step * W5_TOK;
  const int nk = (int)((it.kend - kbeg + hop - 1) / hop);

  // producer: the three 1-KiB pieces of each operand's 16 x 192 B image (LDS slot s = 64 i + lane: token row s / 12, 16-byte chunk s % 12)
  unsigned aoff[3], boff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int s = 64 * i + lane, row = s / 12, ch = s % 12;
    aoff[i] = (unsigned)(row * (int)it.lda + it.m0 + ch * 8) * 2u;
    boff[i] = (unsigned)(row * (int)it.ldb + it.n0 + ch * 8) * 2u;
  }
  const char* const A = reinterpret_cast<const char*>(it.A);
  const char* const B = reinterpret_cast<const char*>(it.B);
  auto issue = [&](char* st, int64_t k0) {
    const char* a = A + k0 * it.lda * 2;
    const char* b = B + k0 * it.ldb * 2;
    const uint32_t l0 = (uint32_t)(uintptr_t)(lptr5_t)st;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l0 + (uint32_t)i * 1024u), "v"(aoff[i]), "s"(a) : "memory", "m0");
#pragma unroll
    for (int i = 0; i < 3; ++i)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l0 + (uint32_t)(W5_HALF + i * 1024)), "v"(boff[i]), "s"(b) : "memory", "m0");
  };

  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fused bias gradient (n0 == 0 tiles): lanes 0 .. 47 own a feature pair of the dY rows
  const bool do_colsum = it.colsum != nullptr && it.n0 == 0;   // wave-uniform
  float cs0 = 0.f, cs1 = 0.f;

#pragma unroll
  for (int s2 = 0; s2 < S - 1; ++s2)
    if (s2 < nk) issue(ring + s2 * W5_STAGE, kbeg + (int64_t)s2 * hop);
  int cs = 0, ps = S - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed once at most the (S - 2) younger stages (6 wave instructions each) are outstanding
    if (S == 2 || kt == nk - 1) w5_wait<0>();
    else if (S == 3 || kt == nk - 2) w5_wait<6>();
    else w5_wait<12>();
    const char* As = ring + cs * W5_STAGE;
    const char* Bs = As + W5_HALF;
    // the stage refilled next was consumed one iteration ago (its fragment reads have returned: their MFMAs have issued)
#ifdef CSTS_W5_DIAG_NODMA      // diagnostics builds (tools/scripts/build_variant.sh): only the first refill is issued
    if (kt + S - 1 < nk && kt < 1) issue(ring + ps * W5_STAGE, kbeg + (int64_t)(kt + S - 1) * hop);
#else
    if (kt + S - 1 < nk) issue(ring + ps * W5_STAGE, kbeg + (int64_t)(kt + S - 1) * hop);
#endif
    cs = (cs + 1 == S) ? 0 : cs + 1;
    ps = (ps + 1 == S) ? 0 : ps + 1;
#ifdef CSTS_W5_DIAG_NOMATH     // diagnostics builds: the stream without fragment reads and MFMAs
    if (kt >= 0) continue;
#endif
    if (do_colsum && lane < 48) {
#pragma unroll
      for (int kk = 0; kk < W5_TOK; ++kk) {
        const bf16x2 t = *reinterpret_cast<const bf16x2*>(As + kk * W5_ROWB + lane * 4);
        cs0 += (float)t[0];
        cs1 += (float)t[1];
      }
    }
    bf16x8 a[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = w5_frag(As, i * 32, lane);
#pragma unroll
    for (int j = 0; j < 3; ++j) b[j] = w5_frag(Bs, j * 32, lane);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = CSTS_MFMA16(b[j], a[i], acc[i][j], 0, 0, 0);   // operands swapped: see the epilogue
  }

  if (do_colsum && lane < 48) {
    it.colsum[it.m0 + 2 * lane] = cs0;
    it.colsum[it.m0 + 2 * lane + 1] = cs1;
  }
  // epilogue: X is the MFMA row operand, so acc[i][j][r] is dW[m][n] with m = the lane's row (lane & 31) of unit i and
  // n = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of unit j: 16-byte stores
  const int hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float* row = it.C + (int64_t)(it.m0 + i * 32 + (lane & 31)) * it.ldc + it.n0 + 4 * hi;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        *reinterpret_cast<f32x4*>(row + j * 32 + 8 * t) = f32x4{acc[i][j][4 * t], acc[i][j][4 * t + 1], acc[i][j][4 * t + 2], acc[i][j][4 * t + 3]};
  }
}

}  // namespace

// tile 96 x 96, bf16 dY and X; every item is a whole tile (M % 96 == 0, N % 96 == 0), its token range a multiple of 16, rows 16-byte
// aligned.  nitems = 8 x (longest XCD list), the lists interleaved (position slot * 8 + xcd), padding slots with A == NULL.
extern "C" int csts_wgrad_grouped5(const csts_wgrad_item* device_items, int nitems, hipStream_t stream) {
  CSTS_REQUIRE(device_items != nullptr && nitems > 0 && nitems % 8 == 0, "no items, or not a whole number of XCD rounds");
  // CSTS_WGRAD5_RING=3: three stages per wave (two in flight, 144 KB of LDS: nothing fits beside it)
  static const bool ring3 = [] { const char* e = getenv("CSTS_WGRAD5_RING"); return e && e[0] == '3'; }();
  const int depth = nitems / 8, grid = 8 * ((depth + W5_W - 1) / W5_W);
  if (ring3) hipLaunchKernelGGL((wgrad5_kernel<3>), dim3((unsigned)grid), dim3(64 * W5_W), 0, stream, device_items, nitems);
  else hipLaunchKernelGGL((wgrad5_kernel<2>), dim3((unsigned)grid), dim3(64 * W5_W), 0, stream, device_items, nitems);
  CSTS_LAUNCH_CHECK();
  return 0;
}
