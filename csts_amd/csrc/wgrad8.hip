// Grouped weight gradients on 8-wave workgroups: dW[M,N] tiles of 192 x 384 (M = the Linear's output features, N = its
// input features), one (tile, token-chunk) work item per workgroup, same item table as wgrad_grouped_kernel (gemm.hip).
//
// Why: the 128 x 128 items of wgrad_grouped_kernel run at ~750 TF/s with 35 % of their L2 requests hitting
// (rocprofv3 TCC_HIT / TCC_MISS, profiles/r2_wgrad_pmc.txt): the tiles that share a dY / X column slice drift apart over
// their 128-step token loops and miss each other in the 4 MiB L2, so every tile streams its own 2 x 256-byte row segments
// from the Infinity Cache -- 64 FLOP per byte through the L2 -> CU path, which tops out at ~12 TB/s on this chip.  A
// 192 x 384 tile gets 128 FLOP per staged byte (half the bytes through that path) and needs no luck in L2: the reuse is
// inside the workgroup.  All Linear layers of the 384- and 768-channel stages have output features % 192 == 0 and input
// features % 384 == 0 (1152 / 384 / 1536 x 384, 384 x 1536, 2304 / 768 / 3072 x 768, 768 x 3072).
//
// TN form, bf16 x bf16: both operands are token-major, so a 64-token k-tile is staged [token][feature] (straight 16-byte
// copies through registers, the next k-tile's global loads in flight under the MFMAs) and fed to v_mfma_f32_32x32x16_bf16
// through ds_read_b64_tr_b16.  Wave grid 2 x 4, wave tile 96 x 96 (9 accumulators).  The fp32 tile goes straight from the
// accumulators to memory: lane = column, so every store instruction writes two full 128-byte lines.
#include "common.h"

namespace {

constexpr int W8_TM = 192, W8_TN = 384, W8_BK = 64, W8_THR = 512;
constexpr int W8_LDA = W8_TM + 32, W8_LDB = W8_TN + 32;      // row strides 448 B / 832 B: conflict-free transposing reads
constexpr int W8_A_CH = W8_TM / 8, W8_B_CH = W8_TN / 8;      // 16-byte chunks per token row
constexpr int W8_NA = W8_BK * W8_A_CH / W8_THR, W8_NB = W8_BK * W8_B_CH / W8_THR;   // chunks per thread: 3, 6

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// MFMA 32x32x16 fragment of rows obase .. obase + 31 of an operand kept [k][out] in LDS (k-substep ks)
__device__ __forceinline__ bf16x8 frag_t(const bf16* S, int ld, int obase, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const bf16* a = &S[(ks * 16 + 8 * (g >> 1) + q) * ld + obase + 16 * (g & 1) + 4 * pp];
  const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
  const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * ld));
  const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0), b1 = __builtin_bit_cast(bf16x4, t1);
  bf16x8 r;
  r[0] = b0[0]; r[1] = b0[1]; r[2] = b0[2]; r[3] = b0[3];
  r[4] = b1[0]; r[5] = b1[1]; r[6] = b1[2]; r[7] = b1[3];
  return r;
}

__global__ __launch_bounds__(W8_THR, 2) void wgrad8_kernel(const csts_wgrad_item* __restrict__ items) {
  __shared__ __attribute__((aligned(16))) bf16 smem[W8_BK * (W8_LDA + W8_LDB)];
  bf16* As = smem;
  bf16* Bs = smem + W8_BK * W8_LDA;
  const csts_wgrad_item it = items[blockIdx.x];
  if (it.A == nullptr) return;          // padding slot (the host equalises the per-XCD lists); block-uniform
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(it.A);
  const bf16* __restrict__ B = reinterpret_cast<const bf16*>(it.B);
  const int64_t m0 = it.m0, n0 = it.n0, kbeg = it.kbeg, kend = it.kend;
  const int nk = (int)((kend - kbeg + W8_BK - 1) / W8_BK);

  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this thread's chunks of a k-tile: token row and feature offset are the same for every k-tile
  int akr[W8_NA], aoc[W8_NA], bkr[W8_NB], boc[W8_NB];
#pragma unroll
  for (int i = 0; i < W8_NA; ++i) { const int c = tid + W8_THR * i; akr[i] = c / W8_A_CH; aoc[i] = (c % W8_A_CH) * 8; }
#pragma unroll
  for (int i = 0; i < W8_NB; ++i) { const int c = tid + W8_THR * i; bkr[i] = c / W8_B_CH; boc[i] = (c % W8_B_CH) * 8; }
  uint4 ra[W8_NA], rb[W8_NB];
  auto load = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < W8_NA; ++i) {
      const int64_t k = k0 + akr[i];
      ra[i] = (k < kend) ? *reinterpret_cast<const uint4*>(A + k * it.lda + m0 + aoc[i]) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W8_NB; ++i) {
      const int64_t k = k0 + bkr[i];
      rb[i] = (k < kend) ? *reinterpret_cast<const uint4*>(B + k * it.ldb + n0 + boc[i]) : make_uint4(0, 0, 0, 0);
    }
  };

  // fused bias gradient (n0 == 0 tiles): 384 threads own a column pair of the dY tile and a quarter of its 64 token rows
  const bool do_colsum = it.colsum != nullptr && n0 == 0 && tid < 384;
  const int cp = tid % 96, csl = tid / 96;
  float cs0 = 0.f, cs1 = 0.f;

  load(kbeg);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                       // previous k-tile fully consumed
#pragma unroll
    for (int i = 0; i < W8_NA; ++i) *reinterpret_cast<uint4*>(&As[akr[i] * W8_LDA + aoc[i]]) = ra[i];
#pragma unroll
    for (int i = 0; i < W8_NB; ++i) *reinterpret_cast<uint4*>(&Bs[bkr[i] * W8_LDB + boc[i]]) = rb[i];
    __syncthreads();
    if (kt + 1 < nk) load(kbeg + (int64_t)(kt + 1) * W8_BK);      // next k-tile's global loads fly under the MFMAs below
    if (do_colsum) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const bf16x2 t = *reinterpret_cast<const bf16x2*>(&As[(csl * 16 + kk) * W8_LDA + 2 * cp]);
        cs0 += (float)t[0];
        cs1 += (float)t[1];
      }
    }
#pragma unroll
    for (int ks = 0; ks < W8_BK / 16; ++ks) {
      bf16x8 a[3], b[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = frag_t(As, W8_LDA, wm * 96 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = frag_t(Bs, W8_LDB, wn * 96 + j * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  if (it.colsum != nullptr && n0 == 0) {   // block-uniform: fold the four token slices of every column (fixed order)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);               // [4][192]
    if (tid < 384) { red[csl * W8_TM + 2 * cp] = cs0; red[csl * W8_TM + 2 * cp + 1] = cs1; }
    __syncthreads();
    if (tid < W8_TM && m0 + tid < it.M) it.colsum[m0 + tid] = (red[tid] + red[W8_TM + tid]) + (red[2 * W8_TM + tid] + red[3 * W8_TM + tid]);
  }

  // ---------------- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int64_t n = n0 + wn * 96 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 96 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < it.M && n < it.N) it.C[m * it.ldc + n] = acc[i][j][r];
      }
    }
}

}  // namespace

// tile 192 x 384, bf16 dY and X; every item must be a whole tile's origin (M % 8 == 0, N % 8 == 0, 16-byte aligned rows)
extern "C" int csts_wgrad_grouped8(const csts_wgrad_item* device_items, int nitems, hipStream_t stream) {
  CSTS_REQUIRE(device_items != nullptr && nitems > 0, "no items");
  hipLaunchKernelGGL(wgrad8_kernel, dim3((unsigned)nitems), dim3(W8_THR), 0, stream, device_items);
  CSTS_LAUNCH_CHECK();
  return 0;
}
