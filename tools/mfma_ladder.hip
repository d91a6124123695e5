// Calibration ladder for the MFMA kernels of this repo (MI355X / gfx950): the SAME inner loop as gemm4_kernel<4,1,3,2>
// (8-wave workgroups, wave tile 32 x 96 = 3 accumulators, 64-deep k-tiles of 4 x 16, 2 workgroups per CU) with its
// ingredients switched on one at a time -- what does each of them cost on this chip, with every CU busy?
//
//   mode 0: MFMAs on register operands only                      (the matrix pipes' practical peak under full load)
//   mode 1: + fragments re-read from LDS every 16-deep sub-step  (ds_read_b128, conflict-free swizzled addresses)
//   mode 2: + one s_barrier per k-tile
//   mode 3: + LDS-DMA refill of a 2-stage ring from an L2-resident buffer, counted wait before the barrier (= the k-loop of
//             gemm4, without prologue / epilogue / tile bookkeeping)
//   mode 4: as 3, but the refill streams through a large buffer (Infinity Cache / HBM instead of L2)
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_ladder.hip -o gpurun_out/mfma_ladder ; run: ./mfma_ladder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 128, BN = 192, STAGE = (BM + BN) * 128, S = 2, NTHR = 512, LPT = 5;   // 5 LDS-DMA instructions per thread per k-tile

template <int MODE>
__global__ __launch_bounds__(NTHR, 4) void ladder_kernel(const bf16* __restrict__ src, int64_t src_elems, float* __restrict__ out, int nk) {
  __shared__ __attribute__((aligned(1024))) char smem[S * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int key = (lane >> 1) & 7, hi = lane >> 5;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = (lane & 31) * 128 + (((ks * 2 + hi) ^ key) << 4);
  // fill LDS once so that every mode multiplies real (random) numbers
  for (int i = tid; i < S * STAGE / 16; i += NTHR) {
    const int64_t e = ((int64_t)blockIdx.x * 7919 + i) * 8 % (src_elems - 8);
    *reinterpret_cast<bf16x8*>(smem + i * 16) = *reinterpret_cast<const bf16x8*>(src + (e & ~7ll));
  }
  __syncthreads();
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 fa, fb[3];
  fa = *reinterpret_cast<const bf16x8*>(smem + (wm * 32) * 128 + foff[0]);
#pragma unroll
  for (int j = 0; j < 3; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(smem + BM * 128 + (wn * 96 + j * 32) * 128 + foff[0]);
  // this workgroup's private slice of the source: MODE 3 loops over 4 k-tiles (L2-resident), MODE 4 walks far
  const int64_t tile_elems = STAGE / 2;
  const int64_t span = (MODE == 4) ? (src_elems / tile_elems) : 4;
  int64_t pos = ((int64_t)blockIdx.x * 13) % span;
  int cstage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE >= 2) __builtin_amdgcn_s_barrier();
    const char* As = smem + cstage * STAGE;
    const char* Bs = As + BM * 128;
    char* st = smem + (cstage ^ 1) * STAGE;
    cstage ^= 1;
    // mode 3: 128 distinct 40 KiB tiles in all (5 MiB: L2-resident on every XCD); mode 4: every k-tile from somewhere new
    const bf16* g = src + ((MODE == 4) ? pos : ((pos & 3) + 4 * (int64_t)(blockIdx.x & 31))) * tile_elems;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE >= 1) {
        fa = *reinterpret_cast<const bf16x8*>(As + (wm * 32) * 128 + foff[ks]);
#pragma unroll
        for (int j = 0; j < 3; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 96 + j * 32) * 128 + foff[ks]);
      }
      if (MODE >= 3) {
#pragma unroll
        for (int i = 0; i < LPT; ++i)
          if (i * 4 / LPT == ks) {
            const int piece = wave * LPT + i;
            __builtin_amdgcn_global_load_lds((gptr_t)(g + piece * 512 + lane * 8), (lptr_t)(st + piece * 1024), 16, 0, 0);
          }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa, acc[j], 0, 0, 0);
    }
    pos = (pos + 1) % span;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) out[blockIdx.x * NTHR + tid] = s;     // keep the accumulators alive
}

template <int MODE>
double run(const bf16* src, int64_t n, float* out, int grid, int nk) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(ladder_kernel<MODE>, dim3(grid), dim3(NTHR), 0, 0, src, n, out, nk);
  hipDeviceSynchronize();
  double best = 1e30;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(ladder_kernel<MODE>, dim3(grid), dim3(NTHR), 0, 0, src, n, out, nk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = (double)grid * 8 /*waves*/ * nk * 4 * 3 * (32.0 * 32 * 16 * 2);
  return flop / (best * 1e-3) / 1e12;
}

int main() {
  const int64_t n = (int64_t)1 << 30;                       // 2 GiB of bf16: beyond the 256 MiB Infinity Cache
  bf16* src; float* out;
  hipMalloc(&src, n * 2); hipMalloc(&out, 4 << 20);
  std::vector<unsigned short> h(1 << 24);
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // random bf16 in +-[1, 2)
  for (int64_t o = 0; o < n; o += (int64_t)h.size()) hipMemcpy(src + o, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int nk = 512;
  for (int wgs = 1; wgs <= 2; ++wgs) {
    const int grid = 256 * wgs;
    printf("%d workgroup(s) of 8 waves per CU (grid %d), %d k-tiles each, wave tile 32 x 96:\n", wgs, grid, nk);
    printf("  mode 0  MFMA only (register operands)          %8.1f TF/s\n", run<0>(src, n, out, grid, nk));
    printf("  mode 1  + LDS fragment reads                    %8.1f TF/s\n", run<1>(src, n, out, grid, nk));
    printf("  mode 2  + s_barrier per k-tile                  %8.1f TF/s\n", run<2>(src, n, out, grid, nk));
    printf("  mode 3  + LDS-DMA ring refill from L2           %8.1f TF/s\n", run<3>(src, n, out, grid, nk));
    printf("  mode 4  + refill streams through 2 GiB (HBM)    %8.1f TF/s\n", run<4>(src, n, out, grid, nk));
  }
  return 0;
}
