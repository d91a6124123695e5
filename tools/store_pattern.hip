// Which store pattern does a GEMM epilogue want on MI355X?  Every workgroup (8 waves) writes one 128 x 192 output tile of a
// bf16 [M][N] matrix (N = 1536, M = 8192 * reps), nothing else, as
//   pattern 0: gemm4's register epilogue -- per wave-instruction 32 rows x 32 bytes (lanes l, l + 32 adjacent 16-byte pieces)
//   pattern 1: 16 rows x 64 bytes per wave-instruction (one 32 x 32 block transposed through LDS)
//   pattern 2: 8 rows x 128 bytes per wave-instruction (two blocks side by side: whole cache lines)
//   pattern 3: as 0 for an fp32 matrix (32 rows x 32 bytes, 4 instructions per block)      pattern 4: fp32, 8 rows x 128 bytes
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_pattern.hip -o tools/bin/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16;

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(void* __restrict__ out, int N, int ntn) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, hi = lane >> 5;
  const int64_t t = blockIdx.x, m0 = (t / ntn) * 128 + wm * 32, n0 = (t % ntn) * 192 + wn * 96;
  const uint4 v = make_uint4(tid, tid + 1, tid + 2, tid + 3);
  if (PAT == 0) {                 // bf16: per block 2 instructions, lane: row lane&31, 16 bytes at column 16 p + 8 hi
    u16* o = reinterpret_cast<u16*>(out);
    for (int ni = 0; ni < 3; ++ni)
      for (int pr = 0; pr < 2; ++pr)
        *reinterpret_cast<uint4*>(o + (m0 + (lane & 31)) * N + n0 + ni * 32 + 16 * pr + 8 * hi) = v;
  } else if (PAT == 1) {          // bf16: per block 2 instructions, 16 rows x 64 bytes
    u16* o = reinterpret_cast<u16*>(out);
    for (int ni = 0; ni < 3; ++ni)
      for (int it = 0; it < 2; ++it)
        *reinterpret_cast<uint4*>(o + (m0 + 16 * it + (lane >> 2)) * N + n0 + ni * 32 + 8 * (lane & 3)) = v;
  } else if (PAT == 2) {          // bf16: 96 columns = 192 bytes per row: 12 lanes per row -> 5 1/3 rows per instruction; use 8 x 128 B + 16 x 64 B
    u16* o = reinterpret_cast<u16*>(out);
    for (int it = 0; it < 4; ++it)
      *reinterpret_cast<uint4*>(o + (m0 + 8 * it + (lane >> 3)) * N + n0 + 8 * (lane & 7)) = v;
    for (int it = 0; it < 2; ++it)
      *reinterpret_cast<uint4*>(o + (m0 + 16 * it + (lane >> 2)) * N + n0 + 64 + 8 * (lane & 3)) = v;
  } else if (PAT == 3) {          // fp32: per block 4 instructions of 32 rows x 32 bytes
    float* o = reinterpret_cast<float*>(out);
    for (int ni = 0; ni < 3; ++ni)
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<uint4*>(o + (m0 + (lane & 31)) * N + n0 + ni * 32 + 8 * q + 4 * hi) = v;
  } else {                        // fp32: per block 4 instructions of 8 rows x 128 bytes
    float* o = reinterpret_cast<float*>(out);
    for (int ni = 0; ni < 3; ++ni)
      for (int it = 0; it < 4; ++it)
        *reinterpret_cast<uint4*>(o + (m0 + 8 * it + (lane >> 3)) * N + n0 + ni * 32 + 4 * (lane & 7)) = v;
  }
}

template <int PAT> void run(void* out, int reps, const char* what, int esz) {
  const int N = 1536, M = 8192 * reps, ntn = N / 192, grid = (M / 128) * ntn;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(store_kernel<PAT>, dim3(grid), dim3(512), 0, 0, out, N, ntn);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(store_kernel<PAT>, dim3(grid), dim3(512), 0, 0, out, N, ntn);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("  M = %6d: %-52s %7.1f us  %6.2f TB/s\n", M, what, best * 1e3, (double)M * N * esz / (best * 1e-3) / 1e12);
}

int main() {
  void* out; hipMalloc(&out, (size_t)8192 * 16 * 1536 * 4);
  for (int reps : {1, 4, 16}) {
    run<0>(out, reps, "bf16, 32 rows x 32 B per instruction (gemm4 today)", 2);
    run<1>(out, reps, "bf16, 16 rows x 64 B per instruction", 2);
    run<2>(out, reps, "bf16, 8 rows x 128 B (+ 16 x 64 B for the last 32 columns)", 2);
    run<3>(out, reps, "fp32, 32 rows x 32 B per instruction (gemm4 today)", 4);
    run<4>(out, reps, "fp32, 8 rows x 128 B per instruction", 4);
  }
  return 0;
}
