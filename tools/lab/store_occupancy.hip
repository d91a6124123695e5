// How many storing waves per CU does an output-bound GEMM epilogue need?  Every workgroup (8 waves) writes 128 x 192 tiles of a
// bf16 [M][1536] matrix in gemm4's register-epilogue pattern (32 rows x 32 bytes per wave instruction) and nothing else; the
// number of resident workgroups per CU is capped by a dummy dynamic-LDS allocation; the grid is PERSISTENT (256 x resident
// workgroups, each walking its share of the tiles), as the GEMM is.
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/diag/store_occupancy tools/store_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16;
__global__ __launch_bounds__(512) void k(void* __restrict__ out, int N, int ntn, int ntiles) {
  extern __shared__ char dummy[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, hi = lane >> 5;
  if (tid == 0) dummy[0] = 1;
  const uint4 v = make_uint4(tid, tid + 1, tid + 2, tid + 3);
  u16* o = reinterpret_cast<u16*>(out);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t m0 = (int64_t)(t / ntn) * 128 + wm * 32, n0 = (t % ntn) * 192 + wn * 96;
    for (int ni = 0; ni < 3; ++ni)
      for (int pr = 0; pr < 2; ++pr)
        *reinterpret_cast<uint4*>(o + (m0 + (lane & 31)) * N + n0 + ni * 32 + 16 * pr + 8 * hi) = v;
  }
}
int main() {
  const int N = 1536, M = 131072, ntn = N / 192, ntiles = (M / 128) * ntn;
  void* out; hipMalloc(&out, (size_t)M * N * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds[4] = {160 * 1024, 80 * 1024, 53 * 1024, 40 * 1024};
  for (int i = 0; i < 4; ++i) {
    const int wpc = i + 1;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds[i]);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(256 * wpc), dim3(512), lds[i], 0, out, N, ntn, ntiles);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 7; ++r) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(256 * wpc), dim3(512), lds[i], 0, out, N, ntn, ntiles);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%d workgroup(s) of 8 waves per CU (persistent, %d tiles each): %7.1f us  %5.2f TB/s\n", wpc, ntiles / (256 * wpc), best * 1e3,
           (double)M * N * 2 / (best * 1e-3) / 1e12);
  }
  return 0;
}
