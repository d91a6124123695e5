// Issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 as a function of how consecutive MFMAs share accumulators (one wave per
// SIMD, 4 waves per CU, every CU busy; operands in registers, no memory traffic).
//   RR6   : 6 accumulators round-robin (0 1 2 3 4 5 0 1 ...)   -- the dV / dK products of the attention backward
//   PAIR  : 0 0 1 1 2 2 ...                                      -- chains of 2
//   CHAIN : 0 0 0 0 0 0 1 1 1 1 1 1                              -- chains of 6 (the S / dP products)
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/diag/mfma_chain tools/mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[6], b[2];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(0.001f * (tid + i + j));
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.002f * (tid - i + j));
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const int ai = MODE == 0 ? s % 6 : MODE == 1 ? s / 2 : s / 6;
      acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % 6], b[s & 1], acc[ai], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  sink[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int MODE> void run(const char* name) {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8); hipMalloc(&sink, 256 * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / iters / 12.0;
  const double tf = 1024.0 * iters * 12 * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-6s %6.1f shader-clock ticks per MFMA; kernel %.3f ms -> %.0f TFLOP/s chip-wide, %.2f GHz implied by the ticks\n", name, per, ms, tf,
         (double)c / (ms * 1e-3) / 1e9);
}
int main() { run<0>("RR6"); run<1>("PAIR"); run<2>("CHAIN"); return 0; }
