// Does the f16 form of the 32x32x16 MFMA run the chip at another clock than the bf16 form?  Same cycles per instruction (MI355X guide),
// but a CU-filling matrix stream is power-limited (tools/mfma_valu_overlap.hip: 1.2-1.4 GHz), and an f16 multiplier array toggles 11-bit
// significands against bf16's 8.  One wave per SIMD, every CU busy, 6 accumulators round-robin, pseudo-random operands of the same values
// in both formats, 3 runs each, interleaved.
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/diag/mfma_f16_vs_bf16 tools/mfma_f16_vs_bf16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ float prand(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return (float)(x & 0xffff) / 32768.f - 1.f; }
template <bool F16>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
  const int tid = threadIdx.x + 256 * blockIdx.x;
  bf16x8 ab[6], bb[2]; f16x8 ah[6], bh[2];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) { const float v = prand(tid * 64 + i * 8 + j); ab[i][j] = (__bf16)v; ah[i][j] = (_Float16)v; }
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) { const float v = prand(tid * 64 + 48 + i * 8 + j); bb[i][j] = (__bf16)v; bh[i][j] = (_Float16)v; }
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      if constexpr (F16) acc[s % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s % 6], bh[s & 1], acc[s % 6], 0, 0, 0);
      else acc[s % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[s % 6], bb[s & 1], acc[s % 6], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  sink[tid] = s;
  if (tid == 0) out[0] = t1 - t0;
}
template <bool F16> void run() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8); hipMalloc(&sink, 256 * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<F16>), dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<F16>), dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
  printf("%-5s %5.1f ticks per MFMA, kernel %.3f ms, %.0f TFLOP/s chip-wide, shader clock %.3f GHz\n", F16 ? "f16" : "bf16", (double)c / iters / 12.0, ms,
         1024.0 * iters * 12 * 32768.0 / (ms * 1e-3) / 1e12, (double)c / (ms * 1e-3) / 1e9);
  hipFree(out); hipFree(sink);
}
int main() { for (int r = 0; r < 3; ++r) { run<false>(); run<true>(); } return 0; }
