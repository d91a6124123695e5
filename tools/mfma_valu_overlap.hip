// How much vector work hides under v_mfma_f32_32x32x16_bf16 for ONE wave per SIMD, as a function of WHERE the MFMA's accumulator
// lives: AGPRs ("a" constraint) or arch VGPRs ("v": what -mllvm -amdgpu-mfma-vgpr-form selects).  Per MFMA gap: F filler instructions
// on registers the MFMAs never touch (v_fma_f32, or v_exp_f32 for every other filler when EXP).  6 accumulators round-robin, every CU
// busy with one 256-thread workgroup, no memory traffic.  Prints shader-clock ticks per MFMA.
// build: hipcc --offload-arch=gfx950 -O3 -w -o tools/diag/mfma_valu_overlap tools/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <bool AGPR, int F, bool EXP>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[6], b[2];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(0.001f * ((tid * 7 + i * 3 + j) % 97) - 0.04f);
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.002f * ((tid * 5 - i + j) % 89) - 0.08f);
  f32x16 acc[6];
  for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.5f + 0.001f * (tid + i);
  const float c = 0.999f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[s % 6]) : "v"(a[s % 6]), "v"(b[s & 1]));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[s % 6]) : "v"(a[s % 6]), "v"(b[s & 1]));
#pragma unroll
      for (int f = 0; f < F; ++f) {
        if (EXP && (f & 1)) asm volatile("v_exp_f32 %0, %0" : "+v"(x[f % 8]));
        else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[f % 8]) : "v"(c));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15");
  float s = 0.f;
  for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += x[i];
  sink[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <bool AGPR, int F, bool EXP> void run() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8); hipMalloc(&sink, 256 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<AGPR, F, EXP>), dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<AGPR, F, EXP>), dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
  printf("acc in %s, %d fillers per gap (%s): %6.1f ticks per MFMA, kernel %.3f ms, %.2f GHz\n", AGPR ? "AGPR" : "VGPR", F,
         EXP ? "fma/exp alternating" : "fma", (double)c / iters / 12.0, ms, (double)c / (ms * 1e-3) / 1e9);
  hipFree(out); hipFree(sink);
}
template <bool AGPR> void sweep() {
  run<AGPR, 0, false>(); run<AGPR, 2, false>(); run<AGPR, 4, false>(); run<AGPR, 5, false>(); run<AGPR, 6, false>(); run<AGPR, 8, false>();
  run<AGPR, 12, false>(); run<AGPR, 4, true>(); run<AGPR, 6, true>(); run<AGPR, 8, true>();
}
int main() { sweep<true>(); sweep<false>(); return 0; }
