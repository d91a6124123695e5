// Shader clock while a kernel runs: a one-wave kernel reads the shader-clock counter (s_memtime) and the 100 MHz real-time counter
// (s_memrealtime) around a ~4 us spin and writes their ratio in MHz.  Launched between the kernels of a step (tools/clock_probe.py)
// it tells at which clock the step's latency-bound kernels actually run.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/clock_probe.hip -o tools/diag/libclock_probe.so
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(float* out) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < 400) r1 = __builtin_amdgcn_s_memrealtime();      // 4 us
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(r1 - r0) * 100.f;
}
extern "C" int clock_probe(float* out, hipStream_t s) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, out);
  return (int)hipGetLastError();
}
