// LDS read rates of the two access patterns of the attention kernels on gfx950, 4 waves per CU issuing back to back:
//   mode 0: ds_read_b128, lane = row (row stride LD bf16), 16-byte column chunk per half-wave  (A operand of S = Q K^T)
//   mode 1: ds_read_b64_tr_b16 pairs with the address pattern of pv() (transposed A operand of dV = P^T dO)
//   mode 2: plain ds_read_b64 at the SAME addresses as mode 1 (is the transposing read slower than a plain one?)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_read_rate tools/lds_read_rate.hip ; run: /tmp/lds_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE, int LD>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) short smem[128 * LD];
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  for (int i = tid; i < 128 * LD; i += 256) smem[i] = (short)i;
  __syncthreads();
  const int g1 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  const short* row = smem + (lane & 31) * LD + 8 * h;
  const short* tr = smem + (4 * h + qq) * LD + 16 * g1 + 4 * pp;
  f32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        if (MODE == 0) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(row + u * 32 * LD + 16 * s);
          acc += v;
        } else if (MODE == 1) {
          const int s2 = s / 3, d = s % 3;
          const short* Yb = tr + (u * 32 + 16 * s2) * LD + 32 * d;
          const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)Yb);
          const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Yb + 8 * LD));
          acc[0] += a[0]; acc[1] += b[1];
        } else {
          const int s2 = s / 3, d = s % 3;
          const short* Yb = tr + (u * 32 + 16 * s2) * LD + 32 * d;
          const s16x4 a = *reinterpret_cast<const s16x4*>(Yb);
          const s16x4 b = *reinterpret_cast<const s16x4*>(Yb + 8 * LD);
          acc[0] += a[0]; acc[1] += b[1];
        }
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
  if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int MODE, int LD> void run(const char* name, int bytes_per_iter_per_wave) {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8); hipMalloc(&sink, 256 * 256 * 4);
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, LD>), dim3(256), dim3(256), 0, 0, out, sink, iters);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
  const double per_iter = (double)c / iters;
  printf("%-44s LD %3d: %7.1f cycles per 24 operand slots per wave (4 waves/CU) -> %5.1f B/clk/CU\n", name, LD, per_iter,
         4.0 * bytes_per_iter_per_wave / per_iter);
  hipFree(out); hipFree(sink);
}
int main() {
  run<0, 104>("ds_read_b128 row-per-lane", 24 * 1024);
  run<1, 104>("2 x ds_read_b64_tr_b16 (pv pattern)", 24 * 1024);
  run<2, 104>("2 x ds_read_b64 at the same addresses", 24 * 1024);
  run<0, 112>("ds_read_b128 row-per-lane", 24 * 1024);
  run<1, 112>("2 x ds_read_b64_tr_b16 (pv pattern)", 24 * 1024);
  run<2, 112>("2 x ds_read_b64 at the same addresses", 24 * 1024);
  run<0, 100>("ds_read_b128 row-per-lane", 24 * 1024);
  run<1, 100>("2 x ds_read_b64_tr_b16 (pv pattern)", 24 * 1024);
  run<1, 108>("2 x ds_read_b64_tr_b16 (pv pattern)", 24 * 1024);
  run<0, 108>("ds_read_b128 row-per-lane", 24 * 1024);
  return 0;
}
