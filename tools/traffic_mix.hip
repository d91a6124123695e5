// What does the HBM system give a kernel with the TRAFFIC MIX of the short-K activation GEMMs of the CSTS step, without any
// matrix math?  Every workgroup (8 waves, gemm4's tile and store pattern: 128 rows x N columns in 192-column slabs) handles row
// tiles of a bf16 problem  C[M][N] (+ a second output H[M][N]) = f(A[M][K]):
//   mode 0: store C only            mode 1: store C and H            mode 2: read A, store C        mode 3: read A, store C and H
//   mode 4: read A and R[M][N] fp32 (residual), store C fp32 (the fp32 residual-stream form)
// one tile per workgroup (grid = tiles) or persistent (grid = 512, tiles strided) -- both printed.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/traffic_mix.hip -o tools/bin/traffic_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short u16;

template <int MODE>
__global__ __launch_bounds__(512) void mix_kernel(const u16* __restrict__ A, const float* __restrict__ R, void* __restrict__ Cv,
                                                  u16* __restrict__ H, int M, int N, int K, int ntiles) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, hi = lane >> 5;
  const int nslab = N / 192;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t m0 = (int64_t)(t / nslab) * 128 + wm * 32, n0 = (t % nslab) * 192 + wn * 96;
    uint4 v = make_uint4(tid, tid + 1, tid + 2, tid + 3);
    if (MODE >= 2) {      // the tile's A rows: 128 x K bf16, 16 bytes per lane per load, spread over the 8 waves
      const int chunks = 128 * K / 8;           // 16-byte pieces
      const int64_t a0 = (int64_t)(t / nslab) * 128 * K;
      for (int c = tid; c < chunks; c += 512) {
        const uint4 a = *reinterpret_cast<const uint4*>(A + a0 + (int64_t)c * 8);
        v.x ^= a.x; v.y ^= a.y; v.z ^= a.z; v.w ^= a.w;
      }
    }
    if (MODE == 5 || MODE == 6) {     // fp32 residual form with whole lines: 5 = 8 rows x 128 B per wave instruction (32 lanes of a row
      // segment), 6 = one row of 96 columns = 384 B per 24 lanes... use 2 rows x 512 B: the wave tile's 96 fp32 columns are 384 B,
      // so mode 6 walks the WORKGROUP's full 192-column slab row (768 B = 48 lanes) -- 64 lanes cover 1 1/3 rows: linearised
      float* o = reinterpret_cast<float*>(Cv);
      if (MODE == 5) {
        for (int ni = 0; ni < 3; ++ni)
          for (int it = 0; it < 4; ++it) {
            const int64_t at = (m0 + 8 * it + (lane >> 3)) * N + n0 + ni * 32 + 4 * (lane & 7);
            const uint4 r = *reinterpret_cast<const uint4*>(R + at);
            *reinterpret_cast<uint4*>(o + at) = make_uint4(v.x ^ r.x, v.y ^ r.y, v.z ^ r.z, v.w ^ r.w);
          }
      } else {
        // the workgroup's 128 x 192 fp32 slab = 128 rows x 768 B, linearised over the 512 threads: 16 B per thread, 48 threads per row
        const int64_t mt = (int64_t)(t / nslab) * 128, nt = (t % nslab) * 192;
        for (int c = tid; c < 128 * 48; c += 512) {
          const int64_t at = (mt + c / 48) * N + nt + 4 * (c % 48);
          const uint4 r = *reinterpret_cast<const uint4*>(R + at);
          *reinterpret_cast<uint4*>(o + at) = make_uint4(v.x ^ r.x, v.y ^ r.y, v.z ^ r.z, v.w ^ r.w);
        }
      }
    } else if (MODE == 4) {
      float* o = reinterpret_cast<float*>(Cv);
      for (int ni = 0; ni < 3; ++ni)
        for (int q = 0; q < 4; ++q) {
          const int64_t at = (m0 + (lane & 31)) * N + n0 + ni * 32 + 8 * q + 4 * hi;
          const uint4 r = *reinterpret_cast<const uint4*>(R + at);
          *reinterpret_cast<uint4*>(o + at) = make_uint4(v.x ^ r.x, v.y ^ r.y, v.z ^ r.z, v.w ^ r.w);
        }
    } else {
      u16* o = reinterpret_cast<u16*>(Cv);
      for (int ni = 0; ni < 3; ++ni)
        for (int pr = 0; pr < 2; ++pr) {
          const int64_t at = (m0 + (lane & 31)) * N + n0 + ni * 32 + 16 * pr + 8 * hi;
          if (MODE == 1 || MODE == 3) *reinterpret_cast<uint4*>(H + at) = v;
          *reinterpret_cast<uint4*>(o + at) = v;
        }
    }
  }
}

template <int MODE> void run(const u16* A, const float* R, void* C, u16* H, int M, int N, int K, bool persistent, const char* what) {
  const int ntiles = (M / 128) * (N / 192), grid = persistent ? (ntiles < 512 ? ntiles : 512) : ntiles;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(mix_kernel<MODE>, dim3(grid), dim3(512), 0, 0, A, R, C, H, M, N, K, ntiles);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mix_kernel<MODE>, dim3(grid), dim3(512), 0, 0, A, R, C, H, M, N, K, ntiles);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  double bytes = 0;
  if (MODE == 0) bytes = 2.0 * M * N; if (MODE == 1) bytes = 4.0 * M * N; if (MODE == 2) bytes = 2.0 * M * N + 2.0 * M * K * (N / 192);
  if (MODE == 3) bytes = 4.0 * M * N + 2.0 * M * K * (N / 192); if (MODE >= 4) bytes = 8.0 * M * N + 2.0 * M * K * (N / 192);
  printf("  %6d x %4d x %4d  %-46s %-14s %7.1f us  %5.2f TB/s\n", M, N, K, what, persistent ? "persistent 512" : "1 tile per wg", best * 1e3,
         bytes / (best * 1e-3) / 1e12);
}

int main() {
  const size_t cap = (size_t)262144 * 768 * 4;
  u16* A; float* R; void* C; u16* H;
  hipMalloc(&A, cap); hipMalloc(&R, cap); hipMalloc(&C, cap); hipMalloc(&H, cap);
  hipMemset(A, 1, cap); hipMemset(R, 1, cap);
  const int shapes[][3] = {{262144, 384, 192}, {131072, 384, 96}, {131072, 768, 192}, {131072, 384, 384}, {32768, 768, 192}, {8192, 1536, 384}};
  for (auto& s : shapes)
    for (int pers = 0; pers < 2; ++pers) {
      run<0>(A, R, C, H, s[0], s[1], s[2], pers, "store C (bf16)");
      run<1>(A, R, C, H, s[0], s[1], s[2], pers, "store C + H (bf16): fc1 + GELU outputs");
      run<2>(A, R, C, H, s[0], s[1], s[2], pers, "read A, store C");
      run<3>(A, R, C, H, s[0], s[1], s[2], pers, "read A, store C + H: fc1 + GELU traffic");
      run<4>(A, R, C, H, s[0], s[1], s[2], pers, "read A + fp32 residual, store fp32 C: proj / fc2");
      run<5>(A, R, C, H, s[0], s[1], s[2], pers, "  same, 8 rows x 128 B per wave instruction");
      run<6>(A, R, C, H, s[0], s[1], s[2], pers, "  same, slab rows linearised (768 B runs)");
    }
  return 0;
}
