// Calibration for the TN (weight-gradient) k-loop on MI355X: tile 128 x 128, 64-token k-tiles kept [token][feature] in LDS,
// fragments through ds_read_b64_tr_b16, 4 waves (wave tile 64 x 64), 3 workgroups per CU -- the loop of
// wgrad_grouped_kernel<false, 2> -- with the k-tile brought in two ways:
//   R: through registers (global_load_dwordx4 -> ds_write_b128 between two barriers; the next tile's loads in flight under
//      the MFMAs): what the kernel does today;
//   D: by LDS-DMA (global_load_lds_dwordx4) into a 2-stage ring, one barrier per k-tile, source-side chunk swizzle so that
//      the transposing reads of the unpadded image are bank-conflict free.
// Operands: "L2" = every workgroup re-reads a small set of tiles (5 MiB in all); "stream" = new memory every k-tile.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wgrad_ladder.hip -o tools/bin/wgrad_ladder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int LD_R = 128 + 32;     // padded row (elements) of the register-staged image: conflict-free transposing reads
constexpr int TILE_B = 64 * 128 * 2;   // one operand k-tile, bytes (unpadded)

// fragment of rows obase .. +31 of an operand kept [k][out]; SWZ: unpadded 256-byte rows, 16-byte chunks XORed with (row & 3) << 2
template <bool SWZ>
__device__ __forceinline__ bf16x8 frag_t(const bf16* S, int obase, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int row = ks * 16 + 8 * (g >> 1) + q, col = obase + 16 * (g & 1) + 4 * pp;
  const bf16 *a0, *a1;
  if (SWZ) {
    const int c0 = ((col >> 3) ^ ((row & 3) << 2)) * 8 + (col & 7), c1 = ((col >> 3) ^ (((row + 4) & 3) << 2)) * 8 + (col & 7);
    a0 = S + row * 128 + c0;
    a1 = S + (row + 4) * 128 + c1;
  } else {
    a0 = S + row * LD_R + col;
    a1 = a0 + 4 * LD_R;
  }
  const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
  const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0), b1 = __builtin_bit_cast(bf16x4, t1);
  bf16x8 r;
  r[0] = b0[0]; r[1] = b0[1]; r[2] = b0[2]; r[3] = b0[3]; r[4] = b1[0]; r[5] = b1[1]; r[6] = b1[2]; r[7] = b1[3];
  return r;
}

template <bool STREAM>
__device__ __forceinline__ const bf16* tile_at(const bf16* src, int64_t ntiles, int64_t p, int which) {
  const int64_t t = STREAM ? p : ((p & 3) + 4 * (int64_t)(blockIdx.x & 63));
  return src + ((t * 2 + which) % ntiles) * (TILE_B / 2);
}
template <bool STREAM>
__device__ __forceinline__ void reg_load(const bf16* src, int64_t ntiles, int64_t p, int tid, uint4 (&ra)[4], uint4 (&rb)[4]) {
  const bf16* a = tile_at<STREAM>(src, ntiles, p, 0);
  const bf16* b = tile_at<STREAM>(src, ntiles, p, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
    ra[i] = *reinterpret_cast<const uint4*>(a + row * 128 + ch * 8);
    rb[i] = *reinterpret_cast<const uint4*>(b + row * 128 + ch * 8);
  }
}

template <bool DMA, bool STREAM>
__global__ __launch_bounds__(256, 3) void wg_kernel(const bf16* __restrict__ src, int64_t ntiles, float* __restrict__ out, int nk) {
  constexpr int SM = DMA ? 2 * 2 * TILE_B : 2 * 64 * LD_R * 2;
  __shared__ __attribute__((aligned(1024))) char smem[SM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int64_t pos = ((int64_t)blockIdx.x * 13) % ntiles;
  auto tile_ptr = [&](int64_t p, int which) {    // operand tile = 64 token rows x 128 features, stored contiguously (256 B rows)
    const int64_t t = STREAM ? p : ((p & 3) + 4 * (int64_t)(blockIdx.x & 63));
    return src + ((t * 2 + which) % ntiles) * (TILE_B / 2);
  };
  if (DMA) {
    // piece = 4 token rows (1 KiB); lane -> row 4 piece + (lane >> 4), LDS chunk lane & 15, global chunk XOR-swizzled
    auto issue = [&](char* st, int64_t p) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int piece = wave * 8 + i;                        // 32 pieces: 16 of A then 16 of B
        const int which = piece >> 4, pr = piece & 15, row = pr * 4 + (lane >> 4), ch = (lane & 15) ^ ((row & 3) << 2);
        __builtin_amdgcn_global_load_lds((gptr_t)(tile_ptr(p, which) + row * 128 + ch * 8), (lptr_t)(st + which * TILE_B + pr * 1024), 16, 0, 0);
      }
    };
    issue(smem, pos);
    int cs = 0;
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bf16* As = reinterpret_cast<const bf16*>(smem + cs * 2 * TILE_B);
      const bf16* Bs = As + TILE_B / 2;
      pos = (pos + 1) % ntiles;
      if (kt + 1 < nk) issue(smem + (cs ^ 1) * 2 * TILE_B, pos);
      cs ^= 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = frag_t<true>(As, wm * 64 + i * 32, ks, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = frag_t<true>(Bs, wn * 64 + j * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  } else {
    bf16* As = reinterpret_cast<bf16*>(smem);
    bf16* Bs = As + 64 * LD_R;
    uint4 ra[4], rb[4];
    reg_load<STREAM>(src, ntiles, pos, tid, ra, rb);
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
        *reinterpret_cast<uint4*>(&As[row * LD_R + ch * 8]) = ra[i];
        *reinterpret_cast<uint4*>(&Bs[row * LD_R + ch * 8]) = rb[i];
      }
      __syncthreads();
      pos = (pos + 1) % ntiles;
      if (kt + 1 < nk) reg_load<STREAM>(src, ntiles, pos, tid, ra, rb);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = frag_t<false>(As, wm * 64 + i * 32, ks, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = frag_t<false>(Bs, wn * 64 + j * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <bool DMA, bool STREAM>
double run(const bf16* src, int64_t ntiles, float* out, int grid, int nk) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((wg_kernel<DMA, STREAM>), dim3(grid), dim3(256), 0, 0, src, ntiles, out, nk);
  hipDeviceSynchronize();
  double best = 1e30;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((wg_kernel<DMA, STREAM>), dim3(grid), dim3(256), 0, 0, src, ntiles, out, nk);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return (double)grid * nk * (128.0 * 128 * 64 * 2) / (best * 1e-3) / 1e12;
}

int main() {
  const int64_t n = (int64_t)1 << 30;
  bf16* src; float* out;
  hipMalloc(&src, n * 2); hipMalloc(&out, 4 << 20);
  std::vector<unsigned short> h(1 << 24);
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  for (int64_t o = 0; o < n; o += (int64_t)h.size()) hipMemcpy(src + o, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int64_t ntiles = n / (TILE_B / 2);
  const int nk = 128, grid = 256 * 3 * 4;
  printf("TN k-loop, tile 128 x 128 x 64, 4 waves, 3 workgroups per CU, %d k-tiles per workgroup, grid %d:\n", nk, grid);
  printf("  register-staged, operands from L2      %8.1f TF/s\n", run<false, false>(src, ntiles, out, grid, nk));
  printf("  LDS-DMA ring,    operands from L2      %8.1f TF/s\n", run<true, false>(src, ntiles, out, grid, nk));
  printf("  register-staged, operands streamed     %8.1f TF/s\n", run<false, true>(src, ntiles, out, grid, nk));
  printf("  LDS-DMA ring,    operands streamed     %8.1f TF/s\n", run<true, true>(src, ntiles, out, grid, nk));
  return 0;
}
