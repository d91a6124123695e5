// MFMA GEMM family for the CSTS path (K3/K9 of SURVEY.md 2.3): every Linear / fusion-conv /
// patch-embed contraction, forward (NT), data-gradient (NN) and weight-gradient (TN).
//
//   NT: C[m,n] = sum_k A[m,k] * B[n,k]      (x @ W^T; reference nn.Linear, attention.py:130,159; common.py:27-33)
//   NN: C[m,n] = sum_k A[m,k] * B[k,n]      (dY @ W)
//   TN: C[m,n] = sum_k A[k,m] * B[k,n]      (dY^T @ X, split over k with fp32 atomics)
//
// Tile 128x128x32, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 accumulators.
// bf16 mode: v_mfma_f32_32x32x16_bf16, operands converted to bf16 while staging (fp32 master weights are
// read directly, no shadow copy).  f32 mode: v_mfma_f32_32x32x2_f32 (exact fp32, the parity mode).
// Operands whose reduction dim is NOT the contiguous one are kept [k][out] in LDS and fed to the MFMA through
// ds_read_b64_tr_b16 (bf16) / plain ds_read_b32 (f32), so staging is always a straight 16-byte copy.
#include "common.h"
#include "gemm_shared.h"
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT_ = 256;
// LDS row strides (elements)
constexpr int KC_LD_BF = BK + 8;     // 80 B rows
constexpr int OC_LD_BF = 128 + 32;   // 320 B rows: conflict-free ds_read_b64_tr_b16 (bank = 16q + 2p + 8g)
constexpr int KC_LD_F = BK + 1;      // 33 floats: conflict-free ds_read_b32 over rows
constexpr int OC_LD_F = 128 + 4;

template <bool F32> struct Cmp;
template <> struct Cmp<false> { typedef bf16 T; };
template <> struct Cmp<true> { typedef float T; };


// load 8 consecutive source elements (guarded) as floats
__device__ __forceinline__ void load8(const void* p, int dt, int64_t idx, int nvalid, bool vec, float (&o)[8]) {
  if (nvalid >= 8 && vec) {
    ld8_as_f32(p, dt, idx, o);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (j < nvalid) ? ld_as_f32(p, dt, idx + j) : 0.f;
  }
}

template <bool F32> struct Stage {  // register-resident copy of this thread's two chunks of one operand tile
  float v[2][8];
};

template <bool KC>
__device__ __forceinline__ void stage_load(const void* P, int dt, int64_t ld, bool vec, int64_t out0, int64_t out_lim,
                                           int64_t k0, int64_t k_lim, int tid, float (&v)[2][8]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + NT_ * i;
    if (KC) {
      const int row = c >> 2, kc = (c & 3) * 8;
      const int64_t r = out0 + row, k = k0 + kc;
      int nvalid = (r < out_lim) ? (int)min((int64_t)8, max((int64_t)0, k_lim - k)) : 0;
      load8(P, dt, r * ld + k, nvalid, vec, v[i]);
    } else {
      const int krow = c >> 4, oc = (c & 15) * 8;
      const int64_t k = k0 + krow, o = out0 + oc;
      int nvalid = (k < k_lim) ? (int)min((int64_t)8, max((int64_t)0, out_lim - o)) : 0;
      load8(P, dt, k * ld + o, nvalid, vec, v[i]);
    }
  }
}

template <bool KC, bool F32>
__device__ __forceinline__ void stage_store(typename Cmp<F32>::T* S, int tid, const float (&v)[2][8]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + NT_ * i;
    if (F32) {
      float* Sf = reinterpret_cast<float*>(S);
      if (KC) {
        const int row = c >> 2, kc = (c & 3) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) Sf[row * KC_LD_F + kc + j] = v[i][j];
      } else {
        const int krow = c >> 4, oc = (c & 15) * 8;
        float4* d = reinterpret_cast<float4*>(&Sf[krow * OC_LD_F + oc]);
        d[0] = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        d[1] = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
      }
    } else {
      bf16* Sb = reinterpret_cast<bf16*>(S);
      bf16x8 w;
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = (bf16)v[i][j];
      if (KC) {
        const int row = c >> 2, kc = (c & 3) * 8;
        *reinterpret_cast<bf16x8*>(&Sb[row * KC_LD_BF + kc]) = w;
      } else {
        const int krow = c >> 4, oc = (c & 15) * 8;
        *reinterpret_cast<bf16x8*>(&Sb[krow * OC_LD_BF + oc]) = w;
      }
    }
  }
}

// bf16 MFMA fragment (8 k-values for output index obase + (lane&31)), k-substep ks (16 wide)
template <bool KC>
__device__ __forceinline__ bf16x8 frag_bf16(const bf16* S, int obase, int ks, int lane) {
  if (KC) {
    return *reinterpret_cast<const bf16x8*>(&S[(obase + (lane & 31)) * KC_LD_BF + ks * 16 + 8 * (lane >> 5)]);
  } else {
    // S is [k][out]; transposing read: each 16-lane group fetches a 4(k) x 16(out) block column-major.
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const bf16* a = &S[(ks * 16 + 8 * (g >> 1) + q) * OC_LD_BF + obase + 16 * (g & 1) + 4 * p];
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * OC_LD_BF));
    bf16x4 b0 = __builtin_bit_cast(bf16x4, t0), b1 = __builtin_bit_cast(bf16x4, t1);
    bf16x8 r;
    r[0] = b0[0]; r[1] = b0[1]; r[2] = b0[2]; r[3] = b0[3];
    r[4] = b1[0]; r[5] = b1[1]; r[6] = b1[2]; r[7] = b1[3];
    return r;
  }
}
// f32 MFMA 32x32x2 fragment: one value, k = 2*ks + (lane>>5)
template <bool KC>
__device__ __forceinline__ float frag_f32(const float* S, int obase, int ks, int lane) {
  if (KC) return S[(obase + (lane & 31)) * KC_LD_F + 2 * ks + (lane >> 5)];
  return S[(2 * ks + (lane >> 5)) * OC_LD_F + obase + (lane & 31)];
}

template <bool A_KC, bool B_KC, bool F32>
__global__ __launch_bounds__(NT_) void gemm_kernel(Params p) {
  typedef typename Cmp<F32>::T T;
  constexpr int A_ELEMS = F32 ? (A_KC ? BM * KC_LD_F : BK * OC_LD_F) : (A_KC ? BM * KC_LD_BF : BK * OC_LD_BF);
  constexpr int B_ELEMS = F32 ? (B_KC ? BN * KC_LD_F : BK * OC_LD_F) : (B_KC ? BN * KC_LD_BF : BK * OC_LD_BF);
  __shared__ __attribute__((aligned(16))) T smem[A_ELEMS + B_ELEMS];
  T* As = smem;
  T* Bs = smem + A_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t tile_n = blockIdx.x % p.ntiles_n, tile_m = blockIdx.x / p.ntiles_n;
  const int64_t m0 = tile_m * BM, n0 = tile_n * BN;
  const int64_t kbeg = (int64_t)blockIdx.y * p.k_chunk;
  const int64_t kend = min(p.K, kbeg + p.k_chunk);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[2][8], rb[2][8];
  stage_load<A_KC>(p.A, p.a_dt, p.lda, p.a_vec, m0, p.M, kbeg, kend, tid, ra);
  stage_load<B_KC>(p.B, p.b_dt, p.ldb, p.b_vec, n0, p.N, kbeg, kend, tid, rb);

  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
    stage_store<A_KC, F32>(As, tid, ra);
    stage_store<B_KC, F32>(Bs, tid, rb);
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch next tile into registers; latency hides under the MFMAs below
      stage_load<A_KC>(p.A, p.a_dt, p.lda, p.a_vec, m0, p.M, k0 + BK, kend, tid, ra);
      stage_load<B_KC>(p.B, p.b_dt, p.ldb, p.b_vec, n0, p.N, k0 + BK, kend, tid, rb);
    }
    if (F32) {
      const float* Af = reinterpret_cast<const float*>(As);
      const float* Bf = reinterpret_cast<const float*>(Bs);
#pragma unroll 4
      for (int ks = 0; ks < BK / 2; ++ks) {
        float a0 = frag_f32<A_KC>(Af, wm * 64, ks, lane), a1 = frag_f32<A_KC>(Af, wm * 64 + 32, ks, lane);
        float b0 = frag_f32<B_KC>(Bf, wn * 64, ks, lane), b1 = frag_f32<B_KC>(Bf, wn * 64 + 32, ks, lane);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    } else {
      const bf16* Ab = reinterpret_cast<const bf16*>(As);
      const bf16* Bb = reinterpret_cast<const bf16*>(Bs);
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 a0 = frag_bf16<A_KC>(Ab, wm * 64, ks, lane), a1 = frag_bf16<A_KC>(Ab, wm * 64 + 32, ks, lane);
        bf16x8 b0 = frag_bf16<B_KC>(Bb, wn * 64, ks, lane), b1 = frag_bf16<B_KC>(Bb, wn * 64 + 32, ks, lane);
        acc[0][0] = CSTS_MFMA16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = CSTS_MFMA16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = CSTS_MFMA16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = CSTS_MFMA16(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
  }

  // ---------------- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool first_split = (blockIdx.y == 0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int64_t n = n0 + wn * 64 + nt * 32 + (lane & 31);
      if (n >= p.N) continue;
      const float bias = (p.bias != nullptr && first_split) ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        if (p.split_k > 1 && p.ws != nullptr) {   // deterministic split-K: raw partial, finished by splitk_finish_kernel
          p.ws[((int64_t)blockIdx.y * p.M + m) * p.N + n] = acc[mt][nt][r];
          continue;
        }
        float v = acc[mt][nt][r] + bias;
        if (p.split_k > 1) {
          atomicAdd(reinterpret_cast<float*>(p.C) + m * p.ldc + n, v);
          continue;
        }
        if (p.epilogue == CSTS_EPI_GELU) {
          if (p.aux != nullptr) st_from_f32(p.aux, p.aux_dt, m * p.ldaux + n, v);
          v = gelu_f(v);
        } else if (p.epilogue == CSTS_EPI_DGELU) {
          v *= dgelu_f(ld_as_f32(p.aux, p.aux_dt, m * p.ldaux + n));
        }
        if (p.row_scale != nullptr) v *= p.row_scale[m / p.rows_per_scale];
        if (p.residual != nullptr) {
          if (p.ru_To > 0) {
            float up[1];
            res_up_load<1>(p, res_up_row(p, m), n, up);
            v += up[0];
          } else {
            const int64_t rm = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
            v += ld_as_f32(p.residual, p.r_dt, rm * p.ldr + n);
          }
        }
        st_from_f32(p.C, p.c_dt, m * p.ldc + n, v);
      }
    }
  }
}

// C = epilogue(sum_s ws[s]) for the deterministic split-K path (VEC elements per thread; VEC = 4 needs N % 4 == 0).
// SL slab-lanes share one output vector (small outputs with many slabs would otherwise be one long serial chain per
// thread); lane l sums slabs l, l+SL, ... in order, lanes are combined in order through LDS -> reproducible.
template <int VEC, int SL>
__global__ __launch_bounds__(256) void splitk_finish_kernel(Params p, int nsplit) {
  constexpr int EV = 256 / SL;   // output vectors per block
  __shared__ float red[SL > 1 ? SL : 1][EV][VEC];
  const int64_t total = p.M * p.N;
  if (p.colsum != nullptr && p.colsum_ws != nullptr) {   // fused bias-gradient partials: colsum[m] = sum_s colsum_ws[s][m]
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += (int64_t)gridDim.x * blockDim.x) {
      float t = 0.f;
      for (int sidx = 0; sidx < nsplit; ++sidx) t += p.colsum_ws[(int64_t)sidx * p.M + m];
      p.colsum[m] = t;
    }
  }
  const int e = threadIdx.x % EV, sl = threadIdx.x / EV;
  const int64_t nvec = (total + VEC - 1) / VEC;
  for (int64_t v0 = (int64_t)blockIdx.x * EV; v0 < nvec; v0 += (int64_t)gridDim.x * EV) {
    const int64_t idx = (v0 + e) * VEC;
    const bool live = v0 + e < nvec;
    float v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = 0.f;
    if (live) {
      int sidx = sl;
      if (VEC == 4) {
        for (; sidx + 3 * SL < nsplit; sidx += 4 * SL) {
          float4 t[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(p.ws + (int64_t)(sidx + u * SL) * total + idx);
#pragma unroll
          for (int u = 0; u < 4; ++u) { v[0] += t[u].x; v[1 % VEC] += t[u].y; v[2 % VEC] += t[u].z; v[3 % VEC] += t[u].w; }
        }
        for (; sidx < nsplit; sidx += SL) {
          const float4 t = *reinterpret_cast<const float4*>(p.ws + (int64_t)sidx * total + idx);
          v[0] += t.x; v[1 % VEC] += t.y; v[2 % VEC] += t.z; v[3 % VEC] += t.w;
        }
      } else {
        for (; sidx < nsplit; sidx += SL) v[0] += p.ws[(int64_t)sidx * total + idx];
      }
    }
    if (SL > 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) red[sl][e][j] = v[j];
      __syncthreads();
      if (sl == 0) {
#pragma unroll
        for (int l = 1; l < SL; ++l)
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] += red[l][e][j];
      }
      __syncthreads();
    }
    if (sl != 0 || !live) continue;
    const int64_t m = idx / p.N, n0 = idx - m * p.N;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int64_t n = n0 + j;
      float x = v[j] + (p.bias ? p.bias[n] : 0.f);
      if (p.epilogue == CSTS_EPI_GELU) {
        if (p.aux != nullptr) st_from_f32(p.aux, p.aux_dt, m * p.ldaux + n, x);
        x = gelu_f(x);
      } else if (p.epilogue == CSTS_EPI_DGELU) {
        x *= dgelu_f(ld_as_f32(p.aux, p.aux_dt, m * p.ldaux + n));
      }
      if (p.row_scale != nullptr) x *= p.row_scale[m / p.rows_per_scale];
      if (p.residual != nullptr) {
        const int64_t rm = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
        x += ld_as_f32(p.residual, p.r_dt, rm * p.ldr + n);
      }
      st_from_f32(p.C, p.c_dt, m * p.ldc + n, x);
    }
  }
}
template <int VEC>
static void launch_finish_v(const Params& p, int nsplit, hipStream_t stream) {
  const int64_t nvec = cdiv(p.M * p.N, VEC);
  // slab-lanes only where the output alone cannot fill the chip (a function of the shape only -> reproducible)
  const int sl = (nsplit >= 16 && nvec < 16384) ? 16 : ((nsplit >= 4 && nvec < 131072) ? 4 : 1);
  const dim3 grid((unsigned)std::min<int64_t>(cdiv(nvec, 256 / sl), 8192));
  if (sl == 16) hipLaunchKernelGGL((splitk_finish_kernel<VEC, 16>), grid, dim3(256), 0, stream, p, nsplit);
  else if (sl == 4) hipLaunchKernelGGL((splitk_finish_kernel<VEC, 4>), grid, dim3(256), 0, stream, p, nsplit);
  else hipLaunchKernelGGL((splitk_finish_kernel<VEC, 1>), grid, dim3(256), 0, stream, p, nsplit);
}
static void launch_finish(const Params& p, int nsplit, hipStream_t stream) {
  if (p.N % 4 == 0 && aligned16(p.ws)) launch_finish_v<4>(p, nsplit, stream);
  else launch_finish_v<1>(p, nsplit, stream);
}

// =====================================================================================================
// v2 kernel (bf16 MFMA only): BK = 64, double-buffered LDS (one barrier per k-tile), raw register staging of the
// next tile (global loads stay in flight across the MFMA phase), source dtypes compile-time, 128- or 64-row
// tiles (small grids get twice the workgroups), and an epilogue staged through LDS so that every global
// load/store of C / residual / aux is a coalesced 16-byte access.
// =====================================================================================================
constexpr int KC2_LD = BK2 + 8;            // 144 B rows: conflict-free ds_read_b128 over 16 rows
constexpr int CS_LD = 128 + 4;             // fp32 C staging row stride

template <bool F32S> struct Raw { uint4 v[F32S ? 2 : 1]; };

template <bool F32S>
__device__ __forceinline__ void raw_load(Raw<F32S>& r, const void* base, int64_t idx, bool valid) {
  if (valid) {
    if (F32S) {
      const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(base) + idx);
      r.v[0] = q[0];
      r.v[F32S ? 1 : 0] = q[1];
    } else {
      r.v[0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(base) + idx);
    }
  } else {
    r.v[0] = make_uint4(0, 0, 0, 0);
    if (F32S) r.v[F32S ? 1 : 0] = make_uint4(0, 0, 0, 0);
  }
}
template <bool F32S>
__device__ __forceinline__ bf16x8 raw_to_bf16(const Raw<F32S>& r) {
  if (F32S) {
    const f32x4 a = __builtin_bit_cast(f32x4, r.v[0]), b = __builtin_bit_cast(f32x4, r.v[F32S ? 1 : 0]);
    bf16x8 o;
    o[0] = (bf16)a[0]; o[1] = (bf16)a[1]; o[2] = (bf16)a[2]; o[3] = (bf16)a[3];
    o[4] = (bf16)b[0]; o[5] = (bf16)b[1]; o[6] = (bf16)b[2]; o[7] = (bf16)b[3];
    return o;
  } else {
    return __builtin_bit_cast(bf16x8, r.v[0]);
  }
}

// one operand tile: EXT output rows (128 or 64) x 64 k.  KC: LDS [EXT][72] ; OC: LDS [64][EXT+32]
template <bool KC, bool F32S, int EXT, int NTHR> struct Oper {
  static constexpr int NCH = EXT * 8 / NTHR;            // chunks of 8 elements per thread (tile = EXT x 64)
  static constexpr int CPR = EXT / 8;                   // OC: chunks per k-row
  static constexpr int KCR = NTHR / 8;                  // KC: rows covered per pass
  static constexpr int OCR = NTHR / CPR;                // OC: k-rows covered per pass
  static constexpr int OC_LD = EXT + 32;
  static constexpr int LDS_ELEMS = KC ? EXT * KC2_LD : BK2 * OC_LD;
  Raw<F32S> raw[NCH];
  __device__ __forceinline__ void load(const void* P, int64_t ld, int64_t out0, int64_t out_lim, int64_t k0, int64_t k_lim,
                                       int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (KC) {
        const int64_t r = out0 + (tid >> 3) + KCR * i, k = k0 + (tid & 7) * 8;
        raw_load<F32S>(raw[i], P, r * ld + k, r < out_lim && k < k_lim);
      } else {
        const int64_t k = k0 + tid / CPR + OCR * i, o = out0 + (tid % CPR) * 8;
        raw_load<F32S>(raw[i], P, k * ld + o, k < k_lim && o < out_lim);
      }
    }
  }
  __device__ __forceinline__ void store(bf16* S, int tid) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const bf16x8 w = raw_to_bf16<F32S>(raw[i]);
      if (KC) *reinterpret_cast<bf16x8*>(&S[((tid >> 3) + KCR * i) * KC2_LD + (tid & 7) * 8]) = w;
      else *reinterpret_cast<bf16x8*>(&S[(tid / CPR + OCR * i) * OC_LD + (tid % CPR) * 8]) = w;
    }
  }
  // MFMA 32x32x16 fragment for output rows obase..obase+31, k-substep ks (0..3)
  static __device__ __forceinline__ bf16x8 frag(const bf16* S, int obase, int ks, int lane) {
    if (KC) {
      return *reinterpret_cast<const bf16x8*>(&S[(obase + (lane & 31)) * KC2_LD + ks * 16 + 8 * (lane >> 5)]);
    } else {
      const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
      const bf16* a = &S[(ks * 16 + 8 * (g >> 1) + q) * OC_LD + obase + 16 * (g & 1) + 4 * pp];
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
      const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * OC_LD));
      const bf16x4 b0 = __builtin_bit_cast(bf16x4, t0), b1 = __builtin_bit_cast(bf16x4, t1);
      bf16x8 r;
      r[0] = b0[0]; r[1] = b0[1]; r[2] = b0[2]; r[3] = b0[3];
      r[4] = b1[0]; r[5] = b1[1]; r[6] = b1[2]; r[7] = b1[3];
      return r;
    }
  }
};

// LDS-lean schedule: ONE LDS stage (A+B <= 40 KiB) + the next k-tile prefetched in registers, and the epilogue staged
// 64 rows at a time (33 KiB) -> 4 workgroups per CU.  (gemm3's register epilogue -- swapped MFMA operands, 16-byte stores of
// 32-byte row segments -- was tried here too: correct, but 8 % slower over the step's GEMMs: at 3-4 workgroups per CU the
// staged epilogue's full-line stores matter more than its barriers.)  Counters on MI355X (rocprofv3 PMC, NT 8192x1536x384): with the
// previous double-buffered 74 KiB layout only 2 workgroups fit per CU and waves sat 47 % in s_waitcnt / barriers with
// the MFMA pipe 12 % busy; residency, not per-workgroup pipelining, is what hides the L2/HBM round trips here.
// Wave grid WM x 2, each wave (32*MT) x 64: tiles 64x128 (MT 1), 128x128 (MT 2), 256x128 (MT 4: 128x64 per wave, 2 wg/CU).
// The operand stream through the vector memory path sustains only ~15 B/clk/CU here (measured: ~1 us per 64-deep
// k-tile of a 64x128 tile), so FLOPs per loaded byte -- the tile size -- is what sets the MFMA rate.
template <bool A_KC, bool B_KC, bool A_F32, bool B_F32, int MT, int WM>
__global__ __launch_bounds__(128 * WM, (MT == 4) ? 2 : (MT == 2 ? 3 : 4)) void gemm2_kernel(Params p) {
  constexpr int BM2 = 32 * MT * WM, NTHR = 128 * WM;
  typedef Oper<A_KC, A_F32, BM2, NTHR> OA;
  typedef Oper<B_KC, B_F32, 128, NTHR> OB;
  constexpr int STAGE_BYTES = (OA::LDS_ELEMS + OB::LDS_ELEMS) * 2;
  constexpr int CS_BYTES = 64 * CS_LD * 4;                     // half-tile (64 rows) fp32 staging
  constexpr int SMEM_BYTES = STAGE_BYTES > CS_BYTES ? STAGE_BYTES : CS_BYTES;
  __shared__ __attribute__((aligned(16))) char smem_raw[SMEM_BYTES];
  bf16* As = reinterpret_cast<bf16*>(smem_raw);
  bf16* Bs = As + OA::LDS_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;      // wm in [0, WM)
  // XCD-aware remap (speed only): workgroups are dealt round-robin over the 8 XCDs, so give each XCD a CONTIGUOUS
  // range of tiles -- tiles that share an A panel (same tile_m) then hit the same L2.  Bijective for any grid size.
  int64_t bid = blockIdx.x;
  {
    const int64_t nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int64_t tile_n = bid % p.ntiles_n, tile_m = bid / p.ntiles_n;
  const int64_t m0 = tile_m * BM2, n0 = tile_n * 128;
  const int64_t kbeg = (int64_t)blockIdx.y * p.k_chunk;
  const int64_t kend = min(p.K, kbeg + p.k_chunk);
  const int nk = (int)((kend - kbeg + BK2 - 1) / BK2);

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fused bias gradient (TN only): threads own a column pair of the A tile and a slice of its 64 k-rows
  constexpr int CPAIRS = BM2 / 2, TPP = NTHR / CPAIRS, KSL = BK2 / TPP;
  const bool do_colsum = !A_KC && p.colsum != nullptr && tile_n == 0;
  float cs0 = 0.f, cs1 = 0.f;

  OA oa;
  OB ob;
  oa.load(p.A, p.lda, m0, p.M, kbeg, kend, tid);
  ob.load(p.B, p.ldb, n0, p.N, kbeg, kend, tid);

  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                       // previous tile fully consumed
    oa.store(As, tid);
    ob.store(Bs, tid);
    __syncthreads();
    if (kt + 1 < nk) {                     // next tile's global loads fly under the MFMAs below
      oa.load(p.A, p.lda, m0, p.M, kbeg + (int64_t)(kt + 1) * BK2, kend, tid);
      ob.load(p.B, p.ldb, n0, p.N, kbeg + (int64_t)(kt + 1) * BK2, kend, tid);
    }
    if (!A_KC && do_colsum) {
      const int cp = tid % CPAIRS, k0s = (tid / CPAIRS) * KSL;
#pragma unroll
      for (int kk = 0; kk < KSL; ++kk) {
        const bf16x2 t = *reinterpret_cast<const bf16x2*>(&As[(k0s + kk) * OA::OC_LD + 2 * cp]);
        cs0 += (float)t[0];
        cs1 += (float)t[1];
      }
    }
#pragma unroll
    for (int ks = 0; ks < BK2 / 16; ++ks) {
      bf16x8 a[MT], b[2];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = OA::frag(As, wm * 32 * MT + i * 32, ks, lane);
      b[0] = OB::frag(Bs, wn * 64, ks, lane);
      b[1] = OB::frag(Bs, wn * 64 + 32, ks, lane);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        acc[i][0] = CSTS_MFMA16(a[i], b[0], acc[i][0], 0, 0, 0);
        acc[i][1] = CSTS_MFMA16(a[i], b[1], acc[i][1], 0, 0, 0);
      }
    }
  }
  __syncthreads();

  if (!A_KC && do_colsum) {   // fold the k-slices of every column pair (fixed order) and emit this block's partial sums
    float* red = reinterpret_cast<float*>(smem_raw);           // [TPP][BM2]
    const int cp = tid % CPAIRS, sl = tid / CPAIRS;
    red[sl * BM2 + 2 * cp] = cs0;
    red[sl * BM2 + 2 * cp + 1] = cs1;
    __syncthreads();
    if (tid < BM2 && m0 + tid < p.M) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < TPP; ++q) t += red[q * BM2 + tid];
      if (p.split_k > 1) p.colsum_ws[(int64_t)blockIdx.y * p.M + m0 + tid] = t;
      else p.colsum[m0 + tid] = t;
    }
    __syncthreads();
  }

  // ---------------- epilogue through LDS, 64 rows (one 32-row MFMA tile of each wave row) at a time
  float* Cs = reinterpret_cast<float*>(smem_raw);
  const bool first_split = (blockIdx.y == 0);
#ifdef CSTS_GEMM_RES_OLD
  constexpr bool RES_LINES = false;
#else
  constexpr bool RES_LINES = true;      // compile-time: carrying both forms of the residual epilogue behind a run-time flag cost the kernel ~0.1 ms per step
#endif
  const int col = (tid & 15) * 8;
  const int64_t n = n0 + col;
  float bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = 0.f;
  if (n < p.N && p.bias != nullptr && first_split && p.ws == nullptr) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
  }
  constexpr int NGRP = BM2 / 64;                 // 64-row groups staged one at a time
#pragma unroll
  for (int g = 0; g < NGRP; ++g) {
    if (g > 0) __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int unit = wm * MT + mt;              // 32-row unit of the tile owned by this wave
      if ((unit >> 1) != g) continue;             // wave-uniform
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[((unit & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS_LD + wn * 64 + nt * 32 + (lane & 31)] = acc[mt][nt][r];
    }
    __syncthreads();
    // Residual-stream form (fp32 C = (acc + bias) * row_scale + residual: proj / fc2 forward, skip projections), whole row
    // groups: straight-line code with every load of the group -- the residual runs and the per-row drop-path scales -- issued
    // BEFORE its first store (vector memory retires in order: a wait for row i + 1's residual is also a wait for row i's stores).
    // Round 4: a lane owns ONE 16-byte run and the 32 lanes of a half-wave cover a whole 512-byte tile row, so every load and
    // store instruction touches whole 128-byte lines.  Before, a lane owned 8 consecutive floats as two float4: each
    // instruction then touched only the first (or second) 16 bytes of every 32 -- half of every line, twice -- and
    // tools/traffic_mix.hip measures what that costs with no arithmetic at all: 2.6-3.2 TB/s against 5.1-5.8 TB/s for the same
    // bytes in whole lines (profiles/r4_traffic_mix.txt).  Same values in the same order: bit-identical.
    if (p.residual != nullptr && p.res_row_mod == 0 && p.ru_To == 0 && p.epilogue == CSTS_EPI_NONE && p.split_k == 1 &&
        p.c_dt == CSTS_F32 && p.r_dt == CSTS_F32 && m0 + g * 64 + 64 <= p.M && RES_LINES) {            // block-uniform
      constexpr int RPP = NTHR / 32;            // rows per pass
      constexpr int NI = 64 / RPP;
      const int col4 = (tid & 31) * 4;
      const int64_t n4 = n0 + col4;
      if (n4 < p.N) {
        const float* __restrict__ res = reinterpret_cast<const float*>(p.residual);
        float* __restrict__ Cf = reinterpret_cast<float*>(p.C);
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias != nullptr) b4 = *reinterpret_cast<const float4*>(p.bias + n4);
        float4 rr[NI];
        float sc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int64_t m = m0 + g * 64 + (tid >> 5) + RPP * i;
          rr[i] = *reinterpret_cast<const float4*>(res + m * p.ldr + n4);
          sc[i] = p.row_scale != nullptr ? p.row_scale[m / p.rows_per_scale] : 1.f;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int row = (tid >> 5) + RPP * i;
          const int64_t m = m0 + g * 64 + row;
          const float4 c = *reinterpret_cast<const float4*>(&Cs[row * CS_LD + col4]);
          float v[4] = {c.x + b4.x, c.y + b4.y, c.z + b4.z, c.w + b4.w};
          if (p.row_scale != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= sc[i];
          }
          *reinterpret_cast<float4*>(Cf + m * p.ldc + n4) = make_float4(v[0] + rr[i].x, v[1] + rr[i].y, v[2] + rr[i].z, v[3] + rr[i].w);
        }
      }
      continue;
    }
    if (n >= p.N) continue;
    if (!RES_LINES && p.residual != nullptr && p.res_row_mod == 0 && p.ru_To == 0 && p.epilogue == CSTS_EPI_NONE && p.split_k == 1 &&
        p.c_dt == CSTS_F32 && p.r_dt == CSTS_F32 && m0 + g * 64 + 64 <= p.M) {            // block-uniform (round-3 form, -DCSTS_GEMM_RES_OLD)
      constexpr int NI = 1024 / NTHR;
      const float* __restrict__ res = reinterpret_cast<const float*>(p.residual);
      float* __restrict__ Cf = reinterpret_cast<float*>(p.C);
      float4 r0[NI], r1[NI];
      float sc[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int64_t m = m0 + g * 64 + (tid >> 4) + (NTHR / 16) * i;
        const float4* q = reinterpret_cast<const float4*>(res + m * p.ldr + n);
        r0[i] = q[0]; r1[i] = q[1];
        sc[i] = p.row_scale != nullptr ? p.row_scale[m / p.rows_per_scale] : 1.f;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = (tid >> 4) + (NTHR / 16) * i;
        const int64_t m = m0 + g * 64 + row;
        const float4 c0 = *reinterpret_cast<const float4*>(&Cs[row * CS_LD + col]);
        const float4 c1 = *reinterpret_cast<const float4*>(&Cs[row * CS_LD + col + 4]);
        float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float rr[8] = {r0[i].x, r0[i].y, r0[i].z, r0[i].w, r1[i].x, r1[i].y, r1[i].z, r1[i].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bias[j];
        if (p.row_scale != nullptr) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= sc[i];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rr[j];
        float4* d = reinterpret_cast<float4*>(Cf + m * p.ldc + n);
        d[0] = make_float4(v[0], v[1], v[2], v[3]);
        d[1] = make_float4(v[4], v[5], v[6], v[7]);
      }
      continue;
    }
    // data gradient through GELU (v *= gelu'(aux), bf16 aux): the four pre-activation runs of the group first, then
    // arithmetic and stores (same reason as above)
    if (p.epilogue == CSTS_EPI_DGELU && p.aux_dt == CSTS_BF16 && p.residual == nullptr && p.row_scale == nullptr && p.split_k == 1 &&
        m0 + g * 64 + 64 <= p.M) {                                                        // block-uniform
      constexpr int NI = 1024 / NTHR;
      const bf16* __restrict__ hx = reinterpret_cast<const bf16*>(p.aux);
      bf16x8 hv[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int64_t m = m0 + g * 64 + (tid >> 4) + (NTHR / 16) * i;
        hv[i] = *reinterpret_cast<const bf16x8*>(hx + m * p.ldaux + n);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int row = (tid >> 4) + (NTHR / 16) * i;
        const int64_t m = m0 + g * 64 + row;
        const float4 c0 = *reinterpret_cast<const float4*>(&Cs[row * CS_LD + col]);
        const float4 c1 = *reinterpret_cast<const float4*>(&Cs[row * CS_LD + col + 4]);
        float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (v[j] + bias[j]) * dgelu_fast((float)hv[i][j]);
        st8_from_f32(p.C, p.c_dt, m * p.ldc + n, v);
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < 1024 / NTHR; ++i) {
      const int row = (tid >> 4) + (NTHR / 16) * i;                         // 0..63 inside the group
      const int64_t m = m0 + g * 64 + row;
      if (m >= p.M) continue;
      float v[8];
      {
        const float4 c0 = *reinterpret_cast<const float4*>(&Cs[row * CS_LD + col]);
        const float4 c1 = *reinterpret_cast<const float4*>(&Cs[row * CS_LD + col + 4]);
        v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
      }
      if (p.split_k > 1 && p.ws != nullptr) {   // deterministic split-K: raw partial slab
        st8_from_f32(p.ws, CSTS_F32, ((int64_t)blockIdx.y * p.M + m) * p.N + n, v);
        continue;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += bias[j];
      if (p.split_k > 1) {
        float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(c + j, v[j]);
        continue;
      }
      if (p.epilogue == CSTS_EPI_GELU) {
        if (p.aux != nullptr) st8_from_f32(p.aux, p.aux_dt, m * p.ldaux + n, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (p.aux_dt == CSTS_BF16) ? gelu_fast(v[j]) : gelu_f(v[j]);
      } else if (p.epilogue == CSTS_EPI_DGELU) {
        float h[8];
        ld8_as_f32(p.aux, p.aux_dt, m * p.ldaux + n, h);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= (p.aux_dt == CSTS_BF16) ? dgelu_fast(h[j]) : dgelu_f(h[j]);
      }
      if (p.row_scale != nullptr) {
        const float sc = p.row_scale[m / p.rows_per_scale];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= sc;
      }
      if (p.residual != nullptr) {
        float rr[8];
        if (p.ru_To > 0) {
          res_up_load<8>(p, res_up_row(p, m), n, rr);
        } else {
          const int64_t rm = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
          ld8_as_f32(p.residual, p.r_dt, rm * p.ldr + n, rr);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rr[j];
      }
      st8_from_f32(p.C, p.c_dt, m * p.ldc + n, v);
    }
  }
}

// =====================================================================================================
// gemm3: PERSISTENT NT kernel (activations x bf16 shadow weights, both k-contiguous) with an LDS-DMA staging ring.
// The model's activation GEMMs have short K (96..3072): a one-tile-per-workgroup launch spends most of its life in the
// fill (first operand round trip), the epilogue (C through LDS, stores) and the drain, with every workgroup of the
// single resident wave in the same phase at the same time (all load, then all multiply, then all store).  Here a
// workgroup owns a LIST of output tiles and runs ONE flattened stream of k-tiles over them:
//   * operand tiles go L2/HBM -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write pass) into a ring of S
//     stages; S-1 k-tiles stay in flight across the ONE raw s_barrier of a k-step (counted s_waitcnt vmcnt(N)), and the
//     prefetch runs ACROSS tile boundaries, so the next tile's first operands land while this tile's epilogue runs;
//   * the epilogue stages C through the ring stage that was consumed last (32 rows at a time), its global stores are
//     fire-and-forget, and the 2-3 workgroups that share a CU drift out of phase, so stores, loads and MFMAs overlap;
//   * tiles are dealt so that an XCD works on a contiguous range of tiles (shared A/B panels stay in its L2).
// The LDS image is lane-linear per wave-instruction (8 rows x 128 B); ds_read_b128 bank conflicts are removed by an
// XOR swizzle applied to the per-lane SOURCE address and, identically, to the fragment read: chunk slot s of row r
// holds k-chunk s ^ ((r >> 1) & 7), which puts the 16 rows a 16-lane group reads on 16 distinct 16-byte slots.
// Preconditions (host-checked): bf16 A and B, 16-byte aligned rows, K % 16 == 0, split_k == 1, gridDim.x % 8 == 0.
// =====================================================================================================
#ifdef CSTS_GEMM3_STAMPS
#define G3_STAMP() do { if (stamp_on && nstamp < 500) stamp_buf[nstamp++] = (unsigned long long)clock64(); } while (0)
#else
#define G3_STAMP() do { } while (0)
#endif

// workgroups per CU that the ring's LDS allows (160 KiB), capped at 3: the register budget is set for exactly that many
template <int MT, int S> struct G3Occ {
  static constexpr int FIT = (160 * 1024) / (S * (64 * MT + 128) * 128);
  static constexpr int WG = FIT < 1 ? 1 : (FIT > 3 ? 3 : FIT);
};

template <int MT, int S>
__global__ __launch_bounds__(256, (G3Occ<MT, S>::WG)) void gemm3_kernel(Params p) {
#ifdef CSTS_GEMM3_STAMPS
  const bool stamp_on = p.stamps != nullptr && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 137);
  unsigned long long* stamp_buf = p.stamps + (blockIdx.x == 0 ? 0 : 512);
  int nstamp = 1;
#endif
  constexpr int BM3 = 64 * MT;                       // tile rows; waves 2 x 2, each (32*MT) x 64
  constexpr int A_BYTES = BM3 * 128, B_BYTES = 128 * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int NA = BM3 / 32, NB = 4, LPT = NA + NB;   // LDS-DMA instructions per thread per k-tile
  static_assert(32 * CS_LD * 4 <= STAGE_BYTES, "C staging (32 rows) must fit one ring stage");
  __shared__ __attribute__((aligned(1024))) char smem_raw[S * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (int)((p.K + BK2 - 1) / BK2), nk_full = (int)(p.K / BK2), k_tail = (int)((p.K % BK2) >> 4);

  // this workgroup's tiles: XCD x (= blockIdx.x % 8: workgroups are dealt round-robin over the XCDs) owns a contiguous
  // range of tiles; its workgroups walk the range with stride = workgroups on the XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, stride = gridDim.x >> 3;
  const int64_t tq = p.ntiles >> 3, tr = p.ntiles & 7;
  const int64_t t_beg = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq), t_end = t_beg + tq + (xcd < tr ? 1 : 0);
  const int64_t t_first = t_beg + slot;
  const int my_tiles = t_first < t_end ? (int)((t_end - t_first + stride - 1) / stride) : 0;
  const int total_k = my_tiles * nk;

  // ---- producer side: per-lane source rows of the pieces this wave stages (8 rows x 128 B per wave-instruction)
  // (scalar matrix base + k offset) + 32-bit lane byte offset: the saddr form of the LDS-DMA instruction (host-checked:
  // M * lda and N * ldb fit 31 bits of bytes)
  unsigned asrc[NA], bsrc[NB];
  int ach[NA], bch[NB];                              // swizzled k-chunk (x8 elements) this lane fetches
#pragma unroll
  for (int i = 0; i < NA; ++i) ach[i] = ((lane & 7) ^ ((((wave * NA + i) * 8 + (lane >> 3)) >> 1) & 7)) * 8;
#pragma unroll
  for (int i = 0; i < NB; ++i) bch[i] = ((lane & 7) ^ ((((wave * NB + i) * 8 + (lane >> 3)) >> 1) & 7)) * 8;
  auto set_tile = [&](int64_t t) {
    const int64_t m0 = (t / p.ntiles_n) * BM3, n0 = (t % p.ntiles_n) * 128;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      asrc[i] = (unsigned)(min(m0 + (wave * NA + i) * 8 + (lane >> 3), p.M - 1) * p.lda) * 2u;
#pragma unroll
    for (int i = 0; i < NB; ++i)
      bsrc[i] = (unsigned)(min(n0 + (wave * NB + i) * 8 + (lane >> 3), p.N - 1) * p.ldb) * 2u;
  };
  int64_t pt = t_first;                              // producer position: tile, k-tile inside it, flattened index
  int pk = 0, pidx = 0, pstage = 0;
  auto produce = [&]() {
    if (pidx >= total_k) return;
    char* st = smem_raw + pstage * STAGE_BYTES;
    const int64_t k0 = (int64_t)pk * BK2;
    const bool full = k0 + BK2 <= p.K;               // wave-uniform; chunks past K are fetched from chunk 0 (never used)
#pragma unroll
    for (int i = 0; i < NA; ++i)
      lds_dma16(reinterpret_cast<const char*>(p.A) + k0 * 2, asrc[i] + ((full || k0 + ach[i] < p.K) ? 2u * ach[i] : 0u),
                st + (wave * NA + i) * 1024);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      lds_dma16(reinterpret_cast<const char*>(p.B) + k0 * 2, bsrc[i] + ((full || k0 + bch[i] < p.K) ? 2u * bch[i] : 0u),
                st + A_BYTES + (wave * NB + i) * 1024);
    ++pidx;
    pstage = (pstage + 1 == S) ? 0 : pstage + 1;
    if (++pk == nk) {
      pk = 0;
      pt += stride;
      if (pidx < total_k) set_tile(pt);
    }
  };
  G3_STAMP();
  if (my_tiles > 0) set_tile(pt);
#pragma unroll
  for (int s = 0; s < S - 1; ++s) produce();
  G3_STAMP();

  // ---- consumer side.  Fragment row = obase + (lane & 31) with obase % 32 == 0: the swizzle key is a per-lane constant
  const int key = (lane >> 1) & 7, hi = lane >> 5;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = (lane & 31) * 128 + (((ks * 2 + hi) ^ key) << 4);
  int cidx = 0, cstage = 0;

  for (int64_t t = t_first; t < t_end; t += stride) {
    // The MFMA operands are SWAPPED (weights as the row operand): acc[mi][ni][r] is C[m][n] with m = the lane's row
    // (lane & 31) of 32-row unit mi and n = 32 * ni + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3) -- every lane owns runs of
    // 4 consecutive columns of ONE output row, so the epilogue needs no LDS transpose and no barrier.
    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int64_t m0 = (t / p.ntiles_n) * BM3 + wm * 32 * MT + (lane & 31), n0 = (t % p.ntiles_n) * 128 + wn * 64 + 4 * hi;
    f32x4 bias4[2][4];                               // loaded ahead of the k-loop: its wait is long over by the epilogue
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t c = n0 + ni * 32 + 8 * q;
        bias4[ni][q] = (p.bias != nullptr && c < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }

    // one k-step: wait for k-tile cidx, barrier, then NKS 16-deep sub-steps; the LDS-DMA instructions that refill the
    // stage consumed one step ago are spread over the sub-steps so that their issue runs under the MFMAs
    auto kstep = [&](auto nks_tag) {
      constexpr int NKS = decltype(nks_tag)::value;
      // k-tile cidx has landed once at most `ahead` younger k-tiles (LPT instructions each) are outstanding; epilogue
      // stores issued meanwhile are younger still, so the count only ever over-waits
      const int ahead = min(S - 2, total_k - 1 - cidx);
      if (S >= 4 && ahead == 2) wait_vm<2 * LPT>();
      else if (S >= 3 && ahead >= 1) wait_vm<LPT>();
      else wait_vm<0>();
      G3_STAMP();
      __builtin_amdgcn_s_barrier();                  // every wave's share landed; everyone is done with the previous stage
      G3_STAMP();
      const char* As = smem_raw + cstage * STAGE_BYTES;
      const char* Bs = As + A_BYTES;
      cstage = (cstage + 1 == S) ? 0 : cstage + 1;
      ++cidx;
      const bool refill = pidx < total_k;            // wave-uniform
      char* st = smem_raw + pstage * STAGE_BYTES;
      const int64_t k0 = (int64_t)pk * BK2;
      const bool full = k0 + BK2 <= p.K;             // chunks past K are fetched from chunk 0 (never used)
      bf16x8 fa[2][MT], fb[2][2];
      auto frags = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          fa[buf][i] = *reinterpret_cast<const bf16x8*>(As + (wm * 32 * MT + i * 32) * 128 + foff[ks]);
        fb[buf][0] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64) * 128 + foff[ks]);
        fb[buf][1] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + 32) * 128 + foff[ks]);
      };
      frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 1 < NKS) frags(ks + 1, (ks + 1) & 1);
        if (refill) {
#pragma unroll
          for (int i = 0; i < LPT; ++i) {
            const bool mine = (NKS == 4) ? (i * 4 / LPT == ks) : (ks == 0);
            if (!mine) continue;
            if (i < NA) {
              lds_dma16(reinterpret_cast<const char*>(p.A) + k0 * 2, asrc[i] + ((full || k0 + ach[i] < p.K) ? 2u * ach[i] : 0u),
                        st + (wave * NA + i) * 1024);
            } else {
              const int b = i - NA;
              lds_dma16(reinterpret_cast<const char*>(p.B) + k0 * 2, bsrc[b] + ((full || k0 + bch[b] < p.K) ? 2u * bch[b] : 0u),
                        st + A_BYTES + (wave * NB + b) * 1024);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[i][0] = CSTS_MFMA16(fb[ks & 1][0], fa[ks & 1][i], acc[i][0], 0, 0, 0);
          acc[i][1] = CSTS_MFMA16(fb[ks & 1][1], fa[ks & 1][i], acc[i][1], 0, 0, 0);
        }
      }
      if (refill) {                                  // advance the producer
        ++pidx;
        pstage = (pstage + 1 == S) ? 0 : pstage + 1;
        if (++pk == nk) {
          pk = 0;
          pt += stride;
          if (pidx < total_k) set_tile(pt);
        }
      }
      G3_STAMP();
    };
    for (int kt = 0; kt < nk_full; ++kt) kstep(std::integral_constant<int, 4>());
    if (k_tail == 1) kstep(std::integral_constant<int, 1>());        // K % 64 = 16, 32 or 48 (K % 16 == 0)
    else if (k_tail == 2) kstep(std::integral_constant<int, 2>());
    else if (k_tail == 3) kstep(std::integral_constant<int, 3>());

    // ---------------- epilogue straight from the accumulators (same arithmetic order as gemm2_kernel)
    G3_STAMP();
    // residual-stream form on a whole tile (see gemm2_kernel): all eight residual runs of the lane's row first, then the
    // arithmetic and the eight 16-byte stores -- no load behind a store
    if (MT == 1 && p.residual != nullptr && p.res_row_mod == 0 && p.ru_To == 0 && p.epilogue == CSTS_EPI_NONE &&
        p.c_dt == CSTS_F32 && p.r_dt == CSTS_F32 && (t / p.ntiles_n) * BM3 + BM3 <= p.M && (t % p.ntiles_n) * 128 + 128 <= p.N) {
      const float* __restrict__ res = reinterpret_cast<const float*>(p.residual);
      float* __restrict__ Cf = reinterpret_cast<float*>(p.C);
      const int64_t m = m0;
      f32x4 rr[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) rr[ni][q] = *reinterpret_cast<const f32x4*>(res + m * p.ldr + n0 + ni * 32 + 8 * q);
      const float rsc = p.row_scale != nullptr ? p.row_scale[m / p.rows_per_scale] : 1.f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[0][ni][4 * q + j] + bias4[ni][q][j];
          if (p.row_scale != nullptr) v *= rsc;
          v += rr[ni][q];
          *reinterpret_cast<f32x4*>(Cf + m * p.ldc + n0 + ni * 32 + 8 * q) = v;
        }
      G3_STAMP();
      continue;
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int64_t m = m0 + mi * 32;
      const bool mok = m < p.M;
      float rsc = 1.f;
      if (p.row_scale != nullptr && mok) rsc = p.row_scale[m / p.rows_per_scale];
      const int64_t rm = (p.residual != nullptr && p.res_row_mod > 0) ? (m % p.res_row_mod) : m;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        f32x4 o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t c = n0 + ni * 32 + 8 * q;
          const bool ok = mok && c < p.N;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][4 * q + j] + bias4[ni][q][j];
          if (p.epilogue == CSTS_EPI_DGELU) {
            f32x4 h = {0.f, 0.f, 0.f, 0.f};
            if (ok) h = ld4_as_f32(p.aux, p.aux_dt, m * p.ldaux + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= (p.aux_dt == CSTS_BF16) ? dgelu_fast(h[j]) : dgelu_f(h[j]);
          }
          o[q] = v;
        }
        if (p.epilogue == CSTS_EPI_GELU) {
          if (p.aux != nullptr) st4x4(p.aux, p.aux_dt, m * p.ldaux + n0 + ni * 32, o, mok, n0 + ni * 32, p.N, hi);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[q][j] = (p.aux_dt == CSTS_BF16) ? gelu_fast(o[q][j]) : gelu_f(o[q][j]);
        }
        if (p.row_scale != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] *= rsc;
        }
        if (p.residual != nullptr) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t c = n0 + ni * 32 + 8 * q;
            if (mok && c < p.N) o[q] += ld4_as_f32(p.residual, p.r_dt, rm * p.ldr + c);
          }
        }
        st4x4(p.C, p.c_dt, m * p.ldc + n0 + ni * 32, o, mok, n0 + ni * 32, p.N, hi);
      }
    }
    G3_STAMP();
  }
  wait_vm<0>();
  G3_STAMP();
#ifdef CSTS_GEMM3_STAMPS
  if (stamp_on) stamp_buf[0] = (unsigned long long)nstamp;
#endif
}

// =====================================================================================================
// Grouped weight gradients: MANY independent dW = dY^T X problems (every Linear of the model) in ONE launch.
// A lone weight gradient of this model has 9..36 output tiles and a reduction over 2 k..262 k tokens, so by itself it
// needs a deep split-K (slabs + finishing pass) just to occupy the chip.  Queued up for a whole backward pass there are
// thousands of (tile, token-chunk) work items: the chunk can be long (8192 tokens: a 128-step main loop) and most
// problems need no split at all.  One workgroup per item; items are independent (a problem with several chunks writes
// one partial slab per chunk, finished by csts_reduce_rows_batched).  Same tile machinery as gemm2_kernel (TN form).
// =====================================================================================================
template <bool A_F32, int MT>
__global__ __launch_bounds__(256, (MT == 4) ? 2 : ((MT == 2) ? 3 : 4)) void wgrad_grouped_kernel(const csts_wgrad_item* __restrict__ items) {
  constexpr int BM2 = 64 * MT, NTHR = 256;
  typedef Oper<false, A_F32, BM2, NTHR> OA;
  typedef Oper<false, false, 128, NTHR> OB;
  constexpr int STAGE_BYTES = (OA::LDS_ELEMS + OB::LDS_ELEMS) * 2;
  constexpr int CS_BYTES = 64 * CS_LD * 4;
  constexpr int SMEM_BYTES = STAGE_BYTES > CS_BYTES ? STAGE_BYTES : CS_BYTES;
  __shared__ __attribute__((aligned(16))) char smem_raw[SMEM_BYTES];
  bf16* As = reinterpret_cast<bf16*>(smem_raw);
  bf16* Bs = As + OA::LDS_ELEMS;
  const csts_wgrad_item it = items[blockIdx.x];
  if (it.A == nullptr) return;          // padding slot (the host equalises the per-XCD lists); block-uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = it.m0, n0 = it.n0, M = it.M, N = it.N;
  const int64_t kbeg = it.kbeg, kend = it.kend;
  const int nk = (int)((kend - kbeg + BK2 - 1) / BK2);

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int CPAIRS = BM2 / 2, TPP = NTHR / CPAIRS, KSL = BK2 / TPP;
  const bool do_colsum = it.colsum != nullptr && n0 == 0;
  float cs0 = 0.f, cs1 = 0.f;

  OA oa;
  OB ob;
  oa.load(it.A, it.lda, m0, M, kbeg, kend, tid);
  ob.load(it.B, it.ldb, n0, N, kbeg, kend, tid);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    oa.store(As, tid);
    ob.store(Bs, tid);
    __syncthreads();
    if (kt + 1 < nk) {
      oa.load(it.A, it.lda, m0, M, kbeg + (int64_t)(kt + 1) * BK2, kend, tid);
      ob.load(it.B, it.ldb, n0, N, kbeg + (int64_t)(kt + 1) * BK2, kend, tid);
    }
    if (do_colsum) {
      const int cp = tid % CPAIRS, k0s = (tid / CPAIRS) * KSL;
#pragma unroll
      for (int kk = 0; kk < KSL; ++kk) {
        const bf16x2 t = *reinterpret_cast<const bf16x2*>(&As[(k0s + kk) * OA::OC_LD + 2 * cp]);
        cs0 += (float)t[0];
        cs1 += (float)t[1];
      }
    }
#pragma unroll
    for (int ks = 0; ks < BK2 / 16; ++ks) {
      bf16x8 a[MT], b[2];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = OA::frag(As, wm * 32 * MT + i * 32, ks, lane);
      b[0] = OB::frag(Bs, wn * 64, ks, lane);
      b[1] = OB::frag(Bs, wn * 64 + 32, ks, lane);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        acc[i][0] = CSTS_MFMA16(a[i], b[0], acc[i][0], 0, 0, 0);
        acc[i][1] = CSTS_MFMA16(a[i], b[1], acc[i][1], 0, 0, 0);
      }
    }
  }
  __syncthreads();

  if (do_colsum) {
    float* red = reinterpret_cast<float*>(smem_raw);           // [TPP][BM2]
    const int cp = tid % CPAIRS, sl = tid / CPAIRS;
    red[sl * BM2 + 2 * cp] = cs0;
    red[sl * BM2 + 2 * cp + 1] = cs1;
    __syncthreads();
    if (tid < BM2 && m0 + tid < M) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < TPP; ++q) t += red[q * BM2 + tid];
      it.colsum[m0 + tid] = t;
    }
    __syncthreads();
  }

  float* Cs = reinterpret_cast<float*>(smem_raw);
  const int col = (tid & 15) * 8;
  const int64_t n = n0 + col;
  constexpr int NGRP = BM2 / 64;
#pragma unroll
  for (int g = 0; g < NGRP; ++g) {
    if (g > 0) __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int unit = wm * MT + mt;
      if ((unit >> 1) != g) continue;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[((unit & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS_LD + wn * 64 + nt * 32 + (lane & 31)] = acc[mt][nt][r];
    }
    __syncthreads();
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < 1024 / NTHR; ++i) {
      const int row = (tid >> 4) + (NTHR / 16) * i;
      const int64_t m = m0 + g * 64 + row;
      if (m >= M) continue;
      const float4 c0 = *reinterpret_cast<const float4*>(&Cs[row * CS_LD + col]);
      const float4 c1 = *reinterpret_cast<const float4*>(&Cs[row * CS_LD + col + 4]);
      float4* dst = reinterpret_cast<float4*>(it.C + m * it.ldc + n);
      dst[0] = c0;
      dst[1] = c1;
    }
  }
}

template <bool A_KC, bool B_KC, bool A_F32, bool B_F32>
void launch2(const Params& p, int bm, dim3 grid, hipStream_t s) {
  if (bm == 256) hipLaunchKernelGGL((gemm2_kernel<A_KC, B_KC, A_F32, B_F32, 4, 2>), grid, dim3(256), 0, s, p);
  else if (bm == 128) hipLaunchKernelGGL((gemm2_kernel<A_KC, B_KC, A_F32, B_F32, 2, 2>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm2_kernel<A_KC, B_KC, A_F32, B_F32, 1, 2>), grid, dim3(256), 0, s, p);
}

// v2 needs every 8-element chunk to be whole and 16-byte aligned
bool v2_ok(const csts_gemm_args* a) {
  if (a->compute != CSTS_BF16) return false;
  auto al = [](const void* ptr, int dt, int64_t ld) { return aligned16(ptr) && ld % (dt == CSTS_F32 ? 4 : 8) == 0; };
  if (!al(a->A, a->a_dt, a->lda) || !al(a->B, a->b_dt, a->ldb)) return false;
  if (a->N % 8 != 0 || a->K % 8 != 0) return false;
  if (a->layout == CSTS_GEMM_TN && a->M % 8 != 0) return false;
  if (!al(a->C, a->c_dt, a->ldc)) return false;
  if (a->aux && !al(a->aux, a->aux_dt, a->ldaux)) return false;
  if (a->residual && !al(a->residual, a->r_dt, a->ldr)) return false;
  if (a->bias && !aligned16(a->bias)) return false;
  // dtype combinations instantiated below
  if (a->layout == CSTS_GEMM_TN && a->b_dt != CSTS_BF16) return false;
  if (a->layout == CSTS_GEMM_NT && a->a_dt != CSTS_BF16) return false;
  return true;
}

// Row tile of the v2 kernel.  Mid-size problems here are latency-bound, not MFMA-bound (K is short: 96..3072), so the
// tile is chosen for resident workgroups per CU rather than for arithmetic intensity; table fitted to sweeps of the
// CSTS shapes on MI355X (tools/gemm_tile_sweep.py, profiles/r1_gemm_tile_sweep.txt).
int pick_tile_rows(const csts_gemm_args* a, int64_t per) {
  const bool a_f32 = a->a_dt == CSTS_F32;
  if (a->tile_rows == 64 || a->tile_rows == 128 || (a->tile_rows == 256 && !a_f32)) return a->tile_rows;
  int mt;
  if (a->layout == CSTS_GEMM_TN) {
    // weight gradients (split-K fills the chip): 128 rows, 256 only where it pads M no further and still leaves
    // one workgroup per CU
    const int64_t pad128 = cdiv(a->M, 128) * 128, pad256 = cdiv(a->M, 256) * 256;
    mt = (!a_f32 && pad256 == pad128 && cdiv(a->M, 256) * per >= 256) ? 256 : 128;
  } else {
    // activations x weights, K = 96..3072: 128 rows once that gives >= 2 workgroups per CU, else 64 (whole-step sweep of the
    // threshold, same box: 384 / 512 / 640 / 768 / 1024 -> 24.60 / 24.58 / 24.66 / 24.66 / 24.66 ms)
    mt = (cdiv(a->M, 128) * per >= 512) ? 128 : 64;
  }
  while (mt > 64 && a->M <= mt / 2) mt /= 2;   // never tile wider than the problem
  return mt;
}

// gemm3 (LDS-DMA ring, NT bf16 x bf16): usable when every 16-byte chunk is whole and K has only whole MFMA sub-steps
bool v3_ok(const csts_gemm_args* a, int split) {
  return v2_ok(a) && a->layout == CSTS_GEMM_NT && a->a_dt == CSTS_BF16 && a->b_dt == CSTS_BF16 && a->K % 16 == 0 &&
         split == 1 && a->colsum == nullptr &&
         a->M * a->lda < (int64_t(1) << 30) && a->N * a->ldb < (int64_t(1) << 30);   // 32-bit lane byte offsets of the LDS-DMA
}
// Library heuristic for the persistent LDS-DMA kernel (64-row tiles, 3-stage ring, 2 workgroups per CU).
bool pick3(const csts_gemm_args* a, int split, int* mt, int* stages) {
  if (a->algo != 0 || a->tile_rows != 0 || !v3_ok(a, split) || a->M < 256 || a->res_up[3] > 0) return false;
  // Measured inside the train step (profiles/r1_v7_gemm_shapes.txt vs r1_v6): the ring wins where a workgroup's k-loop is
  // long and the grid is small (< 2 128-row tiles per CU, K >= 768: fc2 / qkv of the 384- and 768-channel stages,
  // 20-40 % faster); on the large-M short-K shapes the register-staged kernel at 3-4 workgroups per CU stays ahead.
  if (cdiv(a->M, 128) * cdiv(a->N, BN) >= 512 || a->K < 768) return false;
  *mt = 1;
  *stages = 3;
  return true;
}

// Library heuristic for the 8-wave LDS-DMA kernel (gemm4.hip): filled in from tools/gemm4_lab.py measurements.
// Measured on MI355X against the register-staged / 4-wave ring kernels (tools/gemm4_lab.py, profiles/r2_gemm4_lab.txt):
// 128 x 192 tiles on 8 waves, 2 workgroups per CU, win 1.1-1.3 x wherever the grid gives >= 1 tile per workgroup slot; with
// exactly one round of tiles and a long K the 3-stage ring (1 workgroup per CU) is ahead; problems with few 192-wide tiles
// take 128 x 128 tiles on 8 waves.  The 256-row variants (256 x 128 / 192 / 256) measured 5-25 % SLOWER than these on every
// CSTS shape: with K = 384..3072 a workgroup's k-loop is 6-48 steps, and bytes in flight per CU (resident workgroups),
// not FLOP per staged byte, decide.  CSTS_GEMM4=0 switches the family off (A/B runs).
bool pick4(const csts_gemm_args* a, int split, int* variant) {
  static const bool enabled = [] { const char* e = getenv("CSTS_GEMM4"); return !(e && e[0] == '0'); }();
  if (!enabled || a->algo != 0 || a->tile_rows != 0 || !v3_ok(a, split) || a->K % 64 != 0 || a->M < 2048 || a->res_up[3] > 0) return false;
  // Inside the train step (bench.py --dump-gemm, same-box A/B, profiles/r2_gemm4_instep_ab.txt) the family wins 5-23 % on
  // the GEMMs with a bf16 output (qkv, fc1 + GELU, the data gradients that feed bf16) and LOSES 5-15 % on the fp32
  // residual-stream outputs (proj, fc2: C and the residual are 4 bytes per element and the register epilogue reaches them in
  // 32-byte row segments, where gemm2's LDS-staged epilogue moves whole 128-byte lines): bf16 outputs only.
  // Round 4 (end): the fp32 residual-stream outputs have their own form now (gemm4 FORM 5: unswapped MFMA operands, so that the
  // register epilogue touches whole 128-byte row runs): the whole-tile N % 192 == 0, K % 64 == 0 cases without an up-sampled skip take it
  // (inside the replayed step 0.70-1.00 x the time of gemm2 / gemm3 on every routed shape, -0.1 ms per step in all:
  // profiles/r4_gemm4_res_form.txt).  CSTS_GEMM4_RES=0 is the A/B switch.
  if (a->c_dt == CSTS_F32 && a->residual != nullptr) {
    static const bool res_on = [] { const char* e = getenv("CSTS_GEMM4_RES"); return !(e && e[0] == '0'); }();
    if (!res_on || a->r_dt != CSTS_F32 || a->res_row_mod != 0 || a->epilogue != CSTS_EPI_NONE || a->M % 128 != 0) return false;
    // (narrow outputs -- N = 96 on 128 x 128 tiles with the last 32-column block masked, algo 433 -- measured 0.99 x gemm2 inside the step: not routed)
    // Round 5: 64 x 192 tiles on 4-wave workgroups (variant 73) where the problem has at most 128 tiles of 128 x 192 -- half the CUs, or
    // more, idle otherwise (M = 8192 / 2048 rows of the 384- / 768-channel stages: isolated 1.1 - 1.17 x, tools/gemm4_lab.py --shapes
    // tools/shapes_m8192.txt).  CSTS_GEMM4_T64=0 is the A/B switch.
    static const bool t64 = [] { const char* e = getenv("CSTS_GEMM4_T64"); return e && (e[0] == '1' || e[0] == '3'); }();
    if (t64 && a->N % 192 == 0 && a->M % 64 == 0 && a->K >= 384 && (a->M / 128) * (a->N / 192) <= 128 && (a->M / 64) * (a->N / 192) >= 64) {
      *variant = 73;
      return true;
    }
    if (a->N % 192 != 0 || (a->M / 128) * (a->N / 192) < 128) return false;
    *variant = 63;        // the 3-stage ring at one workgroup per CU: FORM 5 needs more than the 128 registers of the two-workgroup variants (62: 1.6 x slower)
    return true;
  }
  if (a->c_dt != CSTS_BF16 || a->residual != nullptr) return false;
  const int64_t rows = cdiv(a->M, 128);
  {   // the same 64 x 192 form for the bf16 outputs of those problems (whole tiles only: the specialised epilogues)
    static const bool t64b = [] { const char* e = getenv("CSTS_GEMM4_T64"); return e && (e[0] == '2' || e[0] == '3'); }();
    if (t64b && a->N % 192 == 0 && a->M % 64 == 0 && a->K >= 384 && rows * (a->N / 192) <= 128 && (a->M / 64) * (a->N / 192) >= 64 && a->row_scale == nullptr) {
      *variant = 73;
      return true;
    }
  }
  if (a->N % 192 == 0 && rows * (a->N / 192) >= 256) {
    *variant = (a->K >= 1536 && rows * (a->N / 192) <= 256) ? 63 : 62;
    // short K, many row tiles, whole tiles: the wave-specialised form (4 producer waves; gemm4.hip) -- OFF by default:
    // -4 .. -11 % on six of ten isolated cases, but +0.18 ms per step inside the two-stream step (one 12-wave workgroup per
    // CU), profiles/r3_shortk_gemm_stamps.txt.  CSTS_GEMM4_SPLIT=1 switches it on for A/B runs.
    // experiment (CSTS_GEMM4_SHORTK=64): K <= 192 on a 4-stage ring -- every k-tile of the NEXT tile is requested before this
    // tile's epilogue stores, and the store-aware waits (gemm4.hip kstep) then never drain a store
    static const int shortk = [] { const char* e = getenv("CSTS_GEMM4_SHORTK"); return e ? atoi(e) : 0; }();
    if (shortk == 64 && a->K <= 192 && a->M >= 32768) { *variant = 64; return true; }
    static const bool split_on = [] { const char* e = getenv("CSTS_GEMM4_SPLIT"); return e && e[0] == '1'; }();
    if (split_on && a->K <= 192 && a->M % 128 == 0 && a->M >= 32768 && a->row_scale == nullptr) *variant = 83;
    return true;
  }
  const int64_t t128 = rows * cdiv(a->N, 128);
  if (a->N % 128 == 0 && t128 >= 128 && t128 <= 512) {
    *variant = a->K >= 1024 ? 33 : 32;
    return true;
  }
  return false;
}

// Library heuristic for the streaming thin-operand kernel (gemm5.hip): every problem it can run with at least one 32-row unit per wave of
// the persistent grid twice over.  CSTS_GEMM5=0 switches it off (A/B runs).
bool pick5(const csts_gemm_args* a, int split) {
  static const bool enabled = [] { const char* e = getenv("CSTS_GEMM5"); return !(e && e[0] == '0'); }();
  if (!enabled || a->algo != 0 || a->tile_rows != 0 || a->M < 16384) return false;
  // the four-chunk ring (K = 384, N <= 384; six waves beside 75 KB of weights): isolated 1.06 - 1.13 x on the bf16 forms, 0.9 - 1.0 x on the fp32
  // residual form; inside the step 20.21 -> 20.24 ms (gpurun_out/r5ak): OFF, CSTS_GEMM5_K384=1 or algo 500 opts in
  static const bool k384 = [] { const char* e = getenv("CSTS_GEMM5_K384"); return e && e[0] == '1'; }();
  if (a->K == 384 && !k384) return false;
  return csts_gemm5_ok(a, split);
}

template <int MT>
void launch3_mt(const Params& p, int stages, dim3 grid, hipStream_t s) {
  if (stages == 2) hipLaunchKernelGGL((gemm3_kernel<MT, 2>), grid, dim3(256), 0, s, p);
  else if (stages == 4 && MT < 4) hipLaunchKernelGGL((gemm3_kernel<MT, (MT < 4 ? 4 : 3)>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm3_kernel<MT, 3>), grid, dim3(256), 0, s, p);
}
// persistent launch: min(tiles, 256 CUs x workgroups that fit a CU's 160 KiB of LDS) workgroups, a multiple of 8
void launch3(Params p, const csts_gemm_args* a, int mt, int stages, int wpc, hipStream_t s) {
  const int64_t ntiles = cdiv(a->M, 64 * mt) * p.ntiles_n;
  p.ntiles = ntiles;
  p.stamps = (a->split_k <= 1 && a->workspace != nullptr && a->ws_bytes >= 8192) ? reinterpret_cast<unsigned long long*>(a->workspace) : nullptr;
  const int stage_bytes = (64 * mt + 128) * 128;
  const int fit = std::max(1, (160 * 1024) / (stages * stage_bytes));
  if (wpc <= 0 || wpc > fit) wpc = fit;
  int64_t g = std::min<int64_t>(cdiv(ntiles, 8) * 8, (int64_t)256 * wpc);
  dim3 grid((unsigned)g, 1, 1);
  if (mt == 1) launch3_mt<1>(p, stages, grid, s);
  else if (mt == 2) launch3_mt<2>(p, stages, grid, s);
  else launch3_mt<4>(p, stages, grid, s);
}

template <bool A_KC, bool B_KC>
void launch(const Params& p, int compute, dim3 grid, hipStream_t s) {
  if (compute == CSTS_F32) hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, true>), grid, dim3(NT_), 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, false>), grid, dim3(NT_), 0, s, p);
}


// ---------------------------------------------------------------------------------------------------------------- tiny problems
// fp32 GEMMs with a handful of outputs or a handful of reduction steps (the EgoNCE path: 256-d projections of B = 4 token
// means, their weight / data gradients, the B x B similarity matrix -- losses.py:26-56, custom_multimodal_builder.py:483-497)
// cost 16-41 us each through the tiled MFMA kernel (LDS staging, 64 x 128 tiles, split-K slabs + a finishing pass for
// ~1 k outputs).  Here: THREAD mode (K <= 16: one thread per output, K FMAs) or WAVE mode (one wave per output, lanes stride
// over K, fixed-order butterfly reduction).  Plain fp32 FMAs in a fixed order: reproducible run to run; bias only.
template <int LAYOUT>
__device__ __forceinline__ float tiny_a(const float* A, int64_t lda, int64_t m, int64_t k) {
  return LAYOUT == CSTS_GEMM_TN ? A[k * lda + m] : A[m * lda + k];
}
template <int LAYOUT>
__device__ __forceinline__ float tiny_b(const float* B, int64_t ldb, int64_t n, int64_t k) {
  return LAYOUT == CSTS_GEMM_NT ? B[n * ldb + k] : B[k * ldb + n];
}
template <int LAYOUT, bool PER_WAVE>
__global__ __launch_bounds__(256) void gemm_tiny_kernel(Params p) {
  const float* A = reinterpret_cast<const float*>(p.A);
  const float* B = reinterpret_cast<const float*>(p.B);
  float* Cm = reinterpret_cast<float*>(p.C);
  const int64_t total = p.M * p.N;
  if (PER_WAVE) {
    const int lane = threadIdx.x & 63;
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= total) return;                       // wave-uniform
    const int64_t m = o / p.N, n = o - m * p.N;
    float s = 0.f;
    for (int64_t k = lane; k < p.K; k += 64) s = __builtin_fmaf(tiny_a<LAYOUT>(A, p.lda, m, k), tiny_b<LAYOUT>(B, p.ldb, n, k), s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) Cm[m * p.ldc + n] = s + (p.bias ? p.bias[n] : 0.f);
  } else {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int64_t m = o / p.N, n = o - m * p.N;
    float s = 0.f;
    for (int64_t k = 0; k < p.K; ++k) s = __builtin_fmaf(tiny_a<LAYOUT>(A, p.lda, m, k), tiny_b<LAYOUT>(B, p.ldb, n, k), s);
    Cm[m * p.ldc + n] = s + (p.bias ? p.bias[n] : 0.f);
  }
}
static bool tiny_ok(const csts_gemm_args* a) {
  static const bool off = getenv("CSTS_GEMM_TINY") != nullptr && getenv("CSTS_GEMM_TINY")[0] == '0';
  if (off || a->compute != CSTS_F32 || a->a_dt != CSTS_F32 || a->b_dt != CSTS_F32 || a->c_dt != CSTS_F32) return false;
  if (a->epilogue != CSTS_EPI_NONE || a->residual || a->row_scale || a->aux || a->colsum || a->algo != 0) return false;
  const int64_t outs = a->M * a->N;
  return (a->K <= 16 && outs <= (1 << 20)) || (outs <= 8192 && a->K <= 8192);
}
template <int LAYOUT>
static void tiny_launch(const Params& p, hipStream_t s) {
  const int64_t outs = p.M * p.N;
  if (p.K <= 16) hipLaunchKernelGGL((gemm_tiny_kernel<LAYOUT, false>), dim3((unsigned)cdiv(outs, 256)), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_tiny_kernel<LAYOUT, true>), dim3((unsigned)cdiv(outs, 4)), dim3(256), 0, s, p);
}

}  // namespace

extern "C" int csts_gemm(const csts_gemm_args* a, hipStream_t stream) {
  CSTS_REQUIRE(a != nullptr, "null args");
  CSTS_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "empty problem");
  CSTS_REQUIRE(a->layout >= CSTS_GEMM_NT && a->layout <= CSTS_GEMM_TN, "bad layout");
  CSTS_REQUIRE(a->A && a->B && a->C, "null operand");
  CSTS_REQUIRE(a->compute == CSTS_F32 || a->compute == CSTS_BF16, "bad compute dtype");
  const int split = a->split_k > 1 ? a->split_k : 1;
  const bool det = split > 1 && a->workspace != nullptr;   // deterministic: partial slabs + finishing pass
  if (split > 1 && !det) {
    CSTS_REQUIRE(a->c_dt == CSTS_F32, "split-k accumulates with fp32 atomics: C must be f32 (pre-zeroed)");
    CSTS_REQUIRE(a->epilogue == CSTS_EPI_NONE && !a->residual && !a->row_scale, "atomic split-k allows bias only");
  }
  if (a->epilogue == CSTS_EPI_DGELU) CSTS_REQUIRE(a->aux != nullptr, "DGELU needs aux (pre-activation)");
  if (tiny_ok(a)) {
    Params q{};
    q.A = a->A; q.B = a->B; q.C = a->C; q.bias = a->bias;
    q.lda = a->lda; q.ldb = a->ldb; q.ldc = a->ldc; q.M = a->M; q.N = a->N; q.K = a->K;
    if (a->layout == CSTS_GEMM_NT) tiny_launch<CSTS_GEMM_NT>(q, stream);
    else if (a->layout == CSTS_GEMM_NN) tiny_launch<CSTS_GEMM_NN>(q, stream);
    else tiny_launch<CSTS_GEMM_TN>(q, stream);
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  Params p;
  p.A = a->A; p.B = a->B; p.C = a->C; p.bias = a->bias; p.aux = a->aux; p.residual = a->residual;
  p.row_scale = a->row_scale;
  p.ws = det ? reinterpret_cast<float*>(a->workspace) : nullptr;
  p.colsum = a->colsum;
  p.colsum_ws = nullptr;
  p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldaux = a->ldaux; p.ldr = a->ldr;
  p.M = a->M; p.N = a->N; p.K = a->K; p.res_row_mod = a->res_row_mod;
  p.rows_per_scale = a->rows_per_scale > 0 ? a->rows_per_scale : 1;
  p.a_dt = a->a_dt; p.b_dt = a->b_dt; p.c_dt = a->c_dt; p.aux_dt = a->aux_dt; p.r_dt = a->r_dt;
  p.epilogue = a->epilogue; p.split_k = split;
  static const int res_lines = [] { const char* e = getenv("CSTS_GEMM_RES_LINES"); return (e && e[0] == '0') ? 0 : 1; }();
  p.res_lines = res_lines;
  p.ru_Ti = p.ru_Hi = p.ru_Wi = p.ru_To = p.ru_Ho = p.ru_Wo = p.ru_lw = p.ru_lh = p.ru_lt = 0;
  if (a->res_up[3] > 0) {
    const int Ti = a->res_up[0], Hi = a->res_up[1], Wi = a->res_up[2], To = a->res_up[3], Ho = a->res_up[4], Wo = a->res_up[5];
    auto p2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    CSTS_REQUIRE(a->residual != nullptr && a->res_row_mod == 0 && split == 1, "res_up needs a residual, no res_row_mod, split_k == 1");
    CSTS_REQUIRE(Ti > 0 && Hi > 0 && Wi > 0 && p2(To) && p2(Ho) && p2(Wo), "res_up: fine grid sizes must be powers of two");
    CSTS_REQUIRE(To >= Ti && Ho >= Hi && Wo >= Wi && a->M % ((int64_t)To * Ho * Wo) == 0, "res_up: M must be B * To * Ho * Wo");
    CSTS_REQUIRE(a->algo % 1000 < 300, "res_up is not available in the persistent LDS-DMA kernels");
    p.ru_Ti = Ti; p.ru_Hi = Hi; p.ru_Wi = Wi; p.ru_To = To; p.ru_Ho = Ho; p.ru_Wo = Wo;
    p.ru_lw = lg(Wo); p.ru_lh = lg(Ho); p.ru_lt = lg(To);
  }
  auto vec_ok = [](const void* ptr, int dt, int64_t ld) {
    return aligned16(ptr) && (ld % (dt == CSTS_F32 ? 4 : 8) == 0);
  };
  p.a_vec = vec_ok(a->A, a->a_dt, a->lda);
  p.b_vec = vec_ok(a->B, a->b_dt, a->ldb);
  const bool use_v2 = v2_ok(a);
  const int bk = use_v2 ? BK2 : BK;
  const int64_t ktiles = cdiv(a->K, bk);
  p.k_chunk = cdiv(ktiles, split) * bk;
  const int64_t nsplit = cdiv(a->K, p.k_chunk);
  if (det) CSTS_REQUIRE(a->ws_bytes >= (size_t)nsplit * a->M * a->N * sizeof(float), "split-k workspace too small");
  if (a->colsum != nullptr) {
    CSTS_REQUIRE(use_v2 && a->layout == CSTS_GEMM_TN, "fused colsum needs the TN bf16 v2 kernel (see csts_gemm_v2_eligible)");
    CSTS_REQUIRE(split == 1 || det, "fused colsum with split-k needs the deterministic workspace");
    if (split > 1) {
      CSTS_REQUIRE(a->ws_bytes >= (size_t)nsplit * a->M * (a->N + 1) * sizeof(float), "workspace too small for colsum partials");
      p.colsum_ws = p.ws + nsplit * a->M * a->N;
    }
  }
  p.ntiles_n = (int)cdiv(a->N, BN);
  int64_t mtiles = cdiv(a->M, BM);
  CSTS_REQUIRE(mtiles * p.ntiles_n < (int64_t)1 << 31, "grid too large");
  if (a->algo % 1000 >= 300 && a->algo % 1000 < 400) {   // forced: 1000 * workgroups-per-CU (0 = as many as LDS allows) + 300 + 10 * (tile_rows / 64) + stages
    CSTS_REQUIRE(v3_ok(a, split), "algo 3xx (persistent LDS-DMA NT kernel) not applicable to this problem");
    const int code = a->algo % 1000 - 300, mt = code / 10, st = code % 10;
    CSTS_REQUIRE((mt == 1 || mt == 2 || mt == 4) && st >= 2 && st <= 4 && !(mt == 4 && st == 4), "bad algo 3xx code");
    launch3(p, a, mt, st, a->algo / 1000, stream);
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  if (a->algo % 1000 >= 400 && a->algo % 1000 < 500) {   // forced: 1000 * workgroups-per-CU + 400 + gemm4 variant code (gemm4.hip)
    CSTS_REQUIRE(v3_ok(a, split), "algo 4xx (8-wave LDS-DMA NT kernel) not applicable to this problem");
    CSTS_REQUIRE(a->K % 64 == 0, "algo 4xx needs K % 64 == 0 (no k-tail instantiations)");
    CSTS_REQUIRE(csts_gemm4_launch(p, a, a->algo % 1000 - 400, a->algo / 1000, stream), "unknown gemm4 variant");
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  if (a->algo % 1000 == 500 || a->algo % 1000 == 503 || a->algo % 1000 == 506) {   // forced: the streaming thin-operand kernel (gemm5.hip; 503 / 506: 96 / 192 columns per workgroup at K = 192)
    CSTS_REQUIRE(csts_gemm5_ok(a, split), "algo 500 (streaming thin-operand NT kernel) not applicable to this problem");
    CSTS_REQUIRE(csts_gemm5_launch(p, a, stream), "gemm5 launch failed");
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  if (pick5(a, split)) {
    CSTS_REQUIRE(csts_gemm5_launch(p, a, stream), "gemm5 launch failed");
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  {
    int v4;
    if (pick4(a, split, &v4)) {
      CSTS_REQUIRE(csts_gemm4_launch(p, a, v4, 0, stream), "gemm4 heuristic picked a variant that is not instantiated");
      CSTS_LAUNCH_CHECK();
      return 0;
    }
    int mt3, st3;
    if (pick3(a, split, &mt3, &st3)) {
      launch3(p, a, mt3, st3, 0, stream);
      CSTS_LAUNCH_CHECK();
      return 0;
    }
  }
  if (use_v2) {
    int mt = pick_tile_rows(a, p.ntiles_n * nsplit);
    mtiles = cdiv(a->M, mt);
    dim3 grid2((unsigned)(mtiles * p.ntiles_n), (unsigned)nsplit, 1);
    const bool af = a->a_dt == CSTS_F32, bf = a->b_dt == CSTS_F32;
    if (a->layout == CSTS_GEMM_NT) {
      if (bf) launch2<true, true, false, true>(p, mt, grid2, stream);
      else launch2<true, true, false, false>(p, mt, grid2, stream);            // bf16 shadow weights
    } else if (a->layout == CSTS_GEMM_NN) {
      if (af && bf) launch2<true, false, true, true>(p, mt, grid2, stream);
      else if (!af && bf) launch2<true, false, false, true>(p, mt, grid2, stream);
      else if (af) launch2<true, false, true, false>(p, mt, grid2, stream);
      else launch2<true, false, false, false>(p, mt, grid2, stream);
    } else {
      if (af) launch2<false, false, true, false>(p, mt, grid2, stream);
      else launch2<false, false, false, false>(p, mt, grid2, stream);
    }
    CSTS_LAUNCH_CHECK();
    if (det) {
      launch_finish(p, (int)nsplit, stream);
      CSTS_LAUNCH_CHECK();
    }
    return 0;
  }
  dim3 grid((unsigned)(mtiles * p.ntiles_n), (unsigned)nsplit, 1);
  switch (a->layout) {
    case CSTS_GEMM_NT: launch<true, true>(p, a->compute, grid, stream); break;
    case CSTS_GEMM_NN: launch<true, false>(p, a->compute, grid, stream); break;
    default: launch<false, false>(p, a->compute, grid, stream); break;
  }
  CSTS_LAUNCH_CHECK();
  if (det) {
    launch_finish(p, (int)nsplit, stream);
    CSTS_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int csts_gemm_v2_eligible(const csts_gemm_args* a) { return a != nullptr && v2_ok(a) ? 1 : 0; }

// Which kernel csts_gemm would launch for these arguments (host-only; used by bench.py to name the kernel a timed call
// ran, exactly as rocprofv3 prints it): v2 = 1 -> gemm2_kernel<A_KC, B_KC, A_F32, B_F32, tile_rows / 64, 2>; v2 = 30 + stages ->
// gemm3_kernel<tile_rows / 64, stages>; 0 -> gemm_kernel; -1 -> gemm_tiny_kernel.
extern "C" int csts_gemm_plan(const csts_gemm_args* a, int* v2, int* tile_rows, int* nsplit) {
  CSTS_REQUIRE(a != nullptr && v2 && tile_rows && nsplit, "null pointer");
  const int split = a->split_k > 1 ? a->split_k : 1;
  const bool use_v2 = v2_ok(a);
  const int bk = use_v2 ? BK2 : BK;
  const int64_t k_chunk = cdiv(cdiv(a->K, bk), split) * bk;
  const int64_t ns = cdiv(a->K, k_chunk);
  int mt3, st3, v4;
  if (tiny_ok(a)) {                     // gemm_tiny_kernel<layout, wave mode>
    *v2 = -1;
    *nsplit = 1;
    *tile_rows = 0;
    return 0;
  }
  if (pick5(a, split)) {               // gemm5 (name: csts_gemm_kernel_name)
    *v2 = 500;
    *nsplit = 1;
    *tile_rows = 0;
    return 0;
  }
  if (pick4(a, split, &v4)) {          // gemm4 variant v4 (name: csts_gemm_kernel_name)
    *v2 = 400 + v4;
    *nsplit = 1;
    *tile_rows = 0;
    return 0;
  }
  if (pick3(a, split, &mt3, &st3)) {   // gemm3_kernel<tile_rows / 64, stages>
    *v2 = 30 + st3;
    *nsplit = 1;
    *tile_rows = 64 * mt3;
    return 0;
  }
  *v2 = use_v2 ? 1 : 0;
  *nsplit = (int)ns;
  *tile_rows = use_v2 ? pick_tile_rows(a, cdiv(a->N, BN) * ns) : BM;
  return 0;
}

// The same decision as a kernel NAME, spelled as rocprofv3 prints it (bench.py attributes its HIP-event timings to it).
extern "C" int csts_gemm_kernel_name(const csts_gemm_args* a, char* buf, int buflen, int* nsplit) {
  CSTS_REQUIRE(a != nullptr && buf != nullptr && buflen > 0 && nsplit != nullptr, "null pointer");
  int v2 = 0, rows = 0;
  if (csts_gemm_plan(a, &v2, &rows, nsplit) != 0) return -1;
  auto tf = [](bool b) { return b ? "true" : "false"; };
  Params q{};          // what the gemm4 epilogue-form choice looks at
  q.M = a->M; q.N = a->N; q.K = a->K; q.c_dt = a->c_dt; q.residual = a->residual; q.row_scale = a->row_scale; q.bias = a->bias;
  q.epilogue = a->epilogue; q.aux = a->aux; q.aux_dt = a->aux_dt;
  if (a->algo % 1000 >= 400 && a->algo % 1000 < 500) {
    *nsplit = 1;
    return csts_gemm4_name(q, a->algo % 1000 - 400, buf, buflen) ? 0 : -1;
  }
  if (a->algo % 1000 == 500 || a->algo % 1000 == 503 || a->algo % 1000 == 506 || v2 == 500) {
    *nsplit = 1;
    return csts_gemm5_name(a, buf, buflen) ? 0 : -1;
  }
  if (v2 < 0) {
    snprintf(buf, buflen, "gemm_tiny_kernel<%d, %s>", a->layout, tf(a->K > 16));
    return 0;
  }
  if (v2 >= 400) return csts_gemm4_name(q, v2 - 400, buf, buflen) ? 0 : -1;
  if (v2 >= 30) snprintf(buf, buflen, "gemm3_kernel<%d, %d>", rows / 64, v2 - 30);
  else if (v2)
    snprintf(buf, buflen, "gemm2_kernel<%s, %s, %s, %s, %d, 2>", tf(a->layout != CSTS_GEMM_TN), tf(a->layout == CSTS_GEMM_NT),
             tf(a->a_dt == CSTS_F32), tf(a->b_dt == CSTS_F32), rows / 64);
  else
    snprintf(buf, buflen, "gemm_kernel<%s, %s, %s>", tf(a->layout != CSTS_GEMM_TN), tf(a->layout == CSTS_GEMM_NT),
             tf(a->compute == CSTS_F32));
  return 0;
}

extern "C" size_t csts_gemm_splitk_workspace(int64_t M, int64_t N, int64_t K, int split_k) {
  if (split_k <= 1) return 0;
  // upper bound over both kernels' k-chunking (v1: BK 32, v2: BK 64)
  const int64_t c1 = cdiv(cdiv(K, BK), split_k) * BK, c2 = cdiv(cdiv(K, BK2), split_k) * BK2;
  return (size_t)std::max(cdiv(K, c1), cdiv(K, c2)) * M * (N + 1) * sizeof(float);   // + colsum partials
}

// items: DEVICE array.  Every item: A = dY [tokens][M] (a_f32 ? fp32 : bf16, lda), B = X [tokens][N] bf16 (ldb), C fp32
// [M][N] tile target (ldc; the weight gradient itself or one chunk's partial slab), tokens [kbeg, kend), tile origin
// (m0, n0), rows = tile height (64 or 128; the whole launch uses one height).  colsum (optional): sum over the chunk's
// tokens of dY[:, m] for the tile's rows (bias gradient), written by the n0 == 0 tiles.
extern "C" int csts_wgrad_grouped(const csts_wgrad_item* device_items, int nitems, int a_f32, int tile_rows, hipStream_t stream) {
  CSTS_REQUIRE(device_items != nullptr && nitems > 0, "no items");
  CSTS_REQUIRE(tile_rows == 64 || tile_rows == 128 || (tile_rows == 256 && !a_f32), "tile_rows must be 64, 128 or (bf16 dY) 256");
  const dim3 grid((unsigned)nitems), block(256);
  if (tile_rows == 256) {
    hipLaunchKernelGGL((wgrad_grouped_kernel<false, 4>), grid, block, 0, stream, device_items);
  } else if (a_f32) {
    if (tile_rows == 128) hipLaunchKernelGGL((wgrad_grouped_kernel<true, 2>), grid, block, 0, stream, device_items);
    else hipLaunchKernelGGL((wgrad_grouped_kernel<true, 1>), grid, block, 0, stream, device_items);
  } else {
    if (tile_rows == 128) hipLaunchKernelGGL((wgrad_grouped_kernel<false, 2>), grid, block, 0, stream, device_items);
    else hipLaunchKernelGGL((wgrad_grouped_kernel<false, 1>), grid, block, 0, stream, device_items);
  }
  CSTS_LAUNCH_CHECK();
  return 0;
}
