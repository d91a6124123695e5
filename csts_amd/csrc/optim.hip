// Multi-tensor optimizer step of the CSTS training iteration: the L2 gradient-norm clip of
// tools/train_avgaze_net.py:105-106 (torch.nn.utils.clip_grad_norm_, max_norm 1.0) and the AdamW update of
// slowfast/models/optimizer.py:85-93 (torch.optim.AdamW, eps 1e-8, weight decay 0 on 1-D parameters and biases),
// over all ~524 parameter tensors in TWO launches instead of ~65, with the bf16 shadow weights the GEMMs read
// refreshed by the same pass.  Pure HBM streaming: 30 bytes per parameter (read p, g, m, v; write p, m, v, w16).
//
// Layout: the host describes the parameter set once as a CHUNK list (chunk -> tensor id, element offset; chunks never
// straddle tensors) plus a per-tensor table; only the gradient pointers change between steps.  Step count, learning
// rate and the clip coefficient live in device memory, so a captured HIP graph replays the schedule correctly.
#include "common.h"
#include <algorithm>

namespace {

constexpr int OPT_THREADS = 256;
constexpr int OPT_VEC = 4;

struct OptTables {
  const int32_t* chunk_tensor;   // [nchunks]
  const int64_t* chunk_off;      // [nchunks] element offset inside the tensor
  const csts_opt_tensor* tensors;  // [ntensors]
  const void* const* grads;      // [ntensors] gradient pointers (null: parameter skipped, like torch)
};

template <bool G16> __device__ __forceinline__ float ldg(const void* g, int64_t i) {
  if constexpr (G16) return (float)reinterpret_cast<const bf16*>(g)[i];
  else return reinterpret_cast<const float*>(g)[i];
}
template <bool G16> __device__ __forceinline__ float4 ldg4(const void* g, int64_t i4) {
  if constexpr (G16) {
    const bf16x4 v = ld_stream(reinterpret_cast<const bf16x4*>(g) + i4);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  } else {
    const f32x4 v = ld_stream(reinterpret_cast<const f32x4*>(g) + i4);
    return make_float4(v[0], v[1], v[2], v[3]);
  }
}
__device__ __forceinline__ float4 ld4s(const float* p, int64_t i4) {
  const f32x4 v = ld_stream(reinterpret_cast<const f32x4*>(p) + i4);
  return make_float4(v[0], v[1], v[2], v[3]);
}

// partial[chunk] = sum g^2 over the chunk (fixed summation order: reproducible)
template <bool G16>
__global__ __launch_bounds__(OPT_THREADS) void opt_sqnorm_kernel(OptTables t, int chunk_elems, float* __restrict__ partial,
                                                                 const float* __restrict__ scaler) {
  __shared__ float red[OPT_THREADS / WAVE];
  const int c = blockIdx.x;
  const int ti = t.chunk_tensor[c];
  const int64_t off = t.chunk_off[c];
  const char* g = reinterpret_cast<const char*>(t.grads[ti]);
  const float inv = scaler != nullptr ? 1.f / scaler[0] : 1.f;      // scaler.unscale_: the norm is that of g / scale
  float s = 0.f;
  if (g != nullptr) {
    const int64_t n = min((int64_t)chunk_elems, t.tensors[ti].n - off);
    g += off * (G16 ? 2 : 4);
    const bool vec = (reinterpret_cast<uintptr_t>(g) & (G16 ? 7 : 15)) == 0;
    if (vec) {
      const int64_t n4 = n / OPT_VEC;
      for (int64_t i = threadIdx.x; i < n4; i += OPT_THREADS) {
        float4 v = ldg4<G16>(g, i);
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
      for (int64_t i = n4 * OPT_VEC + threadIdx.x; i < n; i += OPT_THREADS) { const float x = ldg<G16>(g, i) * inv; s += x * x; }
    } else {
      for (int64_t i = threadIdx.x; i < n; i += OPT_THREADS) { const float x = ldg<G16>(g, i) * inv; s += x * x; }
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < OPT_THREADS / WAVE; ++w) tot += red[w];
    partial[c] = tot;
  }
}

// state[0] = step (float, incremented here), state[1] = total grad norm (written here), state[2] = clip coefficient (times
// 1 / loss scale), state[3] = 1 when this step is skipped (non-finite gradients under a loss scaler), else 0
__global__ __launch_bounds__(1024) void opt_norm_finish_kernel(const float* __restrict__ partial, int nchunks, float max_norm,
                                                               float* __restrict__ state, float* __restrict__ scaler, float growth,
                                                               float backoff, int growth_interval, const float* __restrict__ extra_sq,
                                                               int n_extra) {
  __shared__ float red[1024 / WAVE];
  float s = 0.f;
  for (int i = threadIdx.x; i < nchunks; i += 1024) s += partial[i];
  if (extra_sq != nullptr && (int)threadIdx.x < n_extra) {      // never-materialised (factored) gradients: same units as g, so 1 / scale applies
    const float inv = scaler != nullptr ? 1.f / scaler[0] : 1.f;
    s += extra_sq[threadIdx.x] * inv * inv;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 1024 / WAVE; ++w) tot += red[w];
    const float norm = sqrtf(tot);
    state[1] = norm;
    if (scaler != nullptr) {
      // torch.cuda.amp.GradScaler: unscale_ -> found_inf ? skip the step : step; update(): backoff on a skipped step, growth
      // after growth_interval consecutive good ones
      const float scale = scaler[0];
      const bool bad = !(norm == norm) || norm > 3.0e38f;        // nan / inf
      if (bad) {
        state[2] = 0.f;
        state[3] = 1.f;
        scaler[0] = scale * backoff;
        scaler[1] = 0.f;
        return;
      }
      float tracker = scaler[1] + 1.f;
      if (tracker >= (float)growth_interval) { scaler[0] = scale * growth; tracker = 0.f; }
      scaler[1] = tracker;
      state[0] += 1.f;
      state[3] = 0.f;
      state[2] = (max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f) / scale;
      return;
    }
    state[0] += 1.f;
    state[3] = 0.f;
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    state[2] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
  }
}

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float clip, float lr, float wd, float b1,
                                          float b2, float eps, float step_size, float rsqrt_bc2) {
  g *= clip;
  p -= lr * wd * p;                       // decoupled weight decay (torch: param.mul_(1 - lr * wd))
  m += (1.f - b1) * (g - m);              // lerp(m, g, 1 - b1)
  v = b2 * v + (1.f - b2) * g * g;
  const float denom = sqrtf(v) * rsqrt_bc2 + eps;
  p -= step_size * (m / denom);
}

template <bool G16>
__global__ __launch_bounds__(OPT_THREADS) void opt_adamw_kernel(OptTables t, int chunk_elems, const float* __restrict__ lr_ptr,
                                                                const float* __restrict__ state, float b1, float b2, float eps) {
  const int c = blockIdx.x;
  const int ti = t.chunk_tensor[c];
  const char* g = reinterpret_cast<const char*>(t.grads[ti]);
  if (g == nullptr || state[3] != 0.f) return;      // state[3]: the loss scaler found non-finite gradients -- step skipped
  const csts_opt_tensor tt = t.tensors[ti];
  const int64_t off = t.chunk_off[c];
  const int64_t n = min((int64_t)chunk_elems, tt.n - off);
  const float lr = *lr_ptr, step = state[0], clip = state[2];
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  const float step_size = lr / bc1, rsqrt_bc2 = 1.f / sqrtf(bc2);
  float* p = reinterpret_cast<float*>(tt.p) + off;
  float* m = reinterpret_cast<float*>(tt.m) + off;
  float* v = reinterpret_cast<float*>(tt.v) + off;
  bf16* w16 = tt.w16 ? reinterpret_cast<bf16*>(tt.w16) + off : nullptr;
  g += off * (G16 ? 2 : 4);
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(g) & (G16 ? 7 : 15)) == 0 && (w16 == nullptr || (reinterpret_cast<uintptr_t>(w16) & 7) == 0);
  int64_t done = 0;
  if (vec) {
    const int64_t n4 = n / OPT_VEC;
    const StreamOut sp(p, n * 4), sm(m, n * 4), sv(v, n * 4), sw(w16 ? (void*)w16 : (void*)p, n * 2);     // chunk_elems * 4 < 2^31 (csts_adamw_step)
    auto f4 = [](const float4& x) { f32x4 o; o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w; return o; };
    for (int64_t i = threadIdx.x; i < n4; i += OPT_THREADS) {
      float4 pp = ld4s(p, i);
      const float4 gg = ldg4<G16>(g, i);
      float4 mm = ld4s(m, i), vv = ld4s(v, i);
      adamw_one(pp.x, gg.x, mm.x, vv.x, clip, lr, tt.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
      adamw_one(pp.y, gg.y, mm.y, vv.y, clip, lr, tt.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
      adamw_one(pp.z, gg.z, mm.z, vv.z, clip, lr, tt.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
      adamw_one(pp.w, gg.w, mm.w, vv.w, clip, lr, tt.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
      sp.st16((int)i * 16, f4(pp));
      sm.st16((int)i * 16, f4(mm));
      sv.st16((int)i * 16, f4(vv));
      if (w16) {
        bf16x4 w;
        w[0] = (bf16)pp.x; w[1] = (bf16)pp.y; w[2] = (bf16)pp.z; w[3] = (bf16)pp.w;
        sw.st8((int)i * 8, w);
      }
    }
    done = n4 * OPT_VEC;
  }
  for (int64_t i = done + threadIdx.x; i < n; i += OPT_THREADS) {
    float pp = p[i], mm = m[i], vv = v[i];
    adamw_one(pp, ldg<G16>(g, i), mm, vv, clip, lr, tt.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (w16) w16[i] = (bf16)pp;
  }
}


// ---------------------------------------------------------------- factored AdamW (csts_opt_factored, include/csts_hip.h)
constexpr int FK = 256;       // k-columns per workgroup tile / Gram slab
constexpr int FT = 64;        // max token rows
constexpr int FMAX = 8;       // items per call (the list travels by value in the kernel arguments: graph-capturable, no staging)

struct FactoredList { csts_opt_factored it[FMAX]; int64_t ws_off[FMAX]; };

__device__ __forceinline__ float ld_a(const void* a, int a_dt, int64_t i) {
  return a_dt == CSTS_F32 ? reinterpret_cast<const float*>(a)[i] : (float)reinterpret_cast<const bf16*>(a)[i];
}

// partial Gram slab: ws[item][slab][t * T + t'] = sum_{k in slab} A[t][k] A[t'][k]
__global__ __launch_bounds__(256) void factored_gram_kernel(FactoredList l, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];                     // [T][FK + 1]
  float (*As)[FK + 1] = reinterpret_cast<float (*)[FK + 1]>(fsm);
  const csts_opt_factored& it = l.it[blockIdx.z];
  const int nslab = it.K / FK;
  if ((int)blockIdx.x >= nslab) return;
  const int k0 = blockIdx.x * FK, T = it.T;
  for (int i = threadIdx.x; i < T * (FK / 8); i += 256) {
    const int t = i / (FK / 8), k = (i - t * (FK / 8)) * 8;
    float v[8];
    ld8_as_f32(it.a, it.a_dt, (int64_t)t * it.K + k0 + k, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) As[t][k + j] = v[j];
  }
  __syncthreads();
  float* out = ws + l.ws_off[blockIdx.z] + (int64_t)blockIdx.x * T * T;
  for (int pr = threadIdx.x; pr < T * T; pr += 256) {
    const int t = pr / T, u = pr - t * T;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < FK; ++k) s += As[t][k] * As[u][k];
    out[pr] = s;
  }
}
// out_sq[item] = sum_{t,t'} G1[t,t'] G2[t,t'],  G2 = sum of the slabs, G1 = dY dY^T   (fixed order: reproducible).  Two stages: one
// workgroup per item doing everything was a 430 us serial chain (192 slabs + 2 x 768 dY values per pair and thread); part b of
// FPARTS sums the slabs s = b mod FPARTS and the dY columns of its N / FPARTS range, the finish multiplies the two sums per pair.
constexpr int FPARTS = 32;
__global__ __launch_bounds__(1024) void factored_sq_part_kernel(FactoredList l, const float* __restrict__ ws, float* __restrict__ parts) {
  const csts_opt_factored& it = l.it[blockIdx.y];
  const int T = it.T, nslab = it.K / FK, b = blockIdx.x;
  const float* w = ws + l.ws_off[blockIdx.y];
  float* out = parts + ((int64_t)blockIdx.y * FPARTS + b) * 2 * FT * FT;
  const int nb = (it.N + FPARTS - 1) / FPARTS, n_beg = b * nb, n_end = min(it.N, n_beg + nb);
  for (int pr = threadIdx.x; pr < T * T; pr += 1024) {
    const int t = pr / T, u = pr - t * T;
    float g2 = 0.f;
    for (int s = b; s < nslab; s += FPARTS) g2 += w[(int64_t)s * T * T + pr];
    float g1 = 0.f;
    if (it.a_dt == CSTS_F32) {
      for (int n = n_beg; n < n_end; ++n) g1 += it.dy[(int64_t)t * it.N + n] * it.dy[(int64_t)u * it.N + n];
    } else {             // the MFMA form of the update multiplies dY rounded to the 16-bit type: the norm is that gradient's
      for (int n = n_beg; n < n_end; ++n) g1 += (float)(bf16)it.dy[(int64_t)t * it.N + n] * (float)(bf16)it.dy[(int64_t)u * it.N + n];
    }
    out[pr] = g2;
    out[FT * FT + pr] = g1;
  }
}
__global__ __launch_bounds__(1024) void factored_sq_finish_kernel(FactoredList l, const float* __restrict__ parts, float* __restrict__ out_sq) {
  __shared__ float red[1024 / WAVE];
  const csts_opt_factored& it = l.it[blockIdx.x];
  const int T = it.T;
  const float* w = parts + (int64_t)blockIdx.x * FPARTS * 2 * FT * FT;
  float acc = 0.f;
  for (int pr = threadIdx.x; pr < T * T; pr += 1024) {
    float g2 = 0.f, g1 = 0.f;
    for (int b = 0; b < FPARTS; ++b) { g2 += w[(int64_t)b * 2 * FT * FT + pr]; g1 += w[(int64_t)b * 2 * FT * FT + FT * FT + pr]; }
    acc += g1 * g2;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 1024 / WAVE; ++i) tot += red[i];
    out_sq[blockIdx.x] = (tot == tot) ? fmaxf(tot, 0.f) : __builtin_inff();      // a non-finite factor must reach the loss scaler's check (fmaxf drops NaN)
  }
}
// tile: FR rows n x FK columns k; g[n][k] = sum_t dy[t][n] A[t][k] in fp32, then the AdamW update of csts_adamw_step.  A thread owns
// 4 consecutive columns of FR / 4 rows: one float4 of the operand tile per token row feeds all of them from registers.
constexpr int FR = 32;
__global__ __launch_bounds__(256) void opt_adamw_factored_kernel(FactoredList l, const float* __restrict__ lr_ptr,
                                                                 const float* __restrict__ state, float b1, float b2, float eps) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];                     // [T][FK] operand tile, then [T][FR] dY tile
  if (state[3] != 0.f) return;                                  // step skipped by the loss scaler
  const csts_opt_factored& it = l.it[blockIdx.z];
  float (*As)[FK] = reinterpret_cast<float (*)[FK]>(fsm);
  float (*Ds)[FR] = reinterpret_cast<float (*)[FR]>(fsm + it.T * FK);
  const int k0 = blockIdx.x * FK, n0 = blockIdx.y * FR, T = it.T;
  if (k0 >= it.K || n0 >= it.N) return;
  // operand tile: 8 elements (16 bytes of the 16-bit type, 32 of fp32) per thread and pass
  for (int i = threadIdx.x; i < T * (FK / 8); i += 256) {
    const int t = i / (FK / 8), k = (i - t * (FK / 8)) * 8;
    float v[8];
    ld8_as_f32(it.a, it.a_dt, (int64_t)t * it.K + k0 + k, v);
    *reinterpret_cast<float4*>(&As[t][k]) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(&As[t][k + 4]) = make_float4(v[4], v[5], v[6], v[7]);
  }
  for (int i = threadIdx.x; i < T * FR; i += 256) {
    const int t = i / FR, n = i - t * FR;
    Ds[t][n] = (n0 + n < it.N) ? it.dy[(int64_t)t * it.N + n0 + n] : 0.f;
  }
  __syncthreads();
  const float lr = *lr_ptr, step = state[0], clip = state[2];
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  const float step_size = lr / bc1, rsqrt_bc2 = 1.f / sqrtf(bc2);
  const int kq = (threadIdx.x & 63) * 4;                        // 4 consecutive columns
  const int nr = threadIdx.x >> 6;                              // rows nr, nr + 4, ..., nr + FR - 4
  constexpr int RPT = FR / 4;
  float g[RPT][4];
#pragma unroll
  for (int rr = 0; rr < RPT; ++rr) { g[rr][0] = 0.f; g[rr][1] = 0.f; g[rr][2] = 0.f; g[rr][3] = 0.f; }
  for (int t = 0; t < T; ++t) {
    const float4 av = *reinterpret_cast<const float4*>(&As[t][kq]);
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      const float d = Ds[t][nr + 4 * rr];
      g[rr][0] += d * av.x; g[rr][1] += d * av.y; g[rr][2] += d * av.z; g[rr][3] += d * av.w;
    }
  }
#pragma unroll
  for (int rr = 0; rr < RPT; ++rr) {
    const int n = n0 + nr + 4 * rr;
    if (n >= it.N) continue;
    const int64_t at = (int64_t)n * it.K + k0 + kq;
    float4 pp = *reinterpret_cast<float4*>(it.p + at), mm = *reinterpret_cast<float4*>(it.m + at), vv = *reinterpret_cast<float4*>(it.v + at);
    adamw_one(pp.x, g[rr][0], mm.x, vv.x, clip, lr, it.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
    adamw_one(pp.y, g[rr][1], mm.y, vv.y, clip, lr, it.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
    adamw_one(pp.z, g[rr][2], mm.z, vv.z, clip, lr, it.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
    adamw_one(pp.w, g[rr][3], mm.w, vv.w, clip, lr, it.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
    *reinterpret_cast<float4*>(it.p + at) = pp;
    *reinterpret_cast<float4*>(it.m + at) = mm;
    *reinterpret_cast<float4*>(it.v + at) = vv;
    if (it.w16 != nullptr) {
      bf16x4 w;
      w[0] = (bf16)pp.x; w[1] = (bf16)pp.y; w[2] = (bf16)pp.z; w[3] = (bf16)pp.w;
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(it.w16) + at) = w;
    }
  }
}

// 16-bit operand form: g = dY^T A on the matrix cores (dY rounded to the 16-bit type, fp32 accumulation -- the arithmetic of the TN
// GEMM that materialises dW otherwise), no LDS: a wave owns 32 rows x 64 columns, T <= 64 is <= 4 MFMA k-steps, the operands come
// straight from memory in MFMA layout (dY 98 KB and the k-columns of A: L2-resident), and p / m / v stream through once with the
// optimizer's cache policy.  Accumulator r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31: a half-wave
// touches 128 contiguous bytes of a row per access.
__global__ __launch_bounds__(256) void opt_adamw_factored_mfma_kernel(FactoredList l, const float* __restrict__ lr_ptr,
                                                                      const float* __restrict__ state, float b1, float b2, float eps) {
  if (state[3] != 0.f) return;                                  // step skipped by the loss scaler
  const csts_opt_factored& it = l.it[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * FK + wave * 64;
  if (k0 >= it.K || n0 >= it.N) return;
  const int T = it.T, N = it.N, K = it.K;
  const bf16* __restrict__ a16 = reinterpret_cast<const bf16*>(it.a);
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int nl = n0 + (lane & 31), kl = k0 + (lane & 31);
  for (int t0 = 0; t0 < T; t0 += 16) {
    bf16x8 dv, a0, a1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = t0 + 8 * (lane >> 5) + e;
      const bool ok = t < T;
      dv[e] = (bf16)((ok && nl < N) ? it.dy[(int64_t)t * N + nl] : 0.f);
      a0[e] = ok ? a16[(int64_t)t * K + kl] : (bf16)0.f;
      a1[e] = ok ? a16[(int64_t)t * K + kl + 32] : (bf16)0.f;
    }
    acc[0] = CSTS_MFMA16(dv, a0, acc[0], 0, 0, 0);
    acc[1] = CSTS_MFMA16(dv, a1, acc[1], 0, 0, 0);
  }
  const float lr = *lr_ptr, step = state[0], clip = state[2];
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  const float step_size = lr / bc1, rsqrt_bc2 = 1.f / sqrtf(bc2);
#ifdef CSTS_NO_STREAM_POLICY
  constexpr int LD_AUX = 0, ST_AUX = 0;
#else
  constexpr int LD_AUX = 2, ST_AUX = 17;                        // nontemporal loads, write-through stores (common.h)
#endif
  const int bytes = (int)((int64_t)N * K * 4);
  const auto rp = __builtin_amdgcn_make_buffer_rsrc(it.p, 0, bytes, 0x00020000), rm = __builtin_amdgcn_make_buffer_rsrc(it.m, 0, bytes, 0x00020000),
             rv = __builtin_amdgcn_make_buffer_rsrc(it.v, 0, bytes, 0x00020000),
             rw = __builtin_amdgcn_make_buffer_rsrc(it.w16 != nullptr ? it.w16 : (void*)it.p, 0, bytes / 2, 0x00020000);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float pp[16], mm[16], vv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int off = (n < N) ? (n * K + kl + 32 * j) * 4 : 0;
      pp[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, off, 0, LD_AUX));
      mm[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, off, 0, LD_AUX));
      vv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, off, 0, LD_AUX));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n >= N) continue;
      const int off = (n * K + kl + 32 * j) * 4;
      adamw_one(pp[r], acc[j][r], mm[r], vv[r], clip, lr, it.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pp[r]), rp, off, 0, ST_AUX);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mm[r]), rm, off, 0, ST_AUX);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vv[r]), rv, off, 0, ST_AUX);
      if (it.w16 != nullptr) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16)pp[r]), rw, off / 2, 0, ST_AUX);
    }
  }
}

}  // namespace

constexpr int FT_MFMA = 256;  // token rows the MFMA form of the update takes (no LDS tile, a k-loop over T): the data-parallel chain's W * B * T' rows at 8 ranks
static int factored_fill(const csts_opt_factored* items, int nitems, FactoredList& l, int* max_k, int* max_n, int max_t = FT) {
  CSTS_REQUIRE(items != nullptr && nitems > 0 && nitems <= FMAX, "1 .. 8 items per call");
  *max_k = 0; *max_n = 0;
  int64_t off = 0;
  for (int i = 0; i < nitems; ++i) {
    const csts_opt_factored& it = items[i];
    CSTS_REQUIRE(it.p && it.m && it.v && it.dy && it.a, "null pointer");
    CSTS_REQUIRE(it.T > 0 && it.T <= max_t && it.K > 0 && it.K % FK == 0 && it.N > 0 && it.N % 16 == 0,
                 "T <= 64 (256 for the update with 16-bit operands), K % 256 == 0, N % 16 == 0");
    CSTS_REQUIRE(it.a_dt == CSTS_F32 || it.a_dt == CSTS_HALF, "bad a dtype");
    CSTS_REQUIRE(aligned16(it.p) && aligned16(it.m) && aligned16(it.v) && (it.w16 == nullptr || ((uintptr_t)it.w16 & 7) == 0), "alignment");
    l.it[i] = it;
    l.ws_off[i] = off;
    off += (int64_t)(it.K / FK) * it.T * it.T;
    *max_k = std::max(*max_k, it.K); *max_n = std::max(*max_n, it.N);
  }
  return 0;
}

extern "C" size_t csts_factored_sqnorm_workspace(const csts_opt_factored* items, int nitems) {
  if (items == nullptr || nitems <= 0 || nitems > FMAX) return 0;
  size_t b = 0;
  for (int i = 0; i < nitems; ++i) b += (size_t)(items[i].K / FK) * items[i].T * items[i].T * sizeof(float);
  return b + (size_t)nitems * FPARTS * 2 * FT * FT * sizeof(float);          // Gram slabs + the two-stage finish's partial sums
}
extern "C" int csts_factored_sqnorm(const csts_opt_factored* items, int nitems, float* out_sq, void* workspace, size_t ws_bytes,
                                    hipStream_t stream) {
  FactoredList l;
  int mk, mn;
  if (int rc = factored_fill(items, nitems, l, &mk, &mn)) return rc;
  CSTS_REQUIRE(out_sq != nullptr && workspace != nullptr && aligned16(workspace) && ws_bytes >= csts_factored_sqnorm_workspace(items, nitems), "workspace too small");
  float* slabs = reinterpret_cast<float*>(workspace);
  int mt = 0;
  for (int i = 0; i < nitems; ++i) mt = std::max(mt, items[i].T);
  const size_t sm = (size_t)mt * (FK + 1) * sizeof(float);
  CSTS_REQUIRE(sm <= 64 * 1024 || csts_dyn_lds_optin(reinterpret_cast<const void*>(&factored_gram_kernel), (int)sm), "LDS opt-in failed");
  hipLaunchKernelGGL(factored_gram_kernel, dim3((unsigned)(mk / FK), 1, (unsigned)nitems), dim3(256), sm, stream, l, slabs);
  CSTS_LAUNCH_CHECK();
  size_t slab_floats = 0;
  for (int i = 0; i < nitems; ++i) slab_floats += (size_t)(items[i].K / FK) * items[i].T * items[i].T;
  float* parts = slabs + slab_floats;
  hipLaunchKernelGGL(factored_sq_part_kernel, dim3(FPARTS, (unsigned)nitems), dim3(1024), 0, stream, l, (const float*)slabs, parts);
  CSTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(factored_sq_finish_kernel, dim3((unsigned)nitems), dim3(1024), 0, stream, l, (const float*)parts, out_sq);
  CSTS_LAUNCH_CHECK();
  return 0;
}
extern "C" int csts_adamw_factored(const csts_opt_factored* items, int nitems, const float* state, const float* lr, float beta1,
                                   float beta2, float eps, hipStream_t stream) {
  FactoredList l;
  int mk, mn;
  bool half_ops = items != nullptr && nitems > 0 && nitems <= FMAX;
  for (int i = 0; half_ops && i < nitems; ++i) half_ops = items[i].a_dt == CSTS_HALF && (int64_t)items[i].N * items[i].K * 4 < ((int64_t)1 << 31) && items[i].K % 64 == 0;
  if (int rc = factored_fill(items, nitems, l, &mk, &mn, half_ops ? FT_MFMA : FT)) return rc;
  CSTS_REQUIRE(state != nullptr && lr != nullptr, "null state");
  int mt = 0;
  for (int i = 0; i < nitems; ++i) mt = std::max(mt, items[i].T);
  bool all16 = true;
  for (int i = 0; i < nitems; ++i) all16 = all16 && items[i].a_dt == CSTS_HALF && (int64_t)items[i].N * items[i].K * 4 < ((int64_t)1 << 31) && items[i].K % 64 == 0;
  if (all16) {
    hipLaunchKernelGGL(opt_adamw_factored_mfma_kernel, dim3((unsigned)(mk / FK), (unsigned)cdiv(mn, 32), (unsigned)nitems), dim3(256), 0, stream, l, lr,
                       state, beta1, beta2, eps);
    CSTS_LAUNCH_CHECK();
    return 0;
  }
  const size_t sm = (size_t)mt * (FK + FR) * sizeof(float);
  CSTS_REQUIRE(sm <= 64 * 1024 || csts_dyn_lds_optin(reinterpret_cast<const void*>(&opt_adamw_factored_kernel), (int)sm), "LDS opt-in failed");
  hipLaunchKernelGGL(opt_adamw_factored_kernel, dim3((unsigned)(mk / FK), (unsigned)cdiv(mn, FR), (unsigned)nitems), dim3(256), sm, stream, l, lr,
                     state, beta1, beta2, eps);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_adamw_step(const csts_opt_args* a, hipStream_t stream) {
  CSTS_REQUIRE(a != nullptr, "null args");
  CSTS_REQUIRE(a->nchunks > 0 && a->ntensors > 0 && a->chunk_elems > 0 && a->chunk_elems % 4 == 0 && a->chunk_elems <= (1 << 28), "bad chunking");
  CSTS_REQUIRE(a->chunk_tensor && a->chunk_off && a->tensors && a->grads, "null table");
  CSTS_REQUIRE(a->partial && a->state && a->lr, "null state");
  CSTS_REQUIRE(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps > 0.f, "bad hyper-parameters");
  CSTS_REQUIRE(a->grad_dt == CSTS_F32 || a->grad_dt == CSTS_HALF, "bad gradient dtype");
  CSTS_REQUIRE(a->n_extra_sq >= 0 && a->n_extra_sq <= 1024 && (a->n_extra_sq == 0 || a->extra_sq != nullptr), "bad extra_sq");
  CSTS_REQUIRE(a->scaler == nullptr || (a->growth > 1.f && a->backoff > 0.f && a->backoff < 1.f && a->growth_interval > 0), "bad loss-scaler parameters");
  OptTables t{a->chunk_tensor, a->chunk_off, a->tensors, a->grads};
  const bool g16 = a->grad_dt != CSTS_F32;
  const dim3 grid((unsigned)a->nchunks), block(OPT_THREADS);
  if (g16) hipLaunchKernelGGL(opt_sqnorm_kernel<true>, grid, block, 0, stream, t, a->chunk_elems, a->partial, (const float*)a->scaler);
  else hipLaunchKernelGGL(opt_sqnorm_kernel<false>, grid, block, 0, stream, t, a->chunk_elems, a->partial, (const float*)a->scaler);
  CSTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(opt_norm_finish_kernel, dim3(1), dim3(1024), 0, stream, a->partial, a->nchunks, a->max_grad_norm, a->state,
                     a->scaler, a->growth, a->backoff, a->growth_interval, a->extra_sq, a->n_extra_sq);
  CSTS_LAUNCH_CHECK();
  if (g16) hipLaunchKernelGGL(opt_adamw_kernel<true>, grid, block, 0, stream, t, a->chunk_elems, a->lr, a->state, a->beta1, a->beta2, a->eps);
  else hipLaunchKernelGGL(opt_adamw_kernel<false>, grid, block, 0, stream, t, a->chunk_elems, a->lr, a->state, a->beta1, a->beta2, a->eps);
  CSTS_LAUNCH_CHECK();
  return 0;
}
