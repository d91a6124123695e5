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
    const bf16x4 v = reinterpret_cast<const bf16x4*>(g)[i4];
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  } else return reinterpret_cast<const float4*>(g)[i4];
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
                                                               float backoff, int growth_interval) {
  __shared__ float red[1024 / WAVE];
  float s = 0.f;
  for (int i = threadIdx.x; i < nchunks; i += 1024) s += partial[i];
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
    for (int64_t i = threadIdx.x; i < n4; i += OPT_THREADS) {
      float4 pp = reinterpret_cast<float4*>(p)[i];
      const float4 gg = ldg4<G16>(g, i);
      float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
      adamw_one(pp.x, gg.x, mm.x, vv.x, clip, lr, tt.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
      adamw_one(pp.y, gg.y, mm.y, vv.y, clip, lr, tt.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
      adamw_one(pp.z, gg.z, mm.z, vv.z, clip, lr, tt.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
      adamw_one(pp.w, gg.w, mm.w, vv.w, clip, lr, tt.weight_decay, b1, b2, eps, step_size, rsqrt_bc2);
      reinterpret_cast<float4*>(p)[i] = pp;
      reinterpret_cast<float4*>(m)[i] = mm;
      reinterpret_cast<float4*>(v)[i] = vv;
      if (w16) {
        bf16x4 w;
        w[0] = (bf16)pp.x; w[1] = (bf16)pp.y; w[2] = (bf16)pp.z; w[3] = (bf16)pp.w;
        reinterpret_cast<bf16x4*>(w16)[i] = w;
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

}  // namespace

extern "C" int csts_adamw_step(const csts_opt_args* a, hipStream_t stream) {
  CSTS_REQUIRE(a != nullptr, "null args");
  CSTS_REQUIRE(a->nchunks > 0 && a->ntensors > 0 && a->chunk_elems > 0 && a->chunk_elems % 4 == 0, "bad chunking");
  CSTS_REQUIRE(a->chunk_tensor && a->chunk_off && a->tensors && a->grads, "null table");
  CSTS_REQUIRE(a->partial && a->state && a->lr, "null state");
  CSTS_REQUIRE(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps > 0.f, "bad hyper-parameters");
  CSTS_REQUIRE(a->grad_dt == CSTS_F32 || a->grad_dt == CSTS_HALF, "bad gradient dtype");
  CSTS_REQUIRE(a->scaler == nullptr || (a->growth > 1.f && a->backoff > 0.f && a->backoff < 1.f && a->growth_interval > 0), "bad loss-scaler parameters");
  OptTables t{a->chunk_tensor, a->chunk_off, a->tensors, a->grads};
  const bool g16 = a->grad_dt != CSTS_F32;
  const dim3 grid((unsigned)a->nchunks), block(OPT_THREADS);
  if (g16) hipLaunchKernelGGL(opt_sqnorm_kernel<true>, grid, block, 0, stream, t, a->chunk_elems, a->partial, (const float*)a->scaler);
  else hipLaunchKernelGGL(opt_sqnorm_kernel<false>, grid, block, 0, stream, t, a->chunk_elems, a->partial, (const float*)a->scaler);
  CSTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(opt_norm_finish_kernel, dim3(1), dim3(1024), 0, stream, a->partial, a->nchunks, a->max_grad_norm, a->state,
                     a->scaler, a->growth, a->backoff, a->growth_interval);
  CSTS_LAUNCH_CHECK();
  if (g16) hipLaunchKernelGGL(opt_adamw_kernel<true>, grid, block, 0, stream, t, a->chunk_elems, a->lr, a->state, a->beta1, a->beta2, a->eps);
  else hipLaunchKernelGGL(opt_adamw_kernel<false>, grid, block, 0, stream, t, a->chunk_elems, a->lr, a->state, a->beta1, a->beta2, a->eps);
  CSTS_LAUNCH_CHECK();
  return 0;
}
