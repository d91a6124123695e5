// Shared device/host helpers for the CSTS gfx950 kernel library.
// Everything here is CDNA4-only (wave64, MFMA, 160 KiB LDS); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>

#include "../../include/csts_hip.h"

// The 16-bit activation type of this BUILD.  libcsts_hip.so: bfloat16 (the throughput mode, CSTS_AMD.COMPUTE bf16);
// libcsts_hip_f16.so (-DCSTS_HALF_F16, `make f16`): IEEE half -- the arithmetic of the reference's own mixed precision,
// torch.cuda.amp.autocast = fp16 + GradScaler (tools/train_avgaze_net.py:70,99-109,277).  Same kernels, same MFMA rate
// (v_mfma_f32_32x32x16_f16), same bytes; the enum value CSTS_BF16 then means "the 16-bit type of the build" (CSTS_HALF) and
// csts_half_kind() tells a caller which one a library holds.  The type keeps its short name in the sources.
#ifdef CSTS_HALF_F16
typedef _Float16 bf16;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bf16x2 __attribute__((ext_vector_type(2)));
#define CSTS_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define CSTS_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// Streaming cache policy for tensors that are far larger than the caches and touched once per kernel (tools/launch_floor.hip,
// profiles/r4_stream_policy.txt: c = f(a, b) over 3 x 200 MB: plain loads + stores 5.5 TB/s, nontemporal loads 6.7 TB/s; below
// ~50 MB per tensor the plain forms win -- the producer's lines are still in the memory-side cache).  Used by the optimizer
// (6.4 GB per step, nothing re-read): clip + AdamW 1.20 -> 1.03 ms.
// -DCSTS_NO_STREAM_POLICY gives the plain forms back (same-box A/B builds).
__device__ __forceinline__ f32x4 ld_stream(const f32x4* p) {
#ifdef CSTS_NO_STREAM_POLICY
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}
__device__ __forceinline__ bf16x4 ld_stream(const bf16x4* p) {
#ifdef CSTS_NO_STREAM_POLICY
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}
__device__ __forceinline__ bf16x8 ld_stream(const bf16x8* p) {
#ifdef CSTS_NO_STREAM_POLICY
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}
// Write-through ("sc0 sc1") stores through a buffer descriptor (the compiler schedules and pads them itself; an inline-asm
// global_store_dwordx4 needs a manual wait state after it -- without one the data registers were overwritten too early: NaNs in a
// GEMM epilogue that tried it).  Only for tensors no kernel reads soon: a consumer that would have found the lines in the
// memory-side cache gets slower (profiles/r4_stream_policy.txt).  base: wave-uniform; off: bytes, < 2^31.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
struct StreamOut {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ StreamOut(void* base, int64_t bytes) : rsrc(__builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000)) {}
  __device__ __forceinline__ void st16(int off, f32x4 v) const {
#ifdef CSTS_NO_STREAM_POLICY
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, off, 0, 0);
#else
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, off, 0, 17);
#endif
  }
  __device__ __forceinline__ void st8(int off, bf16x4 v) const {
#ifdef CSTS_NO_STREAM_POLICY
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), rsrc, off, 0, 0);
#else
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), rsrc, off, 0, 17);
#endif
  }
};

// 16-bit element(s) held as raw bits -> float: element 0 / 1 of a 32-bit pair, or one 16-bit value
__device__ __forceinline__ float h16_lo(unsigned v) {
#ifdef CSTS_HALF_F16
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu));
#else
  return __uint_as_float(v << 16);
#endif
}
__device__ __forceinline__ float h16_hi(unsigned v) {
#ifdef CSTS_HALF_F16
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16));
#else
  return __uint_as_float(v & 0xffff0000u);
#endif
}

// ---------------------------------------------------------------- errors
void csts_set_error(const std::string& s);
#define CSTS_FAIL(msg)                                                         \
  do {                                                                         \
    csts_set_error(std::string(__func__) + ": " + (msg));                      \
    return -1;                                                                 \
  } while (0)
#define CSTS_REQUIRE(cond, msg)                                                \
  do {                                                                         \
    if (!(cond)) CSTS_FAIL(std::string(msg) + " [" #cond "]");                 \
  } while (0)
#define CSTS_LAUNCH_CHECK()                                                    \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) CSTS_FAIL(std::string("launch: ") + hipGetErrorString(e__)); \
  } while (0)

// Opt a kernel in to more than 64 KB of dynamic LDS.  The attribute is per (function, device): set once for each pair a
// process launches on and RAISED when a later launch needs more than the size opted in before (a kernel whose dynamic LDS
// depends on an argument: opt_adamw_factored_kernel); a failure is reported to the caller instead of surfacing as an opaque
// launch error.
#include <map>
#include <mutex>
#include <utility>
static inline bool csts_dyn_lds_optin(const void* fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;     // (function, device) -> bytes opted in
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({fn, dev});
  if (it != done.end() && it->second >= bytes) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  done[{fn, dev}] = bytes;
  return true;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- dtype-generic scalar access
// dt is wave-uniform at every call site, so the branch is a scalar branch.
__device__ __forceinline__ float ld_as_f32(const void* p, int dt, int64_t i) {
  return dt == CSTS_F32 ? reinterpret_cast<const float*>(p)[i] : (float)reinterpret_cast<const bf16*>(p)[i];
}
__device__ __forceinline__ void st_from_f32(void* p, int dt, int64_t i, float v) {
  if (dt == CSTS_F32) reinterpret_cast<float*>(p)[i] = v;
  else reinterpret_cast<bf16*>(p)[i] = (bf16)v;
}

// 8 consecutive elements -> 8 floats (16-byte vector loads when aligned)
__device__ __forceinline__ void ld8_as_f32(const void* p, int dt, int64_t i, float (&o)[8]) {
  if (dt == CSTS_F32) {
    const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
    float4 a = q[0], b = q[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  } else {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p) + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)v[j];
  }
}
__device__ __forceinline__ void st8_from_f32(void* p, int dt, int64_t i, const float (&o)[8]) {
  if (dt == CSTS_F32) {
    float4* q = reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i);
    q[0] = make_float4(o[0], o[1], o[2], o[3]);
    q[1] = make_float4(o[4], o[5], o[6], o[7]);
  } else {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16)o[j];
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p) + i) = v;
  }
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU and its derivative (reference: nn.GELU(approximate='none'), common.py:21)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// bf16 mode: erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7: far below the bf16 rounding of the GEMM output it is
// applied to) -- one rcp + one exp instead of libm erff (~3x fewer VALU instructions in the GEMM epilogues; the exp is
// shared with the Gaussian pdf in the derivative).  fp32 (parity) mode keeps the exact functions above.
__device__ __forceinline__ float erf_poly_times_exp(float ax, float e) {
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return 1.f - poly * e;      // erf(|x|)
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float t = x * 0.70710678118654752f, ax = fabsf(t);
  const float e = __expf(-ax * ax);
  const float er = copysignf(erf_poly_times_exp(ax, e), x);
  return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float dgelu_fast(float x) {
  const float t = x * 0.70710678118654752f, ax = fabsf(t);
  const float e = __expf(-ax * ax);                       // = exp(-x^2 / 2)
  const float er = copysignf(erf_poly_times_exp(ax, e), x);
  return 0.5f * (1.0f + er) + x * 0.39894228040143268f * e;
}

// second stage of every cross-block reduction (norm.hip): out[j] = scale * sum_i ws[i][j]
int csts_reduce_rows_launch(const float* ws, float* out, int64_t nrows, int64_t ncols, float scale, hipStream_t s);
