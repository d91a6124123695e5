// Evaluation metric of the gaze heat maps on the device: the min-max rescale of
// tools/test_avgaze_net.py:66-68 / tools/train_avgaze_net.py:125-127 and slowfast/utils/metrics.py:9-74 adaptive_f1
// (best F1 over a dataset-specific threshold sweep, averaged over the fixation frames only).
// The reference materialises two (n_thresholds, B, T, H, W) float tensors (metrics.py:45-51, "It consumes much memory");
// here one workgroup per frame keeps the frame in registers, counts all thresholds at once and writes 2*NT+1 integers.
#include "common.h"

namespace {

constexpr int F1_MAX_THR = 64;

// counts[frame][0..NT) = tp, [NT..2NT) = fg_preds, [2NT] = fg_labels
__global__ __launch_bounds__(256) void f1_count_kernel(const float* __restrict__ preds, const float* __restrict__ labels_hm,
                                                       const float* __restrict__ thr, int nthr, int hw, int rescale,
                                                       int* __restrict__ counts) {
  __shared__ float redf[2][4];
  __shared__ int redi[2 * F1_MAX_THR + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* p = preds + (int64_t)blockIdx.x * hw;
  const float* q = labels_hm + (int64_t)blockIdx.x * hw;
  for (int i = tid; i < 2 * nthr + 1; i += 256) redi[i] = 0;
  float mn = INFINITY, mx = -INFINITY;
  if (rescale) {
    for (int i = tid; i < hw; i += 256) { const float v = p[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    mn = -wave_max(-mn);
    mx = wave_max(mx);
    if (lane == 0) { redf[0][w] = mn; redf[1][w] = mx; }
    __syncthreads();
    mn = fminf(fminf(redf[0][0], redf[0][1]), fminf(redf[0][2], redf[0][3]));
    mx = fmaxf(fmaxf(redf[1][0], redf[1][1]), fmaxf(redf[1][2], redf[1][3]));
  } else {
    __syncthreads();
  }
  const float denom = mx - mn + 1e-6f;           // (p - min) / (max - min + 1e-6), test_avgaze_net.py:67
  int fgl = 0;
  for (int t0 = 0; t0 < nthr; t0 += 8) {         // 8 thresholds per sweep over the frame (registers), frame re-read from L1/L2
    int tp[8], fp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { tp[j] = 0; fp[j] = 0; }
    float th[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) th[j] = thr[min(t0 + j, nthr - 1)];
    for (int i = tid; i < hw; i += 256) {
      const float v = rescale ? (p[i] - mn) / denom : p[i];
      const bool lab = q[i] > 0.001f;             // metrics.py:47
      if (t0 == 0) fgl += lab ? 1 : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool pr = v > th[j];
        fp[j] += pr ? 1 : 0;
        tp[j] += (pr && lab) ? 1 : 0;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (t0 + j < nthr) {
        int a = tp[j], b = fp[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (lane == 0) { atomicAdd(&redi[t0 + j], a); atomicAdd(&redi[nthr + t0 + j], b); }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) fgl += __shfl_xor(fgl, o, 64);
  if (lane == 0) atomicAdd(&redi[2 * nthr], fgl);
  __syncthreads();
  for (int i = tid; i < 2 * nthr + 1; i += 256) counts[(int64_t)blockIdx.x * (2 * nthr + 1) + i] = redi[i];
}

// out = {f1, recall, precision, threshold index (as float)} of the best threshold; frames with tracked[frame] == 0 are skipped
__global__ __launch_bounds__(64) void f1_finish_kernel(const int* __restrict__ counts, const uint8_t* __restrict__ tracked,
                                                       int nframes, int nthr, float* __restrict__ out) {
  __shared__ float f1s[F1_MAX_THR], recs[F1_MAX_THR], pres[F1_MAX_THR];
  const int t = threadIdx.x;
  if (t < nthr) {
    float rs = 0.f, ps = 0.f;
    int n = 0;
    for (int f = 0; f < nframes; ++f) {
      if (!tracked[f]) continue;
      const int* c = counts + (int64_t)f * (2 * nthr + 1);
      const float tp = (float)c[t], fgp = (float)c[nthr + t], fgl = (float)c[2 * nthr];
      rs += tp / (fgl + 1e-6f);
      ps += tp / (fgp + 1e-6f);
      ++n;
    }
    const float r = n ? rs / n : NAN, pcs = n ? ps / n : NAN;     // torch.mean of an empty selection is nan
    recs[t] = r; pres[t] = pcs;
    f1s[t] = (2.f * r * pcs) / (r + pcs + 1e-6f);
  }
  __syncthreads();
  if (t == 0) {
    int best = 0;
    for (int i = 1; i < nthr; ++i)
      if (f1s[i] > f1s[best]) best = i;                           // torch.argmax: first maximum
    out[0] = f1s[best]; out[1] = recs[best]; out[2] = pres[best]; out[3] = (float)best;
  }
}

}  // namespace

extern "C" size_t csts_adaptive_f1_workspace(int64_t nframes, int nthr) { return (size_t)nframes * (2 * nthr + 1) * sizeof(int); }

extern "C" int csts_adaptive_f1(const float* preds, const float* labels_hm, const uint8_t* tracked, const float* thresholds,
                                int nthr, int64_t nframes, int hw, int rescale, float* out, void* workspace, size_t ws_bytes,
                                hipStream_t stream) {
  CSTS_REQUIRE(preds && labels_hm && tracked && thresholds && out && workspace, "null pointer");
  CSTS_REQUIRE(nthr > 0 && nthr <= F1_MAX_THR, "1..64 thresholds");
  CSTS_REQUIRE(nframes > 0 && nframes < ((int64_t)1 << 31) && hw > 0, "bad frame count / size");
  CSTS_REQUIRE(ws_bytes >= csts_adaptive_f1_workspace(nframes, nthr), "workspace too small");
  int* counts = reinterpret_cast<int*>(workspace);
  hipLaunchKernelGGL(f1_count_kernel, dim3((unsigned)nframes), dim3(256), 0, stream, preds, labels_hm, thresholds, nthr, hw, rescale, counts);
  CSTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(f1_finish_kernel, dim3(1), dim3(64), 0, stream, counts, tracked, (int)nframes, nthr, out);
  CSTS_LAUNCH_CHECK();
  return 0;
}
