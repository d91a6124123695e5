// On-device input pipeline of the CSTS path: what the reference does on the CPU (offline or in
// DataLoader workers) immediately before the model is called.
//   frames_normalize : uint8 THWC frames -> (x/255 - mean)/std, C T H W        (slowfast/datasets/utils.py:290-307,
//                                                                               ego4d_avgaze_forecast.py:294-296)
//   stft_logpower    : waveform -> log(|STFT|^2 + eps), librosa.stft(n_fft 511, hop 120, win 240 Hann, center, zero pad)
//                                                                              (data/preprocess.py:276-290)
//   audio_windows    : T windows of 256 spectrogram columns around the sampled frames (ego4d_avgaze_forecast.py:214-219)
//   gaze_heatmaps    : 19x19 OpenCV-Gaussian gaze maps, clipped and renormalised      (ego4d_avgaze_forecast.py:318-326,404-422)
// All HBM/latency-bound byte shuffling except the STFT, a 240-tap direct DFT per (frame, bin) from a twiddle table in LDS.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void frames_normalize_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                               int64_t thw, int C, float3 mean, float3 inv_std) {
  // in: [b][t*h*w][C] uint8 ; out: [b][C][t*h*w] fp32
  const int64_t b = blockIdx.y;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < thw; p += (int64_t)gridDim.x * 256) {
    const uint8_t* px = in + (b * thw + p) * C;
    const float m[3] = {mean.x, mean.y, mean.z}, s[3] = {inv_std.x, inv_std.y, inv_std.z};
    for (int c = 0; c < C; ++c) out[(b * C + c) * thw + p] = ((float)px[c] / 255.0f - m[c < 3 ? c : 2]) * s[c < 3 ? c : 2];
  }
}

// one workgroup per (frame, batch): the window's samples (times the periodic Hann window) in LDS, one thread per bin
__global__ __launch_bounds__(256) void stft_logpower_kernel(const float* __restrict__ wav, float* __restrict__ spec, int n,
                                                            int n_fft, int hop, int win, int nframes, float eps) {
  extern __shared__ float sm[];          // [win] windowed samples, [n_fft] cos, [n_fft] sin
  float* xs = sm;
  float* ct = sm + win;
  float* st = ct + n_fft;
  const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int lpad = (n_fft - win) / 2;     // librosa.util.pad_center of the window inside n_fft
  const int64_t s0 = (int64_t)f * hop + lpad - n_fft / 2;      // first waveform sample under the window (center=True)
  for (int k = tid; k < win; k += 256) {
    const int64_t s = s0 + k;
    const float w = 0.5f - 0.5f * cospif(2.0f * (float)k / (float)win);     // scipy hann, fftbins=True
    xs[k] = (s >= 0 && s < n) ? wav[(int64_t)b * n + s] * w : 0.f;
  }
  for (int m = tid; m < n_fft; m += 256) {
    float sv, cv;
    sincospif(2.0f * (float)m / (float)n_fft, &sv, &cv);
    ct[m] = cv; st[m] = sv;
  }
  __syncthreads();
  const int nbins = n_fft / 2 + 1;
  for (int bin = tid; bin < nbins; bin += 256) {
    float re = 0.f, im = 0.f;
    int idx = (int)(((int64_t)bin * lpad) % n_fft);            // phase index (bin * j) mod n_fft, j = lpad + k
    for (int k = 0; k < win; ++k) {
      re += xs[k] * ct[idx];
      im -= xs[k] * st[idx];
      idx += bin;
      if (idx >= n_fft) idx -= n_fft;
    }
    spec[((int64_t)b * nbins + bin) * nframes + f] = logf(re * re + im * im + eps);
  }
}

// out[b][0][t][bin][j] = spec[b][bin][center[b][t] - 128 + j]
__global__ __launch_bounds__(256) void audio_windows_kernel(const float* __restrict__ spec, const int* __restrict__ centers,
                                                            float* __restrict__ out, int T, int nbins, int cols, int width) {
  const int bt = blockIdx.y, b = bt / T;
  const int c0 = centers[bt] - width / 2;
  const int64_t total = (int64_t)nbins * width;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int bin = (int)(i / width), j = (int)(i - (int64_t)bin * width);
    out[(int64_t)bt * total + i] = spec[((int64_t)b * nbins + bin) * cols + c0 + j];
  }
}

// Python round(): half to even
__device__ __forceinline__ int round_half_even(float v) { return (int)rintf(v); }

__global__ __launch_bounds__(256) void gaze_heatmap_kernel(const float* __restrict__ labels, int label_stride,
                                                           float* __restrict__ hm, int H, int W, int ksize, float sigma) {
  __shared__ float k1[64];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t fr = blockIdx.x;
  const int r = (ksize - 1) / 2;
  if (tid < ksize) {
    const float d = (float)tid - (float)(ksize - 1) * 0.5f;
    k1[tid] = expf(-(d * d) / (2.f * sigma * sigma));
  }
  __syncthreads();
  float ks = 0.f;
  for (int i = 0; i < ksize; ++i) ks += k1[i];
  const int mu_x = round_half_even(labels[fr * label_stride] * (float)W);
  const int mu_y = round_half_even(labels[fr * label_stride + 1] * (float)H);
  const int left = max(mu_x - r, 0), right = min(mu_x + r, W - 1), top = max(mu_y - r, 0), bottom = min(mu_y + r, H - 1);
  const bool placed = !(left >= right || top >= bottom);
  float part = 0.f;
  float vals[16];
  const int per = (H * W + 255) / 256;           // <= 16 for 64x64
  for (int q = 0; q < per; ++q) {
    const int i = tid + 256 * q;
    float v = 0.f;
    if (i < H * W) {
      const int y = i / W, x = i - y * W;
      if (placed && x >= left && x <= right && y >= top && y <= bottom)
        v = (k1[r - mu_y + y] / ks) * (k1[r - mu_x + x] / ks);
    }
    vals[q] = v;
    part += v;
  }
  part = wave_sum(part);
  if (lane == 0) red[w] = part;
  __syncthreads();
  const float s = red[0] + red[1] + red[2] + red[3];
  for (int q = 0; q < per; ++q) {
    const int i = tid + 256 * q;
    if (i < H * W) hm[fr * H * W + i] = (s == 0.f) ? 1.f / (float)(H * W) : vals[q] / s;
  }
}

}  // namespace

extern "C" int csts_frames_normalize(const uint8_t* frames_thwc, float* out_cthw, int B, int64_t thw, int C, const float mean[3],
                                     const float std[3], hipStream_t stream) {
  CSTS_REQUIRE(frames_thwc && out_cthw && mean && std && B > 0 && thw > 0 && C >= 1 && C <= 3, "bad args");
  const float3 m = make_float3(mean[0], mean[1], mean[2]), is = make_float3(1.f / std[0], 1.f / std[1], 1.f / std[2]);
  hipLaunchKernelGGL(frames_normalize_kernel, dim3((unsigned)std::min<int64_t>(cdiv(thw, 256), 4096), B), dim3(256), 0, stream,
                     frames_thwc, out_cthw, thw, C, m, is);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_stft_frames(int n, int n_fft, int hop) { return 1 + (n + 2 * (n_fft / 2) - n_fft) / hop; }

extern "C" int csts_stft_logpower(const float* wav, float* spec, int B, int n, int n_fft, int hop, int win, float eps,
                                  hipStream_t stream) {
  CSTS_REQUIRE(wav && spec && B > 0 && n > 0 && n_fft > 0 && hop > 0 && win > 0 && win <= n_fft && n_fft <= 2048, "bad args");
  const int nframes = csts_stft_frames(n, n_fft, hop);
  CSTS_REQUIRE(nframes > 0, "signal shorter than one frame");
  const size_t sm = (size_t)(win + 2 * n_fft) * sizeof(float);
  hipLaunchKernelGGL(stft_logpower_kernel, dim3(nframes, B), dim3(256), sm, stream, wav, spec, n, n_fft, hop, win, nframes, eps);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_audio_windows(const float* spec, const int* centers, float* out, int B, int T, int nbins, int cols, int width,
                                  hipStream_t stream) {
  CSTS_REQUIRE(spec && centers && out && B > 0 && T > 0 && nbins > 0 && cols >= width && width > 0, "bad args");
  hipLaunchKernelGGL(audio_windows_kernel, dim3((unsigned)std::min<int64_t>(cdiv((int64_t)nbins * width, 256), 1024), B * T), dim3(256),
                     0, stream, spec, centers, out, T, nbins, cols, width);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" int csts_gaze_heatmaps(const float* labels, int label_stride, float* heatmaps, int64_t nframes, int H, int W, int ksize,
                                  hipStream_t stream) {
  CSTS_REQUIRE(labels && heatmaps && nframes > 0 && label_stride >= 2, "bad args");
  CSTS_REQUIRE(H * W <= 4096 && ksize >= 3 && ksize <= 63 && (ksize & 1), "map <= 64x64, odd kernel <= 63");
  const float sigma = 0.3f * ((float)(ksize - 1) * 0.5f - 1.f) + 0.8f;      // cv2.getGaussianKernel(sigma <= 0)
  hipLaunchKernelGGL(gaze_heatmap_kernel, dim3((unsigned)nframes), dim3(256), 0, stream, labels, label_stride, heatmaps, H, W, ksize,
                     sigma);
  CSTS_LAUNCH_CHECK();
  return 0;
}
