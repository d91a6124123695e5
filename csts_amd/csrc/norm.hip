// LayerNorm forward / backward (K2 of SURVEY.md 2.3) and the shared partial-sum reducer.
// Reference: block norms nn.LayerNorm(C, eps=1e-6) (attention.py:192,214) and the per-head
// nn.LayerNorm(hd, eps=1e-5) applied to pooled q/k/v (attention.py:108,112,116) -- same kernel, rows = B*N*h.
// One wave per row, row kept in registers (C <= 768 -> <= 12 values per lane), two-pass statistics in fp32.
// HBM-bound: algorithmic bytes = rows*C*(in + out) (+ 8 B/row of statistics).
#include "common.h"

namespace {

constexpr int MAXV = 12;  // C <= 768

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, int x_dt, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y, int y_dt,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                     int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t row = wave; row < rows; row += nwaves) {
    float v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      v[j] = c < C ? ld_as_f32(x, x_dt, row * C + c) : 0.f;
      s += v[j];
    }
    const float mu = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      const float d = c < C ? v[j] - mu : 0.f;
      q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      if (c < C) st_from_f32(y, y_dt, row * C + c, (v[j] - mu) * rs * gamma[c] + beta[c]);
    }
    if (lane == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// per-wave partial dgamma/dbeta are summed through LDS and written to ws[block][2*C]
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, int dy_dt, const void* __restrict__ x,
                                                     int x_dt, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     void* __restrict__ dx, int dx_dt, float* __restrict__ ws,
                                                     int64_t rows, int C) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][2*C]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + w;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float dg[NV], db[NV], gm[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    dg[j] = 0.f; db[j] = 0.f;
    const int c = lane + 64 * j;
    gm[j] = c < C ? gamma[c] : 0.f;
  }
  for (int64_t row = wave; row < rows; row += nwaves) {
    const float mu = mean[row], rs = rstd[row];
    float xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        const float d = ld_as_f32(dy, dy_dt, row * C + c);
        xh[j] = (ld_as_f32(x, x_dt, row * C + c) - mu) * rs;
        g[j] = d * gm[j];
        dg[j] += d * xh[j];
        db[j] += d;
      } else { xh[j] = 0.f; g[j] = 0.f; }
      s1 += g[j];
      s2 += g[j] * xh[j];
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + 64 * j;
      if (c < C) st_from_f32(dx, dx_dt, row * C + c, rs * (g[j] - s1 - xh[j] * s2));
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = lane + 64 * j;
    if (c < C) { red[w * 2 * C + c] = dg[j]; red[w * 2 * C + C + c] = db[j]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x)
    ws[(int64_t)blockIdx.x * 2 * C + i] = red[i] + red[2 * C + i] + red[4 * C + i] + red[6 * C + i];
}

// out[j] = sum_i ws[i][j]   (deterministic second stage of every cross-block reduction in the library)
// block = 32 columns x RL row-lanes; fixed summation order -> bitwise reproducible
template <int RL>
__global__ void reduce_rows_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t nrows, int64_t ncols,
                                   float scale) {
  __shared__ float red[RL][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t j = (int64_t)blockIdx.x * 32 + tx;
  float s = 0.f;
  if (j < ncols)
    for (int64_t i = ty; i < nrows; i += RL) s += ws[i * ncols + j];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < ncols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < RL; ++r) t += red[r][tx];
    out[j] = t * scale;
  }
}

}  // namespace

int csts_reduce_rows_launch(const float* ws, float* out, int64_t nrows, int64_t ncols, float scale, hipStream_t s) {
  const dim3 grid((unsigned)cdiv(ncols, 32));
  if (nrows > 64) hipLaunchKernelGGL(reduce_rows_kernel<32>, grid, dim3(1024), 0, s, ws, out, nrows, ncols, scale);
  else if (nrows > 8) hipLaunchKernelGGL(reduce_rows_kernel<8>, grid, dim3(256), 0, s, ws, out, nrows, ncols, scale);
  else hipLaunchKernelGGL(reduce_rows_kernel<2>, grid, dim3(64), 0, s, ws, out, nrows, ncols, scale);
  return 0;
}

extern "C" int csts_reduce_rows(const float* ws, float* out, int64_t nrows, int64_t ncols, float scale,
                                hipStream_t stream) {
  CSTS_REQUIRE(ws && out && nrows > 0 && ncols > 0, "bad args");
  csts_reduce_rows_launch(ws, out, nrows, ncols, scale, stream);
  CSTS_LAUNCH_CHECK();
  return 0;
}

static int ln_grid(int64_t rows) { return (int)std::min<int64_t>(cdiv(rows, 4), 4096); }

extern "C" int csts_layernorm_fwd(const void* x, int x_dt, const float* gamma, const float* beta, void* y, int y_dt,
                                  float* mean, float* rstd, int64_t rows, int C, float eps, hipStream_t stream) {
  CSTS_REQUIRE(x && gamma && beta && y, "null pointer");
  CSTS_REQUIRE(rows > 0 && C > 0 && C <= 64 * MAXV, "C must be in (0, 768]");
  dim3 grid(ln_grid(rows)), block(256);
  if (C <= 128) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, block, 0, stream, x, x_dt, gamma, beta, y, y_dt, mean, rstd, rows, C, eps);
  else if (C <= 192) hipLaunchKernelGGL(ln_fwd_kernel<3>, grid, block, 0, stream, x, x_dt, gamma, beta, y, y_dt, mean, rstd, rows, C, eps);
  else if (C <= 384) hipLaunchKernelGGL(ln_fwd_kernel<6>, grid, block, 0, stream, x, x_dt, gamma, beta, y, y_dt, mean, rstd, rows, C, eps);
  else hipLaunchKernelGGL(ln_fwd_kernel<12>, grid, block, 0, stream, x, x_dt, gamma, beta, y, y_dt, mean, rstd, rows, C, eps);
  CSTS_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t csts_layernorm_bwd_workspace(int64_t rows, int C) {
  const int64_t nb = std::min<int64_t>(cdiv(rows, 4 * 2), 2048);
  return (size_t)nb * 2 * C * sizeof(float);
}

extern "C" int csts_layernorm_bwd(const void* dy, int dy_dt, const void* x, int x_dt, const float* gamma,
                                  const float* mean, const float* rstd, void* dx, int dx_dt, float* dgamma,
                                  float* dbeta, void* workspace, size_t ws_bytes, int64_t rows, int C,
                                  hipStream_t stream) {
  CSTS_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && workspace, "null pointer");
  CSTS_REQUIRE(rows > 0 && C > 0 && C <= 64 * MAXV, "C must be in (0, 768]");
  CSTS_REQUIRE(dbeta == dgamma + C, "dgamma/dbeta must be one contiguous [2*C] buffer");
  const int64_t nb = std::min<int64_t>(cdiv(rows, 4 * 2), 2048);
  CSTS_REQUIRE(ws_bytes >= (size_t)nb * 2 * C * sizeof(float), "workspace too small");
  dim3 grid((unsigned)nb), block(256);
  const size_t sh = (size_t)4 * 2 * C * sizeof(float);
  float* ws = reinterpret_cast<float*>(workspace);
  if (C <= 128) hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, block, sh, stream, dy, dy_dt, x, x_dt, gamma, mean, rstd, dx, dx_dt, ws, rows, C);
  else if (C <= 192) hipLaunchKernelGGL(ln_bwd_kernel<3>, grid, block, sh, stream, dy, dy_dt, x, x_dt, gamma, mean, rstd, dx, dx_dt, ws, rows, C);
  else if (C <= 384) hipLaunchKernelGGL(ln_bwd_kernel<6>, grid, block, sh, stream, dy, dy_dt, x, x_dt, gamma, mean, rstd, dx, dx_dt, ws, rows, C);
  else hipLaunchKernelGGL(ln_bwd_kernel<12>, grid, block, sh, stream, dy, dy_dt, x, x_dt, gamma, mean, rstd, dx, dx_dt, ws, rows, C);
  CSTS_LAUNCH_CHECK();
  csts_reduce_rows_launch(ws, dgamma, nb, 2 * C, 1.f, stream);
  CSTS_LAUNCH_CHECK();
  return 0;
}
