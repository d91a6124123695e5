// What does a kernel boundary cost inside a captured HIP graph on MI355X, and does the cache policy of the producer's stores
// change it?  (Per-XCD L2s: the release at the end of a kernel writes dirty lines back, the acquire of the next one invalidates.)
//   part 1: chains of empty kernels (1 workgroup / 2048 workgroups): microseconds per graph node
//   part 2: chains of [writer(S bytes, store mode) -> reader(S bytes, load mode)] pairs on rotating buffers, S = 3 .. 200 MB:
//           store mode 0 plain, 1 nontemporal (nt), 2 write-through (sc0 sc1); load mode 0 plain, 1 nt
//   part 3: writer-only chains and reader-only chains of the same sizes (what each side costs alone)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_floor.hip -o tools/bin/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <utility>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(int* p) { if (p != nullptr && threadIdx.x == 9999) p[0] = 1; }
__global__ void spin_kernel(int* p, int ticks) {      // busy for `ticks` counts of the 100 MHz realtime counter
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) { }
  if (threadIdx.x == 9999) p[0] = 1;
}

template <int SM>
__global__ __launch_bounds__(256) void writer_kernel(v4i* __restrict__ out, int64_t n16, int seed) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    v4i v = {seed, (int)i, seed ^ (int)i, 7};
    if (SM == 0) out[i] = v;
    else if (SM == 1) __builtin_nontemporal_store(v, out + i);
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(out + i), "v"(v) : "memory");
  }
}
template <int LM>
__global__ __launch_bounds__(256) void reader_kernel(const v4i* __restrict__ in, int64_t n16, int* __restrict__ sink) {
  v4i acc = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    v4i v = LM == 0 ? in[i] : __builtin_nontemporal_load(in + i);
    acc ^= v;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) sink[0] = 1;
}

// c = f(a, b): the streaming kernels of the step (LayerNorm, add2, AdamW): two loads + one store per element
template <int LM, int SM>
__global__ __launch_bounds__(256) void rw_kernel(const v4i* __restrict__ a, const v4i* __restrict__ b, v4i* __restrict__ c, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    v4i x = LM == 0 ? a[i] : __builtin_nontemporal_load(a + i);
    v4i y = LM == 0 ? b[i] : __builtin_nontemporal_load(b + i);
    v4i v = x ^ (y + 3);
    if (SM == 0) c[i] = v;
    else if (SM == 1) __builtin_nontemporal_store(v, c + i);
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(c + i), "v"(v) : "memory");
  }
}
template <int LM, int SM> static void launch_rw(const v4i* a, const v4i* b, v4i* c, int64_t n16, int grid, hipStream_t s) { hipLaunchKernelGGL((rw_kernel<LM, SM>), dim3(grid), dim3(256), 0, s, a, b, c, n16); }

// distinct kernels of a few KB of code each (I-dependent constants in an unrolled chain): does a launch pay for a cold instruction cache?
template <int I>
__global__ __launch_bounds__(256) void code_kernel(const float* __restrict__ in, float* __restrict__ out) {
  float x = in[blockIdx.x * 256 + threadIdx.x];
#pragma unroll
  for (int j = 0; j < 256; ++j) x = x * (1.0f + 1e-6f * (float)(I * 256 + j)) + (float)(j ^ I) * 1e-7f;
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
typedef void (*code_fn)(const float*, float*);
template <int... Is> static void fill_code_fns(code_fn* f, std::integer_sequence<int, Is...>) { ((f[Is] = code_kernel<Is>), ...); }

// Does a consumer find the producer's lines in the XCD's L2?  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8); the
// writer's workgroup b writes chunk b; the reader's workgroup b reads chunk (b + shift) % grid: shift 0 = the same XCD as the writer,
// shift 1 = the next XCD.  Chunks of 16 KB, whole buffer S bytes.
__global__ __launch_bounds__(256) void chunk_writer(v4i* __restrict__ out, int seed) {
  v4i* o = out + (int64_t)blockIdx.x * 1024;
  for (int i = threadIdx.x; i < 1024; i += 256) { v4i v = {seed, i, (int)blockIdx.x, 7}; o[i] = v; }
}
__global__ __launch_bounds__(256) void chunk_reader(const v4i* __restrict__ in, int shift, int* __restrict__ sink) {
  const v4i* p = in + (int64_t)((blockIdx.x + shift) % gridDim.x) * 1024;
  v4i acc = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < 1024; i += 256) acc ^= p[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) sink[0] = 1;
}

static float time_graph(hipGraphExec_t g, hipStream_t s, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(g, s));
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(g, s));
  CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}
template <class F> static hipGraphExec_t capture(hipStream_t s, F f) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  f();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  return ge;
}

template <int SM> static void launch_writer(v4i* o, int64_t n16, int grid, int seed, hipStream_t s) { hipLaunchKernelGGL(writer_kernel<SM>, dim3(grid), dim3(256), 0, s, o, n16, seed); }
template <int LM> static void launch_reader(const v4i* i, int64_t n16, int grid, int* sink, hipStream_t s) { hipLaunchKernelGGL(reader_kernel<LM>, dim3(grid), dim3(256), 0, s, i, n16, sink); }

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  int* sink; CK(hipMalloc(&sink, 4096));
  const int NODES = 256;
  for (int grid : {1, 256, 2048}) {
    hipGraphExec_t g = capture(s, [&] { for (int i = 0; i < NODES; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, s, sink); });
    printf("empty kernel chain, grid %5d: %.2f us per node\n", grid, time_graph(g, s, 20) * 1e3 / NODES);
  }
  // fork / join inside a captured graph: [A -> (B on a second stream || C) -> D] against A -> B -> C -> D, kernels that spin ~DUR us
  {
    hipStream_t s2; CK(hipStreamCreate(&s2));
    hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (int dur : {0, 5, 20}) {
      const int G = 64;
      auto K = [&](hipStream_t st) { hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, st, sink, dur * 100); };   // ~100 clocks of 100 MHz s_memrealtime per us
      hipGraphExec_t serial = capture(s, [&] { for (int i = 0; i < G; ++i) { K(s); K(s); K(s); K(s); } });
      hipGraphExec_t forked = capture(s, [&] {
        for (int i = 0; i < G; ++i) {
          K(s);
          CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0));
          K(s2); K(s);
          CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s, ej, 0));
          K(s);
        }
      });
      printf("4 kernels of ~%2d us: serial %.2f us per group, with one fork / join (2 run side by side) %.2f us per group\n", dur,
             time_graph(serial, s, 10) * 1e3 / G, time_graph(forked, s, 10) * 1e3 / G);
    }
  }
  {
    code_fn fns[64];
    fill_code_fns(fns, std::make_integer_sequence<int, 64>());
    float *a, *b; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20));
    for (int distinct : {1, 2, 8, 64}) {
      hipGraphExec_t g = capture(s, [&] { for (int i = 0; i < 256; ++i) hipLaunchKernelGGL(fns[i % distinct], dim3(512), dim3(256), 0, s, (const float*)a, b); });
      printf("chain of 256 launches (512 workgroups, ~3 KB of code each) cycling over %2d distinct kernels: %.2f us per node\n", distinct, time_graph(g, s, 10) * 1e3 / 256);
    }
  }
  {
    v4i* cb[4];
    for (auto& b : cb) CK(hipMalloc(&b, 64 << 20));
    printf("\nproducer -> consumer through the XCD's L2?  us per [writer, reader] pair (writer alone in brackets), rotating over 4 buffers\n");
    for (int mb : {2, 6, 12, 24, 48}) {
      const int grid = mb * 64;                                   // 16 KB chunks
      float w_only, r[3];
      { hipGraphExec_t g = capture(s, [&] { for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(chunk_writer, dim3(grid), dim3(256), 0, s, cb[i % 4], i); });
        w_only = time_graph(g, s, 5) * 1e3 / 64; }
      int k = 0;
      for (int shift : {0, 1, 4}) {
        hipGraphExec_t g = capture(s, [&] { for (int i = 0; i < 64; ++i) { hipLaunchKernelGGL(chunk_writer, dim3(grid), dim3(256), 0, s, cb[i % 4], i);
                                                                         hipLaunchKernelGGL(chunk_reader, dim3(grid), dim3(256), 0, s, (const v4i*)cb[i % 4], shift, sink); } });
        r[k++] = time_graph(g, s, 5) * 1e3 / 64;
      }
      printf("%3d MB: same XCD %.2f   next XCD %.2f   XCD + 4 %.2f   (writer alone %.2f)\n", mb, r[0], r[1], r[2], w_only);
    }
  }
  const int NBUF = 6;                                           // rotating buffers: 6 x 200 MB > the 256 MB memory-side cache
  const int64_t MAXB = 200ll << 20;
  std::vector<v4i*> buf(NBUF);
  for (auto& b : buf) { CK(hipMalloc(&b, MAXB)); CK(hipMemset(b, 1, MAXB)); }
  const int PAIRS = 48;
  printf("\n%8s | pair us: store plain/nt/wt x load plain | load nt (store plain/nt/wt) | writer-only plain/nt/wt | reader-only plain/nt\n", "MB");
  for (double mb : {1.5, 3.0, 6.0, 12.5, 25.0, 50.0, 100.0, 200.0}) {
    const int64_t n16 = (int64_t)(mb * (1 << 20)) / 16;
    const int grid = (int)std::min<int64_t>((n16 + 255) / 256, 256 * 8);     // <= 8 workgroups per CU, grid-stride beyond
    float pr[2][3], wr[3], rd[2];
    for (int lm = 0; lm < 2; ++lm)
      for (int sm = 0; sm < 3; ++sm) {
        hipGraphExec_t g = capture(s, [&] {
          for (int i = 0; i < PAIRS; ++i) {
            v4i* b = buf[i % NBUF];
            if (sm == 0) launch_writer<0>(b, n16, grid, i, s); else if (sm == 1) launch_writer<1>(b, n16, grid, i, s); else launch_writer<2>(b, n16, grid, i, s);
            if (lm == 0) launch_reader<0>(b, n16, grid, sink, s); else launch_reader<1>(b, n16, grid, sink, s);
          }
        });
        pr[lm][sm] = time_graph(g, s, 5) * 1e3 / PAIRS;
        CK(hipGraphExecDestroy(g));
      }
    for (int sm = 0; sm < 3; ++sm) {
      hipGraphExec_t g = capture(s, [&] {
        for (int i = 0; i < PAIRS; ++i) {
          v4i* b = buf[i % NBUF];
          if (sm == 0) launch_writer<0>(b, n16, grid, i, s); else if (sm == 1) launch_writer<1>(b, n16, grid, i, s); else launch_writer<2>(b, n16, grid, i, s);
        }
      });
      wr[sm] = time_graph(g, s, 5) * 1e3 / PAIRS;
      CK(hipGraphExecDestroy(g));
    }
    for (int lm = 0; lm < 2; ++lm) {
      hipGraphExec_t g = capture(s, [&] {
        for (int i = 0; i < PAIRS; ++i) {
          if (lm == 0) launch_reader<0>(buf[i % NBUF], n16, grid, sink, s); else launch_reader<1>(buf[i % NBUF], n16, grid, sink, s);
        }
      });
      rd[lm] = time_graph(g, s, 5) * 1e3 / PAIRS;
      CK(hipGraphExecDestroy(g));
    }
    printf("%8.1f | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f | %7.1f %7.1f   (ideal pair at 5 TB/s: %.1f us)\n", mb, pr[0][0], pr[0][1], pr[0][2],
           pr[1][0], pr[1][1], pr[1][2], wr[0], wr[1], wr[2], rd[0], rd[1], 2 * mb * 1.048576 / 5.0);
  }
  printf("\nc = f(a, b) chains (kernel k reads the outputs of k-1 and k-2, rotating over 6 buffers): us per kernel\n%8s | ld plain: st plain/nt/wt | ld nt: st plain/nt/wt   (ideal at 5 TB/s)\n", "MB");
  for (double mb : {6.0, 12.5, 25.0, 50.0, 100.0, 200.0}) {
    const int64_t n16 = (int64_t)(mb * (1 << 20)) / 16;
    const int grid = (int)std::min<int64_t>((n16 + 255) / 256, 256 * 8);
    float r[2][3];
    for (int lm = 0; lm < 2; ++lm)
      for (int sm = 0; sm < 3; ++sm) {
        hipGraphExec_t g = capture(s, [&] {
          for (int i = 0; i < PAIRS; ++i) {
            const v4i *a = buf[(i + 4) % NBUF], *b = buf[(i + 5) % NBUF];
            v4i* c = buf[i % NBUF];
            if (lm == 0) { if (sm == 0) launch_rw<0, 0>(a, b, c, n16, grid, s); else if (sm == 1) launch_rw<0, 1>(a, b, c, n16, grid, s); else launch_rw<0, 2>(a, b, c, n16, grid, s); }
            else { if (sm == 0) launch_rw<1, 0>(a, b, c, n16, grid, s); else if (sm == 1) launch_rw<1, 1>(a, b, c, n16, grid, s); else launch_rw<1, 2>(a, b, c, n16, grid, s); }
          }
        });
        r[lm][sm] = time_graph(g, s, 5) * 1e3 / PAIRS;
        CK(hipGraphExecDestroy(g));
      }
    printf("%8.1f | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f   (%.1f)\n", mb, r[0][0], r[0][1], r[0][2], r[1][0], r[1][1], r[1][2], 3 * mb * 1.048576 / 5.0);
  }
  return 0;
}
