// Where does the time of a SHORT dependent kernel go?  A hipGraph chain alternates a writer (stands for a GEMM epilogue: every
// workgroup stores 16 bytes per lane) and a timed reader shaped like gn_silu (32-192 workgroups x 512 threads: kernel arguments
// -> one batch of loads of what the writer just stored + a batch from a never-written parameter table -> two barriers -> store).
// The reader samples s_memtime at each stage (lane 0 of every workgroup); the host prints stage durations in shader cycles and
// the s_memrealtime (100 MHz) span of the kernel's first to last workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/hop_bench.hip -o tools/hop_bench.bin && tools/hop_bench.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(512) void writer(uint4 *x, unsigned seed) {
  const unsigned i = blockIdx.x * 512u + threadIdx.x;
  x[i] = make_uint4(i, seed, i ^ seed, 1u);
}

// the other branch: a bandwidth-heavy kernel (192 workgroups streaming 64 MB) on a second stream
__global__ __launch_bounds__(256) void streamer(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint4 v = src[i];
    acc.x += v.x;
    acc.y ^= v.y;
  }
  if (acc.x == 0x12345u) dst[threadIdx.x] = acc;
}

constexpr int NS = 8;
// ts[wg][0..NS): entry, args usable, fresh-data loads landed, parameter loads landed, barrier 1, barrier 2, stores issued, stores done
__global__ __launch_bounds__(512) void reader(const uint4 *__restrict__ x, const uint4 *__restrict__ par, uint4 *__restrict__ out,
                                              unsigned long long *ts, unsigned long long *wt, int rec, int shift) {
  __shared__ unsigned red[16];
  const unsigned long long t0 = now(), w0 = wall();
  const unsigned i = blockIdx.x * 512u + threadIdx.x;
  const uint4 *px = x + ((blockIdx.x + shift) % gridDim.x) * 512u + threadIdx.x;   // shift 3: rows another XCD's workgroup wrote
  asm volatile("" ::"v"(px));
  const unsigned long long t1 = now();
  uint4 a = *px;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = now();
  uint4 p = par[(i * 7u + (unsigned)rec * 0u) & 0xFFFFu];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t3 = now();
  unsigned s = a.x + a.y + p.x;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const unsigned long long t4 = now();
  s += red[(threadIdx.x >> 6) ^ 1];
  __syncthreads();
  const unsigned long long t5 = now();
  out[i] = make_uint4(s, a.z, p.y, a.w);
  const unsigned long long t6 = now();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t7 = now(), w1 = wall();
  if (threadIdx.x == 0 && rec) {
    unsigned long long *t = ts + (size_t)blockIdx.x * NS;
    t[0] = t0; t[1] = t1; t[2] = t2; t[3] = t3; t[4] = t4; t[5] = t5; t[6] = t6; t[7] = t7;
    wt[blockIdx.x * 2] = w0;
    wt[blockIdx.x * 2 + 1] = w1;
  }
}

int main() {
  const int maxwg = 256;
  uint4 *x, *par, *out;
  unsigned long long *ts, *wt;
  CK(hipMalloc(&x, (size_t)maxwg * 512 * 16));
  CK(hipMalloc(&par, (size_t)65536 * 16));
  CK(hipMalloc(&out, (size_t)maxwg * 512 * 16));
  CK(hipMalloc(&ts, (size_t)maxwg * NS * 8));
  CK(hipMalloc(&wt, (size_t)maxwg * 2 * 8));
  CK(hipMemset(par, 1, (size_t)65536 * 16));
  uint4 *big;   // 512 MB streamed between timed pairs in the "evict" mode: what a step's worth of weight traffic does to the caches
  CK(hipMalloc(&big, (size_t)512 << 20));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const char *names[NS - 1] = {"args usable", "fresh loads", "param loads", "barrier 1", "barrier 2", "store issue", "store done"};
  hipStream_t st2;
  CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  for (int wgs : {32, 192}) {
    for (int mode = 0; mode < 4; ++mode) {
      const int evict = mode == 1, shift = mode >= 2 ? 3 : 0, busy = mode == 3;
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 60; ++i) {
        if (evict && i % 10 == 0) CK(hipMemsetAsync(big, i, (size_t)512 << 20, st));
        hipLaunchKernelGGL(writer, dim3(wgs), dim3(512), 0, st, x, (unsigned)i);
        hipLaunchKernelGGL(reader, dim3(wgs), dim3(512), 0, st, x, par, out, ts, wt, i == 59 ? 1 : 0, shift);
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphDestroy(g));
      std::vector<double> acc(NS - 1, 0.0);
      double span = 0, life = 0;
      const int reps = 20;
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) {
        if (busy)
          for (int k = 0; k < 12; ++k) hipLaunchKernelGGL(streamer, dim3(192), dim3(256), 0, st2, big + (size_t)(k % 7) * (4 << 20), big, 4 << 20);   // 12 x 64 MB
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipStreamSynchronize(st2));
        std::vector<unsigned long long> h((size_t)wgs * NS), w((size_t)wgs * 2);
        CK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(w.data(), wt, w.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long wmin = ~0ull, wmax = 0;
        for (int b = 0; b < wgs; ++b) {
          for (int k = 0; k < NS - 1; ++k) acc[k] += double(h[(size_t)b * NS + k + 1] - h[(size_t)b * NS + k]) / wgs;
          life += double(w[b * 2 + 1] - w[b * 2]) / wgs;
          wmin = std::min(wmin, w[b * 2]);
          wmax = std::max(wmax, w[b * 2 + 1]);
        }
        span += double(wmax - wmin);
      }
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%3d workgroups x 512, %s: per-stage shader cycles (mean over workgroups):", wgs, mode == 0 ? "same rows, back to back    " : mode == 1 ? "512 MB memset every 10    " : mode == 2 ? "rows of another workgroup " : "other rows + busy 2nd strm");
      for (int k = 0; k < NS - 1; ++k) printf("  %s %.0f", names[k], acc[k] / reps);
      printf("\n      wave life %.2f us, first entry -> last exit %.2f us (100 MHz wall clock); whole graph %.1f us for 60 writer+reader pairs\n", life / reps / 100.0,
             span / reps / 100.0, us / reps);
      CK(hipGraphExecDestroy(ge));
    }
  }
  return 0;
}
