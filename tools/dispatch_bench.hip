// How fast does this box retire chains of dependent short kernels replayed from hipGraphs, on 1 / 2 / 4 streams?
// Three bodies: an empty kernel, a kernel with one dependent memory round trip per workgroup (load -> store), and a
// kernel with a serial chain of `hops` dependent loads (the K loop of a latency-bound GEMM tile).
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_bench.hip -o /tmp/dispatch_bench && /tmp/dispatch_bench
#include <hip/hip_runtime.h>

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

__global__ void k_empty(int *p) {
  if (p == nullptr) p[0] = 1;
}

// every workgroup: `hops` dependent 16-byte loads per lane from a 64 MB buffer, then one store
__global__ void k_hops(const uint4 *src, uint4 *dst, int hops, unsigned mask) {
  unsigned i = (blockIdx.x * 256u + threadIdx.x) & mask;
  uint4 v = make_uint4(0, 0, 0, 0);
  for (int h = 0; h < hops; ++h) {
    uint4 t = src[i];
    v.x += t.x;
    v.y += t.y;
    i = (i + 977u * 256u + (t.x & 1u)) & mask;
  }
  dst[blockIdx.x * 256u + threadIdx.x] = v;
}

int main() {
  const int N = 233;
  const size_t elems = (64u << 20) / sizeof(uint4);
  uint4 *src = nullptr, *dst = nullptr;
  CK(hipMalloc(&src, elems * sizeof(uint4)));
  CK(hipMalloc(&dst, elems * sizeof(uint4)));
  CK(hipMemset(src, 0, elems * sizeof(uint4)));
  const unsigned mask = (unsigned)elems - 1;
  struct Body {
    const char *name;
    int grid, hops;
  } bodies[] = {{"empty", 256, -1}, {"1 hop, 192 WGs", 192, 1}, {"1 hop, 704 WGs", 704, 1}, {"6 hops, 192 WGs", 192, 6}, {"24 hops, 192 WGs", 192, 24},
                {"24 hops, 384 WGs", 384, 24}};
  for (const Body &b : bodies) {
    for (int ns : {1, 2, 4}) {
      std::vector<hipStream_t> st(ns);
      std::vector<hipGraphExec_t> ge(ns);
      for (int s = 0; s < ns; ++s) {
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
          if (b.hops < 0) hipLaunchKernelGGL(k_empty, dim3(b.grid), dim3(256), 0, st[s], (int *)dst);
          else hipLaunchKernelGGL(k_hops, dim3(b.grid), dim3(256), 0, st[s], src, dst + (size_t)s * 1048576, b.hops, mask);
        }
        CK(hipStreamEndCapture(st[s], &g));
        CK(hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
      }
      const int reps = 20;
      for (int s = 0; s < ns; ++s) CK(hipGraphLaunch(ge[s], st[s]));
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r)
        for (int s = 0; s < ns; ++s) CK(hipGraphLaunch(ge[s], st[s]));
      for (int s = 0; s < ns; ++s) CK(hipStreamSynchronize(st[s]));
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%-18s streams=%d: %.2f us per kernel per chain, %.2f us per kernel overall\n", b.name, ns, us / reps / N, us / reps / N / ns);
      for (int s = 0; s < ns; ++s) {
        CK(hipGraphExecDestroy(ge[s]));
        CK(hipStreamDestroy(st[s]));
      }
    }
  }
  return 0;
}
