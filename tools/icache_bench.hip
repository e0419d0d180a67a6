// What does straight-line code executed ONCE per wave cost when the instruction cache is cold?
// A denoise step is a chain of ~230 dependent short launches of ~25 DIFFERENT kernels whose code (2-17 KB each, mostly
// unrolled prologue / epilogue executed once) is evicted between two uses by a step's worth of traffic.  This measures a
// hipGraph chain of kernels with `KB` kilobytes of straight-line VALU code (no memory traffic besides one store):
//   hot  = the same kernel every launch (its code stays in the instruction cache)
//   cold = 48 distinct instantiations of the same body in rotation (48 x KB of code between two uses of one)
//   hipcc --offload-arch=gfx950 -O3 tools/icache_bench.hip -o /tmp/icache_bench && /tmp/icache_bench
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

// 128 four-byte instructions = 512 bytes of code per block: a fetch-paced mix (7 one-cycle s_nop per 4-cycle v_fmac, ~1.4 cycles
// per instruction, close to the SALU / VALU / waitcnt mix of a GEMM prologue)
#define B8 "v_fmac_f32 %0, %1, %1\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
#define B64 B8 B8 B8 B8 B8 B8 B8 B8
#define B128 B64 B64

template <int HALF_KB, int ID> __global__ void k_code(float *out, float seed) {
  float a = seed + ID, b = seed * 0.5f;
#pragma unroll
  for (int i = 0; i < HALF_KB; ++i) asm volatile(B128 : "+v"(a) : "v"(b));
  if (a == 12345.f) out[threadIdx.x] = a;   // never true: the chain is kept alive without a store
}

template <int HALF_KB, int... IDS> void fill(std::vector<void (*)(float *, float)> &v, std::integer_sequence<int, IDS...>) {
  (v.push_back(k_code<HALF_KB, IDS>), ...);
}

template <int HALF_KB> int run(float *buf, int grid, int threads) {
  std::vector<void (*)(float *, float)> ks;
  fill<HALF_KB>(ks, std::make_integer_sequence<int, 48>());
  const int N = 240;
  for (int cold = 0; cold < 2; ++cold) {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(ks[cold ? i % 48 : 0], dim3(grid), dim3(threads), 0, st, buf, 1.0f);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const int reps = 20;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("code %5.1f KB  grid %4d x %4d threads  %s: %6.2f us per kernel\n", HALF_KB * 0.5, grid, threads, cold ? "cold (48 kernels in rotation)" : "hot  (one kernel)             ",
           us / reps / N);
    CK(hipGraphExecDestroy(ge));
    CK(hipStreamDestroy(st));
  }
  return 0;
}

int main() {
  float *buf;
  CK(hipMalloc(&buf, 1 << 20));
  for (int grid : {32, 192}) {
    for (int threads : {256}) {
      if (run<1>(buf, grid, threads)) return 1;
      if (run<4>(buf, grid, threads)) return 1;
      if (run<8>(buf, grid, threads)) return 1;
      if (run<16>(buf, grid, threads)) return 1;
      if (run<32>(buf, grid, threads)) return 1;
    }
  }
  return 0;
}
