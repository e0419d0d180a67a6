// Where does a K step of the macro-tile GEMM go?  Builds conv_gemm_mt.hip with -DSF_MT_STAMPS (in-kernel s_memtime stamps, workgroup 0)
// and prints, per wave, the mean s_memtime ticks per K step spent in: counted wait | barrier | DMA issue | fragment reads + MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I syncfusion_amd/csrc -DSF_MT_STAMPS -mllvm -amdgpu-kernarg-preload-count=16 tools/mt_stamps.hip -o build/mt_stamps
//   build/mt_stamps B L C N taps variant [two]     (two = 1: a second identical launch runs beside it on another stream)
#include "../syncfusion_amd/csrc/conv_gemm_mt.hip"

#include <cstdio>
#include <vector>

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 7) return 1;
  const int B = atoi(argv[1]), L = atoi(argv[2]), C = atoi(argv[3]), N = atoi(argv[4]), taps = atoi(argv[5]), variant = atoi(argv[6]);
  const int two = argc > 7 ? atoi(argv[7]) : 0;
  const int K = taps * C, M = B * L;
  void *x, *w, *out[2], *res;
  float *bias;
  CK(hipMalloc(&x, (size_t)M * C * 2));
  CK(hipMalloc(&w, (size_t)N * K * 2));
  CK(hipMalloc(&out[0], (size_t)M * N * 2));
  CK(hipMalloc(&out[1], (size_t)M * N * 2));
  CK(hipMalloc(&res, (size_t)M * N * 2));
  CK(hipMalloc(&bias, N * 4));
  CK(hipMemset(x, 0x3c, (size_t)M * C * 2));
  CK(hipMemset(w, 0xbc, (size_t)N * K * 2));
  CK(hipMemset(res, 0x3d, (size_t)M * N * 2));
  CK(hipMemset(bias, 0, N * 4));
  sf::ConvGemmArgs a;
  a.src = x;
  a.src_ld = C;
  a.w = w;
  a.bias = bias;
  a.N = N;
  a.K = K;
  a.cin = C;
  a.taps = taps;
  a.pad = taps / 2;
  a.Lsrc = a.Lout = L;
  a.M = M;
  a.out = out[0];
  a.out_ld = a.n_store = N;
  a.res = res;
  a.res_ld = N;
  hipStream_t s[2];
  CK(hipStreamCreate(&s[0]));
  CK(hipStreamCreate(&s[1]));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20;
  for (int it = 0; it < iters + 3; ++it) {
    if (it == 3) CK(hipEventRecord(e0, s[0]));
    a.out = out[0];
    CK((sf::launch_mt_v<sf::bf16, 0, false>(a, variant, s[0])));
    if (two) {
      a.out = out[1];
      CK((sf::launch_mt_v<sf::bf16, 0, false>(a, variant, s[1])));
    }
  }
  CK(hipEventRecord(e1, s[0]));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[8][8];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(sf::g_mt_stamps), sizeof(h)));
  printf("M %d N %d K %d variant %d%s: %.2f us per launch (stream 0)\n", M, N, K, variant, two ? " (+ a second launch beside it)" : "", ms * 1e3f / iters);
  printf("wave  steps    wait  barrier  issue  reads+mfma | per step | prologue  epilogue    kernel   (s_memtime ticks)\n");
  for (int wv = 0; wv < 8; ++wv) {
    const double nk = (double)h[wv][7];
    printf("%4d  %5.0f  %6.1f  %7.1f  %5.1f  %10.1f | %8.1f | %8llu  %8llu  %8llu\n", wv, nk, h[wv][0] / nk, h[wv][1] / nk, h[wv][2] / nk, h[wv][3] / nk,
           (h[wv][0] + h[wv][1] + h[wv][2] + h[wv][3]) / nk, h[wv][4], h[wv][5], h[wv][6]);
  }
  return 0;
}
