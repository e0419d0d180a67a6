// Stand-alone timing of the 8-channel-level kernels (syncfusion_amd/csrc/conv_d0.hip) on synthetic tensors:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I syncfusion_amd/csrc [-D...] tools/d0_bench.hip -o build/d0_bench && build/d0_bench B L rw
// prints the average launch time of d0_conv and d0_tail (SF_D0_WAVES picks the workgroup size).  Timing only: parity is tests/.
#include "../syncfusion_amd/csrc/conv_d0.hip"

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
  const int B = argc > 1 ? atoi(argv[1]) : 64, L = argc > 2 ? atoi(argv[2]) : 11264, rw = argc > 3 ? atoi(argv[3]) : 2048;
  const int iters = argc > 4 ? atoi(argv[4]) : 50;
  const int nchw = (L + rw - 1) / rw;
  const size_t rows = (size_t)B * L;
  void *x, *h, *o, *ctx;
  float *w32, *w3, *vec, *sA, *sB, *ss;
  CK(hipMalloc(&x, rows * 16));
  CK(hipMalloc(&h, rows * 16));
  CK(hipMalloc(&o, rows * 16));
  CK(hipMalloc(&ctx, rows * 16));
  CK(hipMemset(x, 0x3c, rows * 16));
  CK(hipMemset(h, 0x3d, rows * 16));
  CK(hipMemset(ctx, 0x3c, rows * 16));
  std::vector<float> hw(8 * 24, 0.01f), hv(64, 0.5f), hs((size_t)B * nchw * 16, 1.0f), hss((size_t)B * 16, 0.1f);
  CK(hipMalloc(&w32, hw.size() * 4));
  CK(hipMalloc(&w3, hw.size() * 4));
  CK(hipMalloc(&vec, hv.size() * 4));
  CK(hipMalloc(&sA, hs.size() * 4));
  CK(hipMalloc(&sB, hs.size() * 4));
  CK(hipMalloc(&ss, hss.size() * 4));
  CK(hipMemcpy(w32, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w3, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(vec, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sA, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ss, hss.data(), hss.size() * 4, hipMemcpyHostToDevice));
  sf::ConvThinArgs a;
  a.src = x;
  a.w = w32;
  a.w32 = w32;
  a.out = h;
  a.bias = vec;
  a.gamma = vec + 8;
  a.beta = vec + 16;
  a.stats_in = sA;
  a.stats_out = sB;
  a.B = B;
  a.L = a.Ls = L;
  a.C = a.N = 8;
  a.taps = 3;
  a.src_ld = a.out_ld = a.res_ld = 8;
  a.pro = 1;
  a.G = 8;
  a.nch_in = nchw;
  a.chunk_in = rw;
  a.rw = rw;
  a.nchw = nchw;
  sf::ThinTailArgs t;
  t.h = h;
  t.x = x;
  t.ctx = ctx;
  t.ctx_ld = 8;
  t.w2 = w32;
  t.w3 = w3;
  t.w2_32 = w32;
  t.w3_32 = w3;
  t.c2real = 2;
  t.out = o;
  t.bias2 = vec;
  t.bias3 = vec + 24;
  t.gamma = vec + 8;
  t.beta = vec + 16;
  t.stats_in = sB;
  t.ss = ss;
  t.ss_ld = 16;
  t.stats_out = sA;
  t.B = B;
  t.L = L;
  t.C = t.C2 = 8;
  t.G = 8;
  t.nch_in = nchw;
  t.chunk_in = rw;
  t.rw = rw;
  t.nchw = nchw;
  if (!sf::d0_conv_supported(a) || !sf::d0_tail_supported(t)) {
    fprintf(stderr, "shape not supported\n");
    return 1;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms[2] = {0.f, 0.f};
  for (int which = 0; which < 2; ++which) {
    for (int i = 0; i < 5; ++i) CK(which ? sf::launch_d0_tail(sf::BF16, t, nullptr) : sf::launch_d0_conv(sf::BF16, a, nullptr));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) CK(which ? sf::launch_d0_tail(sf::BF16, t, nullptr) : sf::launch_d0_conv(sf::BF16, a, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[which], e0, e1));
  }
  const char *wv = getenv("SF_D0_WAVES");
  printf("B %d L %d rw %d (%d workgroups, %s waves): conv %.1f us  tail %.1f us\n", B, L, rw, B * nchw, wv ? wv : "4", ms[0] * 1e3f / iters,
         ms[1] * 1e3f / iters);
  return 0;
}
