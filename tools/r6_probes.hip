// Round-6 hardware probes (run on the GPU box):
//   1. does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (the split-operand GEMMs feed it lo parts that may be tiny)?
//   2. hipExtStreamCreateWithCUMask: which mask bit is which XCD, does a kernel launched on a masked stream stay on the masked XCDs, and does a
//      hipGraph captured on / replayed on a masked stream keep the mask?
//   hipcc --offload-arch=gfx950 -O3 tools/r6_probes.hip -o /tmp/r6_probes && /tmp/r6_probes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// every lane feeds a = va (all 8 k), b = vb: D[i][j] = 16 * va * vb
__global__ void k_mfma_f16(float va, float vb, float *out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)va;
    b[i] = (_Float16)vb;
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
__global__ void k_mfma_bf16(float va, float vb, float *out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)va;
    b[i] = (__bf16)vb;
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

// one record per workgroup: the XCD it ran on (bits 0-3) and its HW_ID bits 8-15 (cu_id, sh_id, se_id) in bits 8-15; spins a little so
// that the workgroups of a launch coexist
__global__ void k_xcc(int *out, int spin) {
  if (threadIdx.x == 0) {
    unsigned id = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11));   // bits [3:0]
    unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (8 << 6) | (7 << 11));     // bits [15:8]: cu_id[11:8], sh_id[12], se_id[15:13]
    out[blockIdx.x] = (int)((id & 15) | ((hw & 255) << 8));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {
  }
}

#include <set>
static void histo(const char *what, const std::vector<int> &v) {
  int h[16] = {0};
  std::set<int> cus;
  for (int x : v) {
    h[x & 15]++;
    cus.insert(x & 0xffff);
  }
  printf("  %-58s XCD histogram:", what);
  for (int i = 0; i < 8; ++i) printf(" %d", h[i]);
  printf("   distinct (xcd, se, sh, cu): %d\n", (int)cus.size());
}

int main() {
  float *d = nullptr, h = 0.f;
  CK(hipMalloc(&d, 64));
  printf("== 1. fp16 / bf16 MFMA with subnormal inputs (expect 16 * a * b) ==\n");
  struct {
    float a, b;
  } cases[] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {1.0e-6f, 1.0f}, {6.0e-8f, 1.0f}, {3.0e-5f, 3.0e-5f}, {1.0e-6f, 1024.0f}};
  for (auto &c : cases) {
    hipLaunchKernelGGL(k_mfma_f16, dim3(1), dim3(64), 0, 0, c.a, c.b, d);
    CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    const float ea = (float)(_Float16)c.a, eb = (float)(_Float16)c.b;
    printf("  f16  a=%.3e (as f16 %.6e) b=%.3e : got %.6e expect %.6e\n", c.a, ea, c.b, h, 16.f * ea * eb);
  }
  hipLaunchKernelGGL(k_mfma_bf16, dim3(1), dim3(64), 0, 0, 1.0e-39f, 1.0e20f, d);
  CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
  printf("  bf16 a=1e-39 (subnormal) b=1e20 : got %.6e expect %.6e\n", h, 16.f * (float)(__bf16)1.0e-39f * (float)(__bf16)1.0e20f);

  printf("== 2. CU-masked streams ==\n");
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("  multiProcessorCount %d\n", ncu);
  const int G = 1024;
  int *dx = nullptr;
  CK(hipMalloc(&dx, G * sizeof(int)));
  std::vector<int> hx(G);
  auto run = [&](hipStream_t s, const char *what) -> int {
    CK(hipMemsetAsync(dx, 0xff, G * sizeof(int), s));
    hipLaunchKernelGGL(k_xcc, dim3(G), dim3(64), 0, s, dx, 20000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(hx.data(), dx, G * sizeof(int), hipMemcpyDeviceToHost));
    histo(what, hx);
    return 0;
  };
  hipStream_t s0;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  if (run(s0, "unmasked stream")) return 1;
  const int words = (ncu + 31) / 32;
  // hypothesis A: bit k <-> XCD k % 8 (interleaved);  hypothesis B: bit k <-> XCD k / (ncu / 8) (contiguous)
  for (int hyp = 0; hyp < 2; ++hyp) {
    for (int half = 0; half < 2; ++half) {
      std::vector<uint32_t> mask(words, 0);
      for (int k = 0; k < ncu; ++k) {
        const int xcd = hyp == 0 ? (k % 8) : (k / (ncu / 8));
        if ((xcd < 4) == (half == 0)) mask[k / 32] |= 1u << (k % 32);
      }
      hipStream_t sm;
      hipError_t e = hipExtStreamCreateWithCUMask(&sm, (uint32_t)words, mask.data());
      if (e != hipSuccess) {
        printf("  hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e));
        return 0;
      }
      char what[128];
      snprintf(what, sizeof what, "masked stream, %s bits, %s half", hyp == 0 ? "interleaved (k%%8<4)" : "contiguous", half == 0 ? "first" : "second");
      if (run(sm, what)) return 1;
      if (hyp == 0) {
        // graph captured on the masked stream, replayed (a) on the masked stream, (b) on the plain stream
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(sm, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_xcc, dim3(G), dim3(64), 0, sm, dx, 20000);
        CK(hipStreamEndCapture(sm, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, sm));
        CK(hipStreamSynchronize(sm));
        CK(hipMemcpy(hx.data(), dx, G * sizeof(int), hipMemcpyDeviceToHost));
        histo("  graph captured on it, replayed on the masked stream", hx);
        CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        CK(hipMemcpy(hx.data(), dx, G * sizeof(int), hipMemcpyDeviceToHost));
        histo("  same graph replayed on the UNMASKED stream", hx);
        // graph captured on the plain stream, replayed on the masked one
        hipGraph_t g2;
        hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_xcc, dim3(G), dim3(64), 0, s0, dx, 20000);
        CK(hipStreamEndCapture(s0, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge2, sm));
        CK(hipStreamSynchronize(sm));
        CK(hipMemcpy(hx.data(), dx, G * sizeof(int), hipMemcpyDeviceToHost));
        histo("  graph captured on the PLAIN stream, replayed on the masked one", hx);
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphExecDestroy(ge2);
        (void)hipGraphDestroy(g);
        (void)hipGraphDestroy(g2);
      }
      (void)hipStreamDestroy(sm);
    }
  }
  // which CU is mask bit k?  single-bit masks (every workgroup of the launch must then run on that one CU)
  printf("== 3. single-bit CU masks: where do the workgroups run? (xcd, se, sh, cu) ==\n");
  for (int k : {0, 1, 2, 7, 8, 9, 16, 31, 32, 33, 64, 128, 255}) {
    std::vector<uint32_t> mask(words, 0);
    mask[k / 32] = 1u << (k % 32);
    hipStream_t sm;
    if (hipExtStreamCreateWithCUMask(&sm, (uint32_t)words, mask.data()) != hipSuccess) break;
    CK(hipMemsetAsync(dx, 0xff, 64 * sizeof(int), sm));
    hipLaunchKernelGGL(k_xcc, dim3(64), dim3(64), 0, sm, dx, 2000);
    CK(hipStreamSynchronize(sm));
    CK(hipMemcpy(hx.data(), dx, 64 * sizeof(int), hipMemcpyDeviceToHost));
    std::set<int> where(hx.begin(), hx.begin() + 64);
    printf("  bit %3d ->", k);
    int n = 0;
    for (int w : where) {
      if (n++ < 6) printf(" (x%d se%d sh%d cu%d)", w & 15, (w >> 13) & 7, (w >> 12) & 1, (w >> 8) & 15);
    }
    printf("%s  [%d distinct]\n", where.size() > 6 ? " ..." : "", (int)where.size());
    (void)hipStreamDestroy(sm);
  }
  return 0;
}
