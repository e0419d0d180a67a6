// Host-side plumbing shared by the engines behind the C ABI: error reporting, named-weight lookup,
// a device arena for packed weights and a bump allocator over the caller's workspace.
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/syncfusion_amd.h"
#include "kernels.h"

namespace sf {

void set_error(const char *fmt, ...);
const char *get_error();

struct EngineError {
  int code;
};
// Throwing is confined to the engine; every extern "C" entry point catches EngineError.
[[noreturn]] void fail(int code, const char *fmt, ...);

#define SF_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) ::sf::fail(SF_ERR_HIP, "%s -> %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

class WeightMap {
 public:
  WeightMap(const sf_tensor *w, int n);
  // device pointer of a named fp32 parameter; fails with SF_ERR_MISSING_WEIGHT when absent or mis-sized
  const float *get(const std::string &name, int64_t numel) const;
  bool has(const std::string &name) const { return map_.count(name) != 0; }

 private:
  std::unordered_map<std::string, std::pair<const void *, int64_t>> map_;
};

// Grows in 64 MiB chunks; everything is freed with the engine.
class DeviceArena {
 public:
  ~DeviceArena();
  void *alloc(int64_t bytes);
  template <typename T> T *alloc_n(int64_t n) { return static_cast<T *>(alloc(n * (int64_t)sizeof(T))); }
  int64_t total() const { return total_; }

 private:
  std::vector<void *> chunks_;
  char *cur_ = nullptr;
  int64_t left_ = 0, total_ = 0;
};

// Bump allocator over the caller's workspace.  With base == nullptr it only measures.
class Workspace {
 public:
  Workspace(void *base, int64_t cap) : base_(static_cast<char *>(base)), cap_(cap) {}
  void *alloc(int64_t bytes) {
    int64_t off = align_up(used_, 256);
    used_ = off + bytes;
    if (!base_) return nullptr;
    if (used_ > cap_) fail(SF_ERR_WORKSPACE, "workspace too small: need > %lld bytes, have %lld", (long long)used_, (long long)cap_);
    return base_ + off;
  }
  template <typename T> T *alloc_n(int64_t n) { return static_cast<T *>(alloc(n * (int64_t)sizeof(T))); }
  int64_t used() const { return align_up(used_, 256); }
  bool dry() const { return base_ == nullptr; }

 private:
  char *base_;
  int64_t cap_, used_ = 0;
};

// Debug tap table (tests only).
struct DebugTaps {
  float *buf = nullptr;
  int64_t cap = 0, used = 0;
  struct Entry {
    std::string name;
    int64_t offset, rows;
    int cols;
  };
  std::vector<Entry> entries;
  void reset() {
    used = 0;
    entries.clear();
  }
  // copies rows x cols of a DT activation (row stride ld) as fp32
  void tap(const std::string &name, int dt, const void *x, int ld, int64_t rows, int cols, hipStream_t s);
};

// One packed convolution / linear layer.
struct ConvW {
  void *w = nullptr;       // [N][K] in the compute type (MFMA path) or fp32 (direct path)
  float *bias = nullptr;   // [N] fp32 or null
  int N = 0, K = 0, cin = 0, cin2 = 0, taps = 1;
  int kreal = 0;           // un-padded reduction length (algorithmic FLOP accounting)
  bool direct = false;     // thin layer -> conv_direct (fp32 weights)
  void *wt = nullptr;      // direct layers only: the same [N][K] matrix in the compute type (conv_thin's MFMA operand)
  void *wcb = nullptr;     // k = 3 convolutions of the deep levels: the same weights in MFMA fragment order (conv_cb.hip)
  void *wfr = nullptr;     // the [N][K] matrix in MFMA fragment order [N / 32][K / 16][64][8] (conv_gemm_rs.hip), K <= 2048
  void *wfrx = nullptr;    // fp32x engine: the split weights in MFMA fragment order (conv_gemm_rs.hip), K <= 1280
  void *wx = nullptr;      // fp32x engine: the fp32 [N][K] matrix as split fp16 operands [N][K / 32][hi 32 | lo' 32] (launch_pack_wx)
};

// ---- training step: the data-gradient GEMM of a stride-1 "same" convolution, shared by the forward entry that packs its weight images
// (capi_misc.cpp) and the backward entry that launches it (capi_train.cpp): da[row][c] = sum over (tap', n) of dy[row + tap' - (taps - 1 - pad)][n]
// * W[n][c][taps - 1 - tap'] ----
inline bool conv1d_dgrad_split_ok(bool x3, int N, int taps) { return x3 && (N % 32) == 0 && (((int64_t)taps * N) % 32) == 0; }
inline ConvGemmArgs conv1d_dgrad_args(const float *dy, int B, int L, int C, int N, int taps, int pad, float *out) {
  ConvGemmArgs a;
  a.src = dy;
  a.src_ld = N;
  a.N = C;
  a.K = taps * N;
  a.cin = N;
  a.taps = taps;
  a.stride = 1;
  a.pad = taps - 1 - pad;
  a.Lsrc = a.Lout = L;
  a.M = B * L;
  a.out = out;
  a.out_ld = C;
  a.n_store = C;
  a.solo = 1;
  a.wx_mode = X3_BF16;   // gradients span the whole fp32 exponent range: bf16 hi / lo
  return a;
}
// Name lookup + packing helper shared by the Encoder1d and VideoOnsetNet engines.
struct Packer {
  DeviceArena &arena;
  const WeightMap &wm;
  hipStream_t s;
  int dt;  // compute type of MFMA-path weights

  float *copy_f32(const std::string &name, int64_t numel);
  // conv weight (N, Cin, taps...) -> [N][taps][cin_pad] (+ per-N scale); direct -> fp32, exact Cin; K padded to kpad_to
  ConvW conv(const std::string &wname, const float *bias_dev, int N, int Cin, int taps, bool direct, int cin_pad,
             const float *nscale, int k_multiple = 1);
  ConvW linear(const std::string &pre, int N, int K, bool bias);
};

}  // namespace sf
