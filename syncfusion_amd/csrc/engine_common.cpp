#include "engine_common.h"

#include <cstring>

namespace sf {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
const char *get_error() { return g_err.c_str(); }

void fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  throw EngineError{code};
}

WeightMap::WeightMap(const sf_tensor *w, int n) {
  for (int i = 0; i < n; ++i) {
    if (!w[i].name) continue;
    map_[w[i].name] = {w[i].data, w[i].numel};
  }
}

const float *WeightMap::get(const std::string &name, int64_t numel) const {
  auto it = map_.find(name);
  if (it == map_.end()) fail(SF_ERR_MISSING_WEIGHT, "missing parameter '%s'", name.c_str());
  if (it->second.second != numel)
    fail(SF_ERR_MISSING_WEIGHT, "parameter '%s' has %lld elements, expected %lld", name.c_str(), (long long)it->second.second,
         (long long)numel);
  if (!it->second.first) fail(SF_ERR_MISSING_WEIGHT, "parameter '%s' has a null pointer", name.c_str());
  return static_cast<const float *>(it->second.first);
}

DeviceArena::~DeviceArena() {
  for (void *p : chunks_) (void)hipFree(p);
}

void *DeviceArena::alloc(int64_t bytes) {
  bytes = align_up(bytes > 0 ? bytes : 1, 256);
  if (bytes > left_) {
    int64_t chunk = bytes > (64ll << 20) ? bytes : (64ll << 20);
    void *p = nullptr;
    SF_HIP(hipMalloc(&p, (size_t)chunk));
    chunks_.push_back(p);
    cur_ = static_cast<char *>(p);
    left_ = chunk;
  }
  void *r = cur_;
  cur_ += bytes;
  left_ -= bytes;
  total_ += bytes;
  return r;
}

float *Packer::copy_f32(const std::string &name, int64_t numel) {
  const float *src = wm.get(name, numel);
  float *dst = arena.alloc_n<float>(numel);
  SF_HIP(hipMemcpyAsync(dst, src, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
  return dst;
}

ConvW Packer::conv(const std::string &wname, const float *bias_dev, int N, int Cin, int taps, bool direct, int cin_pad,
                   const float *nscale, int k_multiple) {
  ConvW c;
  c.N = N;
  c.taps = taps;
  c.direct = direct;
  c.cin = direct ? Cin : cin_pad;
  c.cin2 = 0;
  const int kreal = taps * c.cin;
  c.K = pad_to(kreal, k_multiple);
  const float *w = wm.get(wname, (int64_t)N * Cin * taps);
  const int wdt = direct ? F32 : dt;
  c.w = arena.alloc((int64_t)N * c.K * dsize(wdt));
  if (c.K != kreal) SF_HIP(hipMemsetAsync(c.w, 0, (int64_t)N * c.K * dsize(wdt), s));
  SF_HIP(launch_pack_conv(wdt, w, N, Cin, 0, Cin, taps, c.cin, nscale, c.w, c.K, 0, s));
  c.bias = const_cast<float *>(bias_dev);
  return c;
}

ConvW Packer::linear(const std::string &pre, int N, int K, bool bias) {
  ConvW c;
  c.N = N;
  c.K = K;
  c.cin = K;
  c.taps = 1;
  const float *w = wm.get(pre + ".weight", (int64_t)N * K);
  c.w = arena.alloc((int64_t)N * K * dsize(dt));
  SF_HIP(launch_pack_rows(dt, w, N, K, K, nullptr, c.w, K, s));
  if (bias) c.bias = copy_f32(pre + ".bias", N);
  return c;
}

void DebugTaps::tap(const std::string &name, int dt, const void *x, int ld, int64_t rows, int cols, hipStream_t s) {
  if (!buf) return;
  int64_t n = rows * cols;
  if (used + n > cap) fail(SF_ERR_WORKSPACE, "debug buffer too small at tap '%s'", name.c_str());
  SF_HIP(launch_to_f32(dt, x, ld, rows, cols, buf + used, s));
  entries.push_back({name, used, rows, cols});
  used += n;
}

}  // namespace sf
