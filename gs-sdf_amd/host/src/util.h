// util.h — glue between libtorch tensors and the C ABI (include/gsdf_hip.h).
#pragma once
#include <c10/hip/HIPStream.h>

#include <chrono>
#include <map>
#include <torch/torch.h>

#include "gsdf_hip.h"

namespace gsdf_host {

inline gsdf_stream_t cur_stream() { return (gsdf_stream_t)c10::hip::getCurrentHIPStream().stream(); }

inline void check(int rc, const char *what) { TORCH_CHECK(rc == GSDF_OK, what, " failed (", rc, "): ", gsdf_last_error()); }

// contiguous fp32 device tensor (copies only when the caller handed a strided view)
inline torch::Tensor f32c(const torch::Tensor &t, const char *name) {
  TORCH_CHECK(t.defined(), name, ": undefined tensor");
  TORCH_CHECK(t.is_cuda(), name, ": expected a device tensor (the HIP path has no CPU fallback)");
  TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, ": expected float32");
  return t.contiguous();
}
inline const float *fp(const torch::Tensor &t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }
inline float *fpm(torch::Tensor &t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }

inline torch::Tensor empty_like_opts(const torch::Tensor &ref, at::IntArrayRef shape, torch::ScalarType dt) {
  return torch::empty(shape, ref.options().dtype(dt).requires_grad(false));
}
inline torch::Tensor zeros_like_opts(const torch::Tensor &ref, at::IntArrayRef shape, torch::ScalarType dt) {
  return torch::zeros(shape, ref.options().dtype(dt).requires_grad(false));
}
inline int64_t read_i64(const torch::Tensor &dev_scalar) { return dev_scalar.item<int64_t>(); }  // the one host sync

// Counts the host waits for between two launches, without a copy kernel and a stream synchronisation (include/gsdf_hip.h:
// gsdf_host_words_alloc): words of pinned, device-mapped host memory.  arm(i) before the launch that writes word i, pass dev(i) as the
// operator's count output, queue whatever does not depend on the count, then wait(i).  The poll falls back to a stream synchronisation
// after ~2 ms (a store that is only made visible by the end of the queue's work still gets read).
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#endif
}

class HostWords {
 public:
  explicit HostWords(int n) : n_(n) { check(gsdf_host_words_alloc(n, &host_, &dev_), "host_words_alloc"); }
  ~HostWords() { (void)gsdf_host_words_free(host_); }
  HostWords(const HostWords &) = delete;
  HostWords &operator=(const HostWords &) = delete;
  int64_t *dev(int i) const { return dev_ + i; }
  void arm(int i) { __atomic_store_n(host_ + i, kArmed, __ATOMIC_RELEASE); }
  int64_t wait(int i) const {
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t spins = 1;; ++spins) {
      const int64_t v = __atomic_load_n(host_ + i, __ATOMIC_ACQUIRE);
      if (v != kArmed) return v;
      if ((spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
        c10::hip::getCurrentHIPStream().synchronize();
        const int64_t w = __atomic_load_n(host_ + i, __ATOMIC_ACQUIRE);
        TORCH_CHECK(w != kArmed, "HostWords: the count was never written");
        return w;
      }
      cpu_relax();
    }
  }
  static bool enabled() {
    static const bool on = [] { const char *e = getenv("GSDF_HOST_COUNTS"); return !(e && e[0] == '0'); }();   // GSDF_HOST_COUNTS=0: device scalars + item()
    return on;
  }

 private:
  static constexpr int64_t kArmed = INT64_MIN;
  int n_;
  int64_t *host_ = nullptr, *dev_ = nullptr;
};

// One count through the calling thread's own word (the operator-level functions: cull -> M -> fill, count -> I -> encode): `launch` receives
// the pointer to pass as the operator's count output and returns after queueing it.
// `upper`: the largest count the caller can make sense of (it is about to be used as an allocation size): anything outside [0, upper] means the
// word was written by someone else or not at all, and is refused instead of allocated.
template <class F>
inline int64_t count_via_host_word(const torch::Tensor &like, F &&launch, int64_t upper = INT64_MAX) {
  int64_t v;
  if (!HostWords::enabled()) {
    torch::Tensor n = torch::empty({1}, like.options().dtype(torch::kInt64).requires_grad(false));
    launch(n.data_ptr<int64_t>());
    v = read_i64(n);
  } else {
    // one word per (thread, device): two threads never arm the same word, and a thread that moves between devices gets a word per device
    // (never freed: a thread's exit may come after the HIP runtime's)
    static thread_local std::map<int, HostWords *> words;
    const int dev = like.is_cuda() ? (int)like.get_device() : (int)c10::hip::current_device();
    HostWords *&w = words[dev];
    if (w == nullptr) w = new HostWords(1);
    w->arm(0);
    launch(w->dev(0));
    v = w->wait(0);
  }
  TORCH_CHECK(v >= 0 && v <= upper, "count_via_host_word: implausible count ", v, " (expected 0..", upper, ")");
  return v;
}

}  // namespace gsdf_host
