// stream_gate.h — "the tensor produced on another stream is complete": an event + an armed flag (internal to libgsdf_torch.so)
#pragma once
#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

namespace gsdf_extras {
struct StreamGate {
  at::cuda::CUDAEvent event;
  bool armed = false;
  void record_here() {   // on the current stream
    event.record(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
    armed = true;
  }
};
}  // namespace gsdf_extras
