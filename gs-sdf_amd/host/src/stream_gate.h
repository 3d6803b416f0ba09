// stream_gate.h — "the tensor produced on another stream is complete": an event + an armed flag (internal to libgsdf_torch.so)
#pragma once
#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

namespace gsdf_extras {
struct StreamGate {
  at::cuda::CUDAEvent event;
  bool armed = false;
  const void *payload = nullptr;   // the tensor the event stands for, when it was recorded ahead of the autograd pass that hands it over (joint_sdf_loss_analytic,
                                   // first_order_in_forward): a consumer that receives another buffer (the engine copied it) must wait for that one instead
  void record_here() {   // on the current stream
    event.record(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
    armed = true;
  }
};
}  // namespace gsdf_extras
