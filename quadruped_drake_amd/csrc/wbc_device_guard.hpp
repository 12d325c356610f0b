// wbc_device_guard.hpp -- every entry point of the C ABI that touches the GPU runs on ITS handle's device and leaves the calling
// thread's current HIP device as it found it (include/wbc.h, "Current device").  A process that holds handles on several GPUs --
// or shares the thread with torch, whose current device is this same thread-local setting -- is not moved by a wbc_* call.
#pragma once
#include <hip/hip_runtime.h>

namespace wbc {

struct DeviceGuard {
  int prev = -1;
  bool moved = false;
  hipError_t err;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      moved = (err == hipSuccess);
    }
  }
  ~DeviceGuard() {
    if (moved) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

}  // namespace wbc

// `FAIL` is the translation unit's (what, hipError_t) -> int error reporter
#define WBC_ON_DEVICE(dev, FAIL)              \
  wbc::DeviceGuard device_guard_(dev);        \
  if (device_guard_.err != hipSuccess) return FAIL("hipSetDevice", device_guard_.err)
