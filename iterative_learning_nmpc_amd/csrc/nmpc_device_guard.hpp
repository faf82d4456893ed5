// nmpc_device_guard.hpp -- every C-ABI entry point runs on the device its handle (or its tensors) live on and
// leaves the caller's current device as it found it (torch keeps its own notion of the current device; an entry
// point that called hipSetDevice and returned would change it behind torch's back, and one that launched on the
// current device with a stream of another device would fail).
#pragma once
#include <hip/hip_runtime.h>

namespace nmpc {

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && dev >= 0 && dev != prev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// device a pointer was allocated on, -1 if it is not a device pointer (handle-less entry points take their device
// from their first tensor)
inline int device_of(const void* p) {
    hipPointerAttribute_t attr;
    if (p && hipPointerGetAttributes(&attr, p) == hipSuccess) return attr.device;
    (void)hipGetLastError();
    return -1;
}

}  // namespace nmpc
