// Host-side error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace mn {
extern thread_local std::string g_last_error;
inline int fail(const std::string& msg) {
  g_last_error = msg;
  return 1;
}
// HIP keeps a per-thread "last error" that other users of the runtime in this process (PyTorch's
// allocator / stream queries) may leave set; drop it on entry so check_launch only reports ours.
inline void begin_call() { (void)hipGetLastError(); }
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return 2;
  }
  return 0;
}
}  // namespace mn
