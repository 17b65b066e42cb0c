// fsk_host.h -- host-side helpers shared by the C-ABI translation units of libfskhip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fskhip.h"
#include "../../include/fskhip_next.h"

namespace fsk {
// records the thread-local message fskhip_last_error() returns and hands `code` back
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int engine_device(const fskhip_engine *e);
}  // namespace fsk

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) return fsk::fail(FSKHIP_E_HIP, "%s: %s", #expr, hipGetErrorString(_e));      \
  } while (0)
