// shared by the translation units of libadvengine.so (not installed, not part of the ABI)
#ifndef ADV_INTERNAL_H
#define ADV_INTERNAL_H
#include <hip/hip_runtime.h>

#include "advengine.h"

// records the hipError_t behind ADV_ELAUNCH for adv_last_hip_error() (thread-local, defined in advengine.hip)
void adv_internal_set_last_hip_error(int e);

static inline int adv_internal_finish_launch() {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    adv_internal_set_last_hip_error(static_cast<int>(e));
    return ADV_ELAUNCH;
  }
  return ADV_OK;
}
#endif
