// Library identification + device probe for libmvgformer_hip (host only).
#include <string.h>

#include "common.h"

extern "C" {

const char* mvg_version(void) { return "mvgformer_amd 0.1 (gfx950)"; }

int mvg_device_info(char* arch_out, int arch_len, int* cu_count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MVG_E_NOGPU;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MVG_E_NOGPU;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return MVG_E_NOGPU;
  if (arch_out && arch_len > 0) {
    strncpy(arch_out, prop.gcnArchName, arch_len - 1);
    arch_out[arch_len - 1] = 0;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  return 0;
}

}  // extern "C"
