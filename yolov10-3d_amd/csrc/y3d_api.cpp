// Host-side plumbing of liby3d_hip.so: error string, ABI version, device query.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/y3d.h"

static thread_local char g_err[512] = "";

void y3d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* y3d_last_error(void) { return g_err; }

int y3d_abi_version(void) { return 1; }

int y3d_device_info(char* name, int name_len, int* compute_units, int* lds_bytes, int* clock_khz) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { y3d_set_error("hipGetDevice: %s", hipGetErrorString(e)); return Y3D_ERR_HIP; }
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) { y3d_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return Y3D_ERR_HIP; }
  if (name && name_len > 0) { strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
  if (clock_khz) *clock_khz = p.clockRate;
  return Y3D_OK;
}

}  // extern "C"
