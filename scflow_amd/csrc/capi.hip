// Library-level entry points of libscflow_hip.so.
#include "scf_common.h"

extern "C" int scf_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* scf_error_string(int code) {
  switch (code) {
    case SCF_OK: return "ok";
    case SCF_EINVAL: return "invalid argument";
    case SCF_EUNSUPPORTED: return "unsupported shape or configuration";
    case SCF_ELAUNCH: return "HIP kernel launch failed";
    case SCF_ENODEVICE: return "no HIP device";
    default: return "unknown error";
  }
}

extern "C" int scf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return SCF_ENODEVICE;
  return n;
}
