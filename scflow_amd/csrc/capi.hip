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


// ---- launch-bound timers (measurement aid, see scf_common.h) ----
ScfTimer*& scf_armed_timer() {
  static thread_local ScfTimer* slot = nullptr;
  return slot;
}

extern "C" int scf_timer_create(scf_timer_t* out) {
  if (!out) return SCF_EINVAL;
  ScfTimer* t = new ScfTimer;
  if (hipEventCreate(&t->start) != hipSuccess) { delete t; return SCF_ELAUNCH; }
  if (hipEventCreate(&t->stop) != hipSuccess) { (void)hipEventDestroy(t->start); delete t; return SCF_ELAUNCH; }
  *out = t;
  return SCF_OK;
}

extern "C" int scf_timer_destroy(scf_timer_t tm) {
  ScfTimer* t = static_cast<ScfTimer*>(tm);
  if (!t) return SCF_EINVAL;
  if (scf_armed_timer() == t) scf_armed_timer() = nullptr;
  (void)hipEventDestroy(t->start);
  (void)hipEventDestroy(t->stop);
  delete t;
  return SCF_OK;
}

extern "C" int scf_timer_arm(scf_timer_t tm) {
  scf_armed_timer() = static_cast<ScfTimer*>(tm);       // nullptr disarms
  return SCF_OK;
}

extern "C" int scf_timer_elapsed_us(scf_timer_t tm, float* us) {
  ScfTimer* t = static_cast<ScfTimer*>(tm);
  if (!t || !us) return SCF_EINVAL;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, t->start, t->stop) != hipSuccess) return SCF_ELAUNCH;
  *us = ms * 1e3f;
  return SCF_OK;
}
