// Library-level entry points of libscflow_hip.so.
#include "scf_common.h"

extern "C" int scf_version(void) { return SCF_VERSION; }

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


// ---- host-side weight packers (plain CPU loops; layouts documented in include/scflow_hip.h) ----
extern "C" int64_t scf_pack_conv_weight_size(int Cout, int Cin, int KH, int KW, int KC) {
  if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || (KC != 2 && KC != 8 && KC != 32)) return SCF_EINVAL;
  const int64_t nchunk = (Cin + KC - 1) / KC, mld = (Cout + 31) / 32 * 32;
  return nchunk * KH * KW * KC * mld;
}

extern "C" int scf_pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, int KC, float* out) {
  const int64_t total = scf_pack_conv_weight_size(Cout, Cin, KH, KW, KC);
  if (!w || !out || total < 0) return SCF_EINVAL;
  const int T = KH * KW;
  const int64_t mld = (Cout + 31) / 32 * 32;
  for (int64_t i = 0; i < total; ++i) out[i] = 0.f;
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < T; ++t) {
        const int64_t chunk = ci / KC, cl = ci % KC;
        out[((chunk * T + t) * KC + cl) * mld + co] = w[((int64_t)co * Cin + ci) * T + t];
      }
  return SCF_OK;
}

extern "C" int64_t scf_pack_conv_weight_a4_size(int Cout, int Cin, int KH, int KW, int groups) {
  if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || (groups != 1 && groups != 2 && groups != 4)) return SCF_EINVAL;
  const int64_t kc = 8 * groups, nchunk = (Cin + kc - 1) / kc, mld = (Cout + 31) / 32 * 32;
  return nchunk * KH * KW * groups * 2 * mld * 4;
}

extern "C" int scf_pack_conv_weight_a4(const float* w, int Cout, int Cin, int KH, int KW, int groups, float* out) {
  const int64_t total = scf_pack_conv_weight_a4_size(Cout, Cin, KH, KW, groups);
  if (!w || !out || total < 0) return SCF_EINVAL;
  const int T = KH * KW, kc = 8 * groups;
  const int64_t mld = (Cout + 31) / 32 * 32;
  for (int64_t i = 0; i < total; ++i) out[i] = 0.f;
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < T; ++t) {
        const int64_t chunk = ci / kc, r = ci % kc, g = r / 8, r8 = r % 8, s = r8 / 2, h = r8 % 2;
        out[((((chunk * T + t) * groups + g) * 2 + h) * mld + co) * 4 + s] = w[((int64_t)co * Cin + ci) * T + t];
      }
  return SCF_OK;
}

extern "C" int64_t scf_pack_conv_weight_taps_size(int Cout, int Cin, int KH, int KW) {
  if (Cout <= 0 || Cin <= 0 || Cin > 4 || KH <= 0 || KW <= 0) return SCF_EINVAL;
  const int64_t kp = ((int64_t)Cin * KH * KW + 7) / 8 * 8, mld = (Cout + 31) / 32 * 32;
  return kp * mld;
}

extern "C" int scf_pack_conv_weight_taps(const float* w, int Cout, int Cin, int KH, int KW, float* out) {
  const int64_t total = scf_pack_conv_weight_taps_size(Cout, Cin, KH, KW);
  if (!w || !out || total < 0) return SCF_EINVAL;
  const int T = KH * KW;
  const int64_t mld = (Cout + 31) / 32 * 32;
  for (int64_t i = 0; i < total; ++i) out[i] = 0.f;
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < T; ++t) out[((int64_t)ci * T + t) * mld + co] = w[((int64_t)co * Cin + ci) * T + t];
  return SCF_OK;
}

// ---- launch-bound timers (measurement aid, see scf_common.h) ----
ScfTimer*& scf_armed_timer() {
  static thread_local ScfTimer* slot = nullptr;
  return slot;
}

extern "C" int scf_timer_create(scf_timer_t* out) {
  if (!out) return SCF_EINVAL;
  ScfTimer* t = new ScfTimer;
  // hipEventDisableSystemFence: a default event bound to a dispatch folds a system-scope release into that dispatch's end
  // -- the timed kernel itself gets longer (tools/lab/event_overhead.hip, profiles/r5_event_overhead.txt: first wave begin
  // -> last wave end 19.56 us with default events, 18.59 us with these; the kernel trace of the step showed 19.78 -> 20.16 us
  // when timers were armed, notebook R5.5).  Nothing on the host reads what the timed kernels write: no fence is needed.
  if (hipEventCreateWithFlags(&t->start, hipEventDisableSystemFence) != hipSuccess) { delete t; return SCF_ELAUNCH; }
  if (hipEventCreateWithFlags(&t->stop, hipEventDisableSystemFence) != hipSuccess) { (void)hipEventDestroy(t->start); delete t; return SCF_ELAUNCH; }
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
