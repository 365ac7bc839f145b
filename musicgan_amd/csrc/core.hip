// Error reporting and version of libmusicgan_hip.so.
#include "mg_common.h"

namespace {
thread_local char g_err[512] = "";
}

void mg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mg_cu_count() {
  static int cus[MG_MAX_DEVICES] = {};
  const int d = mg_current_device();
  if (cus[d] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
    cus[d] = v;
  }
  return cus[d];
}

extern "C" int mg_version(void) { return 106; }  // 106: + mg_stem_pair, mg_stem_pair_gx, mg_head_pair, mg_blend_up_bwd, mg_gp_apply, MG_C1_ACCUM; 105: + mg_pt_write_samples (round 5); 104: + mg_smallnet, mg_conv3x3_small, mg_stft_1024_pcm, mg_stft_generic, mg_pcm_to_mono, mg_crc32_f64 (round 4)
extern "C" const char* mg_last_error(void) { return g_err; }
