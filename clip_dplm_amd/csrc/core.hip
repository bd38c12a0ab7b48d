// core.hip — ABI identification for libclipk.so.
#include "common.h"

extern "C" int clipk_version(void) { return CLIPK_ABI_VERSION; }
extern "C" const char* clipk_arch(void) { return "gfx950"; }
extern "C" const char* clipk_status_string(int status) {
  switch (status) {
    case CLIPK_OK: return "ok";
    case CLIPK_ERR_BAD_ARG: return "bad argument (null / misaligned pointer or non-positive dimension)";
    case CLIPK_ERR_UNSUPPORTED: return "unsupported shape for the gfx950 kernels";
    case CLIPK_ERR_LAUNCH: return "HIP launch error";
    default: return "unknown clipk status";
  }
}

// ---- kernel-selection options: explicit API instead of environment variables (a stray variable must never change
// which kernel a product launch takes).  Relaxed atomics: an option is a plain int read at launch time.
#include <atomic>
#include <string.h>
namespace {
struct OptDef { const char* name; int dflt; };
const OptDef kOpts[OPT_COUNT] = {
    {"gemm_kernel", -1}, {"gemm_epi_generic", 0}, {"gemm_bm", 0}, {"gemm_stages", 1}, {"gemm_nwg", 0},
    {"gemm_stagger", 0}, {"epi_nt", 0}, {"wgrad_kernel", -1}, {"attn_whole_fwd", -1}, {"attn_fused_bwd", -1},
    {"attn_fused_waves", 0}, {"wgrad_splits", 0}, {"simce_kernel", -1}, {"gemm_abl", 0},
    {"attn_row_stores", 0}, {"gemm_f32_splits", 0},
};
std::atomic<int> g_opts[OPT_COUNT];
std::atomic<bool> g_opts_init{false};
void opts_init() {
  if (!g_opts_init.load(std::memory_order_acquire)) {
    for (int i = 0; i < OPT_COUNT; ++i) g_opts[i].store(kOpts[i].dflt, std::memory_order_relaxed);
    g_opts_init.store(true, std::memory_order_release);
  }
}
int opt_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOpts[i].name) == 0) return i;
  return -1;
}
}  // namespace

namespace { std::atomic<const unsigned*> g_drop_epoch{nullptr}; }
const unsigned* clipk_drop_epoch() { return g_drop_epoch.load(std::memory_order_relaxed); }
extern "C" int clipk_set_dropout_epoch(const uint32_t* epoch_dev) {
  g_drop_epoch.store(reinterpret_cast<const unsigned*>(epoch_dev), std::memory_order_relaxed);
  return CLIPK_OK;
}

int clipk_opt_get(int which) {
  opts_init();
  return g_opts[which].load(std::memory_order_relaxed);
}
extern "C" int clipk_set_option(const char* name, int value) {
  opts_init();
  const int i = opt_index(name);
  if (i < 0) return CLIPK_ERR_BAD_ARG;
#ifndef CLIPK_EXPERIMENTS
  if (i == OPT_GEMM_ABL && value != 0) return CLIPK_ERR_UNSUPPORTED;   // result-changing: not in product builds
#endif
  g_opts[i].store(value, std::memory_order_relaxed);
  return CLIPK_OK;
}
extern "C" int clipk_get_option(const char* name, int* value) {
  opts_init();
  const int i = opt_index(name);
  if (i < 0 || !value) return CLIPK_ERR_BAD_ARG;
  *value = g_opts[i].load(std::memory_order_relaxed);
  return CLIPK_OK;
}
extern "C" int clipk_reset_options(void) {
  g_opts_init.store(false, std::memory_order_release);
  opts_init();
  return CLIPK_OK;
}
