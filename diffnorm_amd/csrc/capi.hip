// Error plumbing and misc entry points of libdiffnorm_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void dn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dn_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------ run-time options
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

namespace {
struct OptDef { const char* name; const char* env; const char* env_old; };
// name (dn_set_option) <- environment variable that seeds it at first use
const OptDef kOpts[dn::OPT_COUNT] = {
    {"taps_inner", "DN_TAPS_INNER", "DN_FAT_TAPS_INNER"},  // K order of a causal conv's taps: 0 term-outer everywhere, 1 tap-inner on the 256-row tiles (default), 2 routed by shape
    {"fuse_norm", "DN_FUSE_NORM", nullptr},                // 1: residual-closing contractions on the whole-row tile that also emits the next RMSNorm
    {"no_split_norm", "DN_NO_SPLIT_NORM", nullptr},        // 1: stand-alone RMSNorm kernels instead of the split norm
    {"kblock", "DN_KBLOCK", nullptr},                      // 0 never / 1 always K-blocked operands (unset: where the consuming tile gains)
    {"wgrad_stream", "DN_WGRAD_STREAM", nullptr},          // 0: weight gradients on the caller's stream
    {"wgrad_tn", "DN_WGRAD_TN", nullptr},                  // 0: weight gradients from transposed operand copies
    {"wgrad_groups", "DN_WGRAD_GROUPS", nullptr},          // 0: one weight-gradient launch per WaveNet block
    {"qkv_192", "DN_QKV_192", nullptr},                    // 1: the q/kv projection (N = 1536 = 8 x 192) on the 256 x 192 tile (A/B timing)
    {"mid2", "DN_MID2", nullptr},                          // bit 0 / bit 1: the q/kv / GEGLU projection on the 256 x 128 two-workgroups-per-CU tile
    {"wgrad_stages", "DN_WGRAD_STAGES", nullptr},          // 5: the weight-gradient kernel on a 160 KiB ring of five stages (default 4 stages = 128 KiB)
    {"tile_192", "DN_TILE_192", nullptr},                  // 0: never choose the 256 x 192 tile by score (default: where a lone launch fills the chip better on it)
    {"fused_geglu", "DN_FUSED_GEGLU", nullptr},            // 0: training forward runs the GEGLU as a pass over the projection's output (default: in its epilogue)
    {"attn_waves8", "DN_ATTN_WAVES8", nullptr},            // 0: attention (2-byte modes) on four waves of 32 queries per workgroup (default: eight waves of 16)
    {"cond_stream", "DN_COND_STREAM", nullptr},            // 0: training: the data gradient of the conditioning projection from a transposed copy (default: the master matrix streamed as it lies)
    {"wgrad_k192", "DN_WGRAD_K192", nullptr},              // 1: the weight-gradient kernel on 192 k-columns per tile where that fills the chip better (default off: faster alone, slower beside the data-gradient chain)
    {"wgrad_prio", "DN_WGRAD_PRIO", nullptr},              // 1: the weight-gradient stream at the lowest priority (read when the stream is created; measured level, default off)
};
std::atomic<int> g_opt[dn::OPT_COUNT];
std::once_flag g_opt_once;
int env_value(const OptDef& d) {
  const char* e = getenv(d.env);
  if (!e && d.env_old) e = getenv(d.env_old);
  return e ? atoi(e) : dn::DN_OPT_UNSET;
}
void init_options() {
  for (int i = 0; i < dn::OPT_COUNT; ++i) g_opt[i].store(env_value(kOpts[i]), std::memory_order_relaxed);
}
int find_option(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < dn::OPT_COUNT; ++i)
    if (strcmp(name, kOpts[i].name) == 0) return i;
  return -1;
}
}  // namespace

int dn::option(dn::Opt o) {
  std::call_once(g_opt_once, init_options);
  return g_opt[o].load(std::memory_order_relaxed);
}

extern "C" int dn_set_option(const char* name, int32_t value) {
  const int i = find_option(name);
  DN_CHECK_ARG(i >= 0, "dn_set_option: unknown option '%s'", name ? name : "(null)");
  std::call_once(g_opt_once, init_options);
  g_opt[i].store(value == DN_OPTION_DEFAULT ? env_value(kOpts[i]) : value, std::memory_order_relaxed);
  return DN_OK;
}

extern "C" int dn_get_option(const char* name, int32_t* value, int32_t* is_set) {
  const int i = find_option(name);
  DN_CHECK_ARG(i >= 0 && value, "dn_get_option: unknown option '%s'", name ? name : "(null)");
  const int v = dn::option(static_cast<dn::Opt>(i));
  *value = v == dn::DN_OPT_UNSET ? 0 : v;
  if (is_set) *is_set = v != dn::DN_OPT_UNSET;
  return DN_OK;
}
extern "C" int dn_version(void) { return 100; }
