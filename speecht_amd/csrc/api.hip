// libspeecht_hip.so: version, thread-local error text, launch trace and tuning overrides (diagnostics).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>
#include <string.h>

#include "st_common.h"

namespace st {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- launch trace: which kernel variant / split policy a call took (tests assert on it), and -- in timed mode
// (st_trace_begin_timed) -- how long each traced launch took on its stream inside the real launch sequence: a pair
// of start / stop events handed to hipExtLaunchKernel (LaunchTimer, st_common.h), resolved when the trace is collected.
static std::atomic<int> g_trace_on{0};      // 0 off, 1 names, 2 names + per-launch start / stop events
static std::mutex g_trace_mu;
static std::vector<std::string> g_lines;
constexpr int kMaxTimed = 1 << 14;
struct TimedLaunch { int line; hipEvent_t e0, e1; };
static std::vector<TimedLaunch> g_timed;
static thread_local int g_last_line = -1;
static std::string g_timed_filter;                 // timed mode: only lines containing this text get events ("" = all)
bool trace_on() { return g_trace_on.load(std::memory_order_relaxed) != 0; }
void trace(const char* fmt, ...) {
  if (!trace_on()) return;
  char line[320];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(line, sizeof(line), fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> lock(g_trace_mu);
  if (g_lines.size() < (1u << 16)) {
    g_last_line = (int)g_lines.size();
    g_lines.emplace_back(line);
  } else {
    g_last_line = -1;
  }
}
LaunchTimer::LaunchTimer(hipStream_t) : start_(nullptr), stop_(nullptr) {
  // the line this launch belongs to is CONSUMED here: a later launch on this thread that traced nothing of its own must not
  // attach its events to it (and trace_begin forgets it: an index into the previous collection's lines)
  const int line = g_last_line;
  g_last_line = -1;
  if (g_trace_on.load(std::memory_order_relaxed) != 2 || line < 0) return;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess) return;
  if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return; }
  std::lock_guard<std::mutex> lock(g_trace_mu);
  if ((int)g_timed.size() >= kMaxTimed || line >= (int)g_lines.size() ||
      (!g_timed_filter.empty() && g_lines[line].find(g_timed_filter) == std::string::npos)) {
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return;
  }
  g_timed.push_back(TimedLaunch{line, e0, e1});
  start_ = e0;
  stop_ = e1;
}

// ---- tuning overrides: 0 = the library's policy.  Set explicitly by perf scripts through st_set_tuning;
// the launch path reads plain ints (no environment look-ups).
static const char* const kTuneNames[TUNE_COUNT] = {"gemm_tile", "gemm_splits", "fwd_splits", "xcd_gm", "no_fast",
                                                   "bf16_tile", "bf16_wgrad_splits", "bf16_sched", "streamk", "transform_wgs", "bf16_wgrad_target", "streamk_slots", "no_fused_transforms",
                                                   "streamk_test_drop", "bf16_lag_copies", "bf16_wgrad_ring", "bf16_taps_panel", "bf16_wgrad_bias_pass", "bf16_wgrad_plain_order", "filters_idft_valu", "no_row_split", "no_g3", "g3_tile"};
static std::atomic<int> g_tune[TUNE_COUNT];
int tuning(int key) { return g_tune[key].load(std::memory_order_relaxed); }
}  // namespace st

extern "C" {
int st_version(void) { return 200; }
const char* st_last_error(void) { return st::g_err; }

static int trace_begin(int mode) {
  std::lock_guard<std::mutex> lock(st::g_trace_mu);
  st::g_lines.clear();
  for (auto& t : st::g_timed) { hipEventDestroy(t.e0); hipEventDestroy(t.e1); }
  st::g_timed.clear();
  st::g_last_line = -1;                 // (this thread's; other threads consume theirs at their next LaunchTimer)
  st::g_trace_on.store(mode);
  return ST_OK;
}
int st_trace_begin(void) { return trace_begin(1); }
int st_trace_begin_timed(void) { return trace_begin(2); }
int st_trace_timed_filter(const char* text) {
  std::lock_guard<std::mutex> lock(st::g_trace_mu);
  st::g_timed_filter = text ? text : "";
  return ST_OK;
}

size_t st_trace_end(char* host_buf, size_t capacity) {
  st::g_trace_on.store(0);
  std::lock_guard<std::mutex> lock(st::g_trace_mu);
  // timed mode: wait for each launch's stop event and append the kernel's duration to its line (first collection only)
  for (auto& t : st::g_timed) {
    float ms = -1.f;
    if (hipEventSynchronize(t.e1) != hipSuccess || hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) ms = -1.f;
    char tail[48];
    snprintf(tail, sizeof(tail), " ms=%.5f", ms);
    if (t.line >= 0 && t.line < (int)st::g_lines.size()) st::g_lines[t.line] += tail;
    hipEventDestroy(t.e0);
    hipEventDestroy(t.e1);
  }
  st::g_timed.clear();
  size_t need = 1;
  for (const auto& l : st::g_lines) need += l.size() + 1;
  if (host_buf && capacity > 0) {
    size_t at = 0;
    for (const auto& l : st::g_lines) {
      if (at + l.size() + 1 >= capacity) break;
      memcpy(host_buf + at, l.data(), l.size());
      at += l.size();
      host_buf[at++] = '\n';
    }
    host_buf[at] = 0;
  }
  return need;
}

// CRC-32C (Castagnoli) on the HOST, slicing-by-8: the checksum of TensorFlow's checkpoint bundles
// (speecht_amd/tf_checkpoint.py reads / writes the reference's speechT.ckpt-N files, speech_model.py:122,251-260).
uint32_t st_host_crc32c(const void* host_data, size_t n, uint32_t crc) {
  static uint32_t table[8][256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
  });
  const unsigned char* p = reinterpret_cast<const unsigned char*>(host_data);
  crc = ~crc;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= crc;
    crc = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
          table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) crc = (crc >> 8) ^ table[0][(crc ^ *p++) & 0xFF];
  return ~crc;
}

int st_set_tuning(const char* name, int value) {
  ST_REQUIRE(name, "st_set_tuning: null name");
  for (int i = 0; i < st::TUNE_COUNT; ++i)
    if (!strcmp(name, st::kTuneNames[i])) {
      st::g_tune[i].store(value);
      return ST_OK;
    }
  st::set_error("st_set_tuning: unknown knob '%s'", name);
  return ST_EINVAL;
}
}
