// Library identification + error plumbing.
#include "dz_common.h"

int g_dz_last_hip_error = 0;

extern "C" const char* dz_version(void) { return "dqnzoo_hip 0.1 (gfx950)"; }
extern "C" int dz_last_hip_error(void) { return g_dz_last_hip_error; }
extern "C" const char* dz_built_arch(void) { return "gfx950"; }

// One-launch decision kernels (dz_act_one.h): polling rounds per seam before a workgroup gives
// up.  200 000 rounds of >= 64 cycles are >= 5 ms: three orders of magnitude above a decision.
int g_dz_act_spin_limit = 200000;
extern "C" int dz_act_debug_spin_limit(int limit) {
  const int old = g_dz_act_spin_limit;
  if (limit >= 0) g_dz_act_spin_limit = limit;
  return old;
}

extern "C" int dz_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(dz_field_t);
    case 1: return (int)sizeof(dz_prio_sample_args_t);
    case 2: return (int)sizeof(dz_rainbow_layout_t);
    case 3: return (int)sizeof(dz_rainbow_args_t);
    case 4: return (int)sizeof(dz_dense_layout_t);
    case 5: return (int)sizeof(dz_dense_args_t);
    case 6: return (int)sizeof(dz_iqn_layout_t);
    case 7: return (int)sizeof(dz_iqn_args_t);
    case 8: return (int)sizeof(dz_insert_field_t);
    case 9: return (int)sizeof(dz_next_sample_t);
    case 10: return (int)sizeof(dz_replay_insert_args_t);
    case 11: return (int)sizeof(dz_rainbow_act_args_t);
    default: return -1;
  }
}

// ---- event profiler ----------------------------------------------------------
#include <string.h>
#include <time.h>
bool g_dz_prof_on = false;
namespace {
constexpr int kMaxMarks = 96;
hipEvent_t g_ev[kMaxMarks + 1];
const char* g_names[kMaxMarks];
int g_nmarks = 0;
bool g_ev_created = false;
// dz_prof_enable(2): the marks take the HOST's clock instead of recording events -- what the
// enqueue of each launch costs the calling thread (tools/window_events.py)
bool g_host_clock = false;
long long g_host_ns[kMaxMarks + 1];
long long host_ns() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
}  // namespace

void dz_prof_begin(hipStream_t s) {
  g_nmarks = 0;
  if (g_host_clock) { g_host_ns[0] = host_ns(); return; }
  (void)hipEventRecord(g_ev[0], s);
}
void dz_prof_mark(hipStream_t s, const char* name) {
  if (g_nmarks >= kMaxMarks) return;
  g_names[g_nmarks] = name;
  if (g_host_clock) g_host_ns[g_nmarks + 1] = host_ns();
  else (void)hipEventRecord(g_ev[g_nmarks + 1], s);
  ++g_nmarks;
}
// Replay entry points (0 sample, 1 gather, 2 priority update): an event pair
// around the launch, taken inside the C function so that no host-side work of
// the caller lands between the two records.
namespace { hipEvent_t g_rep_ev[3][2]; bool g_rep_seen[3] = {false, false, false}; }
void dz_prof_pair(int which, int end, hipStream_t s) {
  if (!g_dz_prof_on || which < 0 || which > 2) return;
  (void)hipEventRecord(g_rep_ev[which][end], s);
  if (end) g_rep_seen[which] = true;
}
extern "C" int dz_prof_read_replay(float* ms_out) {
  DZ_REQUIRE(ms_out);
  for (int i = 0; i < 3; ++i) {
    ms_out[i] = -1.f;
    if (g_ev_created && g_rep_seen[i])
      DZ_HIP_CHECK(hipEventElapsedTime(&ms_out[i], g_rep_ev[i][0], g_rep_ev[i][1]));
  }
  return DZ_OK;
}
extern "C" int dz_prof_enable(int on) {
  if (on && !g_ev_created) {
    for (int i = 0; i <= kMaxMarks; ++i) DZ_HIP_CHECK(hipEventCreate(&g_ev[i]));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 2; ++j) DZ_HIP_CHECK(hipEventCreate(&g_rep_ev[i][j]));
    g_ev_created = true;
  }
  g_dz_prof_on = on != 0;
  g_host_clock = on == 2;
  g_nmarks = 0;
  return DZ_OK;
}
extern "C" int dz_prof_read(int max_marks, float* ms_out, char* names_out) {
  DZ_REQUIRE(ms_out && names_out && max_marks > 0);
  const int n = g_nmarks < max_marks ? g_nmarks : max_marks;
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    if (g_host_clock) ms = (float)(g_host_ns[i + 1] - g_host_ns[i]) * 1e-6f;
    else DZ_HIP_CHECK(hipEventElapsedTime(&ms, g_ev[i], g_ev[i + 1]));
    ms_out[i] = ms;
    strncpy(names_out + 32 * i, g_names[i], 31);
    names_out[32 * i + 31] = 0;
  }
  return n;
}
