// kernels_support.hip -- launch-side support shared by the kernel translation units: event timing and tuning knobs.
#include "k_common.h"

// ---- HIP-event timing of the kernel classes (bench.py's roofline leg) ----
static int g_timing = 0;
struct TimedLaunch {
  hipEvent_t a, b;
  int cls, count;
};
static TimedLaunch g_ev[8192];
static int g_nev = 0;
static double g_ms[T_CLASSES];
static long g_launches[T_CLASSES];

int timing_begin(hipStream_t st, int cls, int count) {
  if (!g_timing || g_nev >= 8192) return -1;
  int i = g_nev++;
  HIP_CHECK(hipEventCreate(&g_ev[i].a));
  HIP_CHECK(hipEventCreate(&g_ev[i].b));
  g_ev[i].cls = cls;
  g_ev[i].count = count;
  HIP_CHECK(hipEventRecord(g_ev[i].a, st));
  return i;
}
void timing_end(hipStream_t st, int i) {
  if (i >= 0) HIP_CHECK(hipEventRecord(g_ev[i].b, st));
}
static void timing_collect() {
  for (int i = 0; i < g_nev; i++) {
    float ms = 0;
    HIP_CHECK(hipEventSynchronize(g_ev[i].b));
    HIP_CHECK(hipEventElapsedTime(&ms, g_ev[i].a, g_ev[i].b));
    g_ms[g_ev[i].cls] += ms;
    g_launches[g_ev[i].cls] += g_ev[i].count;
    HIP_CHECK(hipEventDestroy(g_ev[i].a));
    HIP_CHECK(hipEventDestroy(g_ev[i].b));
  }
  g_nev = 0;
}
extern "C" void ramd_timing_enable(int enable) { g_timing = enable; }
/* the same bracket for the host code (rnn_core.c: the exchange step between ranks, class RAMD_T_XCHG) */
extern "C" int ramd_timing_begin(ramd_stream_t st, int cls) { return timing_begin((hipStream_t)st, cls, 1); }
extern "C" void ramd_timing_end(ramd_stream_t st, int i) { timing_end((hipStream_t)st, i); }
extern "C" double ramd_timing_ms(int which, long *launches, int reset) {
  timing_collect();
  if (which < 0 || which >= T_CLASSES) return 0.0;
  double ms = g_ms[which];
  if (launches) *launches = g_launches[which];
  if (reset) {
    for (int c = 0; c < T_CLASSES; c++) {
      g_ms[c] = 0;
      g_launches[c] = 0;
    }
  }
  return ms;
}

// Tuning knobs (RECUR_AMD_*) are read from the environment ONCE, the first time a launcher
// asks for them, and frozen: the product path does not call getenv per launch.
int env_int(const char *name, int dflt) {
  struct Knob {
    const char *name;
    int set, value;
  };
  static Knob knobs[48];
  static int n_knobs = 0;
  for (int i = 0; i < n_knobs; i++)
    if (knobs[i].name == name || strcmp(knobs[i].name, name) == 0)
      return knobs[i].set ? knobs[i].value : dflt;
  const char *e = getenv(name);
  Knob k = {name, (e && *e) ? 1 : 0, (e && *e) ? atoi(e) : 0};
  if (n_knobs < 48) knobs[n_knobs++] = k;
  return k.set ? k.value : dflt;
}
