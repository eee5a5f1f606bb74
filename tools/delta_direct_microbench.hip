// Microbenchmark for k_delta_direct (recur_amd/csrc/k_delta_direct.h): the weight-delta GEMM of the north star
// (hidden 1024, 256 streams, depth 20: I = 1068, H = 1028) on random data, alone on the chip, against a host
// reference on sampled elements; variants: waves per workgroup, ring depth, coefficient path, epilogue mode.
//   hipcc --offload-arch=gfx950 -O3 -Irecur_amd/csrc tools/delta_direct_microbench.hip -o build/delta_direct_microbench
//   build/delta_direct_microbench [reps]
#include "k_delta_direct.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static unsigned long long rs = 88172645463325252ull;
static inline float urand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 11) * (1.0 / 9007199254740992.0)); }

// something MFMA-dense between the launches (the chain launch of the real generation), so that the clock is the loaded one
__global__ __launch_bounds__(256) void k_burn(float *out, int iters) {
  dd_f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; i++) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
  }
  if (c0[0] + c1[0] == 12345.f) out[0] = c0[1];
}

template <int NW, int P, int NPW>
static double run(const char *name, DdArgs a, int reps, int burn, float *scratch) {
  CK(hipFuncSetAttribute((const void *)(k_delta_direct<NW, P, NPW>), hipFuncAttributeMaxDynamicSharedMemorySize, dd_lds_bytes(NW, NPW)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = a.tm * a.tn;
  auto once = [&]() {
    if (burn) hipLaunchKernelGGL(k_burn, dim3(256), dim3(256), 0, 0, scratch, burn);
    hipLaunchKernelGGL((k_delta_direct<NW, P, NPW>), dim3(grid), dim3(64 * NW), dd_lds_bytes(NW, NPW), 0, a);
  };
  for (int i = 0; i < 20; i++) once();
  CK(hipDeviceSynchronize());
  double burn_us = 0;
  if (burn) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_burn, dim3(256), dim3(256), 0, 0, scratch, burn);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    burn_us = ms * 1e3 / reps;
  }
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) once();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps - burn_us;
  const double gf = 2.0 * a.I * (64.0 * a.tn) * a.nrows * a.D * 1e-9;
  printf("%-34s %8.2f us per launch  %6.1f TFLOP/s (%.2f of 157.3)%s\n", name, us, gf / us * 1e3, gf / us * 1e3 / 157.3,
         burn ? "  [burn between]" : "");
  return us;
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  /* DDM_I=1100: a multi-head net's input rows (73 symbols): 16 row tiles and 76 rest rows, two pieces per workgroup */
  const int S = 256, D = 20, hidden = 1024, I = getenv("DDM_I") ? atoi(getenv("DDM_I")) : 1068, H = 1028, O = 44;
  const size_t plane = (size_t)S * I;
  std::vector<float> hx((size_t)D * plane + 128), he((size_t)(D + 1) * plane), hc((size_t)D * S, 1.0f), hw((size_t)I * H), hm((size_t)I * H);
  for (auto &v : hx) { float u = urand(); v = u < 0.55f ? 0.0f : (u - 0.55f) * 2.0f; }
  for (size_t p = 0; p < (size_t)(D + 1) * S; p++)
    for (int c = 0; c < I; c++) he[p * I + c] = (c >= 1 && c <= hidden) ? (urand() - 0.5f) * 0.01f : 0.0f;
  for (auto &v : hw) v = (urand() - 0.5f) * 0.1f;
  for (auto &v : hm) v = (urand() - 0.5f) * 0.001f;
  if (const char *pat = getenv("DDM_PATTERN")) { /* debugging: X = 1, E = 0 (pattern 1) or X = 0, E = 1 (2): every output must be 0 */
    for (auto &v : hx) v = pat[0] == '1' ? 1.0f : 0.0f;
    for (size_t p = 0; p < (size_t)(D + 1) * S; p++)
      for (int c = 0; c < I; c++) he[p * I + c] = (c >= 1 && c <= hidden && pat[0] == '2') ? 1.0f : 0.0f;
  }
  float *dx, *de, *dc, *dw, *dm, *dd, *dho, *dhom, *dhod, *scratch;
  int *dnex; float *dsc;
  const int tm = 16, tn = hidden / 64, rest = I - 64 * tm;
  const bool two = rest > 64 || getenv("DDM_TWO");
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&de, he.size() * 4)); CK(hipMalloc(&dc, hc.size() * 4));
  CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dm, hm.size() * 4)); CK(hipMalloc(&dd, hw.size() * 4));
  CK(hipMalloc(&dnex, S * 4)); CK(hipMalloc(&dsc, S * 4)); CK(hipMalloc(&scratch, 1024));
  CK(hipMalloc(&dho, (size_t)H * O * 4)); CK(hipMalloc(&dhom, (size_t)H * O * 4)); CK(hipMalloc(&dhod, (size_t)H * O * 4));
  CK(hipMemset(dho, 0, (size_t)H * O * 4)); CK(hipMemset(dhom, 0, (size_t)H * O * 4)); CK(hipMemset(dhod, 0, (size_t)H * O * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(de, he.data(), he.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
  std::vector<int> hnex(S, D);
  std::vector<float> hsc(S, 1.0f);
  CK(hipMemcpy(dnex, hnex.data(), S * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsc, hsc.data(), S * 4, hipMemcpyHostToDevice));
  auto set_general = [&](int general) { /* one stream of the last quad stopped early: the coefficient path */
    hnex[S - 2] = general ? D - 1 : D;
    CK(hipMemcpy(dnex, hnex.data(), S * 4, hipMemcpyHostToDevice));
  };
  DdArgs a = {};
  a.x = dx; a.e = de + 1; a.coef = dc; a.n_exec = dnex; a.ih_scale = dsc; a.w = dw + 1; a.m = dm + 1; a.delta = dd + 1;
  a.plane = plane; a.I = I; a.H = H; a.Scap = S; a.hidden_size = hidden;
  a.nrows = S; a.D = D; a.uidx = 7; a.tm = tm; a.tn = tn; a.rest = rest; a.rgroups = 4 * (two ? 2 : 1); a.mode = 0;
  { /* the library's shares for a SIMD's two waves (kernels_bptt.hip): DDM_FAST_PCT, default 66, 0 = halves */
    const int pct = getenv("DDM_FAST_PCT") ? atoi(getenv("DDM_FAST_PCT")) : 66, n_pair = 2 * (D * (S / 4 / 8));
    if (pct > 0 && n_pair >= 80) a.fast_its = (n_pair * pct / 100 + 2) / 5 * 5;
  }
  a.rate = 1e-5f; a.momentum = 0.95f; a.mw = 0.5f;
  a.ho_w = dho; a.ho_m = dhom; a.ho_delta = dhod; a.ho_delta_out = nullptr; a.ho_n4 = (unsigned)((size_t)H * O / 4); a.ho_rate = 1e-5f;

  // ---- correctness (mode 0, ones and general paths) on sampled elements + every rest row of a few columns
  auto check = [&](const char *what) {
    std::vector<float> out((size_t)I * H);
    CK(hipMemcpy(out.data(), dd, out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    int nbad = 0;
    auto ref = [&](int m, int n) {
      double s = 0;
      for (int t = 0; t < D; t++) {
        int slot = a.uidx - t; if (slot < 0) slot += D;
        for (int r = 0; r < S; r++)
          s += (double)hx[((size_t)slot * S + r) * I + m] * hc[(size_t)t * S + r] * he[((size_t)t * S + r) * I + n];
      }
      return s;
    };
    for (int k = 0; k < 3000; k++) {
      int m = k < 1500 ? (int)(urand() * I) : 1024 + (k % (I - 1024)), n = (int)(urand() * H);
      if (m >= I) m = I - 1;
      double want = (n >= 1 && n <= hidden) ? ref(m, n) : 0.0, got = out[(size_t)m * H + n];
      worst = fmax(worst, fabs(got - want)); scale = fmax(scale, fabs(want));
      if (fabs(got - want) > 1e-4 && nbad++ < 12) printf("   [%d][%d] got %g want %g (ratio %g)\n", m, n, got, want, got / want);
    }
    if (getenv("DDM_PATTERN")) {
      long nz = 0; int shown = 0;
      for (int m = 0; m < I; m++) for (int n = 0; n < H; n++) if (out[(size_t)m * H + n] != 0.0f) { nz++; if (shown++ < 16) printf("   nonzero [%d][%d] = %g\n", m, n, out[(size_t)m * H + n]); }
      printf("   %ld nonzero outputs\n", nz);
    }
    if (getenv("DDM_MAP")) { /* which (rest row, column mod 4) pairs are wrong */
      for (int m = 1024; m < I; m++) {
        int badc[4] = {0, 0, 0, 0};
        for (int n = 1; n <= 64; n++) {
          double want = ref(m, n), got = out[(size_t)m * H + n];
          if (fabs(got - want) > 1e-5) badc[(n - 1) & 3]++;
        }
        printf("   rest row %2d: wrong columns by (column - 1) %% 4: %d %d %d %d\n", m - 1024, badc[0], badc[1], badc[2], badc[3]);
      }
    }
    printf("check %-28s max |err| %.3e of max |ref| %.3e -> %s\n", what, worst, scale, worst <= 2e-5 * scale ? "ok" : "MISMATCH");
    return worst <= 2e-5 * scale;
  };
  bool ok = true;
  CK(hipFuncSetAttribute((const void *)(k_delta_direct<8, 5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, dd_lds_bytes(8, 1)));
  CK(hipFuncSetAttribute((const void *)(k_delta_direct<8, 5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, dd_lds_bytes(8, 2)));
  auto both = [&](const char *tag) {
    char nm[64];
    CK(hipMemset(dd, 0xff, hw.size() * 4));
    if (two) hipLaunchKernelGGL((k_delta_direct<8, 5, 2>), dim3(tm * tn), dim3(512), dd_lds_bytes(8, 2), 0, a);
    else hipLaunchKernelGGL((k_delta_direct<8, 5, 1>), dim3(tm * tn), dim3(512), dd_lds_bytes(8, 1), 0, a);
    CK(hipDeviceSynchronize());
    snprintf(nm, sizeof nm, "NW 8, P 5, %d piece(s), %s", two ? 2 : 1, tag);
    ok &= check(nm);
  };
  both("ones");
  set_general(1);
  both("coefficient path, all 1.0");
  for (int r = 1; r < S; r += 7) for (int t = 0; t < D; t++) hc[(size_t)t * S + r] = 0.5f;
  CK(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  both("coefficients 0.5");
  // some coefficients 0 with inf behind them
  for (int t = 12; t < D; t++) for (int r = 0; r < S; r += 3) { hc[(size_t)t * S + r] = 0.0f; he[((size_t)t * S + r) * I + 5] = INFINITY; }
  CK(hipMemcpy(de, he.data(), he.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  for (int t = 12; t < D; t++) for (int r = 0; r < S; r += 3) he[((size_t)t * S + r) * I + 5] = 0.0f; /* (the reference's view: those rows do not count) */
  both("coefficients 0 over inf");
  if (!ok) return 1;

  // ---- timing
  for (int burn = 0; burn <= 1; burn++) {
    const int b = burn ? 3000 : 0;
    for (int general = 0; general <= 1; general++) {
      set_general(general);
      printf("-- %s, %s\n", general ? "coefficient path" : "all-ones path", burn ? "an MFMA-dense launch between the launches" : "back to back");
      for (int mode = 0; mode <= 2; mode += 2) {
        a.mode = mode;
        char nm[64];
        snprintf(nm, sizeof nm, "NW 8 P 5 pieces %d mode %d", two ? 2 : 1, mode);
        if (two) run<8, 5, 2>(nm, a, reps, b, scratch);
        else run<8, 5, 1>(nm, a, reps, b, scratch);
      }
    }
  }
  return 0;
}
