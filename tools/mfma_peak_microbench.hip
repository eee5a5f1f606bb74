// Microbenchmark: sustained fp32 MFMA rate of the whole chip with nothing else going on
// (4 or 8 waves per CU, independent accumulators), for the two fp32 shapes.  It is the
// ceiling k_delta_dma is priced against in DESIGN.md: the nominal 157.3 TFLOP/s assumes
// 2.4 GHz, the chip sustains less under this load.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_microbench.hip -o build/dev/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
  /* lane-dependent operands: realistic toggling in the multiplier array */
  a = a * (0.37f + (threadIdx.x * 2654435761u >> 8) * (1.0f / 16777216.0f));
  b = b * (0.11f - (threadIdx.x * 40503u >> 3 & 0xffff) * (1.0f / 65536.0f));
  if (SHAPE == 32) {
    f32x16 c[NACC];
    for (int j = 0; j < NACC; j++)
      for (int i = 0; i < 16; i++) c[j][i] = 0;
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 16; u++)
#pragma unroll
        for (int j = 0; j < NACC; j++) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < NACC; j++)
      for (int i = 0; i < 16; i++) s += c[j][i];
    if (s == 12345.f) out[threadIdx.x] = s;
  } else {
    f32x4v c[NACC];
    for (int j = 0; j < NACC; j++) c[j] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int u = 0; u < 32; u++)
#pragma unroll
        for (int j = 0; j < NACC; j++) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < NACC; j++)
      for (int i = 0; i < 4; i++) s += c[j][i];
    if (s == 12345.f) out[threadIdx.x] = s;
  }
}
// a latency-bound filler: one wave per CU chases a pointer for a few microseconds
__global__ void k_idle(const int *chain, int *out, int hops) {
  int i = threadIdx.x;
  for (int h = 0; h < hops; h++) i = chain[i];
  if (i == 12345) out[0] = i;
}
int main() {
  float *o;
  (void)hipMalloc(&o, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto run = [&](int shape, int wgs_per_cu, const char *name) {
    const int iters = 2000, NACC = 4;
    float ms = 0;
    /* about a second of back-to-back launches; the rate of the last one is reported */
    for (int w = 0; w < 300 / wgs_per_cu; w++) {
      const bool last = w == 300 / wgs_per_cu - 1;
      if (last) (void)hipEventRecord(e0, 0);
      if (shape == 32) hipLaunchKernelGGL((k<32, NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, o, iters, 1.0f, 0.5f);
      else hipLaunchKernelGGL((k<16, NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, o, iters, 1.0f, 0.5f);
      if (last) (void)hipEventRecord(e1, 0);
    }
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per wave: iters * (16 * NACC MFMAs of 4096 flop) or (32 * NACC of 2048 flop)
    double flop = (double)256 * wgs_per_cu * 4 * iters * 16.0 * NACC * 4096.0;
    printf("%-34s %8.3f ms  %7.1f TFLOP/s  => %.2f GHz at 256 flop/cycle/CU\n", name, ms, flop / ms / 1e9,
           flop / ms / 1e9 * 1e12 / (256.0 * 256.0) / 1e9);
  };
  // the same MFMA work in the position k_delta_dma has in a generation: one launch of
  // 2560 MFMAs per wave (69 us at 2.37 GHz) after ~180 us of latency-bound small kernels
  {
    int *chain, *io;
    (void)hipMalloc(&chain, 4096);
    (void)hipMalloc(&io, 64);
    int h[1024];
    for (int i = 0; i < 1024; i++) h[i] = (i * 37 + 11) & 1023;
    (void)hipMemcpy(chain, h, 4096, hipMemcpyHostToDevice);
    const int gens = 600;
    hipEvent_t ea[gens], eb[gens];
    for (int g = 0; g < gens; g++) { (void)hipEventCreate(&ea[g]); (void)hipEventCreate(&eb[g]); }
    for (int fill = 0; fill < 2; fill++) {
      for (int g = 0; g < gens; g++) {
        if (fill) for (int t = 0; t < 20; t++) hipLaunchKernelGGL(k_idle, dim3(256), dim3(64), 0, 0, chain, io, 12);
        (void)hipEventRecord(ea[g], 0);
        hipLaunchKernelGGL((k<32, 4>), dim3(256), dim3(256), 0, 0, o, 40, 1.0f, 0.5f);
        (void)hipEventRecord(eb[g], 0);
      }
      (void)hipDeviceSynchronize();
      double sum = 0;
      for (int g = gens / 2; g < gens; g++) { float ms; (void)hipEventElapsedTime(&ms, ea[g], eb[g]); sum += ms; }
      double us = 1e3 * sum / (gens / 2);
      printf("2560 MFMAs per wave, %-32s %7.2f us per launch => %.2f GHz\n",
             fill ? "after 20 latency-bound launches:" : "back to back:", us, 2560.0 * 64.0 / us / 1e3);
    }
  }
  run(32, 1, "32x32x2 f32, 4 waves/CU");
  run(32, 2, "32x32x2 f32, 8 waves/CU");
  run(16, 1, "16x16x4 f32, 4 waves/CU");
  run(16, 2, "16x16x4 f32, 8 waves/CU");
  return 0;
}
