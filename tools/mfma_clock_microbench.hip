// Microbenchmark (round 3): what a dense f32 MFMA stream sustains on this part, per MFMA and in shader clocks.
// Waves issue N back-to-back v_mfma_f32_32x32x2_f32 (or 16x16x4) on four independent accumulators and read
// s_memtime (shader clock) and s_memrealtime (100 MHz) before and after.  Grid of 1 workgroup (one CU busy)
// against 256 (every CU busy): the difference is the clock the chip holds under a full fp32 matrix load.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_microbench.hip -o build/dev/mfma_clock && build/dev/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long clk0, clk1, rt0, rt1; };
template <int SHAPE16, int N>
__global__ __launch_bounds__(512) void k(Stamp *st, float *out) {
  const int lane = threadIdx.x & 63;
  f32x16 c[4];
  f32x4 d[4];
  for (int j = 0; j < 4; j++) {
    for (int i = 0; i < 16; i++) c[j][i] = 0;
    d[j] = {0, 0, 0, 0};
  }
  const float a = 0.37f + lane * 1e-3f, b = 0.11f - lane * 1e-3f;
  __syncthreads();
  const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < N / 16; t++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (SHAPE16) {
        d[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d[u & 3], 0, 0, 0);
        d[(u + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, d[(u + 2) & 3], 0, 0, 0);
      } else {
        c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[u & 3], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 4; j++) {
    for (int i = 0; i < 16; i++) s += c[j][i];
    s += d[j][0] + d[j][3];
  }
  const unsigned long long clk1 = __builtin_readcyclecounter(), rt1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) st[blockIdx.x] = {clk0, clk1, rt0, rt1};
  if (s == 12345.f) out[threadIdx.x] = s;
}
template <int SHAPE16>
static void run(const char *what, int grid, int threads) {
  constexpr int N = 8192;
  Stamp *st;
  float *out;
  hipMalloc(&st, sizeof(Stamp) * 256);
  hipMalloc(&out, 4096);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL((k<SHAPE16, N>), dim3(grid), dim3(threads), 0, 0, st, out);
    hipDeviceSynchronize();
  }
  Stamp h[256];
  hipMemcpy(h, st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost);
  double us = 0, clk = 0;
  for (int i = 0; i < grid; i++) {
    us += (h[i].rt1 - h[i].rt0) / 100.0;
    clk += (double)(h[i].clk1 - h[i].clk0);
  }
  us /= grid;
  clk /= grid;
  /* 32x32x2: 16 passes = 64 cycles; the 16x16x4 loop issues two of 32 cycles per count */
  printf("%-58s %8.2f us  %10.0f counter ticks  = %6.1f ticks/us; %6.2f ns per 64-cycle MFMA slot (2.4 GHz: 26.67)\n", what, us, clk,
         clk / us, us * 1000.0 / N);
  hipFree(st);
  hipFree(out);
}
int main() {
  run<0>("32x32x2, 1 workgroup x 4 waves (one per SIMD)", 1, 256);
  run<0>("32x32x2, 1 workgroup x 8 waves (two per SIMD)", 1, 512);
  run<0>("32x32x2, 256 workgroups x 4 waves", 256, 256);
  run<0>("32x32x2, 256 workgroups x 8 waves", 256, 512);
  run<0>("32x32x2, 64 workgroups x 4 waves", 64, 256);
  run<0>("32x32x2, 128 workgroups x 4 waves", 128, 256);
  run<1>("16x16x4 pairs, 1 workgroup x 4 waves", 1, 256);
  run<1>("16x16x4 pairs, 256 workgroups x 4 waves", 256, 256);
  return 0;
}
