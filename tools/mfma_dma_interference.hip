// Microbenchmark: what an operand load costs the matrix pipe of the SIMD that issues it.
// Every wave (4 per CU, one per SIMD) issues 2560 fp32 MFMAs -- the work of a k_delta_dma
// compute wave -- and, spread evenly between them, N loads of 1 KB per 64 MFMAs: by LDS-DMA,
// or as a plain 16-byte-per-lane global load, or that load followed by a ds_write_b128.
// No barriers, nothing waits for the data: the time added per load is issue cost.
// (A first version with separate mover waves showed something else: beside a wave that issues
// MFMAs back to back, the other wave of the SIMD does not get to issue at all.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_dma_interference.hip -o build/dev/mfma_dma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int MODE, int N> /* N loads per 64 MFMAs (N divides 16 or is 0) */
__global__ __launch_bounds__(256) void k(const float *src, float *out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16 c[4];
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 16; i++) c[j][i] = 0;
  float a = 0.37f + lane * 1e-3f, b = 0.11f - lane * 1e-3f;
  const float *g = src + ((size_t)blockIdx.x * 4 + wave) * 16384 + lane * 4;
  float *d = lds + wave * 4096;
  f32x4 v[16];
  for (int i = 0; i < 16; i++) v[i] = f32x4{0, 0, 0, 0};
  for (int t = 0; t < 40; t++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int j = 0; j < 4; j++) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j], 0, 0, 0);
      if (N > 0 && u % (16 / (N > 16 ? 16 : N)) == 0) {
#pragma unroll
        for (int rep = 0; rep < (N > 16 ? N / 16 : 1); rep++) {
          const int slot = (u + rep) & 15;
          const float *p = g + ((t & 3) * 16 + slot) * 256;
          if (MODE == 0) {
            __builtin_amdgcn_global_load_lds((glb_void_t *)p, (lds_void_t *)(d + slot * 256), 16, 0, 0);
          } else {
            if (MODE == 2) *reinterpret_cast<f32x4 *>(d + slot * 256 + lane * 4) = v[slot]; /* the previous round's */
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[slot]) : "v"(p) : "memory");
          }
        }
      }
    }
    if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 16; i++) s += c[j][i];
  for (int i = 0; i < 16; i++) s += v[i].x;
  if (s == 12345.f) out[threadIdx.x] = s + lds[threadIdx.x];
}

int main() {
  float *src, *o;
  (void)hipMalloc(&src, (size_t)256 << 20);
  (void)hipMemset(src, 0, (size_t)256 << 20);
  (void)hipMalloc(&o, 8192);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  double base = 0;
  auto time = [&](auto kern, const char *name, int n) {
    float ms = 0;
    for (int r = 0; r < 20; r++) {
      const bool last = r == 19;
      if (last) (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(256), 65536, 0, src, o);
      if (last) (void)hipEventRecord(e1, 0);
    }
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    double us = 1e3 * ms;
    if (n == 0) base = us;
    printf("%-46s %7.2f us", name, us);
    if (n) printf("   +%.0f cycles at 2.35 GHz per load", (us - base) * 2350.0 / (40.0 * n));
    printf("\n");
  };
  (void)hipFuncSetAttribute((const void *)k<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
#define RUN(M, NN, NAME)                                                                              \
  (void)hipFuncSetAttribute((const void *)k<M, NN>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
  time(k<M, NN>, NAME, NN)
  RUN(0, 0, "2560 MFMAs per wave, no loads");
  RUN(0, 4, "LDS-DMA, 4 KB per wave and 64 MFMAs");
  RUN(0, 8, "LDS-DMA, 8 KB (k_delta_dma's rate)");
  RUN(0, 16, "LDS-DMA, 16 KB");
  RUN(1, 8, "global_load_dwordx4, 8 KB");
  RUN(1, 16, "global_load_dwordx4, 16 KB");
  RUN(2, 8, "global_load_dwordx4 + ds_write_b128, 8 KB");
  RUN(2, 16, "global_load_dwordx4 + ds_write_b128, 16 KB");
  return 0;
}
