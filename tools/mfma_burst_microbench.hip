// Microbenchmark: the one-launch chain's MFMA burst by itself -- 128 x v_mfma_f32_16x16x4_f32 on two alternating
// accumulators, B from a 128-register panel, A from four rotating registers, one wave per SIMD, 256 workgroups -- timed per
// burst with BOTH clocks: s_memrealtime (100 MHz, constant) and s_memtime (the shader clock): how long a burst takes and at
// what clock the chip runs while every matrix pipe is busy.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_burst_microbench.hip -o build/dev/mfma_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int MODE> /* 0: one B register, one A register; 1: the panel (128 B registers, 4 A registers) */
__global__ __launch_bounds__(256) void k(const float *w, unsigned long long *stamps, float *out, int bursts) {
  float wreg[128];
#pragma unroll
  for (int i = 0; i < 128; i++) wreg[i] = w[(size_t)i * 256 + threadIdx.x];
  float a[4];
#pragma unroll
  for (int i = 0; i < 4; i++) a[i] = w[32768 + i * 256 + threadIdx.x];
  f32x4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  unsigned long long rt0 = 0, ct0 = 0, rt1 = 0, ct1 = 0;
  for (int b = 0; b < bursts; b++) {
    if (b == bursts / 2) {
      rt0 = __builtin_amdgcn_s_memrealtime();
      ct0 = __builtin_amdgcn_s_memtime();
    }
#pragma unroll
    for (int u = 0; u < 64; u++) {
      if (MODE == 0) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], wreg[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], wreg[1], acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 3], wreg[2 * u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 3], wreg[2 * u + 1], acc1, 0, 0, 0);
      }
    }
    asm volatile("" : "+v"(acc0), "+v"(acc1));
  }
  rt1 = __builtin_amdgcn_s_memrealtime();
  ct1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    stamps[4 * blockIdx.x + 0] = rt1 - rt0;
    stamps[4 * blockIdx.x + 1] = ct1 - ct0;
  }
  float s = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  if (s == 12345.f) out[threadIdx.x] = s;
}
int main() {
  float *w, *o;
  unsigned long long *st;
  (void)hipMalloc(&w, 4 * (32768 + 1024));
  (void)hipMalloc(&o, 4096);
  (void)hipMalloc(&st, 8 * 4 * 256);
  std::vector<float> h(32768 + 1024);
  for (size_t i = 0; i < h.size(); i++) h[i] = 0.001f * (float)((i * 2654435761u >> 12) & 1023) - 0.5f;
  (void)hipMemcpy(w, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; mode++)
    for (int bursts : {40, 400, 4000}) {
      for (int rep = 0; rep < 3; rep++) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, w, st, o, bursts);
        else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, w, st, o, bursts);
      }
      (void)hipDeviceSynchronize();
      unsigned long long hs[4 * 256];
      (void)hipMemcpy(hs, st, sizeof hs, hipMemcpyDeviceToHost);
      double rt = 0, ct = 0;
      for (int i = 0; i < 256; i++) { rt += (double)hs[4 * i]; ct += (double)hs[4 * i + 1]; }
      rt /= 256; ct /= 256;
      const int nb = bursts - bursts / 2;
      printf("%s, %4d bursts of 128 MFMAs: %.3f us per burst (%.2f ns per MFMA), shader clock %.3f GHz, %.1f clocks per MFMA\n",
             mode ? "128-register panel, 4 A registers" : "one B register pair, one A register ", bursts,
             rt * 0.01 / nb, rt * 10.0 / nb / 128, ct / (rt * 10.0), ct / nb / 128);
    }
  return 0;
}
