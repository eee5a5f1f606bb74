// Microbenchmark: per-CU LDS-DMA delivery rate for the BPTT chain kernel's access pattern
// (32 x 32 output tile, 128-deep K stages, four loader waves beside four compute waves),
// with and without barriers, fragment reads, MFMAs and a dependent chain of planes.
// It is what showed that the step kernel (11.05 us) was not at a hardware limit: the same
// data movement with the same MFMAs takes 6.5-7.0 us per launch.
//   hipcc --offload-arch=gfx950 -O3 -DNLOAD=4 -DSTAGES=3 -DNST=8 tools/dma_rate_microbench.hip -o build/dma_rate
//   RANDOM_DATA=1 build/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
#ifndef NLOAD
#define NLOAD 4      // loader waves
#endif
#ifndef STAGES
#define STAGES 3
#endif
#ifndef NST
#define NST 8        // stages per launch
#endif
#ifndef MISALIGN
#define MISALIGN 0   // 1: operand rows start at column 1 (4-byte-aligned 16-byte chunks), as in the real kernel
#endif
constexpr int CK = 128, SF = 64 * CK; // 32 KB stage
struct Big { const float *p[60]; int n[16]; };
template <int MODE>
__global__ __launch_bounds__(64 * NLOAD + 256) void k(
#ifdef BIGARG
    Big big,
#endif
    const float *A, const float *W, float *out, int I, int H) {
#ifdef BIGARG
  W = big.p[7]; I = big.n[3]; H = big.n[5]; out = const_cast<float *>(big.p[59]);
  if (big.n[15] == 99) A = big.p[33];
#endif
  __shared__ __attribute__((aligned(16))) float smem[STAGES * SF];
  const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
  const int mt = q % 8, nt = (q / 8) * 8 + xcd;
  const int m0 = mt * 32, n0 = 1 + nt * 32;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool loader = wave >= 4;
  const int lw = wave - 4;
  constexpr int DPW = 32 / NLOAD;
  if (loader) {
    const float *src[DPW];
    for (int j = 0; j < DPW; j++) {
      int i = lw * DPW + j;
      int row = 2 * (i & 15) + (lane >> 5);
      int c = (lane & 31) ^ (row & 15);
      const float *base = (i < 16) ? A + (size_t)(m0 + row) * I : W + (size_t)(n0 + row) * H;
      if (MODE == 2) base = A; // hot line: everything from the same few lines
      src[j] = base + MISALIGN + 4 * c;
    }
    auto issue = [&](int st) {
      for (int j = 0; j < DPW; j++) {
        float *dst = smem + (st % STAGES) * SF + (lw * DPW + j) * 256;
        const float *g = src[j] + (MODE == 2 ? 0 : st * CK);
        __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)dst, 16, 0, 0);
      }
    };
    for (int p = 0; p < STAGES - 1; p++) issue(p);
    for (int st = 0; st < NST; st++) {
      int ahead = min(STAGES - 2, NST - 1 - st);
      // conservative waits (vmcnt immediates): in-order, so waiting for <= ahead*DPW outstanding
      if (ahead * DPW >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (ahead * DPW >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead * DPW >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (ahead * DPW >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead * DPW >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (MODE != 1) __builtin_amdgcn_s_barrier();
      if (st + STAGES - 1 < NST) issue(st + STAGES - 1);
    }
#ifndef QUAD
    if (MODE == 6) __syncthreads();
#endif
  } else {
    float acc = 0;
    for (int st = 0; st < NST && MODE < 4; st++) {
      if (MODE != 1) __builtin_amdgcn_s_barrier();
      if (MODE == 3) { // read fragments like the chain does
        const float *p = smem + (st % STAGES) * SF;
        for (int gi = 0; gi < 4; gi++) {
          int c = 2 * (4 * wave + gi) + (lane >> 5);
          int lm = lane & 31;
          const float4 a = *(const float4 *)(p + lm * CK + ((c ^ (lm & 15)) * 4));
          const float4 b = *(const float4 *)(p + 32 * CK + lm * CK + ((c ^ (lm & 15)) * 4));
          acc += a.x + b.y;
        }
      }
    }
#ifdef QUAD
    // every compute wave owns a 16 x 16 quadrant of the tile over the whole K (16x16x4 MFMA):
    // no cross-wave reduction at the end, twice the LDS fragment reads
    if (MODE >= 4) {
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      f32x4v c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
      const int wm = wave >> 1, wn = wave & 1, m = lane & 15, kq = lane >> 4;
      for (int st = 0; st < NST; st++) {
        __builtin_amdgcn_s_barrier();
        const float *p = smem + (st % STAGES) * SF;
        float4 a[8], b[8];
        for (int i = 0; i < 8; i++) {
          int c = 4 * i + kq;
          int ra = wm * 16 + m, rb = wn * 16 + m;
          a[i] = *(const float4 *)(p + ra * CK + ((c ^ (ra & 15)) * 4));
          b[i] = *(const float4 *)(p + 32 * CK + rb * CK + ((c ^ (rb & 15)) * 4));
        }
        for (int i = 0; i < 8; i++) {
          if (MODE == 4 || (i & 1) == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, c0, 0, 0, 0);
          } else {
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, c1, 0, 0, 0);
          }
        }
      }
      for (int i = 0; i < 4; i++) acc += c0[i] + c1[i];
      if (MODE == 6) {
        // no workgroup-wide sum: each lane writes its four (row, column) values
        float *pl = const_cast<float *>(A) + (size_t)256 * I;
        for (int g = 0; g < 4; g++) {
          int row = m0 + wm * 16 + 4 * kq + g, col = n0 + wn * 16 + m;
          pl[(size_t)row * I + col] = (c0[g] + c1[g]) * 0.f;
        }
      }
    }
#else
    if (MODE >= 4) {
      typedef float f32x16 __attribute__((ext_vector_type(16)));
      f32x16 c0, c1;
      for (int i = 0; i < 16; i++) { c0[i] = 0; c1[i] = 0; }
      for (int st = 0; st < NST; st++) {
        __builtin_amdgcn_s_barrier();
        const float *p = smem + (st % STAGES) * SF;
        float4 a[4], b[4];
        for (int gi = 0; gi < 4; gi++) {
          int c = 2 * (4 * wave + gi) + (lane >> 5);
          int lm = lane & 31;
          a[gi] = *(const float4 *)(p + lm * CK + ((c ^ (lm & 15)) * 4));
          b[gi] = *(const float4 *)(p + 32 * CK + lm * CK + ((c ^ (lm & 15)) * 4));
        }
        for (int gi = 0; gi < 4; gi++) {
          if (MODE == 4 || (gi & 1) == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].x, b[gi].x, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].y, b[gi].y, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].z, b[gi].z, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].w, b[gi].w, c0, 0, 0, 0);
          } else {
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].x, b[gi].x, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].y, b[gi].y, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].z, b[gi].z, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].w, b[gi].w, c1, 0, 0, 0);
          }
        }
      }
      for (int i = 0; i < 16; i++) acc += c0[i] + c1[i];
      if (MODE == 6) {
        __syncthreads();
        // each compute thread writes 4 floats of the tile into the next plane (row-major rows of I)
        int et = threadIdx.x & 255, row = et >> 3, c4 = (et & 7) * 4;
        float *dst = const_cast<float *>(A) + (size_t)256 * I /* next plane */ + (size_t)(m0 + row) * I + n0 + c4;
        dst[0] = acc * 0.f; dst[1] = 0.f; dst[2] = 0.f; dst[3] = 0.f;
      }
    }
#endif
    if (acc == 12345.f) out[threadIdx.x] = acc;
  }
}
int main() {
  const int I = 1068, H = 1028;
  float *A, *W, *o;
  hipMalloc(&A, (size_t)21 * 256 * I * 4);
  hipMalloc(&W, (size_t)I * H * 4);
  hipMalloc(&o, 4096);
  {
    size_t na = (size_t)21 * 256 * I, nw = (size_t)I * H;
    float *h = (float *)malloc((na > nw ? na : nw) * 4);
    const char *rnd = getenv("RANDOM_DATA");
    unsigned x = 12345;
    for (size_t i = 0; i < na; i++) { x = x * 1664525u + 1013904223u; h[i] = rnd ? ((x >> 8) / 16777216.0f - 0.5f) * 0.1f : 0.0f; }
    hipMemcpy(A, h, na * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < nw; i++) { x = x * 1664525u + 1013904223u; h[i] = rnd ? ((x >> 8) / 16777216.0f - 0.5f) * 0.1f : 0.0f; }
    hipMemcpy(W, h, nw * 4, hipMemcpyHostToDevice);
    free(h);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
#ifdef BIGARG
  Big big; for (int i = 0; i < 60; i++) big.p[i] = W; for (int i = 0; i < 16; i++) big.n[i] = 0;
  big.p[7] = W; big.n[3] = I; big.n[5] = H; big.p[59] = o;
#define BA big,
#else
#define BA
#endif
  auto run = [&](int mode, const char *name) {
    const int R = 400;
    for (int w = 0; w < 2; w++) {
      hipEventRecord(e0, 0);
      for (int r = 0; r < R; r++) {
        const float *a = A + (size_t)(r % 20) * 256 * I;
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * NLOAD + 256), 0, 0, BA a, W, o, I, H);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * NLOAD + 256), 0, 0, BA a, W, o, I, H);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * NLOAD + 256), 0, 0, BA a, W, o, I, H);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * NLOAD + 256), 0, 0, BA a, W, o, I, H);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(64 * NLOAD + 256), 0, 0, BA a, W, o, I, H);
        if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(64 * NLOAD + 256), 0, 0, BA a, W, o, I, H);
        if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(64 * NLOAD + 256), 0, 0, BA a, W, o, I, H);
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("NLOAD %d STAGES %d NST %d  %-28s %.2f us/launch  (%.1f GB/s per CU)\n", NLOAD, STAGES, NST, name,
           1e3 * ms / R, NST * 32.768 / (1e3 * ms / R) );
  };
  run(0, "barrier per stage");
  run(1, "no barriers");
  run(2, "hot lines, barrier");
  run(3, "barrier + fragment reads");
  run(4, "reads + 16 MFMA, 1 acc");
  run(5, "reads + 16 MFMA, 2 acc");
  run(6, "MFMA + dependent chain of planes");
  return 0;
}
