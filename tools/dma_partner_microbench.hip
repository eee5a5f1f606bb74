// Microbenchmark (round 3): what slows a loader wave's LDS-DMA issue beside a SIMD partner that issues
// f32 MFMAs back to back, and does it need the address VGPR?  In k_chain_persist sixteen
// `global_load_lds_dwordx4 v, s[base]` take 0.4 us to issue alone and 1.1 us beside the burst; in k_delta_dma
// the operand stream costs the matrix pipe 13 us per launch.
//
// Workgroup = 8 waves, one per CU: waves 0-3 issue MFMAS f32 MFMAs each (nothing else), waves 4-7 issue
// PIECES LDS-DMA instructions of 1 KB each in batches of 16 (wait for the batch, next batch), either
//   form 1: global_load_lds_dwordx4 voffset, s[base:base+1]         (per-lane byte offset in a VGPR), or
//   form 2: buffer_load_dwordx4 off, s[rsrc], soffset lds            (no VGPR: the resource adds tid x 16).
// Reported: kernel time; the loader's time per batch of 16 from s_memrealtime stamps.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_partner_microbench.hip -o build/dev/dma_partner && build/dev/dma_partner
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_batch_ticks[256][4];

template <int FORM, int MFMAS, int PIECES, int SC1 = 0, int SHAPE16 = 0, int LDSREADS = 0, int SELF = 0>
__global__ __launch_bounds__(512) void k(const float *src, float *out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave < 4) {
    f32x16 c[4];
    for (int j = 0; j < 4; j++)
      for (int i = 0; i < 16; i++) c[j][i] = 0;
    const float a = 0.37f + lane * 1e-3f, b = 0.11f - lane * 1e-3f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 c16[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    float ldsacc = 0;
    const uint32_t myl = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(lds + 16384 + wave * 1024) + lane * 16;
    for (int t = 0; t < MFMAS / 64; t++) {
      if (SHAPE16) { /* the chain's shape: 128 x v_mfma_f32_16x16x4_f32 = the same matrix-pipe time as 64 x 32x32x2 */
#pragma unroll
        for (int u = 0; u < 16; u++) {
          f32x4 av = {a, a, a, a};
          if (LDSREADS == 1) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(av) : "v"(myl + (u & 3) * 64));
          if (LDSREADS == 2) __builtin_amdgcn_s_sleep(1);                       /* a 64-cycle nap per 8 MFMAs */
          if (LDSREADS == 3 && (u & 1)) __builtin_amdgcn_s_sleep(1);            /* ... per 16 MFMAs */
          if (LDSREADS == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(av) : "v"(myl + (u & 3) * 64)); /* read, no wait (WRONG data, timing only) */
#pragma unroll
          for (int j = 0; j < 8; j++) c16[j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j & 3], b, c16[j & 1], 0, 0, 0);
          if (SELF) { /* the multiplying wave issues its own piece: one per 8 MFMAs = 16 per 128, the chain's rate */
            const char *sb = reinterpret_cast<const char *>(src) + ((size_t)blockIdx.x * 4 + wave) * 65536;
            const uint32_t dl = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(lds + wave * 4096) + u * 1024;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" : : "v"(lane * 16 + u * 1024), "s"(sb), "s"(dl) : "memory");
          }
        }
        if (SELF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
          for (int j = 0; j < 4; j++) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j], 0, 0, 0);
      }
    }
    float s = c16[0][0] + c16[1][1] + ldsacc;
    for (int j = 0; j < 4; j++)
      for (int i = 0; i < 16; i++) s += c[j][i];
    if (s == 12345.f) out[threadIdx.x] = s;
    return;
  }
  // loaders: a 64 KB region of `src` per wave, re-read batch after batch (L2 hits), into 16 KB of LDS per wave
  const int w = wave - 4;
  const char *base = reinterpret_cast<const char *>(src) + ((size_t)blockIdx.x * 4 + w) * 65536;
  const uint32_t ldsb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(lds + w * 4096);
  const unsigned voff = lane * 16;
  // buffer resource: base, stride 16 with add_tid_enable (word 3 bit 23), num_records = whole range, raw 32-bit format
  u32x4 rsrc;
  {
    const uint64_t b = (uint64_t)base;
    rsrc[0] = (unsigned)b;
    rsrc[1] = (unsigned)(b >> 32) | (16u << 16);
    rsrc[2] = 0x7fffffffu;
    rsrc[3] = (4u << 15) | (7u << 12) | (1u << 23) | 0x3aacu; /* data_format 32, num_format float, add_tid, dst_sel xyzw */
  }
  unsigned long long ticks = 0;
  int batches = 0;
  for (int p = 0; p < PIECES; p += 16) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const uint32_t dst = ldsb + (uint32_t)(i * 1024);
      if (FORM == 1 && SC1)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" : : "v"(voff + i * 1024), "s"(base), "s"(dst) : "memory");
      else if (FORM == 1)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff + i * 1024), "s"(base), "s"(dst) : "memory");
      else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %0, %1 lds" : : "s"(rsrc), "s"(i * 1024), "s"(dst) : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ticks += t1 - t0;
    batches++;
  }
  if (lane == 0) g_batch_ticks[blockIdx.x][w] = batches ? ticks / batches : 0;
  if (lds[threadIdx.x] == 12345.f) out[threadIdx.x] = 1;
}

int main() {
  float *src, *o;
  (void)hipMalloc(&src, (size_t)256 << 20);
  (void)hipMemset(src, 0, (size_t)256 << 20);
  (void)hipMalloc(&o, 8192);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto run = [&](auto kern, const char *name) {
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    float ms = 0;
    for (int r = 0; r < 10; r++) {
      if (r == 9) (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, src, o);
      if (r == 9) (void)hipEventRecord(e1, 0);
    }
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t[256][4];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_batch_ticks), sizeof(t));
    double sum = 0;
    for (int b = 0; b < 256; b++)
      for (int w = 0; w < 4; w++) sum += (double)t[b][w];
    printf("%-64s kernel %7.2f us   16 pieces issued in %5.2f us (mean over the loader waves)\n", name, 1e3 * ms, sum / 1024 / 100.0);
  };
  run(k<1, 2560, 0>, "2560 MFMAs per compute wave, no loader work");
  run(k<1, 0, 640>, "no MFMAs; 640 pieces per loader wave, VGPR-offset form");
  run(k<2, 0, 640>, "no MFMAs; 640 pieces per loader wave, buffer add-tid form (no VGPR)");
  run(k<1, 2560, 320>, "2560 MFMAs beside 320 pieces per loader wave, VGPR-offset form");
  run(k<2, 2560, 320>, "2560 MFMAs beside 320 pieces per loader wave, buffer add-tid form");
  run(k<1, 2560, 640>, "2560 MFMAs beside 640 pieces per loader wave, VGPR-offset form");
  run(k<2, 2560, 640>, "2560 MFMAs beside 640 pieces per loader wave, buffer add-tid form");
  run(k<1, 2560, 640, 1>, "... VGPR-offset form with sc1 (past L1)");
  run(k<1, 2560, 0, 0, 1>, "16x16x4 MFMAs (the chain's shape), no loader work");
  run(k<1, 2560, 640, 0, 1>, "16x16x4 MFMAs beside 640 pieces");
  run(k<1, 2560, 640, 1, 1>, "16x16x4 MFMAs beside 640 pieces, sc1");
  run(k<1, 2560, 0, 1, 1, 0, 1>, "16x16x4 MFMAs, each wave issues ITS OWN 640 pieces (sc1) between them");
  run(k<1, 2560, 0, 0, 1, 2>, "16x16x4 MFMAs + s_sleep 1 per 8, no loader work");
  run(k<1, 2560, 640, 1, 1, 2>, "16x16x4 MFMAs + s_sleep 1 per 8, beside 640 pieces, sc1");
  run(k<1, 2560, 0, 0, 1, 3>, "16x16x4 MFMAs + s_sleep 1 per 16, no loader work");
  run(k<1, 2560, 640, 1, 1, 3>, "16x16x4 MFMAs + s_sleep 1 per 16, beside 640 pieces, sc1");
  run(k<1, 2560, 0, 0, 1, 4>, "16x16x4 MFMAs + an unwaited ds_read_b128 per 8, no loader work");
  run(k<1, 2560, 640, 1, 1, 4>, "16x16x4 MFMAs + an unwaited ds_read_b128 per 8, beside 640 pieces, sc1");
  run(k<1, 2560, 0, 0, 1, 1>, "16x16x4 MFMAs + a ds_read_b128 per 8, no loader work");
  run(k<1, 2560, 640, 1, 1, 1>, "16x16x4 MFMAs + a ds_read_b128 per 8, beside 640 pieces, sc1");
  return 0;
}
