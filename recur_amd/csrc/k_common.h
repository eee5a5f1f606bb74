// k_common.h -- what the kernel translation units (kernels_*.hip) share: the launch-check macros,
// the View of a training set that every kernel takes, small device helpers (input rows, soft clip,
// block sums, the generator, LDS / LDS-DMA / inline-asm memory helpers) and the declarations of the
// launch-side support functions (kernels_support.hip).  CDNA4 (gfx950), wave64 only; there is no
// other backend.
//
// Heavy lifting is three fp32 MFMA GEMMs, all with M = "streams of the synchronic mini-batch":
//   forward   Hpre[S x H] = X[S x I] . W_ih[I x H]              (recur-nn.c:18-48, 117)
//   chain     E_i[S x I]  = E_h[S x H] . W_ih^T   per BPTT step (recur-nn.c:338-376)
//   delta     dW[I x H]   = sum_t X_t^T . diag(c_t) . E_h,t      (recur-nn.c:344-356, 738)
// kernels_forward.hip holds the first, kernels_chain.hip the second, kernels_bptt.hip the third with
// the rest of rnn_bptt_calc_deltas, kernels_loss.hip the callers' loss functions on the device,
// kernels_apply.hip rnn_apply_learning and the conditioning helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <utility>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "ramd_internal.h"

#define HIP_CHECK(x)                                                              \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "librecur_amd: HIP error %s at %s:%d\n", hipGetErrorString(e_), \
              __FILE__, __LINE__);                                                \
      abort();                                                                    \
    }                                                                             \
  } while (0)

// Every launch is followed by hipGetLastError(): a bad launch configuration at an untested
// shape is reported where it happens, not at the next synchronisation.
static inline void ramd_check_launch(const char *file, int line) {
  hipError_t e_ = hipGetLastError();
  if (e_ != hipSuccess) {
    fprintf(stderr, "librecur_amd: kernel launch failed: %s at %s:%d\n", hipGetErrorString(e_), file,
            line);
    abort();
  }
}
#define RAMD_LAUNCH(...)                       \
  do {                                         \
    hipLaunchKernelGGL(__VA_ARGS__);           \
    ramd_check_launch(__FILE__, __LINE__);     \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// constants of the algorithm (recur-nn.h:28-47)
#define INPUT_MEAN_SOFT_TOP_F 16.0f
#define MAX_TOP_ERROR_FACTOR_F 2.0f
#define MAX_ERROR_GAIN_F 2.0f
#define ERROR_GAIN_CEILING_F 1.0f
#define MIN_ERROR_GAIN_F 1e-8f
#define MAX_MIN_ERROR_FACTOR_F 1e-2f
#define ABS_MIN_ERROR_FACTOR_F 1e-20f

// ----------------------------------------------------------------- helpers --

struct View {
  RamdShape sh;
  RamdBuffers b;
};

// input row (history slot or forward-only input row) of state row r, `back`
// steps into the past (back = 0: the slot rnn_bptt_advance points at)
//
// When every stream of the call sits at the same ring position (the normal case:
// the set advances in lock step) the host passes it in b.uniform_idx and no
// index has to be fetched; a load here would sit on the address path of the
// GEMM operand loads and drain their pipeline.
template <bool UNI = false>
__device__ __forceinline__ float *input_row(const View &v, int r, int back) {
  const RamdShape &s = v.sh;
  if (r < s.Scap) {
    int slot = (UNI ? v.b.uniform_idx : v.b.idx[r]) - back;
    if (slot < 0) slot += s.D;
    return v.b.arena + (slot * s.Scap + r) * s.I; /* 32-bit element offsets: checked on the host */
  }
  return v.b.arena + (s.D * s.Scap + (r - s.Scap)) * s.I;
}

/* the same with the choice made at run time (b.uniform_idx >= 0: no index load in front of the row's) */
__device__ __forceinline__ float *input_row_auto(const View &v, int r, int back) {
  return v.b.uniform_idx >= 0 ? input_row<true>(v, r, back) : input_row<false>(v, r, back);
}

// recur-nn-helpers.h:104-113
__device__ __forceinline__ float soft_clip_dev(float sum, float halfmax) {
  if (halfmax == 0) return sum;
  float x = sum / halfmax;
  float fudge = (float)(0.99 + (double)(x * x) / 100);
  return 2.0f * x / (1 + x * x * fudge);
}

// A wave-wide sum that every lane gets, without the LDS crossbar (round 6): a butterfly over the sixteen lanes of a row in
// four DPP steps, then the four rows' values through scalar registers (v_readlane) -- a dozen instructions of a few cycles
// each where six __shfl_xor steps are six dependent ds_bpermute round trips.  (Every lane of the wave must be active.)
__device__ __forceinline__ float wave_sum_all(float x) {
#define RAMD_DPP_ADD_(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
  RAMD_DPP_ADD_(x, 0xB1);  /* quad_perm [1, 0, 3, 2] */
  RAMD_DPP_ADD_(x, 0x4E);  /* quad_perm [2, 3, 0, 1] */
  RAMD_DPP_ADD_(x, 0x141); /* row_half_mirror */
  RAMD_DPP_ADD_(x, 0x140); /* row_mirror */
#undef RAMD_DPP_ADD_
  const int xi = __builtin_bit_cast(int, x);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
  return (r0 + r1) + (r2 + r3);
}

// deterministic block-wide sum (fixed tree), blockDim.x == 256
__device__ __forceinline__ float block_sum_256(float v, float *red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float4 ld4(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// recur-rng.h:22-31 (Jenkins small fast PRNG, 64 bit) and 179-200 (the sum of twelve
// 16-bit fields), bit for bit: the noise a stream gets must be the one the reference
// would draw from that stream's generator.
struct DevRng {
  unsigned long long a, b, c, d;
};
/* The generator's recurrence is ONE lane's dependent instruction stream (a wave of 64 streams is alone on its SIMD's
 * issue slot), so what a draw costs is its instruction count times the instructions' issue cycles.  The state is
 * therefore worked on as 32-bit halves: rotations are two v_alignbit_b32 each, additions and subtractions carry pairs
 * (v_add_co_u32 + v_addc_co_u32, full rate) -- as 64-bit integers hipcc emits v_lshl_add_u64 for every addition,
 * which issues at a quarter of the rate: the pass's noise (3 x 1027 draws per stream) took 211 us, 493 clocks per
 * value, against 225 for the instruction count of this form. */
struct Rng32 {
  unsigned al, ah, bl, bh, cl, ch, dl, dh;
};
__device__ __forceinline__ Rng32 rng_split(const DevRng &x) {
  Rng32 r = {(unsigned)x.a, (unsigned)(x.a >> 32), (unsigned)x.b, (unsigned)(x.b >> 32),
             (unsigned)x.c, (unsigned)(x.c >> 32), (unsigned)x.d, (unsigned)(x.d >> 32)};
  return r;
}
__device__ __forceinline__ void rng_join(DevRng &x, const Rng32 &r) {
  x.a = ((unsigned long long)r.ah << 32) | r.al;
  x.b = ((unsigned long long)r.bh << 32) | r.bl;
  x.c = ((unsigned long long)r.ch << 32) | r.cl;
  x.d = ((unsigned long long)r.dh << 32) | r.dl;
}
__device__ __forceinline__ void add64_pair(unsigned &rl, unsigned &rh, unsigned al, unsigned ah, unsigned bl, unsigned bh) {
  asm("v_add_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, %4, %5, vcc"
      : "=&v"(rl), "=v"(rh)
      : "v"(al), "v"(bl), "v"(ah), "v"(bh)
      : "vcc");
}
__device__ __forceinline__ void sub64_pair(unsigned &rl, unsigned &rh, unsigned al, unsigned ah, unsigned bl, unsigned bh) {
  asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_subb_co_u32 %1, vcc, %4, %5, vcc"
      : "=&v"(rl), "=v"(rh)
      : "v"(al), "v"(bl), "v"(ah), "v"(bh)
      : "vcc");
}
/* (lo, hi) rotated left by K, K a compile-time constant in 1..63 other than 32 */
template <int K>
__device__ __forceinline__ void rotl64_pair(unsigned &rl, unsigned &rh, unsigned lo, unsigned hi) {
  if (K > 32) {
    const unsigned t = lo;
    lo = hi;
    hi = t;
  }
  constexpr int k = K > 32 ? K - 32 : K;
  rh = __builtin_amdgcn_alignbit(hi, lo, 32 - k);
  rl = __builtin_amdgcn_alignbit(lo, hi, 32 - k);
}
/* recur-rng.h:22-31: the draw's value is the new d (r.dl, r.dh) */
__device__ __forceinline__ void dev_rand64_pair(Rng32 &r) {
  unsigned tl, th, el, eh, nal, nah, nbl, nbh, ncl, nch;
  rotl64_pair<7>(tl, th, r.bl, r.bh);
  sub64_pair(el, eh, r.al, r.ah, tl, th); /* e = a - rot(b, 7) */
  rotl64_pair<13>(tl, th, r.cl, r.ch);
  nal = r.bl ^ tl; /* a = b ^ rot(c, 13) */
  nah = r.bh ^ th;
  rotl64_pair<37>(tl, th, r.dl, r.dh);
  add64_pair(nbl, nbh, r.cl, r.ch, tl, th); /* b = c + rot(d, 37) */
  add64_pair(ncl, nch, r.dl, r.dh, el, eh); /* c = d + e */
  add64_pair(r.dl, r.dh, el, eh, nal, nah); /* d = e + a */
  r.al = nal;
  r.ah = nah;
  r.bl = nbl;
  r.bh = nbh;
  r.cl = ncl;
  r.ch = nch;
}
__device__ __forceinline__ unsigned long long dev_rand64(DevRng &x) {
  Rng32 r = rng_split(x);
  dev_rand64_pair(r);
  rng_join(x, r);
  return x.d;
}
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
/* recur-rng.h:179-201: the sum of the twelve 16-bit fields of three draws.  The sum fits 20 bits, so it is kept in
 * 32 bits and each draw's four fields are two v_dot2_u32_u16 with (1, 1).  The division by 65535 is the reference's
 * (correctly rounded): x * RN(1 / 65535) with one fma correction of the remainder gives the same float for every one
 * of the 786,421 possible sums (checked exhaustively against x / 65535.0f on the host), in three instructions
 * instead of the division's twelve. */
__device__ __forceinline__ float dev_cheap_gaussian_pair(Rng32 &r) {
  unsigned a = 0;
  const u16x2_t ones = {1, 1};
#pragma unroll
  for (int w = 0; w < 3; w++) {
    dev_rand64_pair(r);
    a = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, r.dl), ones, a, false);
    a = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, r.dh), ones, a, false);
  }
  const float x = (float)((int)a - 0xffff * 6);
  const float rc = 1.0f / 65535.0f; /* (a constant: RN(1 / 65535)) */
  const float q0 = __fmul_rn(x, rc);
  const float rem = __builtin_fmaf(-q0, 65535.0f, x);
  return __builtin_fmaf(rem, rc, q0);
}
__device__ __forceinline__ float dev_cheap_gaussian(DevRng &x) {
  Rng32 r = rng_split(x);
  const float v = dev_cheap_gaussian_pair(r);
  rng_join(x, r);
  return v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int CM = 32, CN = 32, CK = 128;
constexpr int C_STAGES = 3;                    /* LDS ring: 2 stages in flight + 1 being read (a 4th buys nothing) */
constexpr int C_STAGE_FLOATS = (CM + CN) * CK; /* 32 KB */

__device__ __forceinline__ uint32_t lds_byte_addr(const void *p) {
  return (uint32_t)(uintptr_t)(lds_void_t *)p;
}
__device__ __forceinline__ f32x4 lds_read_b128(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

/* [k][col] and [k + 1][col] of a 64-column K-major stage; _hi: k + 2, k + 3 (offsets in dwords) */
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lds_read2_b32_w64(uint32_t addr) {
  f32x2 v;
  asm volatile("ds_read2_b32 %0, %1 offset1:64" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ f32x2 lds_read2_b32_w64_hi(uint32_t addr) {
  f32x2 v;
  asm volatile("ds_read2_b32 %0, %1 offset0:128 offset1:192" : "=v"(v) : "v"(addr));
  return v;
}

// Loads through pointers that hipcc cannot prove global (the View read from device memory instead of coming as a
// kernel argument): as generic pointers they become FLAT loads, which count on vmcnt AND lgkmcnt and may return out of
// order with other memory instructions -- every wait behind one is a wait for everything.  Say that they are global.
template <class T> __device__ __forceinline__ const __attribute__((address_space(1))) T *as_global(const T *p) {
  return (const __attribute__((address_space(1))) T *)p;
}
template <class T> __device__ __forceinline__ __attribute__((address_space(1))) T *as_global(T *p) {
  return (__attribute__((address_space(1))) T *)p;
}
__device__ __forceinline__ float4 ld4g(const void *p) {
  const f32x4 q = *(const __attribute__((address_space(1))) f32x4 *)p;
  return make_float4(q.x, q.y, q.z, q.w);
}

// one LDS-DMA piece (64 lanes x 16 bytes, L1 bypassed) with a wave-uniform global base, a
// per-lane byte offset and a wave-uniform LDS destination: no vector-ALU instruction at all
// (the builtin form computes a 64-bit per-lane address first)
/* a pointer the compiler cannot prove wave-uniform (it went through a lambda's captures), for an "s" asm operand */
__device__ __forceinline__ const char *uniform_ptr(const void *p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (const char *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void lds_dma16_sc1(const void *sbase, unsigned voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

__device__ __forceinline__ void lds_dma16(const void *sbase, unsigned voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void lds_dma4(const void *sbase, unsigned voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

// --- inline-asm memory helpers of the one-launch chain.  hipcc neither sees nor waits for these
// accesses: every use is followed by an explicit s_waitcnt that names the registers it protects.
__device__ __forceinline__ void g_store_saddr(unsigned voff, float val, const void *sbase) {
  asm volatile("global_store_dword %0, %1, %2" : : "v"(voff), "v"(val), "s"(sbase) : "memory");
}
__device__ __forceinline__ float g_load_saddr(unsigned voff, const void *sbase) {
  float r;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
template <int OFF> __device__ __forceinline__ f32x4 lds_read_b128_off(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ float lds_read_b32_off(uint32_t addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
/* wait until at most N LDS operations issued after `v`'s read are outstanding (they return in order) */
template <int N> __device__ __forceinline__ void lgkm_wait(f32x4 &v) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}
/* two floats 64 * O0 and 64 * O1 floats from `addr` (bytes) */
template <int O0, int O1> __device__ __forceinline__ f32x2 lds_read2st64(uint32_t addr) {
  static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_read2st64_b32 offsets are 8 bits");
  f32x2 v;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
  return v;
}
/* wait until at most N LDS instructions issued after a fragment's reads are outstanding */
template <int N> __device__ __forceinline__ void frag_wait(f32x4 &cf, f32x2 (&a)[2][2], f32x2 (&e)[2][2]) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%9)"
               : "+v"(cf), "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(e[0][0]), "+v"(e[0][1]),
                 "+v"(e[1][0]), "+v"(e[1][1])
               : "n"(N));
}
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// the sum of ks planes in plane order, B loads in flight at a time
template <int B = 8>
__device__ __forceinline__ float4 sum_planes(const float *src, size_t stride, int ks) {
  float4 sum = zero4();
  for (int z0 = 0; z0 < ks; z0 += B) {
    float4 t[B];
#pragma unroll
    for (int z = 0; z < B; z++) t[z] = ld4(src + (size_t)(z0 + z < ks ? z0 + z : z0) * stride);
#pragma unroll
    for (int z = 0; z < B; z++)
      if (z0 + z < ks) { sum.x += t[z].x; sum.y += t[z].y; sum.z += t[z].z; sum.w += t[z].w; }
  }
  return sum;
}

static inline View make_view(const RamdShape *sh, const RamdBuffers *b) {
  View v;
  v.sh = *sh;
  v.b = *b;
  return v;
}

// segments of the hidden row in the output-layer kernels (k_out_layer, k_text_top)
constexpr int OUT_SEGS = 16;
// 64 x 64 tiles of the wide chain step and the wide forward GEMM (k_chain_wide, k_fwd_wide)
constexpr int WM = 64, WN = 64, WK = 64, W_STAGES = 4;
constexpr int W_STAGE_FLOATS = (WM + WN) * WK; /* 32 KB */

// ---- development builds (tools/mkabl.sh bnd -DBND_STAMPS -- NOT with -DPC_STAMPS: the half-step stamps of one workgroup slow its
// whole row tile down, 6.5 us per chain launch): when every workgroup of a launch starts and ends, by the 100 MHz
// clock all CUs share -- tools/gpu_boundary_stamps.py turns the four launches' marks into what the launch boundaries of a
// generation cost WITHOUT a profiler's per-dispatch instrumentation between them
#ifdef BND_STAMPS
#define BND_DECL(sym, reader)                                                                  \
  __device__ unsigned long long sym[2][1024];                                                  \
  extern "C" void reader(unsigned long long *out) {                                            \
    HIP_CHECK(hipDeviceSynchronize());                                                         \
    HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(sym), sizeof(unsigned long long) * 2048));   \
  }
#define BND_MARK(sym, which) do { if (threadIdx.x == 0 && blockIdx.x < 1024) sym[which][blockIdx.x] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BND_DECL(sym, reader)
#define BND_MARK(sym, which) do { } while (0)
#endif

// ---- launch-side support (kernels_support.hip) ----
// HIP-event timing of the kernel classes (bench.py's roofline leg)
enum { T_CHAIN = 0, T_DELTA = 1, T_FWD = 2, T_APPLY = 3, T_OTHER = 4, T_XCHG = 5, T_CLASSES = 6 }; /* (T_XCHG = RAMD_T_XCHG: the exchange between ranks, bracketed by rnn_core.c) */
#define RAMD_LOCAL __attribute__((visibility("hidden"))) /* shared by the kernel files, not exported */
RAMD_LOCAL int timing_begin(hipStream_t st, int cls, int count = 1);
RAMD_LOCAL void timing_end(hipStream_t st, int i);
// Tuning knobs (RECUR_AMD_*): read from the environment once, then frozen
RAMD_LOCAL int env_int(const char *name, int dflt);

// ---- the BPTT chain (kernels_chain.hip) ----
// the device copy of the View for the kernels that take it by pointer
RAMD_LOCAL const View *device_view(hipStream_t st, const View &v);
// All D steps of the chain for streams [row0, row0 + nrows): the one-launch chain where it applies,
// otherwise a launch per step.  Returns the number of partial sums of squares per (step, stream) it
// left for the extras kernels (0: the extras sum the rows themselves).
/* The top layer's weight delta (single_layer_sgd, recur-nn.c:256-301, for all streams at once: dst[h][o] =
 * sum over streams [row0, row0 + nrows) of hidden[s][h] * o_error[s][o], streams with active[s - row0] == 0
 * left out) as a request to whoever has idle vector ALUs before the weight deltas are needed: the one-launch
 * chain forms it while its weight panels are on their way (chain_ho_delta) and sets `done`. */
struct HoWork {
  float *dst; /* [H][O], overwritten: a slab plane for the optimiser to take, or ho_delta itself */
  const unsigned char *active;
  int row0, nrows;
  int done;
  int workers, idle_only; /* (the launch's own: how many of its workgroups share the rows; only those without other work) */
};
#ifndef HO_BATCH
#define HO_BATCH 8 /* streams whose loads are in flight together, per thread: all of a 256-stream set's */
#endif
/* HO_HR: rows per workgroup at most (5: h_size <= 1280 over 256 workgroups; 9: h_size <= 1152 over the 128 or more
 * that have no chain work when the set is small) */
template <bool MASK, int HO_HR, int BATCH>
__device__ __forceinline__ void chain_ho_sum(const View &v, const HoWork &hw, int h0, int HR, int q4, int g,
                                             float4 (&acc)[HO_HR]) {
  const RamdShape &s = v.sh;
  const float *hp = v.b.hidden + (size_t)hw.row0 * s.H;
  const float *ep = v.b.o_error + (size_t)hw.row0 * s.O + 4 * q4;
  int hc[HO_HR]; /* clamped row indices: every load from a valid address, the value selected afterwards */
#pragma unroll
  for (int rr = 0; rr < HO_HR; rr++) hc[rr] = (rr < HR && h0 + rr < s.H) ? h0 + rr : 0;
  for (int s0 = g; s0 < hw.nrows; s0 += BATCH * 32) {
    float hv[BATCH][HO_HR];
    float4 e4[BATCH];
    unsigned char am[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; i++) {
      const int ss = s0 + i * 32, sc = ss < hw.nrows ? ss : s0;
      e4[i] = ld4(ep + (size_t)sc * s.O);
#pragma unroll
      for (int rr = 0; rr < HO_HR; rr++) hv[i][rr] = hp[(size_t)sc * s.H + hc[rr]];
      am[i] = MASK ? hw.active[sc] : (unsigned char)1;
    }
#pragma unroll
    for (int i = 0; i < BATCH; i++) {
      const bool keep = s0 + i * 32 < hw.nrows && am[i] != 0;
#pragma unroll
      for (int rr = 0; rr < HO_HR; rr++) {
        const float x = keep ? hv[i][rr] : 0.0f;
        acc[rr].x += x * e4[i].x;
        acc[rr].y += x * e4[i].y;
        acc[rr].z += x * e4[i].z;
        acc[rr].w += x * e4[i].w;
      }
    }
  }
}
/* the top layer's update for the rows a workgroup has just summed (recur-nn.c:482-487, 653-676), or w == nullptr */
struct HoApply {
  float *w, *m;
  float rate, momentum, mw;
};
/* `rank`: this workgroup's number among the hw.workers that share the rows */
template <int HO_HR, int BATCH>
__device__ __forceinline__ void chain_ho_delta(const View &v, const HoWork &hw, float *lds, int rank, HoApply ap = HoApply{}) {
  const RamdShape &s = v.sh;
  const int H = s.H, O = s.O, OQ = O >> 2; /* OQ <= 12 (launcher) */
  const int HR = (H + hw.workers - 1) / hw.workers; /* rows per workgroup, <= HO_HR (launcher) */
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q4 = lane & 15, gl = lane >> 4;
  const int h0 = rank * HR;
  float4 acc[HO_HR];
#pragma unroll
  for (int rr = 0; rr < HO_HR; rr++) acc[rr] = zero4();
  if (h0 < H) { /* (the whole workgroup) */
    const int q4c = q4 < OQ ? q4 : 0; /* lanes 12-15 of a group repeat quad 0 and are not stored */
    if (hw.active) chain_ho_sum<true, HO_HR, BATCH>(v, hw, h0, HR, q4c, 4 * wave + gl, acc);
    else chain_ho_sum<false, HO_HR, BATCH>(v, hw, h0, HR, q4c, 4 * wave + gl, acc);
#pragma unroll
    for (int rr = 0; rr < HO_HR; rr++) {
      acc[rr].x += __shfl_xor(acc[rr].x, 16, 64); acc[rr].x += __shfl_xor(acc[rr].x, 32, 64);
      acc[rr].y += __shfl_xor(acc[rr].y, 16, 64); acc[rr].y += __shfl_xor(acc[rr].y, 32, 64);
      acc[rr].z += __shfl_xor(acc[rr].z, 16, 64); acc[rr].z += __shfl_xor(acc[rr].z, 32, 64);
      acc[rr].w += __shfl_xor(acc[rr].w, 16, 64); acc[rr].w += __shfl_xor(acc[rr].w, 32, 64);
    }
    if (gl == 0) {
#pragma unroll
      for (int rr = 0; rr < HO_HR; rr++) *reinterpret_cast<float4 *>(lds + 4 * ((wave * HO_HR + rr) * 16 + q4)) = acc[rr];
    }
  }
  /* (the update's weights and momentum, requested before the barrier: they are there when the sums are) */
  const bool fin = tid < HO_HR * 16 && (tid >> 4) < HR && h0 + (tid >> 4) < H && (tid & 15) < OQ;
  const size_t fin_off = fin ? (size_t)(h0 + (tid >> 4)) * O + 4 * (tid & 15) : 0;
  float4 W4 = zero4(), M4 = zero4();
  if (fin && ap.w) {
    W4 = ld4(ap.w + fin_off);
    M4 = ld4(ap.m + fin_off);
  }
  __syncthreads();
  if (fin) {
    const int rr = tid >> 4, q = tid & 15;
    float4 sum = *reinterpret_cast<const float4 *>(lds + 4 * (rr * 16 + q));
    for (int w = 1; w < 8; w++) {
      const float4 t = *reinterpret_cast<const float4 *>(lds + 4 * ((w * HO_HR + rr) * 16 + q));
      sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
    }
    *reinterpret_cast<float4 *>(hw.dst + fin_off) = sum;
    if (ap.w) {
      float wv[4] = {W4.x, W4.y, W4.z, W4.w}, mv[4] = {M4.x, M4.y, M4.z, M4.w};
      const float dv[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float t = dv[i] * ap.rate, mm = mv[i];
        wv[i] += t + mm * ap.mw;
        mv[i] = (mm + t) * ap.momentum;
      }
      *reinterpret_cast<float4 *>(ap.w + fin_off) = make_float4(wv[0], wv[1], wv[2], wv[3]);
      *reinterpret_cast<float4 *>(ap.m + fin_off) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    }
  }
  __syncthreads(); /* `lds` is the caller's again */
}


/* The extras of every step (column 0 and the input columns of each step's error) and the per-stream control
 * logic (k_extras_control's work, k_extras.h) as a request of the same kind: the one-launch chain runs them in
 * its tail, each workgroup for the stream(s) of its row tile that its column tile number names, from error
 * planes that are still in its XCD's L2 -- no launch and no second pass over HBM.  `done` is set when the
 * launch stands. */
struct XcWork {
  int on;
  int row0;   /* first row of the CALL (the launch may start lower: windowed) -- `active` is indexed from it */
  int nx, nxp;
  const unsigned char *active;
  unsigned flags;
  int done;
  int dense; /* the inputs are dense (every input column counts): extras_dense_tail instead of the gather form */
};
RAMD_LOCAL int ramd_chain_steps(hipStream_t st, const View &v, const RamdShape *sh, const RamdBuffers *b,
                                int row0, int nrows, HoWork *ho, XcWork *xc);
