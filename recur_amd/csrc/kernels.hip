// kernels.hip -- CDNA4 (gfx950) kernels of the Recur RNN core and their thin
// C-ABI launchers (ramd_internal.h).  Written for wave64 / MFMA / LDS directly;
// there is no other backend.
//
// Heavy lifting is three fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32), all with
// M = "streams of the synchronic mini-batch":
//   forward   Hpre[S x H] = X[S x I] . W_ih[I x H]              (recur-nn.c:18-48, 117)
//   chain     E_i[S x I]  = E_h[S x H] . W_ih^T   per BPTT step (recur-nn.c:338-376)
//   delta     dW[I x H]   = sum_t X_t^T . diag(c_t) . E_h,t      (recur-nn.c:344-356, 738)
// Each GEMM is split along K over blockIdx.z into fp32 slabs; a small finalize
// kernel sums the slabs in a fixed order (deterministic) and applies the
// reference's elementwise rule (activation / zero-row mask + sum of squares /
// accumulate).  The top layer (O is 4..44 in the text and classify configs) and
// the per-stream control logic are plain VALU kernels.
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

// deterministic block-wide sum (fixed tree), blockDim.x == 256
__device__ __forceinline__ float block_sum_256(float v, float *red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------ K0: advance --

// rnn_bptt_advance (recur-nn.c:696-704) for a range of training streams
__global__ void k_advance(View v, int row0, int nrows) {
  int r = row0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (r < row0 + nrows && r < v.sh.Scap) {
    int i = v.b.idx[r] + 1;
    if (i == v.sh.D) i -= v.sh.D;
    v.b.idx[r] = i;
  }
}

// ----------------------------------------------------------- K4: assemble --

// Builds the input row of each stream: previous hiddens, bias, real inputs
// (recur-nn.c:104-112) and the emergency soft clip of the whole row
// (maybe_scale_inputs, recur-nn.c:68-81).  One workgroup per stream.
__global__ __launch_bounds__(256) void k_assemble(View v, int row0, int mode,
                                                  const float *dense, int ld, int text_i,
                                                  int global_first, int n_set, int advance) {
  __shared__ float red[4];
  const RamdShape &s = v.sh;
  int j = blockIdx.x;
  int r = row0 + j;
  float *slot;
  if (r < s.Scap) {
    int i = v.b.idx[r];
    if (advance) { /* rnn_bptt_advance (recur-nn.c:696-704) for this stream, done here */
      i = (i + 1 == s.D) ? 0 : i + 1;
      __syncthreads(); /* every thread has read the old index */
      if (threadIdx.x == 0) v.b.idx[r] = i;
    }
    slot = v.b.arena + ((size_t)i * s.Scap + r) * s.I;
  } else {
    slot = input_row(v, r, 0);
  }
  const float *hid = v.b.hidden + (size_t)r * s.H;
  int off = s.hidden_size + 1;
  int hot = -1;
  if (mode == RAMD_IN_ONE_HOT) {
    hot = v.b.hot[r];
  } else if (mode == RAMD_IN_TEXT) {
    // charmodel-predict.c:273, 295-298
    int len = v.b.text_len;
    int spacing = (len - 1) / n_set;
    int o = text_i + (global_first + j) * spacing;
    if (o >= len - 1) o -= len - 1;
    hot = v.b.text[o];
    if (threadIdx.x == 0) v.b.target[r] = v.b.text[o + 1];
  }
  float sum = 0.0f;
  for (int i = threadIdx.x; i < s.I; i += 256) {
    float x;
    if (i == 0) {
      x = 1.0f;
    } else if (i < off) {
      x = hid[i];
    } else if (i < off + s.input_size) {
      int k = i - off;
      if (mode == RAMD_IN_KEEP) x = slot[i];
      else if (mode == RAMD_IN_DENSE) x = dense[(size_t)j * ld + k];
      else x = (k == hot) ? 1.0f : 0.0f;
    } else {
      x = slot[i]; /* padding: stays as it is (zero) */
    }
    slot[i] = x;
    sum += x;
  }
  sum = block_sum_256(sum, red);
  float softclip = s.I * INPUT_MEAN_SOFT_TOP_F;
  if (sum > softclip) {
    float scale = soft_clip_dev(sum, softclip);
    for (int i = threadIdx.x; i < s.I; i += 256) slot[i] *= scale;
  }
}

// ------------------------------------------------------------- MFMA GEMM --
//
// Workgroup = 256 threads = 4 waves in a 2 x 2 grid; wave tile 32 x 32 (one
// v_mfma_f32_32x32x2_f32 accumulator of 16 VGPRs), workgroup tile 64 x 64,
// K tile 32.  Operand tiles go global -> registers -> LDS (double buffered, one
// barrier per K tile).  An operand whose global image is K-contiguous ("KC":
// rows of A / rows of W) sits in LDS as [row][32 + 4] and a lane fetches the
// four k it feeds to four consecutive MFMAs with one ds_read_b128 (row stride
// 36 dwords keeps the 16-lane groups of ds_read_b128 conflict free).  An
// operand whose global image is K-major ("KM": K rows of contiguous m) sits as
// [k][64] and is fetched with four conflict-free ds_read_b32.
//
// MFMA operand maps (f32 32x32x2): lane l supplies A[m = l & 31][k = l >> 5]
// and B[k = l >> 5][n = l & 31]; D register g holds row (g & 3) + 8 (g >> 2) +
// 4 (l >> 5), column l & 31.  Within a group of 8 k, MFMA j uses
// k = 8 g + 4 (l >> 5) + j on both operands.
//
// Global loads are unconditional (out-of-range lanes read a clamped, valid
// address) and everything that depends on the loaded value -- zero fill, the
// h_error mask, the per-stream coefficient -- is applied when the registers
// are written to LDS, i.e. after the MFMAs of the current tile, so the loads
// of tile k+1 stay in flight across the compute of tile k.
//
// Grid: 1-D.  Blocks are dealt round-robin over the 8 XCDs, so block id L runs
// on the XCD labelled L % 8.  The tiles that share a B panel (same n tile and K
// range, different m tile) are given ids with equal L % 8 and therefore meet in
// one XCD's L2 (MI355X_MICROARCH.md "Workgroup dispatch"; speed only).

constexpr int BM = 64, BN = 64, BK = 32, LDK = BK + 4;
constexpr int RAMD_MAX_REST_PLANES = 64;

struct GemmOut {
  float *slab;   // [KS][M][ldc]
  int M, N, ldc;
  size_t zs;     // floats between the planes of two K slices (M * ldc unless the planes are compact)
  int nkt;       // K tiles in total
  int tm, tn, ks;
  int col0;      // first output column (tiles start here; columns below are not produced)
  int row0m;     // first output row (k_gemm only; rows below are not produced)
};

__device__ __forceinline__ float4 ld4(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

struct Raw {
  float4 v;
  float aux;
};

// `live` is wave-uniform.  A dead load still issues (from one fixed valid
// address, so it costs a single cached request): every path then issues the same
// number of loads and hipcc can keep counted s_waitcnt vmcnt(N) across the
// pipeline instead of draining it.
template <bool KM, class Prob, bool IS_A>
__device__ __forceinline__ void tile_load(const Prob &p, int kt, bool live, int base,
                                          Raw (&reg)[2]) {
  const float *dummy = p.v.b.slab;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    int idx = threadIdx.x + i * 256;
    int x0 = KM ? (idx >> 4) : base + (idx >> 3);       /* KM: k row     KC: row      */
    int x1 = KM ? base + 4 * (idx & 15) : 4 * (idx & 7); /* KM: column    KC: k in tile */
    const float *src = IS_A ? p.a_ptr(kt, x0, x1) : p.b_ptr(kt, x0, x1);
    reg[i].v = ld4(live ? src : dummy);
    if (!IS_A && Prob::B_AUX) {
      const float *ax = p.b_aux_ptr(kt, x0);
      reg[i].aux = *(live ? ax : dummy);
    }
  }
}

template <bool KM, class Prob, bool IS_A>
__device__ __forceinline__ void tile_store(const Prob &p, int kt, int base, float *lds,
                                           const Raw (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; i++) {
    int idx = threadIdx.x + i * 256;
    int x0 = KM ? (idx >> 4) : base + (idx >> 3);
    int x1 = KM ? base + 4 * (idx & 15) : 4 * (idx & 7);
    float4 v = IS_A ? p.a_fix(kt, x0, x1, reg[i]) : p.b_fix(kt, x0, x1, reg[i]);
    float *dst = KM ? lds + (idx >> 4) * BM + 4 * (idx & 15) : lds + (idx >> 3) * LDK + 4 * (idx & 7);
    *reinterpret_cast<float4 *>(dst) = v;
  }
}

// the 4 values (k = 8 g + 4 kh + 0..3) of row/column `rc` for this lane
template <bool KM>
__device__ __forceinline__ float4 frag_read(const float *lds, int rc, int g, int kh) {
  if (KM) {
    const float *p = lds + (8 * g + 4 * kh) * BM + rc;
    return make_float4(p[0], p[BM], p[2 * BM], p[3 * BM]);
  }
  return *reinterpret_cast<const float4 *>(lds + rc * LDK + 8 * g + 4 * kh);
}

// Register prefetch depth: tiles k+1 .. k+PF are in flight (in registers or on
// their way) while tile k is being multiplied.  One K tile is 1024 MFMA cycles
// of work per wave but a load round trip under load is 2-3x that, so a single
// tile of look-ahead leaves the kernel latency bound (measured: 13 us for a
// 3.7 us chain step); four tiles per workgroup and two or three workgroups per
// CU keep roughly 100-200 KB in flight per CU.
constexpr int PF = 4;

template <bool A_KM, bool B_KM> struct GemmLds {
  static constexpr int A_FLOATS = A_KM ? BK * BM : BM * LDK;
  static constexpr int B_FLOATS = B_KM ? BK * BN : BN * LDK;
  static constexpr int STAGE = A_FLOATS + B_FLOATS;
};

template <bool A_KM, bool B_KM, class Prob>
__device__ __forceinline__ void gemm_body(const Prob &p, const GemmOut &o, const int L,
                                          float (*lds)[GemmLds<A_KM, B_KM>::STAGE]) {
  constexpr int A_FLOATS = GemmLds<A_KM, B_KM>::A_FLOATS;
  // block id -> (m tile, panel = (n tile, K slice)), XCD aware
  const int xcd = L & 7, q = L >> 3;
  const int mt = q % o.tm, panel = (q / o.tm) * 8 + xcd;
  if (panel >= o.tn * o.ks) return;
  const int nt = panel % o.tn, z = panel / o.tn;
  const int m0 = o.row0m + mt * BM, n0 = o.col0 + nt * BN;
  const int kt0 = (int)(((long)o.nkt * z) / o.ks), kt1 = (int)(((long)o.nkt * (z + 1)) / o.ks);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lm = lane & 31, kh = lane >> 5;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;

  Raw ra[PF][2], rb[PF][2];
#pragma unroll
  for (int j = 0; j < PF; j++) {
    {
      const bool live = kt0 + j < kt1;
      tile_load<A_KM, Prob, true>(p, kt0 + j, live, m0, ra[j]);
      tile_load<B_KM, Prob, false>(p, kt0 + j, live, n0, rb[j]);
    }
  }
  if (kt0 < kt1) {
    tile_store<A_KM, Prob, true>(p, kt0, m0, lds[0], ra[0]);
    tile_store<B_KM, Prob, false>(p, kt0, n0, lds[0] + A_FLOATS, rb[0]);
  }
  __syncthreads();
  for (int ktb = kt0; ktb < kt1; ktb += PF) {
#pragma unroll
    for (int j = 0; j < PF; j++) {
      const int kt = ktb + j;
      if (kt >= kt1) break;
      constexpr int dummy = 0;
      (void)dummy;
      const int cur = j & 1; /* PF is even, so the LDS buffer parity follows j */
      /* register set j held tile kt (already in LDS); refill it with tile kt + PF */
      {
        const bool live = kt + PF < kt1;
        tile_load<A_KM, Prob, true>(p, kt + PF, live, m0, ra[j]);
        tile_load<B_KM, Prob, false>(p, kt + PF, live, n0, rb[j]);
      }
      /* pin the issue order: hipcc otherwise sinks the loads below the MFMAs
       * and then waits for them at once (seen in the .s) */
      __builtin_amdgcn_sched_barrier(0);
      const float *la = lds[cur], *lb = lds[cur] + A_FLOATS;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float4 a = frag_read<A_KM>(la, wm * 32 + lm, g, kh);
        float4 b = frag_read<B_KM>(lb, wn * 32 + lm, g, kh);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < kt1) { /* tile kt + 1 sits in register set (j + 1) % PF */
        tile_store<A_KM, Prob, true>(p, kt + 1, m0, lds[cur ^ 1], ra[(j + 1) % PF]);
        tile_store<B_KM, Prob, false>(p, kt + 1, n0, lds[cur ^ 1] + A_FLOATS, rb[(j + 1) % PF]);
      }
      __syncthreads();
    }
  }
  float *c = o.slab + (size_t)z * o.zs;
  const int col = n0 + wn * 32 + lm;
  if (col < o.N) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      int row = m0 + wm * 32 + (g & 3) + 8 * (g >> 2) + 4 * kh;
      if (row < o.M) c[(size_t)row * o.ldc + col] = acc[g];
    }
  }
}

template <bool A_KM, bool B_KM, class Prob>
__global__ __launch_bounds__(256) void k_gemm(Prob p, GemmOut o) {
  __shared__ __attribute__((aligned(16))) float lds[2][GemmLds<A_KM, B_KM>::STAGE];
  gemm_body<A_KM, B_KM, Prob>(p, o, blockIdx.x, lds);
}

// Two independent small GEMMs (both operands K-major) in one launch: workgroups below
// `first_b` work on problem A, the others on problem B.
template <class ProbA, class ProbB>
__global__ __launch_bounds__(256) void k_gemm_pair(ProbA pa, GemmOut oa, int first_b, ProbB pb,
                                                   GemmOut ob) {
  __shared__ __attribute__((aligned(16))) float lds[2][GemmLds<true, true>::STAGE];
  if ((int)blockIdx.x < first_b)
    gemm_body<true, true, ProbA>(pa, oa, blockIdx.x, lds);
  else
    gemm_body<true, true, ProbB>(pb, ob, blockIdx.x - first_b, lds);
}


// ---- 128 x 128 variant for two K-major operands (the weight-delta GEMM) ----
//
// Same structure as k_gemm, but each wave owns a 64 x 64 sub-tile as 2 x 2
// accumulators (64 VGPRs): every A fragment is reused for two B fragments and
// vice versa, so LDS reads, global loads and barriers per MFMA are halved or
// quartered, and the four independent accumulator chains keep the matrix pipe
// fed from a single wave.  Register prefetch depth 2.
constexpr int BM2 = 128, BN2 = 128, PF2 = 2;

template <class Prob>
__global__ __launch_bounds__(256) void k_gemm2(Prob p, GemmOut o) {
  constexpr int A_FLOATS = BK * BM2, B_FLOATS = BK * BN2;
  __shared__ __attribute__((aligned(16))) float lds[2][A_FLOATS + B_FLOATS];
  const int L = blockIdx.x;
  const int xcd = L & 7, q = L >> 3;
  const int mt = q % o.tm, panel = (q / o.tm) * 8 + xcd;
  if (panel >= o.tn * o.ks) return;
  const int nt = panel % o.tn, z = panel / o.tn;
  const int m0 = mt * BM2, n0 = o.col0 + nt * BN2;
  const int kt0 = (int)(((long)o.nkt * z) / o.ks), kt1 = (int)(((long)o.nkt * (z + 1)) / o.ks);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lm = lane & 31, kh = lane >> 5;
  const float *dummy = p.v.b.slab;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;

  Raw ra[PF2][4], rb[PF2][4];
  auto load = [&](int kt, bool live, Raw (&a)[4], Raw (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int idx = threadIdx.x + i * 256;
      int k = idx >> 5, c = 4 * (idx & 31);
      const float *sa = p.a_ptr(kt, k, m0 + c);
      const float *sb = p.b_ptr(kt, k, n0 + c);
      a[i].v = ld4(live ? sa : dummy);
      b[i].v = ld4(live ? sb : dummy);
      if (Prob::B_AUX) {
        const float *ax = p.b_aux_ptr(kt, k);
        b[i].aux = *(live ? ax : dummy);
      }
    }
  };
  auto store = [&](int kt, float *dst, const Raw (&a)[4], const Raw (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int idx = threadIdx.x + i * 256;
      int k = idx >> 5, c = 4 * (idx & 31);
      *reinterpret_cast<float4 *>(dst + k * BM2 + c) = p.a_fix(kt, k, m0 + c, a[i]);
      *reinterpret_cast<float4 *>(dst + A_FLOATS + k * BN2 + c) = p.b_fix(kt, k, n0 + c, b[i]);
    }
  };
#pragma unroll
  for (int j = 0; j < PF2; j++) load(kt0 + j, kt0 + j < kt1, ra[j], rb[j]);
  if (kt0 < kt1) store(kt0, lds[0], ra[0], rb[0]);
  __syncthreads();
  for (int ktb = kt0; ktb < kt1; ktb += PF2) {
#pragma unroll
    for (int j = 0; j < PF2; j++) {
      const int kt = ktb + j;
      if (kt >= kt1) break;
      const int cur = j & 1;
      load(kt + PF2, kt + PF2 < kt1, ra[j], rb[j]);
      __builtin_amdgcn_sched_barrier(0);
      const float *la = lds[cur], *lb = lds[cur] + A_FLOATS;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            a[i][jj] = la[(8 * g + 4 * kh + jj) * BM2 + wm * 64 + i * 32 + lm];
            b[i][jj] = lb[(8 * g + 4 * kh + jj) * BN2 + wn * 64 + i * 32 + lm];
          }
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int jn = 0; jn < 2; jn++)
              acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][jj], b[jn][jj], acc[i][jn], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < kt1) store(kt + 1, lds[cur ^ 1], ra[(j + 1) % PF2], rb[(j + 1) % PF2]);
      __syncthreads();
    }
  }
  float *c = o.slab + (size_t)z * o.zs;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jn = 0; jn < 2; jn++) {
      const int col = n0 + wn * 64 + jn * 32 + lm;
      if (col < o.N) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
          int row = m0 + wm * 64 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * kh;
          if (row < o.M) c[(size_t)row * o.ldc + col] = acc[i][jn][g];
        }
      }
    }
}

// zero column 0 and columns > hidden_size of an error row: what the reference
// does to h_error at the top of every BPTT step (recur-nn.c:334-337)
__device__ __forceinline__ float4 mask_herr(float4 v, int c, int hs) {
  if (c == 0) v.x = 0.0f;
  if (c + 0 > hs) v.x = 0.0f;
  if (c + 1 > hs) v.y = 0.0f;
  if (c + 2 > hs) v.z = 0.0f;
  if (c + 3 > hs) v.w = 0.0f;
  return v;
}

// Every Prob supplies, for A and for B: x_load (issue the global load of one
// float4, always from a valid address) and x_fix (what to do with the value
// once it has arrived).  (x0, x1) = (row, k in tile) for a KC operand and
// (k in tile, column) for a KM operand.

// forward: A = input rows (KC), B = W_ih [I][H] (KM); K = I
template <bool UNI> struct ProbFwd {
  static constexpr bool B_AUX = false;
  View v;
  int row0, nrows;
  __device__ const float *a_ptr(int kt, int row, int k) const {
    k += kt * BK;
    bool ok = row < nrows && k < v.sh.I;
    return input_row<UNI>(v, row0 + (row < nrows ? row : 0), 0) + (ok ? k : 0);
  }
  __device__ float4 a_fix(int kt, int row, int k, const Raw &r) const {
    return (row < nrows && k + kt * BK < v.sh.I) ? r.v : zero4();
  }
  __device__ const float *b_ptr(int kt, int k, int n) const {
    k += kt * BK;
    bool ok = k < v.sh.I && n < v.sh.H;
    return v.b.ih_w + (ok ? k * v.sh.H + n : 0);
  }
  __device__ float4 b_fix(int kt, int k, int n, const Raw &r) const {
    return (k + kt * BK < v.sh.I && n < v.sh.H) ? r.v : zero4();
  }
  __device__ const float *b_aux_ptr(int, int) const { return v.b.slab; }
};

// output layer: A = hidden rows (KC), B = W_ho [H][O] (KM); K = H
struct ProbOut {
  static constexpr bool B_AUX = false;
  View v;
  int row0, nrows;
  __device__ const float *a_ptr(int kt, int row, int k) const {
    k += kt * BK;
    bool ok = row < nrows && k < v.sh.H;
    return v.b.hidden + (ok ? (row0 + row) * v.sh.H + k : 0);
  }
  __device__ float4 a_fix(int kt, int row, int k, const Raw &r) const {
    return (row < nrows && k + kt * BK < v.sh.H) ? r.v : zero4();
  }
  __device__ const float *b_ptr(int kt, int k, int n) const {
    k += kt * BK;
    bool ok = k < v.sh.H && n < v.sh.O;
    return v.b.ho_w + (ok ? k * v.sh.O + n : 0);
  }
  __device__ float4 b_fix(int kt, int k, int n, const Raw &r) const {
    return (k + kt * BK < v.sh.H && n < v.sh.O) ? r.v : zero4();
  }
  __device__ const float *b_aux_ptr(int, int) const { return v.b.slab; }
};

// top-layer delta: ho_delta[H][O] += hidden^T . o_error over the streams.
// A[k = stream][m] = hidden (KM), B[k = stream][n] = o_error (KM); K = streams.
// `live` holds 1.0 per stream that takes part (the active mask), as floats.
struct ProbHoDelta {
  static constexpr bool B_AUX = true;
  View v;
  int row0, nrows;
  const float *live;
  __device__ const float *a_ptr(int kt, int k, int m) const {
    k += kt * BK;
    bool ok = k < nrows && m < v.sh.H;
    return v.b.hidden + (ok ? (row0 + k) * v.sh.H + m : 0);
  }
  __device__ float4 a_fix(int kt, int k, int m, const Raw &r) const {
    return (k + kt * BK < nrows && m < v.sh.H) ? r.v : zero4();
  }
  __device__ const float *b_ptr(int kt, int k, int n) const {
    k += kt * BK;
    bool ok = k < nrows && n < v.sh.O;
    return v.b.o_error + (ok ? (row0 + k) * v.sh.O + n : 0);
  }
  __device__ const float *b_aux_ptr(int kt, int k) const {
    k += kt * BK;
    return live + (k < nrows ? k : 0);
  }
  __device__ float4 b_fix(int kt, int k, int n, const Raw &r) const {
    return (k + kt * BK < nrows && n < v.sh.O && r.aux != 0.0f) ? r.v : zero4();
  }
};

// Chain "extras": the few columns of the input error that the next BPTT step
// never reads -- the bias row (column 0) and the real-input rows (columns above
// hidden_size) -- for ALL steps at once after the chain has run:
// M = (step, stream), N = 1 + i_size - 1 - hidden_size, K = H.
// A[m][k] = ehi[t][r][k] (KC), B[c][k] = W_ih[c ? hidden_size + c : 0][k] (KC).
struct ProbExtras {
  static constexpr bool B_AUX = false;
  View v;
  int row0, nrows, nx;
  __device__ const float *a_ptr(int kt, int m, int k) const {
    k += kt * BK;
    int M = v.sh.D * nrows;
    bool ok = m < M && k < v.sh.H;
    int mm = m < M ? m : 0;
    int t = mm / nrows, r = mm - t * nrows;
    return v.b.ehi + ((t * v.sh.Scap + row0 + r) * v.sh.I + (ok ? k : 0));
  }
  __device__ float4 a_fix(int kt, int m, int k, const Raw &r) const {
    return (m < v.sh.D * nrows && k + kt * BK < v.sh.H) ? r.v : zero4();
  }
  __device__ const float *b_ptr(int kt, int c, int k) const {
    k += kt * BK;
    bool ok = c < nx && k < v.sh.H;
    int n = (c == 0 || c >= nx) ? 0 : v.sh.hidden_size + c;
    return v.b.ih_w + (n * v.sh.H + (ok ? k : 0));
  }
  __device__ float4 b_fix(int kt, int c, int k, const Raw &r) const {
    return (c < nx && k + kt * BK < v.sh.H) ? r.v : zero4();
  }
  __device__ const float *b_aux_ptr(int, int) const { return v.b.slab; }
};

// delta: K runs over (step t, stream r) in tiles of 32 streams.
// A[k][m] = X_t[r][m] (KM), B[k][n] = coef[t][r] * masked ehi[t][r][n] (KM)
template <bool UNI> struct ProbDelta {
  static constexpr bool B_AUX = true;
  View v;
  int row0, nrows, rtiles;
  __device__ const float *a_ptr(int kt, int k, int m) const {
    int t = kt / rtiles, s = (kt - t * rtiles) * BK + k;
    bool ok = s < nrows && m < v.sh.I;
    return input_row<UNI>(v, row0 + (s < nrows ? s : 0), t) + (ok ? m : 0);
  }
  __device__ float4 a_fix(int kt, int k, int m, const Raw &r) const {
    int t = kt / rtiles, s = (kt - t * rtiles) * BK + k;
    if (!(s < nrows && m < v.sh.I)) return zero4();
    float4 x = r.v;
    if (v.sh.activation == 5) {
      /* RNN_RECLIP20: an input row at the ceiling is skipped like a zero one, its delta row too
       * (recur-nn.c:340-341) */
      x.x = x.x < 20.0f ? x.x : 0.0f;
      x.y = x.y < 20.0f ? x.y : 0.0f;
      x.z = x.z < 20.0f ? x.z : 0.0f;
      x.w = x.w < 20.0f ? x.w : 0.0f;
    }
    return x;
  }
  __device__ const float *b_ptr(int kt, int k, int n) const {
    int t = kt / rtiles, s = (kt - t * rtiles) * BK + k;
    bool ok = s < nrows && n < v.sh.H;
    int row = t * v.sh.Scap + row0 + (s < nrows ? s : 0);
    return v.b.ehi + (row * v.sh.I + (ok ? n : 0));
  }
  __device__ const float *b_aux_ptr(int kt, int k) const {
    int t = kt / rtiles, s = (kt - t * rtiles) * BK + k;
    return v.b.coef + (t * v.sh.Scap + row0 + (s < nrows ? s : 0));
  }
  __device__ float4 b_fix(int kt, int k, int n, const Raw &r) const {
    int t = kt / rtiles, s = (kt - t * rtiles) * BK + k;
    float c = r.aux;
    /* select, never multiply by zero: a step past the break may hold inf */
    if (!(s < nrows && n < v.sh.H) || c == 0.0f) return zero4();
    float4 x = mask_herr(r.v, n, v.sh.hidden_size);
    x.x *= c; x.y *= c; x.z *= c; x.w *= c;
    return x;
  }
};

// plain "sum the K slabs" finalize: dst[r][c] (+)= sum_z slab[z][r][c]
__global__ __launch_bounds__(256) void k_sum_slabs(float *dst, int ld_dst, const float *slab,
                                                   int M, int N, int ks, int accumulate) {
  int q = blockIdx.x * 256 + threadIdx.x;
  int per_row = N >> 2;
  if (q >= M * per_row) return;
  int r = q / per_row, c = (q - r * per_row) * 4;
  const float *p = slab + (size_t)r * N + c;
  float4 a = accumulate ? ld4(dst + (size_t)r * ld_dst + c) : zero4();
  for (int z = 0; z < ks; z++) {
    float4 t = ld4(p + (size_t)z * M * N);
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  }
  *reinterpret_cast<float4 *>(dst + (size_t)r * ld_dst + c) = a;
}

// ------------------------------------------------------ presynaptic noise --

// recur-rng.h:22-31 (Jenkins small fast PRNG, 64 bit) and 179-200 (the sum of twelve
// 16-bit fields), bit for bit: the noise a stream gets must be the one the reference
// would draw from that stream's generator.
struct DevRng {
  unsigned long long a, b, c, d;
};
/* k is a compile-time constant at every call site: two v_alignbit_b32 (full rate) instead of two
 * 64-bit shifts and an or (the 64-bit shifts are quarter rate, and the generator's recurrence is
 * a single lane's dependent instruction stream) */
__device__ __forceinline__ unsigned long long rotl64(unsigned long long x, int k) {
  unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
  if (k >= 32) {
    unsigned t = lo;
    lo = hi;
    hi = t;
    k -= 32;
  }
  if (k == 0) return ((unsigned long long)hi << 32) | lo;
  const unsigned nh = __builtin_amdgcn_alignbit(hi, lo, 32 - k), nl = __builtin_amdgcn_alignbit(lo, hi, 32 - k);
  return ((unsigned long long)nh << 32) | nl;
}
__device__ __forceinline__ unsigned long long dev_rand64(DevRng &x) {
  unsigned long long e = x.a - rotl64(x.b, 7);
  x.a = x.b ^ rotl64(x.c, 13);
  x.b = x.c + rotl64(x.d, 37);
  x.c = x.d + e;
  x.d = e + x.a;
  return x.d;
}
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
/* recur-rng.h:179-201: the sum of the twelve 16-bit fields of three draws.  The sum fits 20
 * bits, so it is kept in 32 bits and each draw's four fields are two v_dot2_u32_u16 with (1, 1)
 * -- this lane is alone on its SIMD's issue slot, every instruction saved is time saved */
__device__ __forceinline__ float dev_cheap_gaussian(DevRng &x) {
  unsigned a = 0;
  const u16x2_t ones = {1, 1};
#pragma unroll
  for (int w = 0; w < 3; w++) {
    unsigned long long bits = dev_rand64(x);
    a = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, (unsigned)bits), ones, a, false);
    a = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, (unsigned)(bits >> 32)), ones, a, false);
  }
  return (float)((int)a - 0xffff * 6) / (0xffff);
}

// MAYBE_ADD_ARRAY_NOISE on hidden[1..h_size) (recur-nn.c:120-121; the pad columns get
// noise too).  The generator is sequential per stream, so one thread
// walks each stream's row; the values are added to K slab 0 of the forward GEMM.
/* One lane per stream: the stream's generator is a sequential recurrence (three rand64 per
 * value).  The row is walked in pieces of 16 values whose old contents are requested BEFORE the
 * piece's 48 generator steps and added and stored after them, so the memory round trips sit in
 * the shadow of the recurrence (element by element, as a read-modify-write per value, this kernel
 * took 360 us for 256 streams of 1028 values). */
__global__ void k_presynaptic_noise(View v, int row0, int nrows, float deviation) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nrows) return;
  const RamdShape &s = v.sh;
  DevRng g = reinterpret_cast<DevRng *>(v.b.rng)[row0 + j];
  float *row = v.b.slab + (size_t)j * s.H;
  /* column 0 gets no noise (recur-nn.c:120-121: i from 1); h_size is a multiple of 4 */
  for (int i0 = 0; i0 < s.H; i0 += 16) {
    const int n4 = min(4, (s.H - i0) / 4);
    float4 old[4];
#pragma unroll
    for (int k = 0; k < 4; k++) old[k] = k < n4 ? ld4(row + i0 + 4 * k) : zero4();
    float nz[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = i0 + k;
      nz[k] = (i >= 1 && i < s.H) ? dev_cheap_gaussian(g) * deviation : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < n4)
        *reinterpret_cast<float4 *>(row + i0 + 4 * k) =
            make_float4(old[k].x + nz[4 * k], old[k].y + nz[4 * k + 1], old[k].z + nz[4 * k + 2], old[k].w + nz[4 * k + 3]);
  }
  reinterpret_cast<DevRng *>(v.b.rng)[row0 + j] = g;
}

/* The same values without touching anything: out[j][1..H) and the generator state after them
 * (see noise_speculate in rnn_core.c: runs on a second stream while the rest of the previous
 * generation is still being computed) */
__global__ void k_noise_speculate(View v, int row0, int nrows, float deviation, float *out, DevRng *state) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nrows) return;
  const RamdShape &s = v.sh;
  DevRng g = reinterpret_cast<DevRng *>(v.b.rng)[row0 + j];
  float *row = out + (size_t)j * s.H;
  for (int i0 = 0; i0 < s.H; i0 += 16) {
    float nz[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = i0 + k;
      nz[k] = (i >= 1 && i < s.H) ? dev_cheap_gaussian(g) * deviation : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i0 + 4 * k < s.H)
        *reinterpret_cast<float4 *>(row + i0 + 4 * k) = make_float4(nz[4 * k], nz[4 * k + 1], nz[4 * k + 2], nz[4 * k + 3]);
  }
  state[j] = g;
}
/* ... and their use by the forward pass: slab plane 0 += values, generators = the states after them */
__global__ __launch_bounds__(256) void k_noise_apply(View v, int row0, int nrows) {
  const RamdShape &s = v.sh;
  const int q = blockIdx.x * 256 + threadIdx.x, per_row = s.H >> 2;
  if (q >= nrows * per_row) return;
  const int j = q / per_row, c = (q - j * per_row) * 4;
  float4 a = ld4(v.b.slab + (size_t)j * s.H + c);
  const float4 n4 = ld4(v.b.noise_spec + (size_t)j * s.H + c);
  a.x += n4.x; a.y += n4.y; a.z += n4.z; a.w += n4.w;
  *reinterpret_cast<float4 *>(v.b.slab + (size_t)j * s.H + c) = a;
  if (c == 0) reinterpret_cast<DevRng *>(v.b.rng)[row0 + j] = reinterpret_cast<const DevRng *>(v.b.rng_spec)[j];
}

// ------------------------------------------------------- bottom layer --

// The optional bottom layer of rnn_opinion (recur-nn.c:88-103): one workgroup per
// stream.  The layer is small (tens to a few hundred nodes each side), so each output
// column is one thread's sequential dot product down the rows, in the reference's
// order (calculate_interlayer, recur-nn.c:18-65, skips zero inputs).  The noise comes
// from the stream's own generator before the hidden layer draws from it.
__global__ __launch_bounds__(256) void k_bottom_forward(View v, int row0, int mode,
                                                        const float *dense, int ld, int text_i,
                                                        int global_first, int n_set,
                                                        float deviation) {
  extern __shared__ float bsh[];
  const RamdShape &s = v.sh;
  float *sin = bsh, *sout = bsh + s.bI;
  int j = blockIdx.x;
  int r = row0 + j;
  float *inp = v.b.binp + (size_t)r * s.bI;
  int hot = -1;
  if (mode == RAMD_IN_ONE_HOT) {
    hot = v.b.hot[r];
  } else if (mode == RAMD_IN_TEXT) { /* charmodel-predict.c:273, 295-298 */
    int len = v.b.text_len;
    int spacing = (len - 1) / n_set;
    int o = text_i + (global_first + j) * spacing;
    if (o >= len - 1) o -= len - 1;
    hot = v.b.text[o];
    if (threadIdx.x == 0) v.b.target[r] = v.b.text[o + 1];
  }
  for (int i = threadIdx.x; i < s.bI; i += 256) {
    float x;
    if (i == 0) x = 1.0f;
    else if (i > s.b_in || mode == RAMD_IN_KEEP) x = inp[i];
    else if (mode == RAMD_IN_DENSE) x = dense[(size_t)j * ld + (i - 1)];
    /* one_hot_opinion's bottom-layer branch clears and indexes the layer's inputs from
     * the bias slot (charmodel-helpers.h:20-23, 30-31): symbol k lights entry k, the
     * last entry is never cleared */
    else if (i == s.b_in) x = v.b.blast[0]; /* ONE buffer for all clones: what the last dense pass of ANY stream left */
    else x = (i == hot) ? 1.0f : 0.0f;
    inp[i] = x;
    sin[i] = x;
    if (i == s.b_in && (mode == RAMD_IN_DENSE || mode == RAMD_IN_KEEP) && j == (int)gridDim.x - 1)
      v.b.blast[0] = x; /* the last stream of the pass is the one whose inputs stay in the buffer */
  }
  __syncthreads();
  for (int x = threadIdx.x; x < s.bO; x += 256) {
    float acc = 0.0f;
    for (int y = 0; y < s.bI; y++) {
      float xi = sin[y];
      if (xi != 0.0f) acc += xi * v.b.bw[y * s.bO + x];
    }
    sout[x] = acc;
  }
  __syncthreads();
  if (deviation != 0.0f && threadIdx.x == 0) {
    DevRng g = reinterpret_cast<DevRng *>(v.b.rng)[r];
    for (int i = 1; i < s.input_size; i++) sout[i] += dev_cheap_gaussian(g) * deviation;
    reinterpret_cast<DevRng *>(v.b.rng)[r] = g;
  }
  __syncthreads();
  float *slot = input_row<false>(v, r, 0) + s.hidden_size + 1;
  float *out = v.b.bout + (size_t)r * s.bO;
  for (int x = threadIdx.x; x < s.bO; x += 256) {
    float o = sout[x];
    out[x] = o;
    if (x < s.input_size) slot[x] = o > 0.0f ? o : 0.0f;
  }
}

// cumulative_input_error of one stream (recur-nn.c:377-382): the input columns of
// every executed step's error, summed in step order.  One workgroup per stream.
__global__ __launch_bounds__(64) void k_bottom_error(View v, int row0, int nxp,
                                                     const unsigned char *active) {
  const RamdShape &s = v.sh;
  int j = blockIdx.x, r = row0 + j;
  int n = (active && !active[j]) ? 0 : v.b.n_exec[r];
  for (int y = threadIdx.x; y < s.bO; y += 64) {
    float sum = 0.0f;
    if (y < s.input_size)
      for (int k = 0; k < n; k++) sum += v.b.ex[((size_t)(k + 1) * s.Scap + r) * nxp + y + 1];
    v.b.berr[(size_t)r * s.bO + y] = sum;
  }
}

// single_layer_sgd on the bottom layer (recur-nn.c:750-757, 256-273) for the streams in
// the reference's order.  bottom->o_error is shared by all the clones and only
// rnn_bptt_clear_deltas ever zeroes it, so stream j's update uses the running total of
// every stream before it (and of every earlier generation): carry_in + the prefix sum.
// One thread per weight; the stream loop is sequential, as the reference's calls are.
__global__ __launch_bounds__(256) void k_bottom_delta(View v, int row0, int nrows, int accumulate,
                                                      const unsigned char *active,
                                                      const float *carry_in, float *carry_out) {
  const RamdShape &s = v.sh;
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= s.bI * s.bO) return;
  int yi = e / s.bO, x = e - yi * s.bO;
  float cum = carry_in[x];
  float acc = accumulate ? v.b.bdelta[e] : 0.0f;
  for (int j = 0; j < nrows; j++) {
    if (active && !active[j]) continue;
    int r = row0 + j;
    cum += v.b.berr[(size_t)r * s.bO + x];
    /* a stream whose error gain was clipped shrinks the accumulator -- all of it, it is the layer's
     * one o_error -- by ih_scale twice (recur-nn.c:391-399); an unclipped stream has ih_scale 1 */
    const float sc = v.b.ih_scale[r];
    if (sc != 1.0f && x < s.input_size) cum *= sc * sc;
    float xi = v.b.binp[(size_t)r * s.bI + yi];
    if (xi != 0.0f) acc += xi * cum;
  }
  v.b.bdelta[e] = acc;
  if (yi == 0) carry_out[x] = cum;
}

// ---------------------------------------------------------- finalize: fwd --

// sums the K slabs, applies the activation (recur-nn.c:123-148) and writes
// the hidden rows.  Element-wise, float4 per thread.
__global__ __launch_bounds__(256) void k_fwd_finalize(View v, int row0, int nrows, int ks) {
  const RamdShape &s = v.sh;
  int q = blockIdx.x * 256 + threadIdx.x; /* float4 index */
  int per_row = s.H >> 2;
  if (q >= nrows * per_row) return;
  int j = q / per_row, c = (q - j * per_row) * 4;
  const float *p = v.b.slab + (size_t)j * s.H + c;
  float4 a = ld4(p);
  for (int z = 1; z < ks; z++) {
    float4 t = ld4(p + (size_t)z * nrows * s.H);
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  }
  float h[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float x = h[i];
    if (s.activation == 2) { /* RNN_RESQRT */
      x = (x > 0.0f) ? sqrtf(x + 1.0f) - 1.0f : 0.0f;
    } else if (s.activation == 5) { /* RNN_RECLIP20 */
      x = x < 20.0f ? x : 20.0f;
      x = (x > 0.0f) ? x : 0.0f;
    } else {
      x = (x > 0.0f) ? x : 0.0f;
    }
    h[i] = x;
  }
  if (c == 0) h[0] = 1.0f; /* the bias node, recur-nn.c:148 */
  *reinterpret_cast<float4 *>(v.b.hidden + (size_t)(row0 + j) * s.H + c) =
      make_float4(h[0], h[1], h[2], h[3]);
}

// -------------------------------------------------------- loss on device --

#pragma clang fp contract(off)
// badmaths.h:14-29, kept operation for operation
__device__ float fast_expf_dev(float x) {
  int count = 0;
  while (fabsf(x) > 0.2) {
    x *= 0.125;
    count++;
  }
  float a = ((x + 3) * (x + 3) + 3) / ((x - 3) * (x - 3) + 3);
  while (count) {
    a *= a;
    a *= a;
    a *= a;
    count--;
  }
  return a;
}

// The output layer of rnn_opinion (recur-nn.c:150-151): out = hidden . W_ho for one state
// row per workgroup.  O is small (tens to a few hundred columns) against H, so this is
// not worth an MFMA launch plus a slab pass: thread (seg, col) walks one sixteenth of the
// hidden units down one column of W_ho (a wave reads whole contiguous rows), sixteen
// waves keep enough rows in flight to cover the L2 latency, and the segments are added
// in order.
constexpr int OUT_SEGS = 16;
__global__ __launch_bounds__(1024) void k_out_layer(View v, int row0) {
  extern __shared__ float osh[]; /* [H] hidden row, then [OUT_SEGS][64] partial sums */
  const RamdShape &s = v.sh;
  const int r = row0 + blockIdx.x;
  const float *hid = v.b.hidden + (size_t)r * s.H;
  for (int i = threadIdx.x; i < s.H; i += 1024) osh[i] = hid[i];
  __syncthreads();
  float *part = osh + s.H;
  const int seg = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per = (s.H + OUT_SEGS - 1) / OUT_SEGS;
  const int y0 = seg * per, y1 = min(s.H, y0 + per);
  float *out = v.b.out + (size_t)r * s.O;
  for (int c0 = 0; c0 < s.O; c0 += 64) {
    int col = c0 + lane;
    float acc0 = 0.0f, acc1 = 0.0f;
    if (col < s.O) {
      const float *w = v.b.ho_w + col;
      int y = y0;
#pragma unroll 4
      for (; y + 1 < y1; y += 2) {
        acc0 += osh[y] * w[(size_t)y * s.O];
        acc1 += osh[y + 1] * w[(size_t)(y + 1) * s.O];
      }
      if (y < y1) acc0 += osh[y] * w[(size_t)y * s.O];
    }
    part[seg * 64 + lane] = acc0 + acc1;
    __syncthreads();
    if (seg == 0 && col < s.O) {
      float sum = part[lane];
      for (int g = 1; g < OUT_SEGS; g++) sum += part[g * 64 + lane];
      out[col] = sum;
    }
    __syncthreads();
  }
}

// The same for o_size == 4 (rnnca: Y, Cb, Cr + padding): a row of W_ho is ONE float4, so a wave
// per state row walks the hidden units 64 at a time (coalesced hidden values, coalesced float4
// weights) and reduces with xor shuffles.  k_out_layer gives every column a lane and every
// sixteenth of the rows a wave, which with 4 columns leaves 60 of 64 lanes idle: 252 us for
// the 13,824 rows of an rnnca frame at hidden 2048.
__global__ __launch_bounds__(256) void k_out_layer_o4(View v, int row0, int nrows) {
  const RamdShape &s = v.sh;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= nrows) return;
  const int r = row0 + j;
  const float *hid = v.b.hidden + (size_t)r * s.H;
  float4 acc = zero4();
  for (int y = lane; y < s.H; y += 64) {
    const float h = hid[y];
    const float4 w = ld4(v.b.ho_w + (size_t)y * 4);
    acc.x += h * w.x; acc.y += h * w.y; acc.z += h * w.z; acc.w += h * w.w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    acc.x += __shfl_xor(acc.x, off, 64);
    acc.y += __shfl_xor(acc.y, off, 64);
    acc.z += __shfl_xor(acc.z, off, 64);
    acc.w += __shfl_xor(acc.w, off, 64);
  }
  if (lane == 0) *reinterpret_cast<float4 *>(v.b.out + (size_t)r * 4) = acc;
}

// net_error_bptt's loss (charmodel-predict.c:18-27): softmax (badmaths.h:71-111),
// best guess and negation (badmaths.h:113-141), +1 on the target; plus the
// running statistics of the epoch loop (charmodel-predict.c:302-304).  One
// thread per stream walks its row in the reference's order, so the sums round
// the same way.
__global__ __launch_bounds__(64) void k_softmax_error(View v, int row0, int nrows) {
  extern __shared__ float ex[]; /* [output_size] exponentials */
  int j = blockIdx.x;
  if (j >= nrows) return;
  const RamdShape &s = v.sh;
  int r = row0 + j;
  // zero fraction of the hidden row (recur-nn.c:438-442), counted by the wave
  const float *hid = v.b.hidden + (size_t)r * s.H;
  int zeros = 0;
  for (int i = threadIdx.x; i < s.H; i += 64) zeros += (hid[i] == 0.0f);
  for (int off = 32; off > 0; off >>= 1) zeros += __shfl_down(zeros, off, 64);
  const float *src = v.b.out + (size_t)r * s.O;
  float *err = v.b.o_error + (size_t)r * s.O;
  int len = s.output_size;
  // max and min are order independent: one pass over the lanes
  float lo = src[0], hi = src[0];
  for (int i = threadIdx.x; i < len; i += 64) {
    hi = fmaxf(hi, src[i]);
    lo = fminf(lo, src[i]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    lo = fminf(lo, __shfl_xor(lo, off, 64));
  }
  float adj = 0.0f;
  if (hi > 50.0f) adj = 50.0f - hi;
  else if (lo < -60.0f) adj = fminf(-60.0f - lo, 50.0f - hi);
  // the exponentials in parallel, their sum in the reference's order (lane 0)
  for (int i = threadIdx.x; i < len; i += 64) ex[i] = fast_expf_dev(src[i] + adj);
  __syncthreads();
  // every lane adds the exponentials in the reference's order (the same value in all of
  // them); the divisions and the arg max (first of equal maxima, badmaths.h:126-139) are
  // spread over the lanes
  float sum = 0.0f;
  for (int i = 0; i < len; i++) sum += ex[i];
  float best_e = -1.0f;
  int best_i = 0x7fffffff;
  const int target = v.b.target[r];
  for (int i = threadIdx.x; i < len; i += 64) {
    float e = ex[i] / sum;
    err[i] = (i == target) ? -e + 1.0f : -e; /* error[next] += 1.0f, charmodel-predict.c:25 */
    if (e > best_e) {
      best_e = e;
      best_i = i;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    float oe = __shfl_xor(best_e, off, 64);
    int oi = __shfl_xor(best_i, off, 64);
    if (oe > best_e || (oe == best_e && oi < best_i)) {
      best_e = oe;
      best_i = oi;
    }
  }
  if (threadIdx.x != 0) return;
  float e = -(ex[target] / sum) + 1.0f;
  float l = 1.0f - e;
  v.b.stat_err[r] += e;
  v.b.stat_ent[r] += (l < 1e-30f) ? -100.0f : log2f(l); /* charmodel-helpers.h:11-13 */
  v.b.stat_correct[r] += (best_i == target);
  v.b.stat_count[r] += 1;
  v.b.stat_zero[r] += zeros / (double)s.hidden_size;
}
// The top of a text generation in one launch, one workgroup (16 waves) per stream: the
// output layer (k_out_layer), the softmax loss against the stream's target
// (k_softmax_error) and the top-layer backprop with its soft clip (k_top_backprop, dense
// form), each exactly as in the separate kernels -- same operation order per value -- with
// the hidden row, the outputs and the output error passed through LDS instead of HBM.
__global__ __launch_bounds__(1024) void k_text_top(View v, int row0, int nrows, int fwd_ks) {
  extern __shared__ float tsh[];
  __shared__ float tred[16];
  const RamdShape &s = v.sh;
  const int r = row0 + blockIdx.x;
  float *shid = tsh;                   /* [H] hidden row                    */
  float *part = shid + s.H;            /* [OUT_SEGS][64] output partial sums */
  float *sout = part + OUT_SEGS * 64;  /* [O] outputs                        */
  float *sex = sout + s.O;             /* [O] exponentials                   */
  float *serr = sex + s.O;             /* [O] output error                   */
  float *hid = v.b.hidden + (size_t)r * s.H;
  if (fwd_ks != 0) {
    // the forward GEMM's K slabs are still in the workspace: sum them, apply the
    // activation and write the hidden row here (what k_fwd_finalize does, recur-nn.c:123-148).
    // fwd_ks < 0: k_fwd_fused left one plane of sums and, for the h_size padding columns,
    // -fwd_ks per-tile partial sums in plane 1.
    const float *p = v.b.slab + (size_t)blockIdx.x * s.H;
    const int npart = fwd_ks < 0 ? -fwd_ks : 0;
    if (fwd_ks < 0) fwd_ks = 1;
    for (int i = threadIdx.x; i < s.H; i += 1024) {
      /* all the slabs' loads in flight at once (a loop with a run-time trip count issues
       * them one L2 latency after another) */
      const size_t plane = (size_t)nrows * s.H;
      float xs[8];
#pragma unroll
      for (int z = 0; z < 8; z++) xs[z] = (z < fwd_ks) ? p[z * plane + i] : 0.0f;
      float x = xs[0];
#pragma unroll
      for (int z = 1; z < 8; z++)
        if (z < fwd_ks) x += xs[z];
      for (int z = 8; z < fwd_ks; z++) x += p[z * plane + i];
      if (npart && i >= s.H - 4) continue; /* the tail columns: below */
      if (s.activation == 2) {
        x = (x > 0.0f) ? sqrtf(x + 1.0f) - 1.0f : 0.0f;
      } else if (s.activation == 5) {
        x = x < 20.0f ? x : 20.0f;
        x = (x > 0.0f) ? x : 0.0f;
      } else {
        x = (x > 0.0f) ? x : 0.0f;
      }
      if (i == 0) x = 1.0f; /* the bias node, recur-nn.c:148 */
      hid[i] = x;
      shid[i] = x;
    }
    if (npart && threadIdx.x < 256) {
      /* k_fwd_fused's four tail columns (hidden value hidden_size and the padding of h_size):
       * wave p adds column p's per-tile partial sums */
      const int p4 = threadIdx.x >> 6, ln = threadIdx.x & 63;
      const float *pd = v.b.slab + (size_t)nrows * s.H + (size_t)blockIdx.x * 4 + p4;
      float x = 0.0f;
      for (int t = ln; t < npart; t += 64) x += pd[(size_t)t * nrows * 4];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
      if (s.activation == 2) {
        x = (x > 0.0f) ? sqrtf(x + 1.0f) - 1.0f : 0.0f;
      } else if (s.activation == 5) {
        x = x < 20.0f ? x : 20.0f;
        x = (x > 0.0f) ? x : 0.0f;
      } else {
        x = (x > 0.0f) ? x : 0.0f;
      }
      if (ln == 0) {
        hid[s.H - 4 + p4] = x;
        shid[s.H - 4 + p4] = x;
      }
    }
  } else {
    for (int i = threadIdx.x; i < s.H; i += 1024) shid[i] = hid[i];
  }
  __syncthreads();
  const int seg = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // ---- output layer (recur-nn.c:150-151)
  {
    const int per = (s.H + OUT_SEGS - 1) / OUT_SEGS;
    const int y0 = seg * per, y1 = min(s.H, y0 + per);
    float *out = v.b.out + (size_t)r * s.O;
    for (int c0 = 0; c0 < s.O; c0 += 64) {
      int col = c0 + lane;
      float acc0 = 0.0f, acc1 = 0.0f;
      if (col < s.O) {
        const float *w = v.b.ho_w + col;
        int y = y0;
        /* 32 rows' weights in flight per batch; the sums keep the order of the plain loop */
        for (; y + 31 < y1; y += 32) {
          float wv[32];
#pragma unroll
          for (int k = 0; k < 32; k++) wv[k] = w[(size_t)(y + k) * s.O];
#pragma unroll
          for (int k = 0; k < 32; k += 2) {
            acc0 += shid[y + k] * wv[k];
            acc1 += shid[y + k + 1] * wv[k + 1];
          }
        }
#pragma unroll 4
        for (; y + 1 < y1; y += 2) {
          acc0 += shid[y] * w[(size_t)y * s.O];
          acc1 += shid[y + 1] * w[(size_t)(y + 1) * s.O];
        }
        if (y < y1) acc0 += shid[y] * w[(size_t)y * s.O];
      }
      part[seg * 64 + lane] = acc0 + acc1;
      __syncthreads();
      if (seg == 0 && col < s.O) {
        float sum = part[lane];
        for (int g = 1; g < OUT_SEGS; g++) sum += part[g * 64 + lane];
        out[col] = sum;
        sout[col] = sum;
      }
      __syncthreads();
    }
  }
  // the backprop below needs this thread's row of W_ho: request it now (narrow output layers),
  // so that it arrives while wave 0 works out the softmax
  constexpr int TOP_PF = 12; /* float4 per row: o_size <= 48 */
  float4 wrow[TOP_PF];
  const bool top_pf = s.O <= 4 * TOP_PF;
  if (top_pf) {
    const int y = threadIdx.x;
    const bool need = y != 0 && y < s.H && shid[y] != 0.0f;
    const float *rowp = v.b.ho_w + (size_t)(need ? y : 0) * s.O;
#pragma unroll
    for (int k = 0; k < TOP_PF; k++) wrow[k] = (need && 4 * k < s.O) ? ld4(rowp + 4 * k) : zero4();
  }
  // ---- softmax loss (charmodel-predict.c:18-27, badmaths.h:71-141): wave 0
  if (seg == 0) {
    const int len = s.output_size;
    float *err = v.b.o_error + (size_t)r * s.O;
    int zeros = 0;
    for (int i = lane; i < s.H; i += 64) zeros += (shid[i] == 0.0f);
    for (int off = 32; off > 0; off >>= 1) zeros += __shfl_down(zeros, off, 64);
    float lo = sout[0], hi = sout[0];
    for (int i = lane; i < len; i += 64) {
      hi = fmaxf(hi, sout[i]);
      lo = fminf(lo, sout[i]);
    }
    for (int off = 32; off > 0; off >>= 1) {
      hi = fmaxf(hi, __shfl_xor(hi, off, 64));
      lo = fminf(lo, __shfl_xor(lo, off, 64));
    }
    float adj = 0.0f;
    if (hi > 50.0f) adj = 50.0f - hi;
    else if (lo < -60.0f) adj = fminf(-60.0f - lo, 50.0f - hi);
    for (int i = lane; i < len; i += 64) sex[i] = fast_expf_dev(sout[i] + adj);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* one wave: its LDS writes are ordered */
    float sum = 0.0f;
    for (int i = 0; i < len; i++) sum += sex[i];
    float best_e = -1.0f;
    int best_i = 0x7fffffff;
    const int target = v.b.target[r];
    for (int i = lane; i < s.O; i += 64) {
      float oe;
      if (i < len) {
        float e = sex[i] / sum;
        oe = (i == target) ? -e + 1.0f : -e;
        err[i] = oe;
        if (e > best_e) {
          best_e = e;
          best_i = i;
        }
      } else {
        oe = err[i]; /* the pad of o_error stays what it was (zero) */
      }
      serr[i] = oe;
    }
    for (int off = 32; off > 0; off >>= 1) {
      float oe = __shfl_xor(best_e, off, 64);
      int oi = __shfl_xor(best_i, off, 64);
      if (oe > best_e || (oe == best_e && oi < best_i)) {
        best_e = oe;
        best_i = oi;
      }
    }
    if (lane == 0) {
      float e = -(sex[target] / sum) + 1.0f;
      float l = 1.0f - e;
      v.b.stat_err[r] += e;
      v.b.stat_ent[r] += (l < 1e-30f) ? -100.0f : log2f(l);
      v.b.stat_correct[r] += (best_i == target);
      v.b.stat_count[r] += 1;
      v.b.stat_zero[r] += zeros / (double)s.hidden_size;
    }
  }
  __syncthreads();
  // ---- top-layer backprop + soft clip (recur-nn.c:199-228, 719-721)
  float sum = 0.0f;
  float ev[3] = {0.0f, 0.0f, 0.0f}; /* h_size <= 3072 per launch condition */
  for (int q = 0, y = threadIdx.x; y < s.H; y += 1024, q++) {
    float e = 0.0f;
    if (y != 0 && shid[y] != 0.0f) {
      if (top_pf && q == 0) {
#pragma unroll
        for (int k = 0; k < TOP_PF; k++) {
          if (4 * k < s.O) {
            e += wrow[k].x * serr[4 * k];
            e += wrow[k].y * serr[4 * k + 1];
            e += wrow[k].z * serr[4 * k + 2];
            e += wrow[k].w * serr[4 * k + 3];
          }
        }
      } else {
        const float *row = v.b.ho_w + (size_t)y * s.O;
        for (int x = 0; x < s.O; x += 4) {
          float4 w = ld4(row + x);
          e += w.x * serr[x];
          e += w.y * serr[x + 1];
          e += w.z * serr[x + 2];
          e += w.w * serr[x + 3];
        }
      }
      sum += fabsf(e);
    }
    ev[q] = e;
  }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (lane == 0) tred[seg] = sum;
  __syncthreads();
  /* the same tree as block_sum_256 within each group of four waves, then the four groups */
  float g0 = (tred[0] + tred[1]) + (tred[2] + tred[3]), g1 = (tred[4] + tred[5]) + (tred[6] + tred[7]);
  float g2 = (tred[8] + tred[9]) + (tred[10] + tred[11]), g3 = (tred[12] + tred[13]) + (tred[14] + tred[15]);
  sum = (g0 + g1) + (g2 + g3);
  float halfmax = s.H * MAX_TOP_ERROR_FACTOR_F;
  float scaled = sum, scale = 1.0f;
  if (sum > halfmax) {
    scale = soft_clip_dev(sum, halfmax);
    scaled = scale * sum;
  }
  float *dst = v.b.ehi + (size_t)r * s.I; /* step 0 plane */
  for (int q = 0, y = threadIdx.x; y < s.H; y += 1024, q++)
    dst[y] = (y == 0 || y > s.hidden_size) ? 0.0f : (sum > halfmax) ? ev[q] * scale : ev[q];
  if (threadIdx.x == 0) {
    v.b.top_raw[r] = sum;
    v.b.top_scaled[r] = scaled;
  }
}

// multi_softmax_error (charmodel-multi-predict.c:17-58) after the opinion: the output row is
// n_classes heads of alphabet_len symbols.  The head of the stream's own class is always
// trained; every other head with probability `leakage`, decided by a draw from the
// stream's generator (none for the own head: the || short-circuits).  A trained head gets
// -softmax with +1 on the next symbol; the others stay zero.  The (start, len) ranges
// the reference builds for rnn_bptt_calc_deltas -- aligned, merged when they touch -- are
// left in ranges[j].  One wave per stream; every lane runs the generator redundantly so
// that the decisions are uniform.
// Four waves per stream: wave 0 makes the leak decisions (the generator is sequential) and the
// range list while all four clear the error row; then the trained heads are shared out over the
// waves, each head's softmax exactly as before (the sum of the exponentials in index order).
// (As one wave per stream this was 35 us for 256 streams of 50 heads.)
constexpr int MS_WAVES = 4, MS_MAXCLS = 256;
__global__ __launch_bounds__(64 * MS_WAVES) void k_multi_softmax_error(View v, int row0, int alen, int ncls,
                                                                       unsigned long long threshold,
                                                                       const int *tclass, int *ranges,
                                                                       int range_stride) {
  extern __shared__ float exs[]; /* [MS_WAVES][alen] */
  __shared__ short trained[MS_MAXCLS];
  __shared__ int ntrained;
  __shared__ float own_err_sh;
  const RamdShape &s = v.sh;
  const int j = blockIdx.x, r = row0 + j, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float *src = v.b.out + (size_t)r * s.O;
  float *err = v.b.o_error + (size_t)r * s.O;
  const int next = v.b.target[r], own = tclass[j];
  for (int i = threadIdx.x; i < s.output_size; i += 64 * MS_WAVES) err[i] = 0.0f;
  if (wave == 0) {
    /* every lane runs the generator redundantly, so that the decisions are wave-uniform */
    int *rg = ranges + (size_t)j * range_stride;
    DevRng g = reinterpret_cast<DevRng *>(v.b.rng)[r];
    int nt = 0, nr = 0, prev_start = 0, prev_len = 0;
    for (int c = 0; c < ncls; c++) {
      bool train = (c == own);
      if (!train) train = dev_rand64(g) < threshold;
      if (!train) continue;
      if (lane == 0) trained[nt] = (short)c;
      nt++;
      const int offset = c * alen;
      int start = offset & ~3, end = (offset + alen + 3) & ~3;
      if (nr && prev_start + prev_len >= start) {
        prev_len = end - prev_start;
        if (lane == 0) rg[2 * (nr - 1) + 1] = prev_len;
      } else {
        prev_start = start;
        prev_len = end - start;
        if (lane == 0) {
          rg[2 * nr] = prev_start;
          rg[2 * nr + 1] = prev_len;
        }
        nr++;
      }
    }
    if (lane == 0) {
      rg[2 * nr] = -1;
      rg[2 * nr + 1] = 0;
      reinterpret_cast<DevRng *>(v.b.rng)[r] = g;
      ntrained = nt;
    }
  }
  __syncthreads(); /* the row is clear, the list is there */
  float *ex = exs + wave * alen;
  const int nt = ntrained;
  for (int k = wave; k < nt; k += MS_WAVES) {
    const int c = trained[k], offset = c * alen;
    const float *gs = src + offset;
    float lo = gs[0], hi = gs[0];
    for (int i = lane; i < alen; i += 64) {
      hi = fmaxf(hi, gs[i]);
      lo = fminf(lo, gs[i]);
    }
    for (int off = 32; off > 0; off >>= 1) {
      hi = fmaxf(hi, __shfl_xor(hi, off, 64));
      lo = fminf(lo, __shfl_xor(lo, off, 64));
    }
    float adj = 0.0f;
    if (hi > 50.0f) adj = 50.0f - hi;
    else if (lo < -60.0f) adj = fminf(-60.0f - lo, 50.0f - hi);
    for (int i = lane; i < alen; i += 64) ex[i] = fast_expf_dev(gs[i] + adj);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* one wave: its LDS writes are ordered */
    float sum = 0.0f;
    for (int i = 0; i < alen; i++) sum += ex[i];
    for (int i = lane; i < alen; i += 64) {
      float e = ex[i] / sum;
      err[offset + i] = (i == next) ? -e + 1.0f : -e;
    }
    if (c == own && lane == 0) own_err_sh = -(ex[next] / sum) + 1.0f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* before this wave's next head rewrites ex */
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float own_err = own_err_sh;
    float l = 1.0f - own_err;
    v.b.stat_err[r] += own_err;
    v.b.stat_ent[r] += (l < 1e-30f) ? -100.0f : log2f(l);
    v.b.stat_count[r] += 1;
  }
}

// train_channel's loss (gstclassify.c:2070-2119) for every stream: the output row is a
// few class groups; a group whose target is valid gets -softmax with +1 on the target,
// the others zeros; if any group was trained the whole error row is multiplied by the
// per-output error weights.  One wave per stream.  gt[j * ngroups + i] < 0 (or out of
// range) = "no training for this group" -- the caller decides that (target unknown,
// ignored windows, the balanced-sampling draw), as the reference's caller does.
__global__ __launch_bounds__(64) void k_grouped_softmax_error(View v, int row0, int ngroups,
                                                              const int *goff, const int *gsize,
                                                              const int *gt, const float *weight) {
  extern __shared__ float ex[]; /* [largest group] */
  const RamdShape &s = v.sh;
  const int j = blockIdx.x, r = row0 + j, lane = threadIdx.x;
  const float *src = v.b.out + (size_t)r * s.O;
  float *err = v.b.o_error + (size_t)r * s.O;
  int trained = 0, wins = 0;
  float wrong = 0.0f;
  for (int i = 0; i < ngroups; i++) {
    const int o = goff[i], n = gsize[i], target = gt[(size_t)j * ngroups + i];
    if (target < 0 || target >= n) {
      for (int q = lane; q < n; q += 64) err[o + q] = 0.0f;
      continue;
    }
    const float *gs = src + o;
    float lo = gs[0], hi = gs[0];
    for (int q = lane; q < n; q += 64) {
      hi = fmaxf(hi, gs[q]);
      lo = fminf(lo, gs[q]);
    }
    for (int off = 32; off > 0; off >>= 1) {
      hi = fmaxf(hi, __shfl_xor(hi, off, 64));
      lo = fminf(lo, __shfl_xor(lo, off, 64));
    }
    float adj = 0.0f;
    if (hi > 50.0f) adj = 50.0f - hi;
    else if (lo < -60.0f) adj = fminf(-60.0f - lo, 50.0f - hi);
    __syncthreads();
    for (int q = lane; q < n; q += 64) ex[q] = fast_expf_dev(gs[q] + adj);
    __syncthreads();
    float sum = 0.0f;
    for (int q = 0; q < n; q++) sum += ex[q];
    float best_e = -1.0f;
    int best_i = 0x7fffffff;
    for (int q = lane; q < n; q += 64) {
      float e = ex[q] / sum;
      err[o + q] = (q == target) ? -e + 1.0f : -e;
      if (e > best_e) {
        best_e = e;
        best_i = q;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      float oe = __shfl_xor(best_e, off, 64);
      int oi = __shfl_xor(best_i, off, 64);
      if (oe > best_e || (oe == best_e && oi < best_i)) {
        best_e = oe;
        best_i = oi;
      }
    }
    wins += (best_i == target);
    wrong += -(ex[target] / sum) + 1.0f;
    trained++;
  }
  if (trained && weight) {
    __syncthreads(); /* one wave: its own stores are ordered; this keeps the compiler honest */
    for (int q = lane; q < s.output_size; q += 64) err[q] *= weight[q];
  }
  if (lane == 0 && trained) {
    v.b.stat_err[r] += wrong;
    v.b.stat_correct[r] += wins;
    v.b.stat_count[r] += trained;
  }
}

// get_cross_entropy's inner step (charmodel-predict.c:71-76): softmax of one state
// row's outputs (badmaths.h:71-111, sums in the reference's order), the probability of
// the row's target symbol, capped_log2f of it added to the row's running total.
__global__ __launch_bounds__(64) void k_xent_accumulate(View v, int r, int count_it) {
  extern __shared__ float ex[];
  const RamdShape &s = v.sh;
  const float *src = v.b.out + (size_t)r * s.O;
  int len = s.output_size;
  float lo = src[0], hi = src[0];
  for (int i = threadIdx.x; i < len; i += 64) {
    hi = fmaxf(hi, src[i]);
    lo = fminf(lo, src[i]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    lo = fminf(lo, __shfl_xor(lo, off, 64));
  }
  float adj = 0.0f;
  if (hi > 50.0f) adj = 50.0f - hi;
  else if (lo < -60.0f) adj = fminf(-60.0f - lo, 50.0f - hi);
  for (int i = threadIdx.x; i < len; i += 64) ex[i] = fast_expf_dev(src[i] + adj);
  __syncthreads();
  if (threadIdx.x != 0 || !count_it) return;
  float sum = 0.0f;
  for (int i = 0; i < len; i++) sum += ex[i];
  float e = ex[v.b.target[r]] / sum;
  v.b.xent[r] += (double)((e < 1e-30f) ? -100.0f : log2f(e));
}

// rnn_char_multi_cross_entropy's inner step (charmodel-multi-predict.c:395-403): block c
// takes head c of the output row -- softmax over that head alone (badmaths.h:71-111), the
// probability of the row's target symbol, capped log2 added to acc[c].
__global__ __launch_bounds__(64) void k_multi_xent_accumulate(View v, int r, int alen, double *acc,
                                                              int count_it) {
  extern __shared__ float ex[];
  const RamdShape &s = v.sh;
  const int c = blockIdx.x;
  const float *src = v.b.out + (size_t)r * s.O + (size_t)c * alen;
  float lo = src[0], hi = src[0];
  for (int i = threadIdx.x; i < alen; i += 64) {
    hi = fmaxf(hi, src[i]);
    lo = fminf(lo, src[i]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    lo = fminf(lo, __shfl_xor(lo, off, 64));
  }
  float adj = 0.0f;
  if (hi > 50.0f) adj = 50.0f - hi;
  else if (lo < -60.0f) adj = fminf(-60.0f - lo, 50.0f - hi);
  for (int i = threadIdx.x; i < alen; i += 64) ex[i] = fast_expf_dev(src[i] + adj);
  __syncthreads();
  if (threadIdx.x != 0 || !count_it) return;
  float sum = 0.0f;
  for (int i = 0; i < alen; i++) sum += ex[i];
  float e = ex[v.b.target[r]] / sum;
  acc[c] += (double)((e < 1e-30f) ? -100.0f : log2f(e));
}

// rnnca's loss (gstrnnca.c:701-714, train_net): fast_sigmoid_array(answer, answer, n) IN
// PLACE on the first n outputs (badmaths.h:33-44), then o_error[i] = a (1 - a) (target - a).
// One thread per (stream, output); the rest of the error row stays as it was (zero).
__global__ void k_sigmoid_mse_error(View v, int row0, int nrows, int n, const float *targets,
                                    int ld) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nrows * n) return;
  int j = q / n, i = q - j * n, r = row0 + j;
  float *out = v.b.out + (size_t)r * v.sh.O;
  float a = 1.0f / (1.0f + fast_expf_dev(-out[i] * 1.0f));
  out[i] = a;
  float slope = a * (1.0f - a);
  v.b.o_error[(size_t)r * v.sh.O + i] = slope * (targets[(size_t)j * ld + i] - a);
}

// fill_frame's fast_sigmoid_array(answer, answer, 3) (gstrnnca.c:813-814) for state rows
__global__ void k_sigmoid_outputs(View v, int r0, int nrows, int n) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nrows * n) return;
  int j = q / n, i = q - j * n;
  float *out = v.b.out + (size_t)(r0 + j) * v.sh.O;
  out[i] = 1.0f / (1.0f + fast_expf_dev(-out[i] * 1.0f));
}

#pragma clang fp contract(fast)

// ---------------------------------------------------- K5/K6: top backprop --

// backprop_single_layer / _sparse + softclip_scale (recur-nn.c:156-228,
// 719-721).  One workgroup per stream.  Writes the (scaled) error both to
// ehi[0] (what the BPTT chain reads) and leaves err_a for the lazy write-back.
__global__ __launch_bounds__(256) void k_top_backprop(View v, int row0, const int *ranges,
                                                      int range_stride,
                                                      const unsigned char *active) {
  extern __shared__ float sh[];
  __shared__ float red[4];
  const RamdShape &s = v.sh;
  int j = blockIdx.x, r = row0 + j;
  if (active && !active[j]) return;
  if (ranges) ranges += (size_t)j * range_stride; /* 0: one list for every stream */
  float *oerr = sh;          /* [O] */
  float *herr = sh + s.O;    /* [H] */
  for (int i = threadIdx.x; i < s.O; i += 256) oerr[i] = v.b.o_error[(size_t)r * s.O + i];
  __syncthreads();
  const float *hid = v.b.hidden + (size_t)r * s.H;
  const float *old = v.b.err_a + (size_t)r * s.I;
  float sum = 0.0f;
  for (int y = threadIdx.x; y < s.H; y += 256) {
    float e;
    if (y == 0) {
      e = 0.0f; /* the reference's loop starts at 1; step 1 of the BPTT zeroes it */
    } else if (hid[y] != 0.0f) {
      const float *row = v.b.ho_w + (size_t)y * s.O;
      e = 0.0f;
      if (ranges) {
        for (int i = 0; ranges[2 * i] >= 0; i++) {
          int start = ranges[2 * i] & ~3, len = (ranges[2 * i + 1] + 3) & ~3;
          for (int x = 0; x < len; x++) e += row[start + x] * oerr[start + x];
          sum += fabsf(e); /* once per range, e keeps running: recur-nn.c:178-191 */
        }
      } else {
        /* a row is O contiguous floats (O % 4 == 0): whole rows as float4, same order */
        for (int x = 0; x < s.O; x += 4) {
          float4 w = ld4(row + x);
          e += w.x * oerr[x];
          e += w.y * oerr[x + 1];
          e += w.z * oerr[x + 2];
          e += w.w * oerr[x + 3];
        }
        sum += fabsf(e);
      }
    } else {
      e = ranges ? old[y] : 0.0f; /* sparse path leaves the stale value */
    }
    herr[y] = e;
  }
  sum = block_sum_256(sum, red);
  float halfmax = s.H * MAX_TOP_ERROR_FACTOR_F;
  float scaled = sum, scale = 1.0f;
  if (sum > halfmax) {
    scale = soft_clip_dev(sum, halfmax);
    scaled = scale * sum;
  }
  float *dst = v.b.ehi + (size_t)r * s.I; /* step 0 plane */
  for (int y = threadIdx.x; y < s.H; y += 256) /* ehi keeps column 0 and the pad at zero */
    dst[y] = (y == 0 || y > s.hidden_size) ? 0.0f : (sum > halfmax) ? herr[y] * scale : herr[y];
  if (threadIdx.x == 0) {
    v.b.top_raw[r] = sum;
    v.b.top_scaled[r] = scaled;
  }
}

// backprop_single_layer_sparse (recur-nn.c:156-196) with half a wave per hidden row: k_top_backprop
// gives every thread a row of W_ho of its own and walks the ranges' columns one by one -- 64
// lanes on 64 different cache lines per load -- which at o_size 3652 (the multi-head nets) took
// 486 us for 256 streams.  Here 32 lanes read a row's range as float4 (a head of 73 symbols is 19
// of them: one instruction per row and range), a wave works on eight rows at a time (four
// instructions, each covering two rows) so that the loads and the shuffle chains of the rows
// overlap, the products are reduced over the 32 lanes with xor shuffles (a fixed tree per range),
// and the range's sum joins the row's running value and |running value| the error sum, range by
// range as the reference does (recur-nn.c:178-191).  Rows whose hidden value is zero keep the
// stale entry of the last BPTT run.  One workgroup of 16 waves per stream; with
// few streams (a GPU's share of a sharded set, the one-net trainer) the rows of a stream are
// shared out over gridDim.y workgroups -- a row is a chain of memory round trips per range, and
// 32 busy CUs of 256 leave most of the latency exposed -- which leave the unscaled values and
// their partial sums of |e| (part) for k_top_backprop_scale.
__global__ __launch_bounds__(1024) void k_top_backprop_ranged(View v, int row0, const int *ranges,
                                                              int range_stride, const unsigned char *active,
                                                              float *part) {
  extern __shared__ __attribute__((aligned(16))) float rsh[];
  __shared__ float red[16];
  __shared__ int rlist[2 * 65];
  const RamdShape &s = v.sh;
  const int j = blockIdx.x, r = row0 + j;
  if (active && !active[j]) return;
  ranges += (size_t)j * range_stride; /* 0: one list for every stream */
  float *oerr = rsh;       /* [O] */
  float *herr = rsh + s.O; /* [H] */
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, l32 = lane & 31;
  for (int i = threadIdx.x; i < s.O; i += 1024) oerr[i] = v.b.o_error[(size_t)r * s.O + i];
  if (threadIdx.x == 0) {
    int n = 0;
    while (n < 64 && ranges[2 * n] >= 0) {
      rlist[2 * n] = ranges[2 * n] & ~3;                 /* start, aligned as k_top_backprop does */
      rlist[2 * n + 1] = (ranges[2 * n + 1] + 3) & ~3;   /* length */
      n++;
    }
    rlist[2 * n] = -1;
  }
  __syncthreads();
  const float *hid = v.b.hidden + (size_t)r * s.H;
  const float *old = v.b.err_a + (size_t)r * s.I;
  float sum = 0.0f; /* this half-wave's rows */
  constexpr int PAIRS = 8; /* sixteen rows at a time */
  /* this workgroup's rows: [ylo, yhi), whole groups of sixteen */
  const int nb = gridDim.y, groups = (s.H + 2 * PAIRS - 1) / (2 * PAIRS);
  const int ylo = (int)(((long)groups * blockIdx.y) / nb) * 2 * PAIRS;
  const int yhi = min(s.H, (int)(((long)groups * (blockIdx.y + 1)) / nb) * 2 * PAIRS);
  for (int y0 = ylo + 2 * PAIRS * wave; y0 < yhi; y0 += 2 * PAIRS * 16) {
    float e[PAIRS];
    bool act[PAIRS];
    const float *rowp[PAIRS];
#pragma unroll
    for (int q = 0; q < PAIRS; q++) {
      const int y = y0 + 2 * q + half; /* this half-wave's row of pair q */
      act[q] = y > 0 && y < yhi && hid[y < s.H ? y : 0] != 0.0f;
      rowp[q] = v.b.ho_w + (size_t)(act[q] ? y : 0) * s.O;
      e[q] = 0.0f;
    }
    for (int i = 0; rlist[2 * i] >= 0; i++) {
      const int start = rlist[2 * i], len4 = rlist[2 * i + 1] >> 2;
      float p[PAIRS];
#pragma unroll
      for (int q = 0; q < PAIRS; q++) p[q] = 0.0f;
      for (int x4 = l32; x4 < len4; x4 += 32) {
        const float4 o4 = ld4(oerr + start + 4 * x4);
#pragma unroll
        for (int q = 0; q < PAIRS; q++) {
          const float4 w4 = ld4(rowp[q] + start + 4 * x4);
          p[q] += (w4.x * o4.x + w4.y * o4.y) + (w4.z * o4.z + w4.w * o4.w);
        }
      }
#pragma unroll
      for (int off = 16; off > 0; off >>= 1)
#pragma unroll
        for (int q = 0; q < PAIRS; q++) p[q] += __shfl_xor(p[q], off, 64);
#pragma unroll
      for (int q = 0; q < PAIRS; q++)
        if (act[q]) {
          e[q] += p[q];
          sum += fabsf(e[q]); /* once per range, e keeps running: recur-nn.c:178-191 */
        }
    }
    if (l32 == 0) {
#pragma unroll
      for (int q = 0; q < PAIRS; q++) {
        const int y = y0 + 2 * q + half;
        if (y < yhi) herr[y] = (y == 0) ? 0.0f : act[q] ? e[q] : old[y]; /* stale value where the row is skipped */
      }
    }
  }
  sum += __shfl_xor(sum, 32, 64); /* the two half-waves' rows */
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
  sum += ((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]));
  if (nb > 1) { /* unscaled values and this workgroup's share of the sum: k_top_backprop_scale goes on */
    float *dst = v.b.ehi + (size_t)r * s.I;
    for (int y = ylo + threadIdx.x; y < yhi; y += 1024) dst[y] = (y == 0 || y > s.hidden_size) ? 0.0f : herr[y];
    if (threadIdx.x == 0) part[(size_t)j * nb + blockIdx.y] = sum;
    return;
  }
  float halfmax = s.H * MAX_TOP_ERROR_FACTOR_F;
  float scaled = sum, scale = 1.0f;
  if (sum > halfmax) {
    scale = soft_clip_dev(sum, halfmax);
    scaled = scale * sum;
  }
  float *dst = v.b.ehi + (size_t)r * s.I; /* step 0 plane */
  for (int y = threadIdx.x; y < s.H; y += 1024) /* ehi keeps column 0 and the pad at zero */
    dst[y] = (y == 0 || y > s.hidden_size) ? 0.0f : (sum > halfmax) ? herr[y] * scale : herr[y];
  if (threadIdx.x == 0) {
    v.b.top_raw[r] = sum;
    v.b.top_scaled[r] = scaled;
  }
}

/* the end of backprop_single_layer_sparse + the soft clip (recur-nn.c:719-721) for streams whose
 * rows were shared out over nb workgroups: the partial sums in order, the scale, the row */
__global__ __launch_bounds__(256) void k_top_backprop_scale(View v, int row0, const unsigned char *active,
                                                            const float *part, int nb) {
  const RamdShape &s = v.sh;
  const int j = blockIdx.x, r = row0 + j;
  if (active && !active[j]) return;
  float sum = 0.0f;
  for (int k = 0; k < nb; k++) sum += part[(size_t)j * nb + k];
  const float halfmax = s.H * MAX_TOP_ERROR_FACTOR_F;
  float scaled = sum;
  if (sum > halfmax) {
    const float scale = soft_clip_dev(sum, halfmax);
    scaled = scale * sum;
    float *dst = v.b.ehi + (size_t)r * s.I;
    for (int y = threadIdx.x; y < s.H; y += 256) dst[y] *= scale; /* (0 stays 0) */
  }
  if (threadIdx.x == 0) {
    v.b.top_raw[r] = sum;
    v.b.top_scaled[r] = scaled;
  }
}

__global__ void k_live_mask(float *dst, const unsigned char *active, int n) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = (!active || active[j]) ? 1.0f : 0.0f;
}

// single_layer_sgd / _sparse for all streams at once (recur-nn.c:256-301):
// hidden^T . o_error comes from the MFMA GEMM (ProbHoDelta); this sums its K
// slabs into ho_delta.  With error ranges only the columns inside a range
// receive anything.
__global__ void k_ho_delta_finalize(View v, const float *slab, int ks, int accumulate,
                                    const int *ranges) {
  const RamdShape &s = v.sh;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  int n = s.H * s.O;
  if (e >= n) return;
  int x = e % s.O;
  float acc = accumulate ? v.b.ho_delta[e] : 0.0f;
  bool live = true;
  if (ranges) {
    live = false;
    for (int i = 0; ranges[2 * i] >= 0; i++) {
      int start = ranges[2 * i] & ~3, len = (ranges[2 * i + 1] + 3) & ~3;
      if (x >= start && x < start + len) live = true;
    }
  }
  if (live) {
    for (int z = 0; z < ks; z++) acc += slab[(size_t)z * n + e];
  }
  v.b.ho_delta[e] = acc;
}

// ------------------------------------------------------ BPTT chain step --
//
// One launch per BPTT step t (recur-nn.c:338-376 for every stream at once):
//     E[t+1][s][y] = on(X_t[s][y]) * sum_k E[t][s][k] W_ih[y][k],   y = 1..hidden_size
// plus the per-stream sum of squares.  The hidden->hidden block is all the next
// step needs, and for a power-of-two hidden size it tiles exactly: 32 x 32
// output tiles, (S/32) x (hidden/32) workgroups = 256 at the 1024 / 256 size,
// one per CU, no split-K slabs and no separate finalize pass.  Column 0 (bias
// row) and the real-input rows only feed the sum of squares and are done for
// all steps together afterwards (k_extras_gather, or the ProbExtras GEMM for very wide nets).
//
// Workgroup = 8 waves.  Waves 0-3 multiply: each takes a quarter of every 128-deep K
// stage (in-workgroup split-K, summed through LDS at the end).  Waves 4-7 only move data:
// operand stages go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round
// trip), three stages deep, with counted s_waitcnt vmcnt and raw s_barrier so that two
// stages stay in flight across barriers.  LDS rows are 128 floats; the 16-byte chunk c
// of row r is stored at chunk position c ^ (r & 15), applied on the DMA's global
// address (the LDS side of a DMA is lane-linear) and again on the ds_read_b128
// address, which makes the 16-lane groups of ds_read_b128 conflict free.
// Fragments are fetched with inline-asm ds_read_b128 one stage ahead of the MFMAs that
// use them (hipcc would otherwise drain vmcnt to 0 before any LDS read that may alias an
// LDS-DMA destination, and would not overlap the reads with the previous stage's MFMAs).

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

// stores the launch-invariant part of the kernel arguments in device memory (see below)
__global__ void k_store_view(View v, View *dst) { *dst = v; }

// The View is read from device memory instead of coming by value: 520 bytes of kernel
// arguments per launch are fetched from the host-visible argument ring, and this kernel is
// launched D times per generation.  Only the ring position changes between generations, and
// that comes as a plain argument.
template <bool UNI, int NS = 0> /* NS > 0: the number of K stages, known at compile time */
__global__ __launch_bounds__(512) void k_chain_main(const View *__restrict__ vp, int uniform_idx,
                                                    int row0, int nrows, int t, int tm, int tn,
                                                    int nstages_arg) {
  View v = *vp;
  v.b.uniform_idx = uniform_idx;
  const int nstages = NS > 0 ? NS : nstages_arg;
  __shared__ __attribute__((aligned(16))) float smem[C_STAGES * C_STAGE_FLOATS];
  const RamdShape &s = v.sh;
  const int L = blockIdx.x;
  const int xcd = L & 7, q = L >> 3;
  const int mt = q % tm, nt = (q / tm) * 8 + xcd; /* the m tiles of one W panel share an XCD */
  if (nt >= tn) return;
  const int m0 = mt * CM, n0 = 1 + nt * CN;       /* output columns start at 1 */
  // 8 waves: 0-3 multiply (one quarter of every K stage each), 4-7 only feed the LDS
  // ring.  An LDS-DMA instruction costs its issuing wave 100-200 cycles; on a wave of its
  // own that cost overlaps the other waves' MFMAs instead of delaying them.
  const int wave8 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int lm = lane & 31, kh = lane >> 5;
  const float *ehi_t = v.b.ehi + ((size_t)t * s.Scap + row0) * s.I;

  // --- LDS-DMA source addresses of this lane: 8 instructions per stage and wave.
  // Instruction i (0..31 over the workgroup) fills rows 2 (i & 15), +1 of A (i < 16) or B.
  const float *src[8];
  int kcol[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    int i = wave * 8 + j;
    int row = 2 * (i & 15) + (lane >> 5);
    int c = (lane & 31) ^ (row & 15); /* global chunk stored at this LDS position */
    const float *base;
    if (i < 16) {
      int r = m0 + row;
      base = ehi_t + (size_t)(r < nrows ? r : nrows - 1) * s.I;
    } else {
      int n = n0 + row;
      base = v.b.ih_w + (size_t)(n < s.I ? n : s.I - 1) * s.H;
    }
    src[j] = base + 1 + 4 * c; /* K runs over the hidden columns 1..hidden_size */
    kcol[j] = 1 + 4 * c;
  }
  // a stage whose 128 columns all lie inside K needs no per-chunk test
  auto issue_one = [&](int stage, int j) {
    float *dst = smem + (stage % C_STAGES) * C_STAGE_FLOATS + (wave * 8 + j) * 256;
    const int k0 = stage * CK;
    const float *g = src[j] + k0;
    /* last, partial stage only: chunks wholly past the hidden columns come from a zero line
     * (a chunk that straddles the end reads pad columns, which are zero in E) */
    if (k0 + CK > s.hidden_size) g = (k0 + kcol[j] <= s.hidden_size) ? g : v.b.zeros;
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)dst, 16, 0, 0);
  };
  auto issue = [&](int stage) {
    if (stage * CK + CK <= s.hidden_size) { /* a full stage: plain address arithmetic, nothing to select */
      float *dst = smem + (stage % C_STAGES) * C_STAGE_FLOATS + wave * 8 * 256;
#pragma unroll
      for (int j = 0; j < 8; j++)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(src[j] + stage * CK), (lds_void_t *)(dst + j * 256), 16,
                                         0, 0);
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) issue_one(stage, j);
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;

  // --- what the epilogue needs from global memory (this thread's 4 output columns of one
  // row: the input values that gate them, and the operands of a short K tail) is fetched
  // now, so that its latency hides under the main loop
  const int etid = threadIdx.x & 255; /* epilogue work is done by the compute waves */
  const int erow_i = etid >> 3, ec4 = (etid & 7) * 4;
  const int er = m0 + erow_i < nrows ? m0 + erow_i : nrows - 1;
  const float *xrow = input_row<UNI>(v, row0 + er, t);
  float xin[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (loader) {
    /* the loaders' first stages go out before anything else in the workgroup touches memory */
#pragma unroll
    for (int p = 0; p < C_STAGES - 1; p++)
      if (p < nstages) issue(p);
  }
  // what the epilogue needs from global memory (the input values that gate this thread's four
  // outputs), requested by the compute waves now so that it has arrived by the end of the loop
  if (!loader) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int n = n0 + ec4 + i;
      xin[i] = xrow[n <= s.hidden_size ? n : s.hidden_size];
    }
  }
  const uint32_t lds0 = lds_byte_addr(smem);
  const uint32_t rowoff = (uint32_t)lm * (CK * 4u);
  if (loader) {
#pragma unroll
    for (int st = 0; st < nstages; st++) {
      // stages st+1 .. st+C_STAGES-2 may stay in flight (8 DMAs per stage and loader wave)
      const int ahead = min(C_STAGES - 2, nstages - 1 - st);
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); /* stage st has landed; stage st-1's buffer is free */
      if (st + C_STAGES - 1 < nstages) issue(st + C_STAGES - 1);
    }
  } else {
    // Compute waves run one stage behind their own LDS reads: the fragments of stage
    // st + 1 are requested right after its barrier and arrive while the 16 dependent
    // MFMAs of stage st execute, so neither the ds_read latency nor the barrier sits
    // between two MFMA blocks.
    auto rd = [&](int st, f32x4 (&a)[4], f32x4 (&b)[4]) {
      const uint32_t abase = lds0 + (uint32_t)((st % C_STAGES) * C_STAGE_FLOATS) * 4u;
      const uint32_t bbase = abase + (uint32_t)(CM * CK) * 4u;
#pragma unroll
      for (int gi = 0; gi < 4; gi++) {
        int c = 2 * (4 * wave + gi) + kh;               /* chunk = 4 consecutive k */
        uint32_t off = rowoff + (uint32_t)((c ^ (lm & 15)) * 16);
        a[gi] = lds_read_b128(abase + off);
        b[gi] = lds_read_b128(bbase + off);
      }
    };
    auto step = [&](int st, f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&an)[4], f32x4 (&bn)[4]) {
      /* every read issued so far has arrived: this stage's fragments are usable, and the
       * loaders may overwrite its buffer after the next barrier */
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (st + 1 < nstages) {
        __builtin_amdgcn_s_barrier(); /* stage st + 1 has landed */
        rd(st + 1, an, bn);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int gi = 0; gi < 4; gi++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].x, b[gi].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].y, b[gi].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].z, b[gi].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].w, b[gi].w, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (NS > 0) {
      f32x4 a0[4], b0[4], a1[4], b1[4];
      __builtin_amdgcn_s_barrier(); /* stage 0 has landed */
      rd(0, a0, b0);
#pragma unroll
      for (int st = 0; st < NS; st += 2) {
        step(st, a0, b0, a1, b1);
        if (st + 1 < NS) step(st + 1, a1, b1, a0, b0);
      }
    } else {
      /* Any number of stages: read, wait, multiply, stage by stage.  The read-ahead form above
       * is only used fully unrolled: in a rolled loop the compiler may copy the ping-pong
       * fragment registers right after the ds_read that fills them -- before the data has
       * arrived -- since it cannot see that an inline-asm load completes later. */
      f32x4 a[4], b[4];
      for (int st = 0; st < nstages; st++) {
        __builtin_amdgcn_s_barrier(); /* stage st has landed */
        rd(st, a, b);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]),
                       "+v"(b[3])
                     :
                     : "memory");
#pragma unroll
        for (int gi = 0; gi < 4; gi++) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].x, b[gi].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].y, b[gi].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].z, b[gi].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].w, b[gi].w, acc, 0, 0, 0);
        }
      }
    }
  }
  // --- sum the four waves' partial tiles through LDS.  The ring buffer that stage
  // `nstages` would have used holds stage nstages - 3, which every wave finished reading
  // two barriers ago and no DMA targets any more: it can be overwritten without a barrier.
  float *red = smem + (nstages % C_STAGES) * C_STAGE_FLOATS; /* [4][32][32] */
  if (!loader) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      int row = (g & 3) + 8 * (g >> 2) + 4 * kh;
      red[(wave * CM + row) * CN + lm] = acc[g];
    }
  }
  __syncthreads();
  if (loader) return;
  const int row = etid >> 3, c4 = (etid & 7) * 4;
  float e[4];
  {
    float4 p0 = ld4(red + (0 * CM + row) * CN + c4), p1 = ld4(red + (1 * CM + row) * CN + c4);
    float4 p2 = ld4(red + (2 * CM + row) * CN + c4), p3 = ld4(red + (3 * CM + row) * CN + c4);
    e[0] = (p0.x + p1.x) + (p2.x + p3.x);
    e[1] = (p0.y + p1.y) + (p2.y + p3.y);
    e[2] = (p0.z + p1.z) + (p2.z + p3.z);
    e[3] = (p0.w + p1.w) + (p2.w + p3.w);
  }
  const int r = m0 + row;
  float sq = 0.0f;
  if (r < nrows) {
    float *dst = v.b.ehi + ((size_t)(t + 1) * s.Scap + row0 + r) * s.I;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int n = n0 + c4 + i;
      if (n <= s.hidden_size) {
        float ev = e[i];
        float xi = xin[i];
        bool on = xi != 0.0f && (s.activation != 5 || xi < 20.0f);
        ev = on ? ev : 0.0f;
        if (on && s.activation == 2) ev /= 2 * (xi + 1.0f);
        dst[n] = ev;
        sq += ev * ev;
      }
    }
  }
  sq += __shfl_xor(sq, 1, 64);
  sq += __shfl_xor(sq, 2, 64);
  sq += __shfl_xor(sq, 4, 64);
  if ((etid & 7) == 0 && r < nrows)
    v.b.esum_part[((size_t)t * (tn + 1) + nt) * s.Scap + row0 + r] = sq;
}

// ------------------------------------- BPTT chain step, 64 x 64 tiles (big sets) --
//
// The same step as k_chain_main for sets with many streams of a wide net (rnnca: 512 streams,
// hidden 2048), where the one-launch chain does not apply (its W panel would not fit the
// registers) and k_chain_main's 32 x 32 tiles are bound by the operand stream: a 32 x 32 tile
// moves 32 KB into LDS per 1024 MFMA cycles of a wave, 1024 workgroups x 512 KB = 512 MB of
// L2 -> LDS traffic per step.  Here the tile is 64 streams x 64 columns, the four multiplying
// waves own a 32 x 32 quadrant each over the WHOLE K (no split-K, no cross-wave reduction), and a
// K stage is 64 deep: 32 KB per 2048 MFMA cycles, half the bytes per MFMA, 256 workgroups = one
// per CU for 512 x 2048.  Staging is k_chain_main's: both operands K-contiguous, 16-byte chunk c
// of row r at position c ^ (r & 15) so that the b128 fragment reads are conflict free, four
// loader waves with LDS-DMA into a four-deep ring, fragments of stage st + 1 read while stage
// st multiplies (fully unrolled: NS stages).  Epilogue as in k_chain_main (zero-row mask, RESQRT
// derivative, store, per-tile sum of squares), one partial per 64 columns.
// Preconditions (launcher): every stream at one ring position, streams % 64 == 0,
// hidden_size == 64 * NS.
constexpr int WM = 64, WN = 64, WK = 64, W_STAGES = 4;
constexpr int W_STAGE_FLOATS = (WM + WN) * WK; /* 32 KB */
template <int NS>
__global__ __launch_bounds__(512) void k_chain_wide(const View *__restrict__ vp, int uniform_idx, int row0,
                                                    int nrows, int t, int tm, int tn) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  View v = *vp;
  v.b.uniform_idx = uniform_idx;
  const RamdShape &s = v.sh;
  const int L = blockIdx.x;
  const int xcd = L & 7, q = L >> 3;
  const int mt = q % tm, nt = (q / tm) * 8 + xcd; /* the m tiles of one W panel share an XCD */
  if (nt >= tn) return;
  const int m0 = mt * WM, n0 = 1 + nt * WN; /* output columns start at 1 */
  const int wave8 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int lm = lane & 31, kh = lane >> 5;
  const float *ehi_t = v.b.ehi + ((size_t)t * s.Scap + row0) * s.I;
  const uint32_t lds0 = lds_byte_addr(wsm);

  if (loader) {
    // instruction i (0..31 over the four loader waves) fills rows 4 (i & 15) .. + 3 of A (i < 16)
    // or B: lane l brings chunk (l & 15) ^ (row & 15) of row 4 (i & 15) + (l >> 4)
    const float *src[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = wave * 8 + j;
      const int row = 4 * (i & 15) + (lane >> 4);
      const int c = (lane & 15) ^ (row & 15);
      const float *base = i < 16 ? ehi_t + (size_t)(m0 + row) * s.I : v.b.ih_w + (size_t)(n0 + row) * s.H;
      src[j] = base + 1 + 4 * c; /* K runs over the hidden columns 1..hidden_size */
    }
    auto issue = [&](int stage) {
      float *dst = wsm + (stage % W_STAGES) * W_STAGE_FLOATS + wave * 8 * 256;
#pragma unroll
      for (int j = 0; j < 8; j++)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(src[j] + stage * WK), (lds_void_t *)(dst + j * 256), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < W_STAGES - 1; p++)
      if (p < NS) issue(p);
#pragma unroll
    for (int st = 0; st < NS; st++) {
      const int ahead = (NS - 1 - st) < (W_STAGES - 2) ? (NS - 1 - st) : (W_STAGES - 2);
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); /* stage st has landed; stage st - 1's buffer is free */
      if (st + W_STAGES - 1 < NS) issue(st + W_STAGES - 1);
    }
    __syncthreads();
    return;
  }

  // ------------------------------------------------------------------ multiply
  const int wm = wave >> 1, wn = wave & 1;
  // the gate values of this thread's 4 x 4 outputs in the epilogue, requested now
  const int etid = threadIdx.x; /* 0..255 */
  const int rq = etid >> 4, c4 = (etid & 15) * 4;
  float xin[4][4];
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const float *xrow = input_row<true>(v, row0 + m0 + 4 * rq + rr, t) + n0 + c4;
#pragma unroll
    for (int i = 0; i < 4; i++) xin[rr][i] = xrow[i];
  }
  f32x16 acc; /* (a second accumulator taking turns with this one measured no difference) */
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;
  const uint32_t arow = (uint32_t)(wm * 32 + lm) * (WK * 4u), brow = (uint32_t)(WM + wn * 32 + lm) * (WK * 4u);
  auto rd = [&](int st, f32x4 (&a)[8], f32x4 (&b)[8]) {
    const uint32_t base = lds0 + (uint32_t)((st % W_STAGES) * W_STAGE_FLOATS) * 4u;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t off = (uint32_t)(((2 * u + kh) ^ (lm & 15)) * 16);
      a[u] = lds_read_b128(base + arow + off);
      b[u] = lds_read_b128(base + brow + off);
    }
  };
  auto step = [&](int st, f32x4 (&a)[8], f32x4 (&b)[8], f32x4 (&an)[8], f32x4 (&bn)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this stage's fragments have arrived */
    if (st + 1 < NS) {
      __builtin_amdgcn_s_barrier(); /* stage st + 1 has landed */
      rd(st + 1, an, bn);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    f32x4 a0[8], b0[8], a1[8], b1[8];
    __builtin_amdgcn_s_barrier(); /* stage 0 has landed */
    rd(0, a0, b0);
#pragma unroll
    for (int st = 0; st < NS; st += 2) {
      step(st, a0, b0, a1, b1);
      if (st + 1 < NS) step(st + 1, a1, b1, a0, b0);
    }
  }
  // the tile through LDS (the ring buffer stage NS would have used was read four barriers ago)
  float *red = wsm + (NS % W_STAGES) * W_STAGE_FLOATS; /* [64][64] */
#pragma unroll
  for (int g = 0; g < 16; g++) {
    const int row = wm * 32 + (g & 3) + 8 * (g >> 2) + 4 * kh;
    red[row * WN + wn * 32 + lm] = acc[g];
  }
  __syncthreads();
  float *dst0 = v.b.ehi + ((size_t)(t + 1) * s.Scap + row0 + m0) * s.I + n0 + c4;
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int row = 4 * rq + rr;
    const float4 e4 = ld4(red + row * WN + c4);
    const float e[4] = {e4.x, e4.y, e4.z, e4.w};
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float ev = e[i];
      const float xi = xin[rr][i];
      const bool on = xi != 0.0f && (s.activation != 5 || xi < 20.0f);
      ev = on ? ev : 0.0f;
      if (on && s.activation == 2) ev /= 2 * (xi + 1.0f);
      dst0[(size_t)row * s.I + i] = ev;
      sq += ev * ev;
    }
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    sq += __shfl_xor(sq, 4, 64);
    sq += __shfl_xor(sq, 8, 64);
    if ((etid & 15) == 0) v.b.esum_part[((size_t)t * (tn + 1) + nt) * s.Scap + row0 + m0 + row] = sq;
  }
}

// ------------------------------------ forward GEMM, 64 x 64 tiles (big sets) --
//
// hidden sums = X . W_ih (recur-nn.c:117-119) for big sets of dense-input nets -- rnnca's frame
// fill is 13,824 forward-only cells of a 2048-hidden net per frame, 118 GFLOP -- on k_chain_wide's
// plan: 64 rows x 64 columns per workgroup, four multiplying waves with a 32 x 32 quadrant each
// over the whole K, four loader waves with LDS-DMA into a four-deep ring of 64-deep stages.
// A (the input rows, K-contiguous) is staged as in the chain: chunk c of row r at position
// c ^ (r & 15), one b128 read = four k.  B is W_ih itself, K-major: a stage is [64 k][64 columns]
// as it lies in memory (one DMA instruction = four k rows of 256 bytes), and a lane fetches its
// four k of a chunk with two ds_read2_b32 (as k_fwd_fused does).  K = i_size is padded to whole
// stages with zeros (chunks and rows past i_size come from a zero line); the column tiles start
// at column 0 and the last one runs past h_size, where nothing is stored.  The sums go to slab
// plane 0 for k_fwd_finalize (noise, activation, bias node).
// Preconditions (launcher): rows % 64 == 0, NS == ceil(i_size / 64).
template <int NS>
__global__ __launch_bounds__(512) void k_fwd_wide(const View *__restrict__ vp, int uniform_idx, int row0,
                                                  int nrows, int tm, int tn) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  View v = *vp;
  v.b.uniform_idx = uniform_idx;
  const RamdShape &s = v.sh;
  /* Blocks are dealt round-robin over the 8 XCDs and an XCD runs 32 workgroups at a time: those 32
   * are a supertile of 4 row tiles x 8 column tiles, so that while they walk K together every
   * stage of the input rows is fetched into that XCD's L2 once per 8 workgroups and every stage of
   * W once per 4 (one column tile after another for all the row tiles, as the chain's mapping
   * does, re-reads the whole input set once per column tile: 3.8 GB for a 13,824-cell frame). */
  const int L = blockIdx.x;
  const int xcd = L & 7, q = L >> 3;
  const int st_i = (q >> 5) * 8 + xcd, in_i = q & 31;
  const int stm = (tm + 3) >> 2;
  const int mt = (st_i % stm) * 4 + (in_i & 3), nt = (st_i / stm) * 8 + (in_i >> 2);
  if (mt >= tm || nt >= tn) return;
  const int m0 = mt * WM, n0 = nt * WN;
  const int wave8 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int lm = lane & 31, kh = lane >> 5;
  const uint32_t lds0 = lds_byte_addr(wsm);

  if (loader) {
    // instruction i (0..31): i < 16: rows 4 i .. + 3 of A (lane l: chunk (l & 15) ^ (row & 15) of
    // row 4 i + (l >> 4)); i >= 16: k rows 4 (i - 16) .. + 3 of B (lane l: 16-byte piece l & 15 of
    // k row 4 (i - 16) + (l >> 4), i.e. columns n0 + 4 (l & 15) ..)
    const float *src[8];
    int kofs[8]; /* A: first k of this lane's chunk within a stage; B: this lane's k row within a stage */
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = wave * 8 + j;
      if (i < 16) {
        const int row = 4 * i + (lane >> 4);
        const int c = (lane & 15) ^ (row & 15);
        src[j] = input_row<false>(v, row0 + m0 + row, 0) + 4 * c;
        kofs[j] = 4 * c;
      } else {
        const int k = 4 * (i - 16) + (lane >> 4);
        src[j] = v.b.ih_w + (size_t)k * s.H + n0 + 4 * (lane & 15);
        kofs[j] = k;
      }
    }
    auto issue = [&](int stage) {
      float *dst = wsm + (stage % W_STAGES) * W_STAGE_FLOATS + wave * 8 * 256;
      const int k0 = stage * WK;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool a_side = wave * 8 + j < 16;
        const float *g = a_side ? src[j] + k0 : src[j] + (size_t)k0 * s.H;
        if ((k0 + WK > s.I && k0 + kofs[j] >= s.I) || (!a_side && n0 + 4 * (lane & 15) >= s.H))
          g = v.b.zeros + 4 * (lane & 15); /* past K, or past the last column of W: zeros */
        __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(dst + j * 256), 16, 0, 0);
      }
    };
#pragma unroll
    for (int p = 0; p < W_STAGES - 1; p++)
      if (p < NS) issue(p);
#pragma unroll
    for (int st = 0; st < NS; st++) {
      const int ahead = (NS - 1 - st) < (W_STAGES - 2) ? (NS - 1 - st) : (W_STAGES - 2);
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); /* stage st has landed; stage st - 1's buffer is free */
      if (st + W_STAGES - 1 < NS) issue(st + W_STAGES - 1);
    }
    __syncthreads();
    return;
  }

  // ------------------------------------------------------------------ multiply
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;
  const uint32_t arow = (uint32_t)(wm * 32 + lm) * (WK * 4u);
  const uint32_t bcol = (uint32_t)(WM * WK + wn * 32 + lm) * 4u; /* B: [k][64 columns] behind A */
  auto rd = [&](int st, f32x4 (&a)[8], f32x2 (&b0)[8], f32x2 (&b1)[8]) {
    const uint32_t base = lds0 + (uint32_t)((st % W_STAGES) * W_STAGE_FLOATS) * 4u;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int c = 2 * u + kh; /* chunk = k 4 c .. 4 c + 3 of the stage, on both operands */
      a[u] = lds_read_b128(base + arow + (uint32_t)((c ^ (lm & 15)) * 16));
      const uint32_t baddr = base + bcol + (uint32_t)(4 * c) * (WN * 4u);
      b0[u] = lds_read2_b32_w64(baddr);     /* k, k + 1 */
      b1[u] = lds_read2_b32_w64_hi(baddr);  /* k + 2, k + 3 */
    }
  };
  auto step = [&](int st, f32x4 (&a)[8], f32x2 (&b0)[8], f32x2 (&b1)[8], f32x4 (&an)[8], f32x2 (&b0n)[8],
                  f32x2 (&b1n)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this stage's fragments have arrived */
    if (st + 1 < NS) {
      __builtin_amdgcn_s_barrier(); /* stage st + 1 has landed */
      rd(st + 1, an, b0n, b1n);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; u++) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b0[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b0[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b1[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b1[u].y, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    f32x4 a0[8], a1[8];
    f32x2 p0[8], q0[8], p1[8], q1[8];
    __builtin_amdgcn_s_barrier(); /* stage 0 has landed */
    rd(0, a0, p0, q0);
#pragma unroll
    for (int st = 0; st < NS; st += 2) {
      step(st, a0, p0, q0, a1, p1, q1);
      if (st + 1 < NS) step(st + 1, a1, p1, q1, a0, p0, q0);
    }
  }
  // the tile through LDS (the ring buffer stage NS would have used was read four barriers ago)
  float *red = wsm + (NS % W_STAGES) * W_STAGE_FLOATS; /* [64][64] */
#pragma unroll
  for (int g = 0; g < 16; g++) {
    const int row = wm * 32 + (g & 3) + 8 * (g >> 2) + 4 * kh;
    red[row * WN + wn * 32 + lm] = acc[g];
  }
  __syncthreads();
  const int etid = threadIdx.x, rq = etid >> 4, c4 = (etid & 15) * 4;
  if (n0 + c4 < s.H) { /* h_size % 4 == 0: whole float4s */
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int row = 4 * rq + rr;
      *reinterpret_cast<float4 *>(v.b.slab + (size_t)(m0 + row) * s.H + n0 + c4) = ld4(red + row * WN + c4);
    }
  }
}

// ------------------------------------------------ BPTT chain, one launch --
//
// All D steps of the chain in ONE launch (hidden 1024: 32 column tiles; up to 8 row tiles of
// 32 streams).  What a kernel boundary costs the launch-per-step form -- 1.5 us of boundary,
// 1.7 us until the first operand stage has landed from a cold L2, the W panel fetched again
// every step -- is most of its 8.9 us; here the 32 workgroups that work on one row tile all
// run on ONE XCD (each reads the XCD it runs on from HW_REG_XCC_ID and draws its column tile
// by a ticket on that XCD: with one workgroup per CU and 32 CUs per XCD every XCD gets exactly
// 32, whatever the dispatch order), so that a step's output rows travel producer -> consumer
// through that XCD's L2 (plain stores, drained; a plain flag word per producer wave; L1-
// bypassing `sc1` polls and LDS-DMA loads), and each row tile is split into two independent
// sub-chains of 16 streams whose steps ALTERNATE on the matrix pipe: while the 32 workgroups
// exchange sub-chain a's step t, they multiply sub-chain b's.  (A row tile's recurrences are
// independent per stream: E[t+1][s] = mask[t][s] . (E[t][s] W^T), recur-nn.c:338-376.)
//
// Workgroup = 8 waves.  Waves 4-7 (one per SIMD) only multiply: the workgroup's W panel
// (32 output columns x 1024 k) lives in their registers for the whole launch as MFMA B
// fragments (v_mfma_f32_16x16x4_f32; each wave a quarter of K: 128 VGPRs), the A operand --
// 16 error rows x 1024 k per sub-chain, 64 KB -- is read from LDS with conflict-free
// ds_read_b128 (16-byte chunk c of row m sits at position c ^ m).  Waves 0-3 do everything
// else for four of the 16 rows each: finish the previous half-step (sum the four K quarters
// from LDS, zero-row mask, RESQRT derivative, sum of squares, store the rows), publish, poll
// the 32 producers of their rows, and pull the next operand into LDS by LDS-DMA.  One
// s_barrier per half-step couples the two groups.  Every poll is bounded; a time-out or a
// surplus ticket raises the abort word (host-mapped), which the library checks at its next
// synchronisation and aborts on -- results are never silently wrong.
constexpr int PC_SUB = 16;                  /* streams per sub-chain                */
constexpr int PC_RED_FLOATS = 4 * PC_SUB * 32;
constexpr int pc_lds_bytes(int K) { return (2 * PC_SUB * K + 2 * PC_RED_FLOATS) * 4 + 64; }
/* s_sleep units (64 cycles) between the barrier and the first poll; the producers publish ~0.35 us after
 * the barrier and the flag is visible in the XCD's L2 ~0.2 us later.  Round 3 (20 steps of 1024 / 256,
 * us per chain): first poll after 16 / 18 / 22 / 24 / 26 / 32 units with gaps from K block 3 on =
 * 114 / 105 / 100.6 / 100.7 / 101.3 / 105. */
#ifndef PC_SLEEP0
#define PC_SLEEP0 22
#endif
#ifndef PC_SLEEP1
#define PC_SLEEP1 1
#endif
#ifndef PC_SLEEP0_ONE
#define PC_SLEEP0_ONE 12 /* 16-stream row tiles: the publish comes in an otherwise empty half-step */
#endif
#ifndef PC_SLEEP0_SMALL
#define PC_SLEEP0_SMALL 14 /* hidden 512 / 256: the half-step is shorter, the publish comes at the same ~0.35 us */
#endif
constexpr unsigned PC_EPOCH = 64;           /* flag values per launch (depth <= 60)  */
#ifndef PC_POLL_SCALAR
#define PC_POLL_SCALAR 0
#endif
/* K blocks (8 MFMAs each) after which the multiplying waves pause for 64 cycles, per hidden size:
 * measured at 1024: after blocks 3-6 100.6 us per chain, 3-7 100.8, 4-7 100.7, 3-8 100.9, 2-6 105.5,
 * 2-9 104.6, every block from 3 on 109.9, blocks 5 / 7 / 9 / 11 103.6-105.1, none (the poll then
 * completes when the burst has ended) 133. */
#ifndef PC_GAPS
#define PC_GAPS (PC_POLL_SCALAR ? 0x0 : 0x78)
#endif
#ifndef PC_GAPS_512
#define PC_GAPS_512 (PC_POLL_SCALAR ? 0x0 : 0x3c)
#endif
#ifndef PC_GAPS_256
#define PC_GAPS_256 (PC_POLL_SCALAR ? 0x0 : 0xe)
#endif
#ifndef PC_FILL_MFMAS
#define PC_FILL_MFMAS 0
#endif
#ifndef PC_FETCH_PRIO
#define PC_FETCH_PRIO 0
#endif
#ifndef PC_GAP_NOPS
#define PC_GAP_NOPS 0
#endif
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

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

struct ChainSync {
  unsigned tickets[8];       /* per XCD, monotonic over launches                     */
  unsigned pad0[24];
  unsigned flags[32][2][4][32]; /* [row tile][sub-chain][fetching wave][column tile]   */
  unsigned abort;            /* raised by any workgroup that gives up                */
};

typedef __attribute__((address_space(1))) unsigned gu32;

#ifdef PC_STAMPS /* development builds only (tools/mkabl.sh -DPC_STAMPS): where a half-step's time goes */
__device__ unsigned long long g_pc_stamps[2][64][8];
#define PC_STAMP(role, k, slot)                                                                    \
  do {                                                                                             \
    if (g == 0 && j == 0 && lane == 0 && (wave8 & 3) == 0 && (k) < 64)                             \
      g_pc_stamps[role][k][slot] = __builtin_amdgcn_s_memrealtime();                               \
  } while (0)
#else
#define PC_STAMP(role, k, slot) do { } while (0)
#endif

/* ONE: row tiles of 16 streams, i.e. only sub-chain a exists and every other half-step is empty
 * (the workgroup multiplies, then finishes and publishes, then waits for the 32 producers of its
 * next operand): 4.5 instead of 6.4 us per step for HALF the streams per workgroup -- worse per
 * stream, but a small set (64 streams at hidden 1024: a GPU's share of 512 on eight) then runs
 * on twice as many CUs.  The launcher picks it when the 16-stream tiles still fit one launch. */
/* PAD (with ONE): the set is not whole row tiles -- only rows [vlo, nvalid) of the launch are its
 * own; the others belong to other streams or to nobody: they are multiplied like the rest (rows
 * do not mix) and never stored. */
template <int ACT, int K, bool ONE = false, bool PAD = false> /* rnn_activation; hidden size: 1024, 512 or 256 */
__global__ __launch_bounds__(512) void k_chain_persist(const View *__restrict__ vp, int uniform_idx,
                                                       int row0, int nrows, int depth, unsigned seq,
                                                       ChainSync *sy, unsigned *host_abort, int nvalid, int vlo) {
  extern __shared__ __attribute__((aligned(16))) float psm[];
  constexpr int BUF = PC_SUB * K;             /* one sub-chain's operand (64 KB at K = 1024) */
  constexpr int NT = K / 32;                  /* column tiles of a row tile            */
  constexpr int KB = K / 64;                  /* 16-k blocks of a wave's K quarter     */
  constexpr int PPR = K / 256;                /* 1 KB LDS-DMA pieces per operand row   */
  float *abuf = psm;                          /* [2][16][K], swizzled chunks           */
  float *red = psm + 2 * BUF;                 /* [2][4 waves][16 rows][32 cols]        */
  unsigned *wg_info = reinterpret_cast<unsigned *>(red + 2 * PC_RED_FLOATS);
  View v = *vp;
  v.b.uniform_idx = uniform_idx;
  const RamdShape &s = v.sh;
  const int wave8 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int TR = ONE ? PC_SUB : 2 * PC_SUB; /* streams of a row tile */
  const int mtiles = nrows / TR;

  // --- which XCD am I on, and which of its 32 seats do I get
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; /* HW_REG_XCC_ID */
    const unsigned t = __hip_atomic_fetch_add(&sy->tickets[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                       (seq - 1u) * 32u;
    wg_info[0] = xcc;
    wg_info[1] = t;
  }
  __syncthreads();
  /* (wave-uniform by construction: say so, or every address built from them goes through the vector ALU) */
  const unsigned seat = __builtin_amdgcn_readfirstlane(wg_info[1]);
  /* row tile = XCD + 8 x (seat / column tiles): the first 8 row tiles spread over the 8 XCDs
   * before any XCD takes a second one; column tile = seat % column tiles */
  const int g = __builtin_amdgcn_readfirstlane((int)wg_info[0]) + 8 * (int)(seat / NT);
  if (seat >= 32u) { /* cannot happen with one workgroup per CU on a 256-CU part */
    if (threadIdx.x == 0) {
      __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(host_abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  if (g >= mtiles) return; /* fewer row tiles than seats: nothing to do here */
  const int j = (int)(seat % NT);
  const int m0 = TR * g, n0 = 1 + 32 * j;     /* output columns start at 1 */
  const unsigned epoch0 = seq * PC_EPOCH;
  const int halfsteps = 2 * depth;
  const int tn = NT;
  const size_t plane_stride = (size_t)s.Scap * s.I;

  if (wave8 >= 4) {
    // ============================================ multiply, finish, publish
    // Waves 4-7, one per SIMD.  Wave wv multiplies the K quarter wv of BOTH 16 x 16 tiles of
    // every half-step and, after the barrier, finishes rows 4 wv .. 4 wv + 3 of the tile
    // pair: the four K quarters summed from LDS, the zero-row mask, the RESQRT derivative, the
    // store, and -- once the stores have drained -- the flag that the 32 consumers of those rows
    // poll.  (The f32 MFMA runs on the SIMD's vector ALU at the vector rate: a partner wave's VALU
    // work does not overlap with it, so the epilogue belongs in the wave that owns the ALU.)
    //
    // Round 3: everything between the barrier and the flag is on the critical path of all 32
    // consumers, so it is written instruction by instruction:
    //   * the row stores, the gate loads and the flag go through an SGPR base + a launch-invariant
    //     per-lane byte offset (inline asm, `global_*` with saddr): no 64-bit address arithmetic,
    //     no register arrays indexed by the sub-chain (hipcc had turned those into a dozen
    //     v_cndmask per store and FLAT stores);
    //   * the flag is a PLAIN store (it stays in this XCD's L2, where the consumers' L1-bypassing
    //     polls find it).  As a `volatile` store hipcc made it `sc0 sc1` and put an
    //     `s_waitcnt vmcnt(0)` BEHIND it (SIMemoryLegalizer's rule for volatile accesses): every
    //     half-step waited a second time, for the flag's own acknowledgement, before its MFMAs;
    //   * the sum of squares of each error row (recur-nn.c:371) is no longer taken here (two
    //     multiplies, ten LDS-crossbar shuffles in five dependent round trips and two more stores
    //     in front of the drain): k_extras_control holds every error row in registers anyway and
    //     sums it there (tn = 0 in its arguments);
    //   * the A fragments come from LDS by inline-asm ds_read_b128 three K blocks ahead of the
    //     MFMAs that use them, with counted lgkmcnt waits (hipcc's own schedule had half of the
    //     sixteen reads directly in front of their first MFMA: 60-100 cycles of idle matrix pipe
    //     each), the first three before the finish, whose drain covers their latency.
    const int wv = __builtin_amdgcn_readfirstlane(wave8) - 4, m = lane & 15, kq = lane >> 4;
    const int col = lane & 31, rh = lane >> 5;
    float wreg[KB][4][2];
    {
      const float *wb = v.b.ih_w + (size_t)(n0 + m) * s.H + 1 + (K / 4) * wv + 4 * kq;
#pragma unroll
      for (int u = 0; u < KB; u++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int h = 0; h < 2; h++) wreg[u][i][h] = wb[(size_t)16 * h * s.H + 16 * u + i];
      /* The panel has to have LANDED before the loop, as far as hipcc can tell: otherwise it puts the
       * `s_waitcnt vmcnt(0)` for these loads in front of the loop's first MFMA, where it waits in EVERY
       * half-step for the gate loads issued just before (inline asm, not on its scoreboard).  An empty
       * asm that reads the registers makes it wait here. */
#pragma unroll
      for (int u = 0; u < KB; u++)
        asm volatile("" : : "v"(wreg[u][0][0]), "v"(wreg[u][0][1]), "v"(wreg[u][1][0]), "v"(wreg[u][1][1]),
                     "v"(wreg[u][2][0]), "v"(wreg[u][2][1]), "v"(wreg[u][3][0]), "v"(wreg[u][3][1]));
    }
    // this thread's two outputs per half-step: rows 4 wv + rh and + 2 of the sub-chain, column
    // n0 + col.  Byte offset of (row, column) within a plane of [Scap][I] floats, relative to
    // the sub-chain's first row: the same for the error planes and the history slots.
    unsigned voff[2];
    bool mine[2][2]; /* PAD: is the row one of the set's own */
#pragma unroll
    for (int q = 0; q < 2; q++) {
      voff[q] = (unsigned)(((size_t)(4 * wv + rh + 2 * q) * s.I + n0 + col) * sizeof(float));
#pragma unroll
      for (int x = 0; x < 2; x++) {
        const int sr = m0 + PC_SUB * x + 4 * wv + rh + 2 * q;
        mine[x][q] = !PAD || (sr >= vlo && sr < nvalid);
      }
    }
    const float *ehi_sub = v.b.ehi + (size_t)(row0 + m0) * s.I; /* plane 0, sub-chain a, row 0 */
    // LDS addresses.  A fragment of K block u: chunk ((K / 16) wv + 4 u + kq) ^ m of row m; the
    // xor only touches the low four bits, i.e. (4 (u & 3) + kq) ^ m: four per-lane addresses per
    // sub-chain, the rest of u is an immediate offset.
    uint32_t a_addr[2][4];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
      for (int i = 0; i < 4; i++)
        a_addr[x][i] = lds_byte_addr(abuf + x * BUF + m * K) + 16u * (uint32_t)((K / 16) * wv + ((4 * i + kq) ^ m));
    const uint32_t red_rd = lds_byte_addr(red) + 4u * (uint32_t)((4 * wv + rh) * 32 + col);
    float xg0 = 0.f, xg1 = 0.f;
    f32x4 af[4];
    __syncthreads(); /* barrier 0: both operands of the first two half-steps have landed */

    // one half-step; XC: which sub-chain it MULTIPLIES (it finishes the other one's previous half-step)
    auto half = [&](auto XC, const int k) -> bool {
      constexpr int x = decltype(XC)::value, xf = x ^ 1;
      const bool multiplies = k < halfsteps && !(ONE && x == 1);
      PC_STAMP(0, k, 0);
      if (multiplies) { /* the first fragments: their latency hides under the finish */
        af[0] = lds_read_b128_off<0>(a_addr[x][0]);
        if (KB > 1) af[1] = lds_read_b128_off<0>(a_addr[x][1]);
        if (KB > 2) af[2] = lds_read_b128_off<0>(a_addr[x][2]);
      }
      if (k >= 1 && !(ONE && x == 0)) {
        // ---- finish half-step k - 1 (sub-chain xf, step (k - 1) >> 1)
        const int t = (k - 1) >> 1;
        float ev[2];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xg0), "+v"(xg1)); /* the gate loads of the last half-step */
        {
          const uint32_t ra = red_rd + 4u * (uint32_t)(xf * PC_RED_FLOATS);
          float p[2][4];
#pragma unroll
          for (int q = 0; q < 2; q++) {
            p[q][0] = q ? lds_read_b32_off<256 + 0 * 2048>(ra) : lds_read_b32_off<0 * 2048>(ra);
            p[q][1] = q ? lds_read_b32_off<256 + 1 * 2048>(ra) : lds_read_b32_off<1 * 2048>(ra);
            p[q][2] = q ? lds_read_b32_off<256 + 2 * 2048>(ra) : lds_read_b32_off<2 * 2048>(ra);
            p[q][3] = q ? lds_read_b32_off<256 + 3 * 2048>(ra) : lds_read_b32_off<3 * 2048>(ra);
          }
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(p[0][0]), "+v"(p[0][1]), "+v"(p[0][2]), "+v"(p[0][3]), "+v"(p[1][0]), "+v"(p[1][1]),
                         "+v"(p[1][2]), "+v"(p[1][3]));
          ev[0] = (p[0][0] + p[0][1]) + (p[0][2] + p[0][3]);
          ev[1] = (p[1][0] + p[1][1]) + (p[1][2] + p[1][3]);
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const float xi = q ? xg1 : xg0;
          const bool on = xi != 0.0f && (ACT != 5 || xi < 20.0f);
          ev[q] = on ? ev[q] : 0.0f;
          if (ACT == 2) ev[q] = on ? ev[q] / (2 * (xi + 1.0f)) : 0.0f;
        }
        const float *obase = ehi_sub + (size_t)(t + 1) * plane_stride + (size_t)xf * PC_SUB * s.I;
        if (mine[xf][0]) g_store_saddr(voff[0], ev[0], obase);
        if (mine[xf][1]) g_store_saddr(voff[1], ev[1], obase);
        PC_STAMP(0, k, 2);
        /* Drain and publish BEFORE the next MFMAs, with the vector ALU idle.  Every way of hiding
         * this wait under the multiply was slower (round 2: stores waited for 8 / 16 / 48 MFMAs
         * later 135 / 136 / 147 us per chain against 135; the finished tile handed through LDS to
         * the fetching waves 150-155 us; wave groups of their own per sub-chain 195 us): beside a
         * wave that issues f32 MFMAs back to back a store's acknowledgement comes 1-4 us late. */
#if PC_FILL_MFMAS
        /* experiment (profiles/r03_fused_delta_negative.txt): how much foreign matrix work fits into the
         * drain window?  PC_FILL_MFMAS MFMAs on a scratch accumulator between the stores and the wait. */
        {
          f32x4 junk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int f = 0; f < PC_FILL_MFMAS; f++) junk = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[0], wreg[f % KB][f & 3][0], junk, 0, 0, 0);
          asm volatile("" : : "v"(junk));
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) g_store_saddr(0u, __uint_as_float(epoch0 + (unsigned)t + 1u), &sy->flags[g][xf][wv][j]);
        PC_STAMP(0, k, 3);
      }
      if (k == halfsteps) return false; /* nothing left to multiply (nobody polls the last flag) */
      if (!multiplies) {                /* ONE: sub-chain b does not exist, an empty half-step */
        __syncthreads();
        return true;
      }
      { /* the gate values X[t][row][n0 + col] for the finish of THIS half-step, one barrier from now */
        const float *gbase = input_row<true>(v, row0 + m0 + PC_SUB * x, k >> 1);
        xg0 = g_load_saddr(voff[0], gbase);
        xg1 = g_load_saddr(voff[1], gbase);
      }
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      static_for<KB>([&](auto UC) {
        constexpr int u = decltype(UC)::value;
        constexpr int ahead = KB - 1 - u < 2 ? KB - 1 - u : 2; /* reads issued after this block's */
        lgkm_wait<ahead>(af[u & 3]);
        if (u + 3 < KB) af[(u + 3) & 3] = lds_read_b128_off<((u + 3) >> 2) * 256>(a_addr[x][(u + 3) & 3]);
        const f32x4 a = af[u & 3];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wreg[u][0][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wreg[u][0][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wreg[u][1][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wreg[u][1][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wreg[u][2][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wreg[u][2][1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wreg[u][3][0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wreg[u][3][1], acc1, 0, 0, 0);
        /* a short pause of the MFMA stream (64 cycles asleep = the last MFMA's 32 + 32 with the vector ALU
         * free): the fetching wave's one v_cmp per poll gets through here and nowhere else */
        if (!ONE && (((K == 1024 ? PC_GAPS : K == 512 ? PC_GAPS_512 : PC_GAPS_256) >> u) & 1)) { /* (ONE: nobody polls beside a burst) */
          __builtin_amdgcn_sched_barrier(0);
#if PC_GAP_NOPS
          asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#else
          __builtin_amdgcn_s_sleep(1);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      /* this wave's K quarter of the 16 x 32 tile: register r of the accumulator is row
       * 4 (lane >> 4) + r, column lane & 15 (+ 16 for the second accumulator) */
      /* (plain stores: hipcc knows how many wait states an MFMA result needs before an LDS write may
       * read it -- an inline-asm ds_write directly behind the last MFMA read the OLD accumulator) */
      {
        float *rdw = red + x * PC_RED_FLOATS + wv * (PC_SUB * 32);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          rdw[(4 * kq + r) * 32 + m] = acc0[r];
          rdw[(4 * kq + r) * 32 + 16 + m] = acc1[r];
        }
      }
      PC_STAMP(0, k, 1);
      __syncthreads(); /* barrier k + 1 */
      return true;
    };
    for (int k = 0;; k += 2) {
      if (!half(std::integral_constant<int, 0>{}, k)) break;
      if (!half(std::integral_constant<int, 1>{}, k + 1)) break;
    }
    return;
  }

  // ======================================================= poll and fetch
  // Waves 0-3: rows 4 lw .. 4 lw + 3 of each sub-chain's operand.  Almost no vector-ALU
  // work (one compare per poll), so the multiplying waves keep the ALU.
  const int lw = __builtin_amdgcn_readfirstlane(wave8); /* scalar: LDS addresses stay off the vector ALU */
  const int col = lane & 31, rh = lane >> 5;
  gu32 *aborted = (gu32 *)&sy->abort;
  bool dead = false;                  /* gave up: keep the barriers going, nothing else */

  // fetch rows 4 lw .. + 3 of sub-chain x, error plane `plane`, into its LDS image: 16 pieces
  // of 1 KB (lane l of piece q lands at chunk position 64 (q & 3) + l of row q >> 2 and
  // therefore brings chunk position ^ row)
  // Per-lane BYTE offsets within a plane are fixed for the whole launch (the same for both
  // sub-chains: b's rows are 16 rows further on); the plane / sub-chain base is wave-uniform.
  // A fetch is then 16 x (s_mov m0, global_load_lds saddr + voffset): no vector-ALU
  // instruction, which beside a multiplying wave would wait for a gap in its MFMAs.
  unsigned voff[4 * PPR]; /* this wave's four rows, PPR pieces of 64 chunks each */
#pragma unroll
  for (int i = 0; i < 4 * PPR; i++) {
    const int r = 4 * lw + i / PPR;
    const int c = (64 * (i % PPR) + lane) ^ r;
    voff[i] = (unsigned)(((size_t)r * s.I + 1 + 4 * c) * sizeof(float));
  }
  const float *sub_base = v.b.ehi + (size_t)(row0 + m0) * s.I; /* plane 0, sub-chain a, row 0 */
  /* (Round 3, measured and removed: every workgroup of a row tile starting its fetch at another row and
   * piece, so that the NT CUs do not all ask the L2 for the same line at the same moment: 102.2 against
   * 100.8 us per chain, and the sixteen instructions still took 1.1-1.25 us to issue beside the burst.) */
  auto fetch = [&](int x, int plane) {
    const char *base = reinterpret_cast<const char *>(sub_base + (size_t)plane * plane_stride +
                                                      (size_t)x * PC_SUB * s.I);
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr(abuf + x * BUF + 4 * lw * K));
#pragma unroll
    for (int i = 0; i < 4 * PPR; i++)
      lds_dma16_sc1(base, voff[i], dst + (uint32_t)(((i / PPR) * K + 256 * (i % PPR)) * sizeof(float)));
  };
  // wait until all NT column tiles have published step t of sub-chain x (rows of this wave).
  // Beside a wave that issues f32 MFMAs back to back this wave gets NO vector-ALU instruction through
  // (round 3, once the multiplying waves' own stalls were gone: the twelve VALU instructions of the
  // compiler's poll loop completed only when the 128-MFMA burst had ended, stamps: flags published
  // 0.4 us after the barrier, "seen" at 2.5 us).  So the poll is
  //   PC_POLL_SCALAR: scalar loads (s_load_dwordx8 glc: past the scalar cache) of the NT flag words and
  //     scalar compares -- no vector instruction at all; or
  //   otherwise: one L1-bypassing vector load per lane and ONE v_cmp, which takes the next of the short
  //     gaps that the multiplying waves leave in their MFMA stream for exactly this (PC_GAPS).
  // Flags compare as unsigned numbers: the launcher restarts the sequence long before it wraps.
  const unsigned poll_off = (unsigned)((lane % NT) * sizeof(unsigned));
  auto give_up = [&]() {
    if (lane == 0) {
      __hip_atomic_store(aborted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    dead = true;
  };
  auto wait_for = [&](int x, int t) {
    const unsigned want = epoch0 + (unsigned)t + 1u;
    const unsigned *fbase = &sy->flags[g][x][lw][0];
    __builtin_amdgcn_s_sleep(ONE ? PC_SLEEP0_ONE : K == 1024 ? PC_SLEEP0 : PC_SLEEP0_SMALL);
    for (unsigned spins = 0;; spins++) {
#if PC_POLL_SCALAR
      unsigned behind = 0u; /* any flag still below `want` */
#pragma unroll
      for (int i = 0; i < NT / 8; i++) {
        u32x8 f;
        asm volatile("s_load_dwordx8 %0, %1, %2 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(f) : "s"(fbase), "n"(32 * i) : "memory");
#pragma unroll
        for (int e = 0; e < 8; e++) behind |= (f[e] - want) >> 31;
      }
      if (!behind) return;
#else
      unsigned got;
      asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(poll_off), "s"(fbase) : "memory");
      if (__all(got >= want)) return;
#endif
      if ((spins & 1023u) == 1023u) { /* rarely: has somebody else given up; have we been here for tens of ms */
        const unsigned ab = __hip_atomic_load(aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__any(ab != 0u) || spins > (1u << 21)) { /* ~1 s: a co-tenant's long kernel may hold CUs for a while */
          give_up();
          return;
        }
      }
      __builtin_amdgcn_s_sleep(PC_SLEEP1);
    }
  };

  fetch(0, 0);
  if (halfsteps > 1 && !ONE) fetch(1, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads(); /* barrier 0 */
  for (int k = 0; k < halfsteps; k++) {
    PC_STAMP(1, k, 0);
    if (k >= 1 && k + 1 < halfsteps && !dead && !(ONE && (k & 1) == 0)) {
      /* half-step k + 1 continues the sub-chain of half-step k - 1, which the multiplying
       * waves of all 32 column tiles are finishing right now */
      wait_for((k - 1) & 1, (k - 1) >> 1);
      PC_STAMP(1, k, 2);
#if PC_FETCH_PRIO
      __builtin_amdgcn_s_setprio(3);
#endif
      if (!dead) fetch((k + 1) & 1, (k + 1) >> 1);
#if PC_FETCH_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      PC_STAMP(1, k, 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PC_STAMP(1, k, 4);
    __syncthreads(); /* barrier k + 1 */
  }
}

// ------------------------------------- assemble + hidden layer in one launch --
//
// The text step's forward pass shaped like a chain step (recur-nn.c:104-148): output tile =
// 32 streams x 32 hidden columns, K = the previous hidden values 1..hidden_size in 128-deep
// stages through the same LDS ring, loader and compute waves as k_chain_main.  What differs:
//   * A = rows of `hidden` (the previous step's activations, K-contiguous, swizzled as in the
//     chain); B = W_ih rows k, columns of the tile: K-major, so a stage is [128 k][32 columns]
//     in LDS and a lane fetches its four k with two ds_read2_b32;
//   * the bias row (input 0 is always 1) and the row of the stream's one-hot input are added in
//     the epilogue -- the K loop never touches the input columns;
//   * the workgroup does k_assemble's work for its block on the side: it writes its 32 x 32
//     block of the new history slot (the previous hidden values), the workgroups of column tile
//     0 also the bias, the input columns, the ring index and the text target; the row sum
//     that decides the emergency soft clip (maybe_scale_inputs, recur-nn.c:68-81) falls out
//     of the A fragments, and the clip is applied to the outputs and the stored row alike;
//   * the pre-activation sums go to slab plane 0 for k_text_top (which applies the activation
//     and writes `hidden`: the A operand must stay intact while other workgroups read it); the
//     h_size padding columns, which only matter when W's padding is non-zero, come as per-tile
//     partial sums in plane 1 ([tn][nrows][4]) that k_text_top adds up.
// Launcher preconditions: every stream at the same ring position, one-hot or text input, no
// presynaptic noise, no bottom layer, hidden_size a multiple of 32.
/* [k][col] and [k + 1][col] of a 32-column K-major stage; _hi: k + 2, k + 3.  The results are
 * used as they come (sub-registers of the asm output): a copy the compiler is free to place
 * before the s_waitcnt would read them too early */
__device__ __forceinline__ f32x2 lds_read2_b32(uint32_t addr) {
  f32x2 v;
  asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ f32x2 lds_read2_b32_hi(uint32_t addr) {
  f32x2 v;
  asm volatile("ds_read2_b32 %0, %1 offset0:64 offset1:96" : "=v"(v) : "v"(addr));
  return v;
}

template <int NS = 0>
__global__ __launch_bounds__(512) void k_fwd_fused(const View *__restrict__ vp, int new_idx, int row0,
                                                   int nrows, int tm, int tn, int nstages_arg,
                                                   int mode, int text_i, int global_first,
                                                   int n_set) {
  View v = *vp;
  const int nstages = NS > 0 ? NS : nstages_arg;
  __shared__ __attribute__((aligned(16))) float smem[C_STAGES * C_STAGE_FLOATS];
  __shared__ float rs_sh[4][CM];
  __shared__ float4 wt_sh[CN];
  const RamdShape &s = v.sh;
  const int L = blockIdx.x;
  const int xcd = L & 7, q = L >> 3;
  const int mt = q % tm, nt = (q / tm) * 8 + xcd;
  if (nt >= tn) return;
  /* column tiles start at column 0 (16-byte aligned rows of W and of the outputs); column 0's
   * sum is never used (the bias node), and the last four columns of h_size -- hidden value
   * hidden_size and the padding -- are not in any tile: they come as partial sums */
  const int m0 = mt * CM, n0 = nt * CN;
  const int wave8 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int lm = lane & 31, kh = lane >> 5;
  const float *hid0 = v.b.hidden + (size_t)row0 * s.H;

  // --- LDS-DMA sources: instruction i < 16 fills rows 2 (i & 15), +1 of A; i >= 16 fills
  // k rows 8 (i - 16) .. + 7 of B, eight lanes (32 columns) per k row
  const float *src[8];
  size_t stage_step[8];
  int kfirst[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    int i = wave * 8 + j;
    if (i < 16) {
      int row = 2 * (i & 15) + (lane >> 5);
      int c = (lane & 31) ^ (row & 15);
      int r = m0 + row;
      src[j] = hid0 + (size_t)(r < nrows ? r : nrows - 1) * s.H + 1 + 4 * c;
      stage_step[j] = CK;
      kfirst[j] = 1 + 4 * c;
    } else {
      int kr = 8 * (i - 16) + (lane >> 3);
      src[j] = v.b.ih_w + (size_t)(1 + kr) * s.H + n0 + 4 * (lane & 7);
      stage_step[j] = (size_t)CK * s.H;
      kfirst[j] = 1 + kr;
    }
  }
  auto issue = [&](int stage) {
    float *dst = smem + (stage % C_STAGES) * C_STAGE_FLOATS + wave * 8 * 256;
    const int k0 = stage * CK;
    if (k0 + CK <= s.hidden_size) {
#pragma unroll
      for (int j = 0; j < 8; j++)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(src[j] + stage * stage_step[j]),
                                         (lds_void_t *)(dst + j * 256), 16, 0, 0);
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { /* last, partial stage: what lies past the hidden values is zero */
      const float *g = (k0 + kfirst[j] <= s.hidden_size) ? src[j] + stage * stage_step[j] : v.b.zeros;
      __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(dst + j * 256), 16, 0, 0);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;

  const int etid = threadIdx.x & 255;
  const int erow = etid >> 3, ec4 = (etid & 7) * 4;
  const int er = m0 + erow < nrows ? m0 + erow : nrows - 1; /* row within the set */
  const int grow = row0 + er;
  const int tail = s.H - 4; /* the four columns outside the tiles: hidden_size = tail or tail + 1 .. */
  if (loader) {
#pragma unroll
    for (int p = 0; p < C_STAGES - 1; p++)
      if (p < nstages) issue(p);
  }
  // --- what the epilogue needs (compute waves): the one-hot index, this thread's four input
  // values x[n0 + ec4 ..] (column 0 is the bias node, 1), the bias row's and the input row's
  // weights under its four columns, and the tail columns of W in the rows of its four inputs
  int hot = -1, text_o = 0;
  float4 a4 = zero4(), wb = zero4(), ws = zero4(), wt_mine = zero4();
  if (!loader) {
    if (mode == RAMD_IN_TEXT) { /* charmodel-predict.c:273, 295-298 */
      int len = v.b.text_len;
      int spacing = (len - 1) / n_set;
      text_o = text_i + (global_first + er) * spacing;
      if (text_o >= len - 1) text_o -= len - 1;
      hot = v.b.text[text_o];
    } else {
      hot = v.b.hot[grow];
    }
    if (hot < 0 || hot >= s.input_size) hot = -1;
    a4 = ld4(hid0 + (size_t)er * s.H + n0 + ec4);
    if (n0 + ec4 == 0) a4.x = 1.0f;
    wb = ld4(v.b.ih_w + n0 + ec4);
    ws = ld4(v.b.ih_w + (size_t)(hot >= 0 ? s.hidden_size + 1 + hot : 0) * s.H + n0 + ec4);
    /* the tail columns of W in this tile's 32 input rows: one row per thread of the first half
     * wave, shared through LDS at the end */
    if (etid < CN) wt_mine = ld4(v.b.ih_w + (size_t)(n0 + etid) * s.H + tail);
  }
  const uint32_t lds0 = lds_byte_addr(smem);
  const uint32_t rowoff = (uint32_t)lm * (CK * 4u);
  float rsum = 0.0f;
  if (loader) {
#pragma unroll
    for (int st = 0; st < nstages; st++) {
      const int ahead = min(C_STAGES - 2, nstages - 1 - st);
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (st + C_STAGES - 1 < nstages) issue(st + C_STAGES - 1);
    }
  } else {
    struct BFrag {
      f32x2 lo[4], hi[4];
    };
    auto rd = [&](int st, f32x4 (&a)[4], BFrag &b) {
      const uint32_t abase = lds0 + (uint32_t)((st % C_STAGES) * C_STAGE_FLOATS) * 4u;
      const uint32_t bbase = abase + (uint32_t)(CM * CK) * 4u;
#pragma unroll
      for (int gi = 0; gi < 4; gi++) {
        int c = 2 * (4 * wave + gi) + kh; /* chunk = 4 consecutive k */
        a[gi] = lds_read_b128(abase + rowoff + (uint32_t)((c ^ (lm & 15)) * 16));
        const uint32_t baddr = bbase + (uint32_t)((4 * c) * CN + lm) * 4u;
        b.lo[gi] = lds_read2_b32(baddr);
        b.hi[gi] = lds_read2_b32_hi(baddr);
      }
    };
    auto step = [&](int st, f32x4 (&a)[4], BFrag &b, f32x4 (&an)[4], BFrag &bn) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (st + 1 < nstages) {
        __builtin_amdgcn_s_barrier();
        rd(st + 1, an, bn);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int gi = 0; gi < 4; gi++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].x, b.lo[gi].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].y, b.lo[gi].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].z, b.hi[gi].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].w, b.hi[gi].y, acc, 0, 0, 0);
        rsum += (a[gi].x + a[gi].y) + (a[gi].z + a[gi].w); /* the row's input sum, on the side */
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (NS > 0) {
      f32x4 a0[4], a1[4];
      BFrag b0, b1;
      __builtin_amdgcn_s_barrier();
      rd(0, a0, b0);
#pragma unroll
      for (int st = 0; st < NS; st += 2) {
        step(st, a0, b0, a1, b1);
        if (st + 1 < NS) step(st + 1, a1, b1, a0, b0);
      }
    } else { /* any number of stages: no read-ahead (see k_chain_main) */
      f32x4 a[4];
      BFrag b;
      for (int st = 0; st < nstages; st++) {
        __builtin_amdgcn_s_barrier();
        rd(st, a, b);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b.lo[0]), "+v"(b.lo[1]),
                       "+v"(b.lo[2]), "+v"(b.lo[3]), "+v"(b.hi[0]), "+v"(b.hi[1]), "+v"(b.hi[2]), "+v"(b.hi[3])
                     :
                     : "memory");
#pragma unroll
        for (int gi = 0; gi < 4; gi++) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].x, b.lo[gi].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].y, b.lo[gi].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].z, b.hi[gi].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi].w, b.hi[gi].y, acc, 0, 0, 0);
          rsum += (a[gi].x + a[gi].y) + (a[gi].z + a[gi].w);
        }
      }
    }
  }
  float *red = smem + (nstages % C_STAGES) * C_STAGE_FLOATS; /* [4][32][32] */
  if (!loader) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      int row = (g & 3) + 8 * (g >> 2) + 4 * kh;
      red[(wave * CM + row) * CN + lm] = acc[g];
    }
    rsum += __shfl_xor(rsum, 32, 64);
    if (kh == 0) rs_sh[wave][lm] = rsum;
    if (etid < CN) wt_sh[etid] = wt_mine;
  }
  __syncthreads();
  if (loader) return;
  const int row = erow, c4 = ec4;
  float4 wt[4];
#pragma unroll
  for (int i = 0; i < 4; i++) wt[i] = wt_sh[c4 + i];
  float4 e;
  {
    float4 p0 = ld4(red + (0 * CM + row) * CN + c4), p1 = ld4(red + (1 * CM + row) * CN + c4);
    float4 p2 = ld4(red + (2 * CM + row) * CN + c4), p3 = ld4(red + (3 * CM + row) * CN + c4);
    e.x = (p0.x + p1.x) + (p2.x + p3.x);
    e.y = (p0.y + p1.y) + (p2.y + p3.y);
    e.z = (p0.z + p1.z) + (p2.z + p3.z);
    e.w = (p0.w + p1.w) + (p2.w + p3.w);
  }
  // the row's input sum: bias + previous hidden values + the one-hot input
  float sum = ((rs_sh[0][row] + rs_sh[1][row]) + (rs_sh[2][row] + rs_sh[3][row])) + 1.0f +
              (hot >= 0 ? 1.0f : 0.0f);
  const float softclip = s.I * INPUT_MEAN_SOFT_TOP_F;
  const float scale = (sum > softclip) ? soft_clip_dev(sum, softclip) : 1.0f;
  const bool live = m0 + row < nrows;
  // this workgroup's share of the four tail columns: its 32 inputs x W[k][tail .. tail + 3]
  float4 pp;
  pp.x = ((a4.x * wt[0].x + a4.y * wt[1].x) + (a4.z * wt[2].x + a4.w * wt[3].x));
  pp.y = ((a4.x * wt[0].y + a4.y * wt[1].y) + (a4.z * wt[2].y + a4.w * wt[3].y));
  pp.z = ((a4.x * wt[0].z + a4.y * wt[1].z) + (a4.z * wt[2].z + a4.w * wt[3].z));
  pp.w = ((a4.x * wt[0].w + a4.y * wt[1].w) + (a4.z * wt[2].w + a4.w * wt[3].w));
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    pp.x += __shfl_xor(pp.x, off, 64);
    pp.y += __shfl_xor(pp.y, off, 64);
    pp.z += __shfl_xor(pp.z, off, 64);
    pp.w += __shfl_xor(pp.w, off, 64);
  }
  if (!live) return;
  float *out = v.b.slab + (size_t)er * s.H;
  float *slot = v.b.arena + ((size_t)new_idx * s.Scap + grow) * s.I;
  {
    float4 o;
    o.x = ((e.x + wb.x) + (hot >= 0 ? ws.x : 0.0f)) * scale;
    o.y = ((e.y + wb.y) + (hot >= 0 ? ws.y : 0.0f)) * scale;
    o.z = ((e.z + wb.z) + (hot >= 0 ? ws.z : 0.0f)) * scale;
    o.w = ((e.w + wb.w) + (hot >= 0 ? ws.w : 0.0f)) * scale;
    *reinterpret_cast<float4 *>(out + n0 + c4) = o;
    *reinterpret_cast<float4 *>(slot + n0 + c4) = make_float4(a4.x * scale, a4.y * scale, a4.z * scale, a4.w * scale);
  }
  if ((etid & 7) == 0) {
    if (nt == 0) {
      /* once per row: the inputs the tiles do not cover -- hidden values tail .. hidden_size and
       * the one-hot input -- times their W rows */
      for (int k = tail; k <= s.hidden_size; k++) {
        float x = hid0[(size_t)er * s.H + k];
        float4 w = ld4(v.b.ih_w + (size_t)k * s.H + tail);
        pp.x += x * w.x; pp.y += x * w.y; pp.z += x * w.z; pp.w += x * w.w;
        slot[k] = x * scale;
      }
      if (hot >= 0) {
        float4 w = ld4(v.b.ih_w + (size_t)(s.hidden_size + 1 + hot) * s.H + tail);
        pp.x += w.x; pp.y += w.y; pp.z += w.z; pp.w += w.w;
      }
    }
    float *pd = v.b.slab + (size_t)nrows * s.H + ((size_t)nt * nrows + er) * 4;
    *reinterpret_cast<float4 *>(pd) = make_float4(pp.x * scale, pp.y * scale, pp.z * scale, pp.w * scale);
  }
  if (nt == 0) { /* the rest of k_assemble's row: input columns, ring index, target */
    const int sub = etid & 7;
    if (sub == 0) {
      v.b.idx[grow] = new_idx;
      if (mode == RAMD_IN_TEXT) v.b.target[grow] = v.b.text[text_o + 1];
    }
    for (int k = sub; k < s.input_size; k += 8) slot[s.hidden_size + 1 + k] = (k == hot) ? scale : 0.0f;
  }
}

// Chain "extras" without a GEMM: for every (step, stream) the error of the bias
// row (column 0) and of the real-input rows, i.e. e = W_ih[y][:] . E_h[t][s][:] for
// the rows y whose input value is non-zero -- the reference's zero-row skip
// (recur-nn.c:338-341) is what makes this cheap: a one-hot text stream has two
// such rows per step, a dense audio frame a few dozen.  One wave per (step,
// stream): the error row sits in registers (5 float4 per lane at h_size 1028),
// the wave walks the non-zero columns (ballot), each dot product is reduced with
// xor shuffles in a fixed order.  It also closes the step's sum of squares:
// the column-tile partials of k_chain_main in index order, then the extras.
// what one (step, stream) item reads before anything depends on anything: its error row,
// this lane's input value of the first 64 extra columns, this lane's column-tile partial
template <int MAXQ> struct ExtrasIn {
  float4 ev[MAXQ];
  float xi, pv;
};
template <int MAXQ> /* float4 per lane: 5 covers h_size <= 1280, 8 h_size <= 2048, 9 h_size <= 2304 */
__device__ __forceinline__ void extras_load(const View &v, int t, int r, int nx, int tn, int lane,
                                            ExtrasIn<MAXQ> &in) {
  const RamdShape &s = v.sh;
  const float *erow = v.b.ehi + (t * s.Scap + r) * s.I;
  const float *x = input_row_auto(v, r, t);
  const int nq = (s.H / 4 + 63) / 64;
  /* Every load is UNCONDITIONAL from a clamped (always valid) address, the select comes after: as
   * `cond ? load : 0` hipcc branched around each load and waited for it at the join, so that the five loads of a
   * row (and the ten of the two weight rows below) went out one L2 round trip after another -- most of this
   * kernel's time until round 3. */
  const int last4 = s.H / 4 - 1;
#pragma unroll
  for (int i = 0; i < MAXQ; i++) {
    const int k4 = lane + 64 * i;
    const float4 e = ld4(erow + 4 * min(k4, last4));
    in.ev[i] = (i < nq && k4 <= last4) ? e : zero4();
  }
  {
    const float xv = x[(lane == 0 || lane >= nx) ? 0 : s.hidden_size + lane];
    in.xi = (lane < nx) ? xv : 0.0f;
    const float pv = v.b.esum_part[((size_t)t * (tn + 1) + (lane < tn ? lane : 0)) * s.Scap + r];
    in.pv = (lane < tn) ? pv : 0.0f;
  }
}
/* sum of squares of the error row an item holds (column 0 and the padding are zero): the same in every lane */
template <int MAXQ> __device__ __forceinline__ float row_sumsq(const ExtrasIn<MAXQ> &in) {
  float a = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXQ; i++)
    a += (in.ev[i].x * in.ev[i].x + in.ev[i].y * in.ev[i].y) + (in.ev[i].z * in.ev[i].z + in.ev[i].w * in.ev[i].w);
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
  return a;
}
/* the same for row r of error plane `plane`, fetched here */
__device__ __forceinline__ float row_sumsq_load(const View &v, int plane, int r, int lane) {
  const RamdShape &s = v.sh;
  const float *erow = v.b.ehi + ((size_t)plane * s.Scap + r) * s.I;
  float a = 0.0f;
  for (int k4 = lane; 4 * k4 < s.H; k4 += 64) {
    const float4 e = ld4(erow + 4 * k4);
    a += (e.x * e.x + e.y * e.y) + (e.z * e.z + e.w * e.w);
  }
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
  return a;
}
template <int MAXQ>
__device__ __forceinline__ float extras_compute(const View &v, int t, int r, int nx, int nxp, int tn,
                                                int lane, const ExtrasIn<MAXQ> &in) {
  const RamdShape &s = v.sh;
  const float *x = input_row_auto(v, r, t);
  float *dst = v.b.ex + ((size_t)(t + 1) * s.Scap + r) * nxp;
  const int nq = (s.H / 4 + 63) / 64;
  float sq = 0.0f;
  for (int c0 = 0; c0 < nx; c0 += 64) {
    int c = c0 + lane;
    int n = (c == 0) ? 0 : s.hidden_size + c;
    float xi = (c0 == 0) ? in.xi : ((c < nx) ? x[n] : 0.0f);
    bool on = xi != 0.0f && (s.activation != 5 || xi < 20.0f);
    if (c < nx) dst[c] = 0.0f;
    unsigned long long live = __ballot(on);
    /* two live columns per round, so that both weight rows are in flight together (a
     * one-hot text stream has exactly two: the bias row and the symbol's row) */
    while (live) {
      int la = __ffsll((long long)live) - 1;
      live &= live - 1;
      int lb = live ? __ffsll((long long)live) - 1 : -1;
      if (lb >= 0) live &= live - 1;
      int ca = c0 + la, cb = c0 + (lb >= 0 ? lb : la);
      int na = (ca == 0) ? 0 : s.hidden_size + ca;
      int nb = (cb == 0) ? 0 : s.hidden_size + cb;
      float xa = __shfl(xi, la, 64), xb = __shfl(xi, lb >= 0 ? lb : la, 64);
      const float *wa = v.b.ih_w + na * s.H;
      const float *wb = v.b.ih_w + nb * s.H;
      float4 wva[MAXQ], wvb[MAXQ];
      const int last4 = s.H / 4 - 1;
#pragma unroll
      for (int i = 0; i < MAXQ; i++) { /* unconditional, clamped: all ten in flight together (see extras_load) */
        const int k4 = lane + 64 * i, k4c = min(k4, last4);
        const float4 ta = ld4(wa + 4 * k4c), tb = ld4(wb + 4 * k4c);
        const bool inb = i < nq && k4 <= last4;
        wva[i] = inb ? ta : zero4();
        wvb[i] = inb ? tb : zero4();
      }
      float acca = 0.0f, accb = 0.0f;
#pragma unroll
      for (int i = 0; i < MAXQ; i++) {
        acca += in.ev[i].x * wva[i].x + in.ev[i].y * wva[i].y + in.ev[i].z * wva[i].z + in.ev[i].w * wva[i].w;
        accb += in.ev[i].x * wvb[i].x + in.ev[i].y * wvb[i].y + in.ev[i].z * wvb[i].z + in.ev[i].w * wvb[i].w;
      }
      for (int off = 32; off > 0; off >>= 1) {
        acca += __shfl_xor(acca, off, 64);
        accb += __shfl_xor(accb, off, 64);
      }
      if (s.activation == 2) {
        acca /= 2 * (xa + 1.0f);
        accb /= 2 * (xb + 1.0f);
      }
      if (lane == 0) {
        dst[ca] = acca;
        if (lb >= 0) dst[cb] = accb;
      }
      sq += acca * acca; /* identical in every lane */
      if (lb >= 0) sq += accb * accb;
    }
  }
  // the step's total: the column-tile partials of k_chain_main in index order (each lane
  // fetches one, every lane adds them in order), then the extras.  tn == 0 (the one-launch
  // chain leaves no partials): the caller adds the hidden columns' part (row_sumsq of the
  // step's OUTPUT row, error plane t + 1) itself.
  float sum = 0.0f;
  for (int p0 = 0; p0 < tn; p0 += 64) {
    int p = p0 + lane;
    float pv = (p0 == 0) ? in.pv
                         : ((p < tn) ? v.b.esum_part[((size_t)t * (tn + 1) + p) * s.Scap + r] : 0.0f);
    int cnt = min(64, tn - p0);
    for (int i = 0; i < cnt; i++) sum += __shfl(pv, i, 64);
  }
  return sum + sq; /* the same in every lane */
}

template <int MAXQ>
__global__ __launch_bounds__(256) void k_extras_gather(View v, int row0, int nrows, int nx, int nxp,
                                                       int tn) {
  const RamdShape &s = v.sh;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (m >= s.D * nrows) return;
  const int t = m / nrows, r = row0 + (m - t * nrows);
  ExtrasIn<MAXQ> in;
  extras_load<MAXQ>(v, t, r, nx, tn, lane, in);
  float es = extras_compute<MAXQ>(v, t, r, nx, nxp, tn, lane, in);
  if (tn == 0) es += row_sumsq_load(v, t + 1, r, lane);
  if (lane == 0) v.b.esum[(size_t)t * s.Scap + r] = es;
}

// Finalize of the extras GEMM: applies the row rule to column 0 and the input
// columns, keeps the raw values in ex[t+1][s][c] (for the h_error / i_error
// images) and adds their squares as the last partial sum.  One wave per (t, s).
__global__ __launch_bounds__(64) void k_extras_finalize(View v, int row0, int nrows, int nx, int nxp,
                                                        int ks, int tn) {
  const RamdShape &s = v.sh;
  int m = blockIdx.x;
  int t = m / nrows, j = m - t * nrows, r = row0 + j;
  int M = s.D * nrows;
  const float *x = input_row<false>(v, r, t);
  float *dst = v.b.ex + ((size_t)(t + 1) * s.Scap + r) * nxp;
  float sq = 0.0f;
  for (int c = threadIdx.x; c < nx; c += 64) {
    float e = 0.0f;
    for (int z = 0; z < ks; z++) e += v.b.slab[((size_t)z * M + m) * nxp + c];
    int n = c == 0 ? 0 : s.hidden_size + c;
    float xi = x[n];
    bool on = xi != 0.0f && (s.activation != 5 || xi < 20.0f);
    e = on ? e : 0.0f;
    if (on && s.activation == 2) e /= 2 * (xi + 1.0f);
    dst[c] = e;
    sq += e * e;
  }
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
  /* the step's total: the column-tile partials of k_chain_main in index order, then
   * the extras (a fixed order, so the break decisions are reproducible); tn == 0: the one-launch
   * chain left no partials, the hidden columns' part is summed from the row */
  const float hsq = tn == 0 ? row_sumsq_load(v, t + 1, r, threadIdx.x) : 0.0f;
  if (threadIdx.x == 0) {
    float sum = hsq;
    for (int p = 0; p < tn; p++) sum += v.b.esum_part[((size_t)t * (tn + 1) + p) * s.Scap + r];
    v.b.esum[(size_t)t * s.Scap + r] = sum + sq;
  }
}

// ------------------------------------------------ weight-delta GEMM by LDS-DMA --
//
// ih_delta[m][n] = sum over (step t, stream r) of X_t[r][m] * coef[t][r] * E_t[r][n]
// (recur-nn.c:343-358 for every executed step of every stream, with the stream's
// ih_scale folded in).  Both operands are K-major: a K tile is 32 streams of one step, a
// row of it 128 consecutive floats of a history slot (A) or of an error plane (B).
//
// Workgroup = 8 waves on one CU: waves 0-3 own a 64 x 64 quadrant of the 128 x 128 tile
// each (2 x 2 accumulators, 64 MFMAs per K tile and wave); waves 4-7 only move data:
// each K tile is 32 + 1 LDS-DMA wave instructions (16 KB of A, 16 KB of B, the tile's 32
// coefficients) into a four-deep ring, three tiles in flight.  One raw s_barrier per K
// tile; nothing else couples the two groups.  A K tile costs a compute wave 64 MFMAs
// (4096 cycles), so the barrier and the LDS read latency between tiles are a few per cent.
// The per-row coefficient is applied to the B fragments after the LDS read (select on
// zero: rows of steps a stream did not execute may hold anything).
//
// Preconditions (checked by the launcher, which otherwise uses k_gemm2): every stream at
// the same ring position, nrows % 32 == 0, hidden_size % 128 == 0, not RECLIP20.  Only
// whole 128-row tiles are computed here; the remaining rows (bias row 0 is in tile 0; the
// input rows above the last whole tile) go through the generic k_gemm with a row offset.
constexpr int DD_STAGES = 4;
constexpr int DD_STAGE_FLOATS = 2 * BK * 128 + 64; /* A, B, 32 coefficients (+ pad) */

/* The rows above the last whole 128-row tile (the input rows of a text net: 44 at the north
 * star) ride along: the tm workgroups that share a column tile and a K slice (mt = 0..tm-1;
 * eight at hidden 1024) each take every tm-th K tile of the slice, fetch the 32 x 64 piece of the history rows'
 * tail for it (columns rows_core .. rows_core + 63: past i_size they run into the next row,
 * which only feeds output rows nobody stores), and multiply it with the error fragments they
 * have in registers anyway: 8 more MFMAs per 16 in one K tile of tm, no second pass over
 * the error planes.  Their partial sums are plane z * tm + mt of `planes`. */
struct DeltaRest {
  float *planes;  /* [ks * tm][rows][ldc] */
  size_t stride;  /* floats between planes */
  int rows;       /* i_size - rows_core, <= RR */
  int col;        /* rows_core */
};
constexpr int DD_REST_FLOATS = BK * 128; /* the rest ring: two tiles of 32 k x 64 rows or one of 32 k x 128 */

/* RR: rows of the rest tile: 0 (none), 64 or 128 (a wave then has 32 or 64 rest rows x its 64 columns) */
template <int RR>
__global__ __launch_bounds__(512) void k_delta_dma(View v, int row0, int nrows, GemmOut o, DeltaRest dr) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const RamdShape &s = v.sh;
  const int L = blockIdx.x;
  // K slice z lives on 8 / ks XCDs (ks divides 8), so each XCD's L2 sees one slice of X
  // and E only; within a slice consecutive tiles alternate between its XCDs
  const int xcd = L & 7, q = L >> 3;
  const int per = 8 / o.ks;
  const int z = xcd / per, tile = q * per + (xcd % per);
  if (tile >= o.tm * o.tn) return;
  const int mt = tile / o.tn, nt = tile % o.tn;
  const int m0 = o.row0m + mt * 128, n0 = o.col0 + nt * 128; /* (row0m: a launch over the upper row tiles only) */
  const int kt0 = (int)(((long)o.nkt * z) / o.ks), kt1 = (int)(((long)o.nkt * (z + 1)) / o.ks);
  const int nst = kt1 - kt0;
  const int wave8 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rtiles = nrows / BK;
  /* stage st carries rest work when st % tm == mt; its tile sits in slot (st / tm) % 2 behind the
   * ring (tm >= 4, so a slot's previous tile was consumed long before the next one is fetched) */
  constexpr bool REST = RR > 0;
  constexpr int RI = RR / 64; /* 32-row groups of rest rows per wave */
  auto is_rest = [&](int st) { return REST && st < nst && st % o.tm == mt; };
  /* byte-free float offset of a stage's rest tile in the rest ring: two slots of 64 rows, one of 128 */
  auto rest_slot = [&](int st) { return RR == 64 ? ((st / o.tm) & 1) * (BK * 64) : 0; };
  float *const rest_ring = dsm + DD_STAGES * DD_STAGE_FLOATS;

  if (wave8 >= 4) {
    // ---------------------------------------------------------------- loaders
    const int w = wave8 - 4;
    // instruction i (0..31): rows 2 (i & 15), +1 of A (i < 16) or B; wave w issues i = 8 w + j
    /* Every DMA is (wave-uniform base, per-lane byte offset that never changes, wave-uniform
     * LDS address): issued as saddr + voffset by inline asm, a fetch costs the loader wave no
     * vector-ALU instruction -- beside a wave that issues f32 MFMAs back to back (they run on
     * the SIMD's vector ALU) such instructions wait for a gap in the MFMA stream. */
    const int ws = __builtin_amdgcn_readfirstlane(w);
    const unsigned voff = (unsigned)(((size_t)(lane >> 5) * s.I + (lane & 31) * 4) * sizeof(float));
    const unsigned voff_c = (unsigned)((lane & 31) * sizeof(float));
    const unsigned voff_r = RR == 128 ? voff : (unsigned)(((size_t)(lane >> 4) * s.I + (lane & 15) * 4) * sizeof(float));
    const uint32_t dsm_lds = __builtin_amdgcn_readfirstlane(lds_byte_addr(dsm));
    auto issue = [&](int st) {
      const int kt = kt0 + st;
      const int t = kt / rtiles, sb = (kt - t * rtiles) * BK;
      int slot = v.b.uniform_idx - t;
      if (slot < 0) slot += s.D;
      /* waves 0, 1 fetch the history rows (A), waves 2, 3 the error rows (B): rows 2 (i & 15), + 1 */
      const float *base = ws < 2 ? v.b.arena + ((size_t)slot * s.Scap + row0 + sb) * s.I + m0
                                 : v.b.ehi + ((size_t)t * s.Scap + row0 + sb) * s.I + n0;
      const uint32_t dst = dsm_lds + (uint32_t)(((st % DD_STAGES) * DD_STAGE_FLOATS + (ws < 2 ? 0 : BK * 128)) * sizeof(float));
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int i15 = (ws & 1) * 8 + j; /* = (w * 8 + j) & 15 */
        lds_dma16(base + (size_t)(2 * i15) * s.I, voff, dst + (uint32_t)(i15 * 256 * sizeof(float)));
      }
      if (ws == 0) { /* the 32 coefficients of the tile: lanes 32-63 repeat them */
        lds_dma4(v.b.coef + (size_t)t * s.Scap + row0 + sb, voff_c,
                 dsm_lds + (uint32_t)(((st % DD_STAGES) * DD_STAGE_FLOATS + 2 * BK * 128) * sizeof(float)));
      }
      if (REST && ws == 3 && is_rest(st)) { /* the tail of the 32 history rows: four rows per instruction */
        const float *rb = v.b.arena + ((size_t)slot * s.Scap + row0 + sb) * s.I + dr.col;
        const uint32_t rdst = dsm_lds + (uint32_t)((DD_STAGES * DD_STAGE_FLOATS + rest_slot(st)) * sizeof(float));
        /* an instruction moves 1 KB: four rows of 64 floats or two of 128 */
#pragma unroll
        for (int j = 0; j < 8 * (RI ? RI : 1); j++)
          lds_dma16(rb + (size_t)((RR == 128 ? 2 : 4) * j) * s.I, voff_r, rdst + (uint32_t)(j * 256 * sizeof(float)));
      }
    };
#pragma unroll
    for (int p = 0; p < DD_STAGES - 1; p++)
      if (p < nst) issue(p);
    for (int st = 0; st < nst; st++) {
      // stages st+1 .. st+DD_STAGES-2 may stay in flight (8 or 9 DMAs per stage)
      const int ahead = min(DD_STAGES - 2, nst - 1 - st);
      if (w == 0) {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (REST && w == 3) {
        /* 8 (RR 64) or 16 (RR 128) more in flight for a rest stage among the ones ahead (at most one:
         * they are tm >= 4 apart) */
        const bool more = (ahead >= 1 && is_rest(st + 1)) || (ahead >= 2 && is_rest(st + 2));
        if (ahead >= 2) {
          if (more && RR == 128) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
          else if (more) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else if (ahead == 1) {
          if (more && RR == 128) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
          else if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      } else {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier(); /* stage st has landed; stage st-1's buffer is free */
      if (st + DD_STAGES - 1 < nst) issue(st + DD_STAGES - 1);
    }
    return;
  }

  // ------------------------------------------------------------------ compute
  const int wm = wave8 >> 1, wn = wave8 & 1;
  const int lm = lane & 31, kh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;
  // The fragments of K group u + 1 (8 k: four MFMA steps of two k each) are read from LDS
  // before the 16 MFMAs of group u are issued and used after them, so the LDS latency
  // and, at a stage boundary, the barrier sit in the shadow of the matrix pipe.
  // Round 3: the reads are inline asm with COUNTED waits.  As plain loads hipcc put an
  // `s_waitcnt lgkmcnt(0)` behind the reads of every second group, in front of the CURRENT group's
  // multiplies -- the read-ahead it was meant to be waited for its own reads twice per K tile -- and
  // spent five vector-ALU instructions per group on LDS addresses.  Now: `ds_read2st64_b32` (two k
  // rows, 128 floats apart, per instruction; immediate offsets in units of 64 floats reach every k of
  // a stage), four address registers per STAGE, and each group waits only for what was issued before
  // the reads of the group after it (lgkmcnt(n), n = that group's instruction count <= 13).
  struct Frag {
    f32x2 a[2][2], e[2][2]; /* [i or jn][k pair]: k = 8 g + 4 kh + 2 pair, + 1 */
    f32x4 cf;
    f32x2 ar[RI ? RI : 1][2]; /* rest rows (REST stages only) */
  };
  f32x16 racc[RI ? RI : 1][2];
#pragma unroll
  for (int i = 0; i < (RI ? RI : 1); i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 16; g++) racc[i][j][g] = 0.0f;
  /* per-lane byte addresses within a stage: A rows i = 0, 1; E columns jn = 0, 1; the coefficients; the rest rows */
  const uint32_t dsm0 = lds_byte_addr(dsm);
  const uint32_t la_lane = dsm0 + 4u * (uint32_t)(4 * kh * 128 + wm * 64 + lm);
  const uint32_t le_lane = dsm0 + 4u * (uint32_t)(BK * 128 + 4 * kh * 128 + wn * 64 + lm);
  const uint32_t lc_lane = dsm0 + 4u * (uint32_t)(2 * BK * 128 + 4 * kh);
  const uint32_t lr_lane = lds_byte_addr(rest_ring) + 4u * (uint32_t)(4 * kh * RR + wm * (RR / 2) + lm);
  uint32_t ad_a0 = 0, ad_a1 = 0, ad_e0 = 0, ad_e1 = 0, ad_c = 0, ad_r = 0; /* of the stage being READ */
  auto stage_addr = [&](int st, bool rest) {
    const uint32_t sb = (uint32_t)((st % DD_STAGES) * DD_STAGE_FLOATS * (int)sizeof(float));
    ad_a0 = la_lane + sb;
    ad_a1 = ad_a0 + 128u;
    ad_e0 = le_lane + sb;
    ad_e1 = ad_e0 + 128u;
    ad_c = lc_lane + sb;
    if (REST && rest) ad_r = lr_lane + (uint32_t)(rest_slot(st) * (int)sizeof(float));
  };
  /* group G of the stage whose addresses are set: 9 LDS instructions (+ 2 RI for a rest stage) */
  auto rd = [&](auto GC, Frag &f, bool rest) {
    constexpr int G = decltype(GC)::value;
    constexpr int RU = RR ? RR / 64 : 1; /* k rows of the rest tile are RR floats apart: RR / 64 offset units */
    if (REST && rest) { /* wave-uniform */
#pragma unroll
      for (int i = 0; i < RI; i++) {
        f.ar[i][0] = lds_read2st64<(8 * G + 0) * RU, (8 * G + 1) * RU>(ad_r + 128u * (uint32_t)i);
        f.ar[i][1] = lds_read2st64<(8 * G + 2) * RU, (8 * G + 3) * RU>(ad_r + 128u * (uint32_t)i);
      }
    }
    f.cf = lds_read_b128_off<32 * G>(ad_c);
    f.a[0][0] = lds_read2st64<2 * (8 * G + 0), 2 * (8 * G + 1)>(ad_a0);
    f.e[0][0] = lds_read2st64<2 * (8 * G + 0), 2 * (8 * G + 1)>(ad_e0);
    f.a[1][0] = lds_read2st64<2 * (8 * G + 0), 2 * (8 * G + 1)>(ad_a1);
    f.e[1][0] = lds_read2st64<2 * (8 * G + 0), 2 * (8 * G + 1)>(ad_e1);
    f.a[0][1] = lds_read2st64<2 * (8 * G + 2), 2 * (8 * G + 3)>(ad_a0);
    f.e[0][1] = lds_read2st64<2 * (8 * G + 2), 2 * (8 * G + 3)>(ad_e0);
    f.a[1][1] = lds_read2st64<2 * (8 * G + 2), 2 * (8 * G + 3)>(ad_a1);
    f.e[1][1] = lds_read2st64<2 * (8 * G + 2), 2 * (8 * G + 3)>(ad_e1);
  };
  /* Every group waits for "at most nine LDS instructions behind my reads": the next group's nine (a
   * rest stage issues its 2 RI extra reads FIRST, so there the wait also covers those: a few dozen
   * cycles once per tm stages).  A count chosen at run time would put the wait into branches, and
   * hipcc then copies the whole fragment (ten v_mov_b64) in front of the multiplies in each of them. */
  auto mm = [&](Frag &f, bool rest) {
    frag_wait<9>(f.cf, f.a, f.e);
    float b[2][4];
    /* v_mul_legacy_f32: 0 * x is 0 for ANY x (a step past the break may hold inf), otherwise the
     * IEEE product: select and multiply in one instruction of the MFMAs' own ALU.  All eight in
     * ONE statement with early-clobber outputs: eight distinct registers that nothing rewrites
     * while the sixteen MFMAs that read them are being issued (as separate statements hipcc
     * recycled two registers between the MFMAs, and results went wrong). */
    asm volatile("v_mul_legacy_f32 %0, %8, %12\n\tv_mul_legacy_f32 %1, %9, %13\n\t"
                 "v_mul_legacy_f32 %2, %10, %14\n\tv_mul_legacy_f32 %3, %11, %15\n\t"
                 "v_mul_legacy_f32 %4, %8, %16\n\tv_mul_legacy_f32 %5, %9, %17\n\t"
                 "v_mul_legacy_f32 %6, %10, %18\n\tv_mul_legacy_f32 %7, %11, %19\n\ts_nop 1"
                 : "=&v"(b[0][0]), "=&v"(b[0][1]), "=&v"(b[0][2]), "=&v"(b[0][3]), "=&v"(b[1][0]),
                   "=&v"(b[1][1]), "=&v"(b[1][2]), "=&v"(b[1][3])
                 : "v"(f.cf[0]), "v"(f.cf[1]), "v"(f.cf[2]), "v"(f.cf[3]), "v"(f.e[0][0][0]), "v"(f.e[0][0][1]),
                   "v"(f.e[0][1][0]), "v"(f.e[0][1][1]), "v"(f.e[1][0][0]), "v"(f.e[1][0][1]), "v"(f.e[1][1][0]),
                   "v"(f.e[1][1][1]));
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jn = 0; jn < 2; jn++)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][jj >> 1][jj & 1], b[jn][jj], acc[i][jn], 0, 0, 0);
    if (REST && rest) { /* wave-uniform: this wave's 32 or 64 rest rows x its 64 columns (their reads are older
                         * than the nine waited for above) */
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int i = 0; i < RI; i++)
#pragma unroll
          for (int jn = 0; jn < 2; jn++)
            racc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.ar[i][jj >> 1][jj & 1], b[jn][jj], racc[i][jn], 0, 0, 0);
    }
  };
  /* groups 0-2 of a stage: read the next group of the same stage, multiply this one */
  auto step = [&](auto GC, Frag &cur, Frag &nxt, bool rest) {
    constexpr int g = decltype(GC)::value;
    rd(std::integral_constant<int, g + 1>{}, nxt, rest);
    __builtin_amdgcn_sched_barrier(0);
    mm(cur, rest);
    __builtin_amdgcn_sched_barrier(0);
  };
  /* group 3: across the stage boundary.  After the last stage group 0 of the same stage is read once more
   * (into the idle fragment), so that "nine instructions behind" holds for every group of the launch. */
  auto step3 = [&](int st, Frag &cur, Frag &nxt, bool rest, bool rest_next) {
    if (st + 1 < nst) {
#ifdef PC_STAMPS
      if (blockIdx.x == 0 && threadIdx.x == 0 && st < 63) g_pc_stamps[0][st + 1][4] = __builtin_amdgcn_s_memrealtime();
#endif
      __builtin_amdgcn_s_barrier(); /* stage st + 1 has landed; stage st - 1's buffer is free */
      asm volatile("" ::: "memory");
#ifdef PC_STAMPS
      if (blockIdx.x == 0 && threadIdx.x == 0 && st < 63) g_pc_stamps[0][st + 1][5] = __builtin_amdgcn_s_memrealtime();
#endif
      stage_addr(st + 1, rest_next);
    } else {
      rest_next = false;
    }
    rd(std::integral_constant<int, 0>{}, nxt, rest_next);
    __builtin_amdgcn_sched_barrier(0);
    mm(cur, rest);
    __builtin_amdgcn_sched_barrier(0);
  };
  Frag f0, f1;
#ifdef PC_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) g_pc_stamps[0][0][4] = __builtin_amdgcn_s_memrealtime();
#endif
  if (nst > 0) {
    __builtin_amdgcn_s_barrier(); /* stage 0 has landed */
    asm volatile("" ::: "memory");
    stage_addr(0, is_rest(0));
    rd(std::integral_constant<int, 0>{}, f0, is_rest(0));
  }
  for (int st = 0; st < nst; st++) {
    const bool r = is_rest(st), rn = is_rest(st + 1);
    step(std::integral_constant<int, 0>{}, f0, f1, r);
    step(std::integral_constant<int, 1>{}, f1, f0, r);
    step(std::integral_constant<int, 2>{}, f0, f1, r);
    step3(st, f1, f0, r, rn);
  }
#ifdef PC_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) g_pc_stamps[0][0][6] = __builtin_amdgcn_s_memrealtime();
#endif
  float *c = o.slab + (size_t)z * o.zs;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jn = 0; jn < 2; jn++) {
      const int col = n0 + wn * 64 + jn * 32 + lm;
#pragma unroll
      for (int g = 0; g < 16; g++) {
        int row = m0 + wm * 64 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * kh;
        c[(size_t)row * o.ldc + col] = acc[i][jn][g];
      }
    }
  if (REST) {
    float *rp = dr.planes + (size_t)(z * o.tm + mt) * dr.stride;
#pragma unroll
    for (int i = 0; i < RI; i++)
#pragma unroll
      for (int jn = 0; jn < 2; jn++) {
        const int col = n0 + wn * 64 + jn * 32 + lm;
#pragma unroll
        for (int g = 0; g < 16; g++) {
          int row = wm * (RR / 2) + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * kh;
          if (row < dr.rows) rp[(size_t)row * o.ldc + col] = racc[i][jn][g];
        }
      }
  }
}

// ----------------------------------------------------- K9: BPTT control --

// The data-dependent part of bptt_and_accumulate_error (recur-nn.c:317-330,
// 383-413), one wave per stream: lane k holds the error sum of step k, a ballot finds
// the step at which the reference's loop would have stopped, lane 0 derives ih_scale
// and the adaptive min_error_factor, and the lanes publish coef[t][r] = ih_scale while
// the step counts, 0 afterwards.
/* es: the stream's error sums by step, es[k * stride] */
__device__ __forceinline__ void bptt_control_wave(const View &v, int r, int j, int lane,
                                                  const unsigned char *active, unsigned flags,
                                                  const float *es_src, size_t es_stride) {
  const RamdShape &s = v.sh;
  const int D = s.D;
  if (active && !active[j]) {
    for (int k = lane; k < D; k += 64) v.b.coef[(size_t)k * s.Scap + r] = 0.0f;
    if (lane == 0) v.b.n_exec[r] = 0; /* no step ran: k_err_writeback leaves its images alone */
    return;
  }
  float top = v.b.top_scaled[r];
  float max_error_sum = MAX_ERROR_GAIN_F * top + 1;
  float error_sum_ceiling = ERROR_GAIN_CEILING_F * top;
  float min_error_gain = MIN_ERROR_GAIN_F * top;
  float mef = v.b.mef[r];
  /* MIN(a, b) of the reference is (a < b) ? a : b: keep NaN behaviour aligned */
  float mef_rate = mef / v.b.lr[r];
  float min_error_sum = (mef_rate < min_error_gain) ? mef_rate : min_error_gain;
  /* the first step whose sum leaves [min, max] ends the loop (recur-nn.c:387-389) */
  int n_exec = D;
  float error_sum = 0.0f;
  for (int k0 = 0; k0 < D; k0 += 64) {
    int k = k0 + lane;
    float es = (k < D) ? es_src[(size_t)k * es_stride] : 0.0f;
    bool stop = k < D && (es <= min_error_sum || es > max_error_sum);
    unsigned long long hit = __ballot(stop);
    int last = hit ? __ffsll((long long)hit) - 1 : min(63, D - 1 - k0);
    error_sum = __shfl(es, last, 64);
    if (hit) {
      n_exec = k0 + last + 1;
      break;
    }
  }
  /* the reference's t counts down from D and is not decremented on a break */
  bool broke = n_exec < D || (error_sum <= min_error_sum || error_sum > max_error_sum);
  int t = broke ? D - n_exec + 1 : 0;
  float scale;
  if (error_sum > error_sum_ceiling) {
    scale = soft_clip_dev(error_sum, max_error_sum);
  } else {
    scale = 1.0f;
    if (flags & 64u) { /* RNN_NET_FLAG_BPTT_ADAPTIVE_MIN_ERROR */
      int depth_error = D / 4 - t;
      if (mef < MAX_MIN_ERROR_FACTOR_F && (min_error_gain != min_error_sum || depth_error < 0)) {
        mef *= (float)(1.0f + depth_error * 1e-3);
      }
      mef = (mef >= ABS_MIN_ERROR_FACTOR_F) ? mef : ABS_MIN_ERROR_FACTOR_F;
    }
  }
  if (lane == 0) {
    v.b.mef[r] = mef;
    v.b.ih_scale[r] = scale;
    v.b.bptt_err[r] = error_sum;
    v.b.n_exec[r] = n_exec;
    v.b.depth_log[r] = D - t;
    v.b.stat_depth[r] += (double)(D - t);
  }
  /* 0x20000000: rnn_bptt_calculate without batching leaves the UNSCALED sum in ih_delta and puts
   * ih_scale into the rate (recur-nn.c:966-975) */
  const float cf = (flags & 0x20000000u) ? 1.0f : scale;
  for (int k = lane; k < D; k += 64) v.b.coef[(size_t)k * s.Scap + r] = (k < n_exec) ? cf : 0.0f;
}

__global__ __launch_bounds__(256) void k_bptt_control(View v, int row0, int nrows,
                                                      const unsigned char *active, unsigned flags,
                                                      int tn) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= nrows) return;
  const int r = row0 + j;
  bptt_control_wave(v, r, j, lane, active, flags, v.b.esum + r, (size_t)v.sh.Scap);
}

// k_extras_gather and k_bptt_control in one launch, one workgroup per stream: the waves
// share out the stream's steps, leave each step's error sum in LDS, and wave 0 then runs
// the control logic on them (nothing else needs the sums of other streams).
template <int MAXQ, int THREADS>
__global__ __launch_bounds__(THREADS) void k_extras_control(View v, int row0, int nrows, int nx,
                                                            int nxp, int tn,
                                                            const unsigned char *active,
                                                            unsigned flags) {
  extern __shared__ float es_sh[]; /* [D] the steps' totals; tn == 0: then [D + 1] the rows' own sums of squares */
  const RamdShape &s = v.sh;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x, r = row0 + j;
  /* tn == 0 (after the one-launch chain, which leaves no per-tile partial sums): the hidden columns'
   * part of step t's sum of squares (recur-nn.c:371) is the sum over the step's OUTPUT row, error
   * plane t + 1 -- the row that the item of step t + 1 holds in registers for its dot products.  So
   * every item also sums its own row, one more item (t = D) does only that, and the totals are put
   * together after the barrier. */
  const int items = tn == 0 ? s.D + 1 : s.D;
  float *hs_sh = es_sh + s.D;
  /* the next item's reads are requested before the current one is worked on */
  ExtrasIn<MAXQ> cur, nxt;
  if (wave < items) extras_load<MAXQ>(v, wave, r, nx, tn, lane, cur);
  for (int t = wave; t < items; t += THREADS / 64) {
    const int tnext = t + THREADS / 64;
    if (tnext < items) extras_load<MAXQ>(v, tnext, r, nx, tn, lane, nxt);
    if (tn == 0) {
      const float hs = row_sumsq<MAXQ>(cur);
      if (lane == 0) hs_sh[t] = hs;
    }
    if (t < s.D) {
      float es = extras_compute<MAXQ>(v, t, r, nx, nxp, tn, lane, cur);
      if (lane == 0) {
        if (tn != 0) v.b.esum[(size_t)t * s.Scap + r] = es;
        es_sh[t] = es;
      }
    }
    cur = nxt;
  }
  __syncthreads();
  if (wave == 0) {
    if (tn == 0) {
      for (int k = lane; k < s.D; k += 64) {
        const float es = hs_sh[k + 1] + es_sh[k];
        es_sh[k] = es;
        v.b.esum[(size_t)k * s.Scap + r] = es;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): the wave's own LDS writes before it reads them back */
    }
    bptt_control_wave(v, r, j, lane, active, flags, es_sh, 1);
  }
}

// ------------------------------------ one stream, small net: forward in one launch --
//
// rnn_opinion for ONE stream of a small net (recur-nn.c:83-154 without noise and bottom layer):
// the input row (bias, previous hidden values, the inputs the caller put there, the emergency
// soft clip of maybe_scale_inputs), hidden = act(x . W_ih) with the zero-row skip, bias node,
// out = hidden . W_ho -- k_assemble + k_gemm + k_fwd_finalize + k_out_layer as one workgroup.
// Thread (g = tid / HC, n = tid % HC) walks the input rows y = g, g + G, .. of column n (a wave
// reads whole contiguous rows of W_ih), the G partial sums per column are added in order.
// Preconditions (launcher): h_size <= 256, i_size <= 512, o_size <= 64.
__global__ __launch_bounds__(1024) void k_fwd_small(View v, int r) {
  __shared__ float xs[512], hsh[256], part[1024], red[16];
  const RamdShape &s = v.sh;
  const int I = s.I, H = s.H, O = s.O, hs = s.hidden_size;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float *slot = input_row<false>(v, r, 0);
  float *hid = v.b.hidden + (size_t)r * H;
  // the input row (k_assemble, mode KEEP) and its sum
  float sum = 0.0f;
  if (tid < I) {
    float x = (tid == 0) ? 1.0f : (tid <= hs) ? hid[tid] : slot[tid];
    xs[tid] = x;
    sum = x;
  }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
  {
    const float softclip = I * INPUT_MEAN_SOFT_TOP_F;
    float scale = 1.0f;
    if (sum > softclip) scale = soft_clip_dev(sum, softclip);
    if (tid < I) {
      const float x = xs[tid] * scale;
      if (sum > softclip) xs[tid] = x;
      slot[tid] = (sum > softclip) ? x : xs[tid];
    }
  }
  __syncthreads();
  // hidden sums: column n, rows y = g, g + G, ...
  const int HC = H <= 128 ? 128 : 256, G = 1024 / HC;
  const int n = tid & (HC - 1), g = tid / HC;
  float acc = 0.0f;
  if (n < H) {
    const float *w = v.b.ih_w + n;
    for (int y = g; y < I; y += G) {
      const float x = xs[y]; /* the same for the whole group: whole waves skip a zero row */
      if (x != 0.0f) acc += x * w[(size_t)y * H];
    }
  }
  part[tid] = acc;
  __syncthreads();
  if (tid < H) {
    float x = part[tid];
    for (int k = 1; k < G; k++) x += part[k * HC + tid];
    if (s.activation == 2) {
      x = (x > 0.0f) ? sqrtf(x + 1.0f) - 1.0f : 0.0f;
    } else if (s.activation == 5) {
      x = x < 20.0f ? x : 20.0f;
      x = (x > 0.0f) ? x : 0.0f;
    } else {
      x = (x > 0.0f) ? x : 0.0f;
    }
    if (tid == 0) x = 1.0f; /* the bias node, recur-nn.c:148 */
    hid[tid] = x;
    hsh[tid] = x;
  }
  __syncthreads();
  // output layer: column o, rows y = g2, g2 + 16, ...
  {
    const int o = tid & 63, g2 = tid >> 6;
    float a = 0.0f;
    if (o < O)
      for (int y = g2; y < H; y += 16) a += hsh[y] * v.b.ho_w[(size_t)y * O + o];
    part[tid] = a;
    __syncthreads();
    if (tid < O) {
      float x = part[tid];
      for (int k = 1; k < 16; k++) x += part[k * 64 + tid];
      v.b.out[(size_t)r * O + tid] = x;
    }
  }
}

// ------------------------------------------ one stream, small net: one launch --
//
// bptt_and_accumulate_error (recur-nn.c:303-450) for ONE stream of a small net as one
// workgroup -- what the per-net calls of an unchanged caller need (text-predict's default net
// has 99 hidden units: D launches of a 100 x 142 matrix-vector product are all launch latency).
// The workgroup walks the steps in the reference's own order.  Thread (y = tid / 8,
// chunk = tid % 8) owns 16 columns of rows y and y + 128 of the recurrent matrix: their
// weights AND their weight-delta sums live in its registers for the whole launch, so a step
// costs it four float4 of the incoming error row and two input values from LDS, 32 + 32
// multiply-adds, and three xor shuffles per row to close the dot products (LDS bandwidth is what
// a single CU runs out of first: an earlier version that fetched the weights from LDS took
// 2.1 us per step, all of it LDS reads).  The input rows of all D steps are brought into LDS
// up front and the barriers inside the loop wait for LDS only: vmcnt counts stores too on this
// architecture, so a global load inside the loop would make every step wait for the previous
// step's plane stores.  The sum of squares closes the step and the loop ends where the
// reference's would (recur-nn.c:387-389).  The planes the other kernels read afterwards
// (error rows, extras, sums: k_err_writeback, k_bottom_error, the log) are written as the
// chain kernels write them, the control logic is the shared bptt_control_wave, and
// ih_delta (+)= ih_scale * the accumulated matrix at the end.
// Preconditions (launcher): h_size <= 128, i_size <= 256, D * i_size floats fit in LDS.
__global__ __launch_bounds__(1024) void k_bptt_small(View v, int r, int accumulate, unsigned flags,
                                                     int nx, int nxp) {
  extern __shared__ __attribute__((aligned(16))) float bsm[];
  const RamdShape &s = v.sh;
  const int I = s.I, H = s.H, hs = s.hidden_size, D = s.D;
  float *h = bsm;             /* [128] error into the step, 0 at column 0 and past hidden_size */
  float *xon = h + 128;       /* [256] the step's input row, 0 where the row is skipped or absent */
  float *en = xon + 256;      /* [256] error out of the step, by row                           */
  float *red = en + 256;      /* [16] + [1]                                                    */
  float *es_sh = red + 20;    /* [D]  sums of squares by step                                  */
  float *xall = es_sh + ((D + 3) & ~3); /* [D][I] the input rows of all the steps              */
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int y0 = tid >> 3, n0 = (tid & 7) * 16; /* rows y0, y0 + 128; columns n0 .. n0 + 15 */
  const float *W = v.b.ih_w;
  const int idx0 = v.b.idx[r];
  /* weights: rows past I and columns past H are zero */
  float w[2][16], acc[2][16];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int y = y0 + 128 * q;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float4 t = (y < I && n0 + 4 * k < H) ? ld4(W + (size_t)y * H + n0 + 4 * k) : zero4();
      w[q][4 * k] = t.x; w[q][4 * k + 1] = t.y; w[q][4 * k + 2] = t.z; w[q][4 * k + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) acc[q][k] = 0.0f;
  }
  {
    const int I4 = I / 4;
    for (int e = tid; e < D * I4; e += 1024) {
      const int t = e / I4, i = e - t * I4;
      int slot = idx0 - t;
      if (slot < 0) slot += D;
      *reinterpret_cast<float4 *>(xall + t * I + 4 * i) = ld4(v.b.arena + ((size_t)slot * s.Scap + r) * I + 4 * i);
    }
    const float *e0 = v.b.ehi + (size_t)r * I; /* plane 0: the top layer's error */
    if (tid < 128) h[tid] = (tid == 0 || tid > hs || tid >= H) ? 0.0f : e0[tid];
    if (tid < 256) xon[tid] = 0.0f;
    for (int i = tid; i < D; i += 1024) es_sh[i] = 0.0f;
  }
  /* thresholds exactly as bptt_control_wave derives them */
  const float top = v.b.top_scaled[r];
  const float max_error_sum = MAX_ERROR_GAIN_F * top + 1;
  const float min_error_gain = MIN_ERROR_GAIN_F * top;
  const float mef_rate = v.b.mef[r] / v.b.lr[r];
  const float min_error_sum = (mef_rate < min_error_gain) ? mef_rate : min_error_gain;
  const size_t plane = (size_t)s.Scap * I;
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  /* sum over the eight lanes of a row without LDS: two quad permutes and a half-row mirror */
#define DPP_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
  const bool two_rows = (wave * 8 + 128) < I; /* this wave's second rows exist */
  __syncthreads();
  if (tid < I) { /* step 0's input row */
    const float xi = xall[tid];
    bool on = xi != 0.0f && (s.activation != 5 || xi < 20.0f);
    xon[tid] = on ? xi : 0.0f;
  }
  LDS_BARRIER();
  for (int t = 0; t < D; t++) {
    float sq = 0.0f;
    {
      float hv[16];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float4 t4 = ld4(h + n0 + 4 * k);
        hv[4 * k] = t4.x; hv[4 * k + 1] = t4.y; hv[4 * k + 2] = t4.z; hv[4 * k + 3] = t4.w;
      }
#pragma unroll
      for (int q = 0; q < 2; q++) {
        if (q == 1 && !two_rows) break; /* wave-uniform */
        const float xi = xon[y0 + 128 * q];
        /* weight deltas of the step (recur-nn.c:343-358) and the error that leaves it
         * (359-376); a skipped row has xi == 0: nothing is added and its error is 0 */
        float e = 0.0f, e1 = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
          acc[q][k] += xi * hv[k];
          acc[q][k + 1] += xi * hv[k + 1];
          e += w[q][k] * hv[k];
          e1 += w[q][k + 1] * hv[k + 1];
        }
        e += e1;
        DPP_ADD(e, 0xB1);  /* quad_perm [1,0,3,2] */
        DPP_ADD(e, 0x4E);  /* quad_perm [2,3,0,1] */
        DPP_ADD(e, 0x141); /* row_half_mirror: the other quad of the eight */
        if (s.activation == 2) e /= 2 * (xi + 1.0f);
        e = (xi != 0.0f) ? e : 0.0f;
        if ((tid & 7) == 0) {
          en[y0 + 128 * q] = e;
          sq += e * e;
        }
      }
    }
    /* the wave's share of the sum of squares: its row leaders sit in lanes 0, 8, .., 56 */
    DPP_ADD(sq, 0x128); /* row_ror 8: lanes 0 and 8 of every row of sixteen */
    {
      const int sqi = __builtin_bit_cast(int, sq);
      float ws = __builtin_bit_cast(float, __builtin_amdgcn_readlane(sqi, 0)) +
                 __builtin_bit_cast(float, __builtin_amdgcn_readlane(sqi, 16));
      ws += __builtin_bit_cast(float, __builtin_amdgcn_readlane(sqi, 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(sqi, 48));
      if (lane == 0) red[wave] = ws;
    }
    LDS_BARRIER();
    float es = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    es += ((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]));
    /* the planes, laid out as the chain and extras kernels leave them; the next step's rows */
    {
      float *eo = v.b.ehi + (size_t)(t + 1) * plane + (size_t)r * I;
      if (tid >= 1 && tid <= hs) eo[tid] = en[tid];
      float *xo = v.b.ex + ((size_t)(t + 1) * s.Scap + r) * nxp;
      if (tid < nxp) xo[tid] = (tid == 0) ? en[0] : (tid < nx) ? en[hs + tid] : 0.0f;
    }
    if (tid < 128) h[tid] = (tid == 0 || tid > hs) ? 0.0f : en[tid];
    if (tid < I && t + 1 < D) {
      const float xi = xall[(t + 1) * I + tid];
      bool on = xi != 0.0f && (s.activation != 5 || xi < 20.0f);
      xon[tid] = on ? xi : 0.0f;
    }
    if (tid == 0) {
      es_sh[t] = es;
      v.b.esum[(size_t)t * s.Scap + r] = es;
    }
    LDS_BARRIER();
    if (es <= min_error_sum || es > max_error_sum) break; /* the same for every thread */
  }
#undef DPP_ADD
#undef LDS_BARRIER
  if (wave == 0) {
    /* the steps that did not run left zeros, which end the scan of bptt_control_wave too */
    bptt_control_wave(v, r, 0, lane, nullptr, flags, es_sh, 1);
    if (lane == 0) {
      red[16] = v.b.ih_scale[r]; /* lane 0 wrote it */
      red[17] = __int_as_float(v.b.n_exec[r]);
    }
  }
  __syncthreads(); /* (also: every plane store of this workgroup has been performed) */
  const float scale = (flags & 0x20000000u) ? 1.0f : red[16]; /* see bptt_control_wave */
  {
    /* bptt->h_error / i_error as the reference leaves them: k_err_writeback's job, from the planes
     * this workgroup has just written (nothing of them was read before: no stale lines) */
    const int nex = __float_as_int(red[17]);
    if (nex > 0 && tid < I) {
      float *A = v.b.err_a + (size_t)r * I, *B = v.b.err_b + (size_t)r * I;
      float *last_written = (nex & 1) ? B : A, *last_read = (nex & 1) ? A : B;
      const float *enp = v.b.ehi + (size_t)nex * plane + (size_t)r * I;
      const float *epp = v.b.ehi + (size_t)(nex - 1) * plane + (size_t)r * I;
      const float *xn = v.b.ex + ((size_t)nex * s.Scap + r) * nxp;
      const float *xp = v.b.ex + ((size_t)(nex - 1) * s.Scap + r) * nxp;
      const int i = tid;
      last_written[i] = (i == 0) ? xn[0] : (i <= hs) ? enp[i] : xn[i - hs];
      if (i < H) {
        last_read[i] = (i == 0 || i > hs) ? 0.0f : epp[i];
      } else if (nex > 1) { /* the top error (nex == 1) only covers h_size entries */
        last_read[i] = xp[i - hs];
      }
    }
  }
  float *d = v.b.ih_delta;
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int y = y0 + 128 * q;
    if (y < I) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (n0 + 4 * k < H) {
          float4 *dp = reinterpret_cast<float4 *>(d + (size_t)y * H + n0 + 4 * k);
          float4 a = make_float4(acc[q][4 * k] * scale, acc[q][4 * k + 1] * scale, acc[q][4 * k + 2] * scale,
                                 acc[q][4 * k + 3] * scale);
          if (accumulate) {
            float4 o = *dp;
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
          }
          *dp = a;
        }
      }
    }
  }
}

// ------------------------------------------------------- finalize: delta --

// the sum of ks planes in plane order, eight loads in flight at a time
__device__ __forceinline__ float4 sum_planes(const float *src, size_t stride, int ks) {
  float4 sum = zero4();
  for (int z0 = 0; z0 < ks; z0 += 8) {
    float4 t[8];
#pragma unroll
    for (int z = 0; z < 8; z++) t[z] = ld4(src + (size_t)(z0 + z < ks ? z0 + z : z0) * stride);
#pragma unroll
    for (int z = 0; z < 8; z++)
      if (z0 + z < ks) { sum.x += t[z].x; sum.y += t[z].y; sum.z += t[z].z; sum.w += t[z].w; }
  }
  return sum;
}

// ih_delta (+)= sum of the K slabs (recur-nn.c:735-748 folded: the per-stream
// ih_scale already multiplies the error rows that went into the GEMM)
__global__ __launch_bounds__(256) void k_delta_finalize(float *delta, const float *slab,
                                                        size_t n4, size_t n, int ks,
                                                        int accumulate, int H, int hidden_size,
                                                        int rows_core, int ks_rest,
                                                        const float *rest, size_t rest_stride,
                                                        float *ho_delta, const float *ho_slab,
                                                        size_t ho_n, int ho_ks) {
  size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t ih_threads = ((n4 + 255) / 256) * 256;
  if (q >= ih_threads) {
    /* the blocks past ih_delta: ho_delta (+)= its K slabs, when the caller has them pending
     * (k_ho_delta_finalize without error ranges) */
    size_t e = q - ih_threads;
    if (ho_slab && 4 * e < ho_n) {
      float4 a = accumulate ? ld4(ho_delta + 4 * e) : zero4();
      for (int z = 0; z < ho_ks; z++) {
        float4 t = ld4(ho_slab + (size_t)z * ho_n + 4 * e);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      *reinterpret_cast<float4 *>(ho_delta + 4 * e) = a;
    }
    return;
  }
  if (q >= n4) return;
  /* rows below rows_core were produced with ks K slices, the others with ks_rest (their own planes) */
  const float *src = slab + 4 * q;
  size_t stride = n;
  if ((int)((4 * q) / (size_t)H) >= rows_core) {
    ks = ks_rest;
    src = rest + (4 * q - (size_t)rows_core * H);
    stride = rest_stride;
  }
  float4 a = accumulate ? ld4(delta + 4 * q) : zero4();
  float4 sum = sum_planes(src, stride, ks);
  /* the GEMM only produced columns 1..hidden_size; the others are exactly zero */
  int c = (int)((4 * q) % (size_t)H);
  a.x += (c + 0 >= 1 && c + 0 <= hidden_size) ? sum.x : 0.0f;
  a.y += (c + 1 >= 1 && c + 1 <= hidden_size) ? sum.y : 0.0f;
  a.z += (c + 2 >= 1 && c + 2 <= hidden_size) ? sum.z : 0.0f;
  a.w += (c + 3 >= 1 && c + 3 <= hidden_size) ? sum.w : 0.0f;
  *reinterpret_cast<float4 *>(delta + 4 * q) = a;
}

// Rebuilds bptt->h_error (err_a) and bptt->i_error (err_b) as the reference
// leaves them: the loop ping-pongs between the two buffers (recur-nn.c:384-386),
// zeroing element 0 and the pad of whichever one it reads (334-337).  Columns
// 1..hidden_size of a step's error live in ehi, column 0 and the input columns
// in ex.
__global__ __launch_bounds__(256) void k_err_writeback(View v, int row0, int nxp) {
  const RamdShape &s = v.sh;
  int r = row0 + blockIdx.x;
  int n = v.b.n_exec[r];
  if (n <= 0) return;
  float *A = v.b.err_a + (size_t)r * s.I, *B = v.b.err_b + (size_t)r * s.I;
  float *last_written = (n & 1) ? B : A; /* step n wrote it in full        */
  float *last_read = (n & 1) ? A : B;    /* step n read it (and zeroed bits) */
  const float *en = v.b.ehi + ((size_t)n * s.Scap + r) * s.I;
  const float *ep = v.b.ehi + ((size_t)(n - 1) * s.Scap + r) * s.I;
  const float *xn = v.b.ex + ((size_t)n * s.Scap + r) * nxp;
  const float *xp = v.b.ex + ((size_t)(n - 1) * s.Scap + r) * nxp;
  int hs = s.hidden_size;
  for (int i = threadIdx.x; i < s.I; i += 256) {
    last_written[i] = (i == 0) ? xn[0] : (i <= hs) ? en[i] : xn[i - hs];
    if (i < s.H) {
      last_read[i] = (i == 0 || i > hs) ? 0.0f : ep[i];
    } else if (n > 1) { /* the top error (n == 1) only covers h_size entries */
      last_read[i] = xp[i - hs];
    }
  }
}

// --------------------------------------------------------- K11: optimiser --

// The seven update rules of rnn_apply_learning (recur-nn.c:454-593) as one
// float4-wide elementwise kernel.  `rs` optionally points at a device float
// that multiplies the rate (ih_scale of the fused single-net path).
// Up to three arrays per launch (top layer, recurrent layer, bottom layer), each with its
// own rate: segment g owns blocks [first[g], first[g + 1]).
struct ApplySegs {
  float *w[3];
  const float *delta[3];
  float *m[3];
  float *aux[3];
  size_t n4[3];
  float rate[3];
  unsigned first[4];
  unsigned end1; /* one past the last block of segment 1 */
  /* when pend.slab is set, segment 1's deltas have not been summed yet: the kernel does what
   * k_delta_finalize would have (non-accumulating form) and stores them as well */
  RamdPendingDelta pend;
};
template <int METHOD>
__global__ __launch_bounds__(256) void k_apply(ApplySegs sg, float momentum, float mw,
                                               const float *rs) {
  const int g = (blockIdx.x >= sg.first[2]) ? 2 : (blockIdx.x >= sg.first[1]) ? 1 : 0;
  /* the recurrent layer's blocks run from the bottom of the matrix up: the rows that sum many
   * planes (the rest rows, last in memory) then start first instead of forming the tail */
  const unsigned bl = (g == 1) ? sg.end1 - 1 - blockIdx.x : blockIdx.x - sg.first[g];
  size_t q = (size_t)bl * 256 + threadIdx.x;
  if (q >= sg.n4[g]) return;
  float *w = sg.w[g], *m = sg.m[g], *aux = sg.aux[g];
  const float *delta = sg.delta[g];
  float rate = sg.rate[g];
  if (rs) rate *= *rs;
  float4 W = ld4(w + 4 * q), M = ld4(m + 4 * q), Dl;
  if (g == 1 && sg.pend.slab) {
    const RamdPendingDelta &pd = sg.pend;
    const float *src = pd.slab;
    size_t off = 4 * q, stride = pd.n;
    int ks = pd.ks;
    if ((int)(off / (size_t)pd.H) >= pd.rows_core) {
      ks = pd.ks_rest;
      src = pd.rest;
      off -= (size_t)pd.rows_core * pd.H;
      stride = pd.rest_stride;
    }
    float4 sum = sum_planes(src + off, stride, ks);
    int c = (int)((4 * q) % (size_t)pd.H);
    Dl.x = (c + 0 >= 1 && c + 0 <= pd.hidden_size) ? sum.x : 0.0f;
    Dl.y = (c + 1 >= 1 && c + 1 <= pd.hidden_size) ? sum.y : 0.0f;
    Dl.z = (c + 2 >= 1 && c + 2 <= pd.hidden_size) ? sum.z : 0.0f;
    Dl.w = (c + 3 >= 1 && c + 3 <= pd.hidden_size) ? sum.w : 0.0f;
    *reinterpret_cast<float4 *>(pd.delta_out + 4 * q) = Dl;
  } else if (g == 0 && sg.pend.ho_slab) {
    const RamdPendingDelta &pd = sg.pend;
    float4 t[8];
#pragma unroll
    for (int z = 0; z < 8; z++) t[z] = (z < pd.ho_ks) ? ld4(pd.ho_slab + z * pd.ho_n + 4 * q) : zero4();
    Dl = t[0];
#pragma unroll
    for (int z = 1; z < 8; z++)
      if (z < pd.ho_ks) { Dl.x += t[z].x; Dl.y += t[z].y; Dl.z += t[z].z; Dl.w += t[z].w; }
    *reinterpret_cast<float4 *>(pd.ho_delta_out + 4 * q) = Dl;
  } else {
    Dl = ld4(delta + 4 * q);
  }
  float4 A = (METHOD == 5 || METHOD == 6) ? ld4(aux + 4 * q) : zero4();
  float wv[4] = {W.x, W.y, W.z, W.w}, dv[4] = {Dl.x, Dl.y, Dl.z, Dl.w};
  float mv[4] = {M.x, M.y, M.z, M.w}, av[4] = {A.x, A.y, A.z, A.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (METHOD == 0) { /* weighted / simplified nesterov / classical: 482-487 */
      float t = dv[i] * rate;
      float mm = mv[i];
      wv[i] += t + mm * mw;
      mv[i] = (mm + t) * momentum;
    } else if (METHOD == 1) { /* nesterov: 501-508 */
      float t = dv[i] * rate;
      wv[i] += t;
      float mm = (mv[i] + t) * momentum;
      mv[i] = mm;
      wv[i] += mm;
    } else if (METHOD == 4) { /* adagrad: 518-524 */
      float d = dv[i];
      float a = mv[i] + d * d;
      wv[i] += d * rate / sqrtf(a);
      mv[i] = a;
    } else if (METHOD == 5) { /* adadelta, abs-value branch: 537-557 */
      const float renewal = 1.0f - momentum;
      float d = dv[i];
      float g = mv[i] * momentum;
      float s = av[i] * momentum;
      g += fabsf(d) * renewal + rate;
      float step = s / g * d;
      s += fabsf(step) * renewal + rate;
      mv[i] = g;
      av[i] = s;
      wv[i] += step;
    } else if (METHOD == 6) { /* rprop: 568-592 */
      const float max_step = 1 * rate;
      const float min_step = (float)(1e-6 * (double)rate);
      float d = dv[i], p = mv[i], step = av[i];
      if (d * p > 0.0f) {
        float g = step * 1.2f;
        step = (g < max_step) ? g : max_step;
      } else if (d * p < 0.0f) {
        float g = step * 0.5f;
        step = (g >= min_step) ? g : min_step;
        d = 0;
      }
      if (d > 0.0f) wv[i] += step;
      else wv[i] -= step;
      av[i] = step;
      mv[i] = d;
    }
  }
  *reinterpret_cast<float4 *>(w + 4 * q) = make_float4(wv[0], wv[1], wv[2], wv[3]);
  *reinterpret_cast<float4 *>(m + 4 * q) = make_float4(mv[0], mv[1], mv[2], mv[3]);
  if (METHOD == 5 || METHOD == 6)
    *reinterpret_cast<float4 *>(aux + 4 * q) = make_float4(av[0], av[1], av[2], av[3]);
}

// rnn_bptt_calculate's two updates in ONE launch: workgroups below `top_blocks` do apply_sgd_top_layer's
// immediate update for one stream (recur-nn.c:941-964), the others apply_learning_with_momentum on the recurrent layer (k_apply<0>'s
// arithmetic, recur-nn.c:482-487) when the call is due to apply it; *rs (ih_scale) multiplies its rate.
__global__ __launch_bounds__(256) void k_fused_updates(View v, int row, float rate, float momentum, float mw,
                                                       unsigned top_blocks, const float *rs) {
  const RamdShape &s = v.sh;
  if (blockIdx.x < top_blocks) {
    int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= s.H * s.O) return;
    int y = e / s.O, x = e - y * s.O;
    float h = v.b.hidden[(size_t)row * s.H + y];
    float mm = v.b.ho_m[e];
    if (h != 0.0f) {
      float d = v.b.o_error[(size_t)row * s.O + x] * (h * rate);
      v.b.ho_w[e] += d + mm * mw;
      mm += d;
      v.b.ho_m[e] = mm * momentum;
    } else {
      v.b.ho_w[e] += mm * mw;
      v.b.ho_m[e] = mm * momentum;
    }
    return;
  }
  const size_t q = (size_t)(blockIdx.x - top_blocks) * 256 + threadIdx.x;
  if (q >= (size_t)s.I * s.H / 4) return;
  float r = rate;
  if (rs) r *= *rs;
  float4 W = ld4(v.b.ih_w + 4 * q), M = ld4(v.b.ih_m + 4 * q);
  const float4 Dl = ld4(v.b.ih_delta + 4 * q);
  float wv[4] = {W.x, W.y, W.z, W.w}, mv[4] = {M.x, M.y, M.z, M.w};
  const float dv[4] = {Dl.x, Dl.y, Dl.z, Dl.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float t = dv[i] * r;
    float mm = mv[i];
    wv[i] += t + mm * mw;
    mv[i] = (mm + t) * momentum;
  }
  *reinterpret_cast<float4 *>(v.b.ih_w + 4 * q) = make_float4(wv[0], wv[1], wv[2], wv[3]);
  *reinterpret_cast<float4 *>(v.b.ih_m + 4 * q) = make_float4(mv[0], mv[1], mv[2], mv[3]);
}

// -------------------------------------------------------- K12: conditioning --

__global__ void k_scale(float *a, size_t n, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] *= scale;
}
__global__ void k_zero_small(float *a, size_t n) { /* recur-nn-helpers.h:126-133 */
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (fabsf(a[i]) > 1e-34f) ? a[i] : 0.0f;
}
__global__ void k_clamp(float *a, size_t n, float lo, float hi) { /* recur-nn.c:848-851 */
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float x = a[i];
    x = (x >= lo) ? x : lo;
    x = (x < hi) ? x : hi;
    a[i] = x;
  }
}
__global__ void k_add_at(float *a, size_t index, float v) { a[index] += v; }

// arg-max of |a| with the reference's tie rule (first index wins,
// recur-nn.c:830-838): each block publishes its best (value, index), block 0
// of a second launch reduces them.
struct BestAbs {
  float v;
  unsigned long long i;
};
__global__ __launch_bounds__(256) void k_absmax_part(const float *a, size_t n, BestAbs *part) {
  __shared__ BestAbs sh[256];
  size_t chunk = (n + gridDim.x - 1) / gridDim.x;
  size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  BestAbs b = {-1.0f, 0};
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    float x = fabsf(a[i]);
    if (x > b.v) { b.v = x; b.i = i; }
  }
  sh[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      BestAbs o = sh[threadIdx.x + off], m = sh[threadIdx.x];
      if (o.v > m.v || (o.v == m.v && o.i < m.i)) sh[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ void k_tall_poppy(float *a, const BestAbs *part, int nparts, float threshold,
                             float scale) {
  BestAbs b = part[0];
  for (int i = 1; i < nparts; i++) {
    BestAbs o = part[i];
    if (o.v > b.v || (o.v == b.v && o.i < b.i)) b = o;
  }
  if (b.v > threshold) a[b.i] *= scale;
}

__global__ void k_zero_f4(float *a, size_t n4) {
  size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n4) *reinterpret_cast<float4 *>(a + 4 * q) = zero4();
}

// ================================================================ launchers ==

static inline View make_view(const RamdShape *sh, const RamdBuffers *b) {
  View v;
  v.sh = *sh;
  v.b = *b;
  return v;
}

// ---- HIP-event timing of the GEMM classes (bench.py's roofline leg) ----
enum { T_CHAIN = 0, T_DELTA = 1, T_FWD = 2, T_APPLY = 3, T_OTHER = 4, T_CLASSES = 5 };
static int g_timing = 0;
struct TimedLaunch {
  hipEvent_t a, b;
  int cls, count;
};
static TimedLaunch g_ev[8192];
static int g_nev = 0;
static double g_ms[T_CLASSES];
static long g_launches[T_CLASSES];

static inline int timing_begin(hipStream_t st, int cls, int count = 1) {
  if (!g_timing || g_nev >= 8192) return -1;
  int i = g_nev++;
  HIP_CHECK(hipEventCreate(&g_ev[i].a));
  HIP_CHECK(hipEventCreate(&g_ev[i].b));
  g_ev[i].cls = cls;
  g_ev[i].count = count;
  HIP_CHECK(hipEventRecord(g_ev[i].a, st));
  return i;
}
static inline void timing_end(hipStream_t st, int i) {
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
extern "C" double ramd_timing_ms(int which, long *launches, int reset) {
  timing_collect();
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
static int env_int(const char *name, int dflt) {
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

// Split-K factor: enough workgroups to give every CU two or three, without
// shredding K into single tiles.
static int pick_ks(int tiles, int nkt, const char *env, size_t slab_floats, size_t out_floats) {
  int forced = env_int(env, 0);
  int ks;
  if (forced > 0) {
    ks = forced;
  } else {
    const int cus = 256;
    double best = 1e30;
    ks = 1;
    for (int k = 1; k <= 16 && k <= nkt; k++) {
      long wgs = (long)tiles * k;
      /* CUs run up to ~3 of these workgroups side by side; count time in
       * "K tiles on the busiest CU" plus a fill/drain charge per workgroup */
      double per_cu = (double)((wgs + cus - 1) / cus);
      double cost = per_cu * ((double)nkt / k) + 2.0 * (per_cu > 3 ? per_cu / 3 : 1) + 0.15 * k;
      if (cost < best) {
        best = cost;
        ks = k;
      }
    }
  }
  if (ks > nkt) ks = nkt;
  if (ks < 1) ks = 1;
  while (ks > 1 && (size_t)ks * out_floats > slab_floats) ks--;
  if (out_floats > slab_floats) { /* the workspace is sized for every output at engine creation */
    fprintf(stderr, "librecur_amd: a GEMM output of %zu floats does not fit the split-K workspace (%zu)\n",
            out_floats, slab_floats);
    abort();
  }
  return ks;
}

static GemmOut make_gemm_out(float *slab, int M, int N, int nkt, int ks, int col0, int ldc,
                             int row0m, int *blocks) {
  GemmOut o;
  o.slab = slab;
  o.M = M;
  o.N = N;
  o.ldc = ldc > 0 ? ldc : N;
  o.zs = (size_t)M * o.ldc;
  o.nkt = nkt;
  o.row0m = row0m;
  o.tm = (M - row0m + BM - 1) / BM;
  o.tn = (N - col0 + BN - 1) / BN;
  o.ks = ks;
  o.col0 = col0;
  int panels = o.tn * ks;
  *blocks = ((panels + 7) / 8) * 8 * o.tm;
  return o;
}

template <bool A_KM, bool B_KM, class Prob>
static void launch_gemm(hipStream_t st, const Prob &p, float *slab, int M, int N, int nkt, int ks,
                        int cls, int col0 = 0, int ldc = 0, int row0m = 0, size_t zs = 0) {
  int blocks;
  GemmOut o = make_gemm_out(slab, M, N, nkt, ks, col0, ldc, row0m, &blocks);
  if (zs) o.zs = zs;
  int ev = timing_begin(st, cls);
  RAMD_LAUNCH((k_gemm<A_KM, B_KM, Prob>), dim3(blocks), dim3(256), 0, st, p, o);
  timing_end(st, ev);
}

template <class Prob>
static void launch_gemm2(hipStream_t st, const Prob &p, float *slab, int M, int N, int nkt, int ks,
                         int cls, int col0, int ldc) {
  GemmOut o;
  o.slab = slab;
  o.M = M;
  o.N = N;
  o.ldc = ldc;
  o.zs = (size_t)M * ldc;
  o.nkt = nkt;
  o.tm = (M + BM2 - 1) / BM2;
  o.tn = (N - col0 + BN2 - 1) / BN2;
  o.ks = ks;
  o.col0 = col0;
  o.row0m = 0;
  int panels = o.tn * ks;
  int blocks = ((panels + 7) / 8) * 8 * o.tm;
  int ev = timing_begin(st, cls);
  RAMD_LAUNCH((k_gemm2<Prob>), dim3(blocks), dim3(256), 0, st, p, o);
  timing_end(st, ev);
}


extern "C" void ramd_launch_advance(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b,
                                    int row0, int nrows) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_advance, dim3((nrows + 255) / 256), dim3(256), 0, st, v, row0, nrows);
}

extern "C" void ramd_launch_assemble(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b,
                                     int row0, int nrows, int mode, const float *dense, int ld,
                                     int text_i, int global_first, int n_set, int advance) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_assemble, dim3(nrows), dim3(256), 0, st, v, row0, mode, dense, ld, text_i,
                     global_first, n_set, advance);
}

extern "C" void ramd_launch_bottom_forward(ramd_stream_t st_, const RamdShape *sh,
                                           const RamdBuffers *b, int row0, int nrows, int mode,
                                           const float *dense, int ld, int text_i,
                                           int global_first, int n_set, float noise) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  size_t shm = (size_t)(sh->bI + sh->bO) * sizeof(float);
  RAMD_LAUNCH(k_bottom_forward, dim3(nrows), dim3(256), shm, st, v, row0, mode, dense, ld,
                     text_i, global_first, n_set, noise);
}

extern "C" void ramd_launch_bottom_deltas(ramd_stream_t st_, const RamdShape *sh, RamdBuffers *b,
                                          int row0, int nrows, int accumulate,
                                          const unsigned char *active) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  int nxp = (sh->I - sh->hidden_size + 3) & ~3;
  RAMD_LAUNCH(k_bottom_error, dim3(nrows), dim3(64), 0, st, v, row0, nxp, active);
  int n = sh->bI * sh->bO;
  const float *cin = b->bcarry + (size_t)b->bcarry_cur * sh->bO;
  float *cout = b->bcarry + (size_t)(b->bcarry_cur ^ 1) * sh->bO;
  RAMD_LAUNCH(k_bottom_delta, dim3((n + 255) / 256), dim3(256), 0, st, v, row0, nrows,
                     accumulate, active, cin, cout);
  b->bcarry_cur ^= 1;
}

extern "C" int ramd_text_top_ok(const RamdShape *sh) {
  return sh->O <= 256 && sh->H <= 3072 && !env_int("RECUR_AMD_NO_TEXT_TOP", 0);
}

/* the device copy of the View for the kernels that take it by pointer, rewritten only when
 * it changes (the ring position is not part of it: those kernels get it as an argument) */
static const View *device_view(hipStream_t st, const View &v) {
  static View *d_view = nullptr;
  static View h_view;
  static bool have = false;
  View cur = v;
  cur.b.uniform_idx = 0;
  if (!d_view) HIP_CHECK(hipMalloc(&d_view, sizeof(View)));
  if (!have || memcmp(&cur, &h_view, sizeof(View)) != 0) {
    RAMD_LAUNCH(k_store_view, dim3(1), dim3(1), 0, st, cur, d_view);
    h_view = cur;
    have = true;
  }
  return d_view;
}

/* ---- the one-launch chain (k_chain_persist): its device state and the abort word ---- */
static ChainSync *g_chain_sync = nullptr;
static unsigned *g_chain_abort_host = nullptr, *g_chain_abort_dev = nullptr;
static unsigned g_chain_seq = 0;
static int g_chain_cus = -1;
/* The first one-launch chain of a process is checked synchronously: where its 256 workgroups cannot all be
 * resident (a CU-masked queue, a partition mode that still reports 256 CUs, a co-tenant holding CUs) it
 * raises the abort word; the launcher then resets it, stops using the kernel for the rest of the process
 * and the caller runs the launch-per-step chain for that very call (the one-launch chain reads error plane
 * 0 and writes planes >= 1 only, so its input is intact).  Later give-ups -- a co-tenant that arrives in
 * mid-run -- are still caught at the next synchronisation (rnn_core.c: dsync), where nothing can be redone. */
static bool g_chain_validated = false, g_chain_broken = false;

#ifdef PC_STAMPS
extern "C" void ramd_chain_stamps(unsigned long long *out) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_stamps), sizeof(unsigned long long) * 2 * 64 * 8));
}
#endif

extern "C" unsigned ramd_chain_abort_word(void) {
  return g_chain_abort_host ? *(volatile unsigned *)g_chain_abort_host : 0u;
}

static bool chain_persist_ok(const RamdShape *sh, const RamdBuffers *b, int nrows) {
  const int hs = sh->hidden_size;
  if (b->uniform_idx < 0 || (hs != 1024 && hs != 512 && hs != 256) || nrows < 1 || nrows % 16 != 0 ||
      sh->D > 60 || g_chain_broken || !env_int("RECUR_AMD_CHAIN_PERSIST", 1))
    return false;
  if (g_chain_cus < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    g_chain_cus = prop.multiProcessorCount;
  }
  return g_chain_cus == 256; /* 8 XCDs x 32 CUs: one workgroup per CU, 32 seats per XCD */
}

template <int ACT, int K>
static void launch_chain_persist_k(hipStream_t st, const View *d_view, const RamdShape *sh,
                                   const RamdBuffers *b, int row0, int nrows, unsigned seq, bool one, int nvalid,
                                   int vlo) {
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_persist<ACT, K, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, pc_lds_bytes(K)));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_persist<ACT, K, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, pc_lds_bytes(K)));
    HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_persist<ACT, K, true, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, pc_lds_bytes(K)));
    attr_set = true;
  }
  if (one && (nvalid < nrows || vlo > 0))
    RAMD_LAUNCH((k_chain_persist<ACT, K, true, true>), dim3(256), dim3(512), pc_lds_bytes(K), st, d_view,
                b->uniform_idx, row0, nrows, sh->D, seq, g_chain_sync, g_chain_abort_dev, nvalid, vlo);
  else if (one)
    RAMD_LAUNCH((k_chain_persist<ACT, K, true>), dim3(256), dim3(512), pc_lds_bytes(K), st, d_view,
                b->uniform_idx, row0, nrows, sh->D, seq, g_chain_sync, g_chain_abort_dev, nrows, 0);
  else
    RAMD_LAUNCH((k_chain_persist<ACT, K, false>), dim3(256), dim3(512), pc_lds_bytes(K), st, d_view,
                b->uniform_idx, row0, nrows, sh->D, seq, g_chain_sync, g_chain_abort_dev, nrows, 0);
}

/* row tiles per launch: 8 XCDs x (32 seats / column tiles) */
static int chain_persist_seats(const RamdShape *sh) {
  const int nt = sh->hidden_size / 32; /* column tiles; the one-launch chain exists for 8, 16 and 32 of them */
  return nt > 0 && nt <= 32 ? 8 * (32 / nt) : 0;
}
/* 16-stream row tiles (one sub-chain per workgroup) when they all still fit one launch: twice the
 * CUs for a small set; otherwise 32-stream tiles, which move more streams per microsecond */
static bool chain_persist_one(const RamdShape *sh, int nrows) {
  return nrows / 16 <= chain_persist_seats(sh) && (nrows % 32 != 0 || env_int("RECUR_AMD_CHAIN_ONE", 1));
}
static int chain_persist_rows(const RamdShape *sh, bool one) { return chain_persist_seats(sh) * (one ? 16 : 32); }

static bool launch_chain_persist(hipStream_t st, const View *d_view, const RamdShape *sh,
                                 const RamdBuffers *b, int row0, int nrows, bool one, int nvalid, int vlo = 0) {
  if (!g_chain_sync) {
    HIP_CHECK(hipMalloc(&g_chain_sync, sizeof(ChainSync)));
    HIP_CHECK(hipMemset(g_chain_sync, 0, sizeof(ChainSync)));
    HIP_CHECK(hipHostMalloc((void **)&g_chain_abort_host, 64, hipHostMallocMapped));
    *g_chain_abort_host = 0;
    HIP_CHECK(hipHostGetDevicePointer((void **)&g_chain_abort_dev, g_chain_abort_host, 0));
  }
  if (g_chain_seq >= (1u << 25)) { /* flags are seq * 64 + step and compare as unsigned numbers: start over */
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipMemset(g_chain_sync, 0, sizeof(ChainSync)));
    g_chain_seq = 0;
  }
  const unsigned seq = ++g_chain_seq;
  int ev = timing_begin(st, T_CHAIN, 1);
#define CHAIN_PERSIST(ACT)                                                                  \
  do {                                                                                      \
    if (sh->hidden_size == 1024) launch_chain_persist_k<ACT, 1024>(st, d_view, sh, b, row0, nrows, seq, one, nvalid, vlo); \
    else if (sh->hidden_size == 512) launch_chain_persist_k<ACT, 512>(st, d_view, sh, b, row0, nrows, seq, one, nvalid, vlo); \
    else launch_chain_persist_k<ACT, 256>(st, d_view, sh, b, row0, nrows, seq, one, nvalid, vlo);             \
  } while (0)
  if (sh->activation == 2) CHAIN_PERSIST(2);
  else if (sh->activation == 5) CHAIN_PERSIST(5);
  else CHAIN_PERSIST(1);
#undef CHAIN_PERSIST
  timing_end(st, ev);
  if (!g_chain_validated) {
    HIP_CHECK(hipStreamSynchronize(st));
    /* (RECUR_AMD_CHAIN_TEST_GIVEUP=1: the tests' way of taking this branch on a healthy device) */
    if (*(volatile unsigned *)g_chain_abort_host || env_int("RECUR_AMD_CHAIN_TEST_GIVEUP", 0)) {
      fprintf(stderr, "librecur_amd: the one-launch BPTT chain gave up on its first launch (code %u: its 256 "
                      "workgroups were not all resident, one per CU); using the launch-per-step chain from here on\n",
              *(volatile unsigned *)g_chain_abort_host);
      *(volatile unsigned *)g_chain_abort_host = 0;
      HIP_CHECK(hipMemset(g_chain_sync, 0, sizeof(ChainSync)));
      g_chain_seq = 0;
      g_chain_broken = true;
      return false;
    }
    g_chain_validated = true;
  }
  return true;
}

/* assemble + hidden layer in one launch for the text step (k_fwd_fused); returns what
 * ramd_launch_text_top wants as fwd_ks (negative: one plane of sums + per-tile padding
 * partials), or 0 when the preconditions do not hold and nothing was launched */
extern "C" int ramd_launch_forward_fused(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b,
                                         int row0, int nrows, int mode, int text_i,
                                         int global_first, int n_set) {
  if (b->uniform_idx < 0 || sh->bI || sh->hidden_size % CN != 0 || row0 + nrows > sh->Scap ||
      mode != RAMD_IN_TEXT /* the only caller that stops after the hidden layer */ || !ramd_text_top_ok(sh) ||
      env_int("RECUR_AMD_NO_FWD_FUSED", 0))
    return 0;
  const int tm = (nrows + CM - 1) / CM, tn = sh->hidden_size / CN;
  /* plane 0: sums; plane 1: [tn][nrows][4] padding partials */
  if ((size_t)nrows * sh->H + (size_t)tn * nrows * 4 > b->slab_floats || tn * 4 > sh->H) return 0;
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  const View *d_view = device_view(st, v);
  const int nstages = (sh->hidden_size + CK - 1) / CK;
  const int blocks = ((tn + 7) / 8) * 8 * tm;
  int ev = timing_begin(st, T_FWD);
  const bool exact = sh->hidden_size % CK == 0 && !env_int("RECUR_AMD_FWD_NS0", 0);
#define FWD_FUSED(NS)                                                                              \
  RAMD_LAUNCH((k_fwd_fused<NS>), dim3(blocks), dim3(512), 0, st, d_view, b->uniform_idx, row0, \
                     nrows, tm, tn, nstages, mode, text_i, global_first, n_set)
  if (exact && nstages == 8) FWD_FUSED(8);
  else if (exact && nstages == 4) FWD_FUSED(4);
  else if (exact && nstages == 2) FWD_FUSED(2);
  else if (exact && nstages == 16) FWD_FUSED(16);
  else FWD_FUSED(0);
#undef FWD_FUSED
  timing_end(st, ev);
  return -tn;
}

extern "C" void ramd_launch_text_top(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b,
                                     int row0, int nrows, int fwd_ks) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  size_t shm = (size_t)(sh->H + OUT_SEGS * 64 + 3 * sh->O) * sizeof(float);
  RAMD_LAUNCH(k_text_top, dim3(nrows), dim3(1024), shm, st, v, row0, nrows, fwd_ks);
}

extern "C" int ramd_launch_forward_hidden(ramd_stream_t st_, const RamdShape *sh,
                                          const RamdBuffers *b, int row0, int nrows, float noise,
                                          int leave_slabs);

/* rnn_opinion's device work for one stream of a small net in one launch (k_fwd_small); returns 0
 * when the shape is not its kind and nothing was launched */
extern "C" int ramd_launch_forward_small(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b, int r) {
  if (sh->H > 256 || sh->I > 512 || sh->O > 64 || sh->bI || !env_int("RECUR_AMD_FWD_SMALL", 1)) return 0;
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  int ev = timing_begin(st, T_FWD);
  RAMD_LAUNCH(k_fwd_small, dim3(1), dim3(1024), 0, st, v, r);
  timing_end(st, ev);
  return 1;
}

extern "C" void ramd_launch_forward(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b,
                                    int row0, int nrows, float noise) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  ramd_launch_forward_hidden(st_, sh, b, row0, nrows, noise, 0);
  if (sh->O == 4 && nrows >= 64) {
    RAMD_LAUNCH(k_out_layer_o4, dim3((nrows + 3) / 4), dim3(256), 0, st, v, row0, nrows);
  } else if (sh->O <= 256 && !env_int("RECUR_AMD_OUT_GEMM", 0)) {
    RAMD_LAUNCH(k_out_layer, dim3(nrows), dim3(1024),
                       (size_t)(sh->H + OUT_SEGS * 64) * sizeof(float), st, v, row0);
  } else { /* wide output layers (multi-head nets, O in the thousands): the MFMA GEMM */
    int tm = (nrows + BM - 1) / BM;
    int tn = (sh->O + BN - 1) / BN, nkt = (sh->H + BK - 1) / BK;
    int ks = pick_ks(tm * tn, nkt, "RECUR_AMD_KS_OUT", b->slab_floats, (size_t)nrows * sh->O);
    ProbOut p = {v, row0, nrows};
    launch_gemm<false, true, ProbOut>(st, p, b->slab, nrows, sh->O, nkt, ks, T_OTHER);
    int n4 = nrows * (sh->O / 4);
    RAMD_LAUNCH(k_sum_slabs, dim3((n4 + 255) / 256), dim3(256), 0, st,
                       b->out + (size_t)row0 * sh->O, sh->O, b->slab, nrows, sh->O, ks, 0);
  }
}

/* the hidden layer only: hidden = act(X . W_ih) (recur-nn.c:117-148).  With leave_slabs the
 * K slabs of the GEMM (noise included) stay in the workspace un-summed for
 * ramd_launch_text_top; the return value is their number. */
extern "C" int ramd_launch_forward_hidden(ramd_stream_t st_, const RamdShape *sh,
                                          const RamdBuffers *b, int row0, int nrows, float noise,
                                          int leave_slabs) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  int tm = (nrows + BM - 1) / BM;
  {
    int tn = (sh->H + BN - 1) / BN, nkt = (sh->I + BK - 1) / BK;
    int ks = pick_ks(tm * tn, nkt, "RECUR_AMD_KS_FWD", b->slab_floats, (size_t)nrows * sh->H);
    const int wide_ns = (sh->I + WK - 1) / WK;
    /* (from 2048 rows: h_size = hidden_size + 4 makes 33 column tiles of 64, and with a few hundred
     * rows that 33rd tile is a second round of workgroups: 97 us against the generic kernel's 60
     * at 512 x 2048; at 13,824 rows it is 1406 us against 1515) */
    if (nrows % WM == 0 && nrows >= 2048 && (wide_ns == 9 || wide_ns == 17 || wide_ns == 33) &&
        (size_t)nrows * sh->H <= b->slab_floats && env_int("RECUR_AMD_FWD_WIDE", 1)) {
      /* big sets: 64 x 64 tiles, operands by LDS-DMA (k_fwd_wide); one plane of sums */
      static bool attr_set = false;
      const size_t shm = (size_t)W_STAGES * W_STAGE_FLOATS * sizeof(float);
      if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_wide<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_wide<17>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_wide<33>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        attr_set = true;
      }
      const View *d_view = device_view(st, v);
      const int wtm = nrows / WM, wtn = (sh->H + WN - 1) / WN;
      const int supertiles = ((wtm + 3) / 4) * ((wtn + 7) / 8); /* of 4 x 8 tiles, 32 blocks each */
      const int wblocks = ((supertiles + 7) / 8) * 8 * 32;
      int ev = timing_begin(st, T_FWD);
      if (wide_ns == 33)
        RAMD_LAUNCH(k_fwd_wide<33>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, wtm, wtn);
      else if (wide_ns == 17)
        RAMD_LAUNCH(k_fwd_wide<17>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, wtm, wtn);
      else
        RAMD_LAUNCH(k_fwd_wide<9>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, wtm, wtn);
      timing_end(st, ev);
      ks = 1;
    } else if (b->uniform_idx >= 0) {
      ProbFwd<true> p = {v, row0, nrows};
      launch_gemm<false, true, ProbFwd<true>>(st, p, b->slab, nrows, sh->H, nkt, ks, T_FWD);
    } else {
      ProbFwd<false> p = {v, row0, nrows};
      launch_gemm<false, true, ProbFwd<false>>(st, p, b->slab, nrows, sh->H, nkt, ks, T_FWD);
    }
    if (noise != 0.0f && b->noise_spec_use) {
      int n4 = nrows * (sh->H / 4);
      RAMD_LAUNCH(k_noise_apply, dim3((n4 + 255) / 256), dim3(256), 0, st, v, row0, nrows);
    } else if (noise != 0.0f) {
      RAMD_LAUNCH(k_presynaptic_noise, dim3((nrows + 63) / 64), dim3(64), 0, st, v, row0, nrows,
                         noise);
    }
    if (leave_slabs) return ks;
    int n4 = nrows * (sh->H / 4);
    RAMD_LAUNCH(k_fwd_finalize, dim3((n4 + 255) / 256), dim3(256), 0, st, v, row0, nrows, ks);
  }
  return 0;
}

extern "C" void ramd_launch_softmax_error(ramd_stream_t st_, const RamdShape *sh,
                                          const RamdBuffers *b, int row0, int nrows) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_softmax_error, dim3(nrows), dim3(64), (size_t)sh->output_size * sizeof(float), st, v,
                     row0, nrows);
}

extern "C" void ramd_launch_xent_accumulate(ramd_stream_t st_, const RamdShape *sh,
                                            const RamdBuffers *b, int row, int count_it) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_xent_accumulate, dim3(1), dim3(64), (size_t)sh->output_size * sizeof(float),
                     st, v, row, count_it);
}

extern "C" void ramd_launch_multi_xent_accumulate(ramd_stream_t st_, const RamdShape *sh,
                                                  const RamdBuffers *b, int row, int alphabet_len,
                                                  int n_classes, double *acc, int count_it) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_multi_xent_accumulate, dim3(n_classes), dim3(64), (size_t)alphabet_len * sizeof(float), st,
              v, row, alphabet_len, acc, count_it);
}

extern "C" void ramd_launch_sigmoid_mse_error(ramd_stream_t st_, const RamdShape *sh,
                                              const RamdBuffers *b, int row0, int nrows, int n,
                                              const float *targets, int ld) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_sigmoid_mse_error, dim3((nrows * n + 255) / 256), dim3(256), 0, st, v, row0, nrows, n,
              targets, ld);
}

extern "C" void ramd_launch_sigmoid_outputs(ramd_stream_t st_, const RamdShape *sh,
                                            const RamdBuffers *b, int r0, int nrows, int n) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_sigmoid_outputs, dim3((nrows * n + 255) / 256), dim3(256), 0, st, v, r0, nrows, n);
}

extern "C" void ramd_launch_multi_softmax_error(ramd_stream_t st_, const RamdShape *sh,
                                                const RamdBuffers *b, int row0, int nrows,
                                                int alphabet_len, int n_classes,
                                                unsigned long long threshold, const int *tclass,
                                                int *ranges, int range_stride) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  if (n_classes > MS_MAXCLS) {
    fprintf(stderr, "librecur_amd: more than %d class heads\n", MS_MAXCLS);
    abort();
  }
  RAMD_LAUNCH(k_multi_softmax_error, dim3(nrows), dim3(64 * MS_WAVES),
                     (size_t)MS_WAVES * alphabet_len * sizeof(float), st, v, row0, alphabet_len, n_classes,
                     threshold, tclass, ranges, range_stride);
}

extern "C" void ramd_launch_grouped_softmax_error(ramd_stream_t st_, const RamdShape *sh,
                                                  const RamdBuffers *b, int row0, int nrows,
                                                  int ngroups, int largest, const int *goff,
                                                  const int *gsize, const int *gt,
                                                  const float *weight) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_grouped_softmax_error, dim3(nrows), dim3(64), (size_t)largest * sizeof(float), st,
                     v, row0, ngroups, goff, gsize, gt, weight);
}

extern "C" void ramd_launch_clear_deltas(ramd_stream_t st_, const RamdShape *sh,
                                         const RamdBuffers *b) {
  hipStream_t st = (hipStream_t)st_;
  size_t ih4 = (size_t)sh->I * sh->H / 4, ho4 = (size_t)sh->H * sh->O / 4;
  RAMD_LAUNCH(k_zero_f4, dim3((unsigned)((ih4 + 255) / 256)), dim3(256), 0, st, b->ih_delta, ih4);
  RAMD_LAUNCH(k_zero_f4, dim3((unsigned)((ho4 + 255) / 256)), dim3(256), 0, st, b->ho_delta, ho4);
  if (sh->bI) { /* recur-nn.c:687-692 */
    size_t b4 = (size_t)sh->bI * sh->bO / 4, c4 = (size_t)2 * sh->bO / 4;
    RAMD_LAUNCH(k_zero_f4, dim3((unsigned)((b4 + 255) / 256)), dim3(256), 0, st, b->bdelta, b4);
    RAMD_LAUNCH(k_zero_f4, dim3((unsigned)((c4 + 255) / 256)), dim3(256), 0, st, b->bcarry, c4);
  }
}

static int g_calc_wrote_images = 0;
/* Multi-GPU: the weight-delta GEMM in two row halves, so that the sum over the ranks of the first half
 * can travel while the second half is still being multiplied (rnn_core.c sets the hook for the call it
 * wants split; the launcher calls it after each half's deltas are complete in ih_delta || ho_delta,
 * with the half's range in floats from ih_delta). */
static void (*g_delta_half_hook)(void *ctx, int half, size_t first_float, size_t n_floats) = nullptr;
static void *g_delta_half_ctx = nullptr;
extern "C" void ramd_set_delta_half_hook(void (*hook)(void *, int, size_t, size_t), void *ctx) {
  g_delta_half_hook = hook;
  g_delta_half_ctx = ctx;
}
/* whether the last ramd_launch_calc_deltas also rebuilt bptt->h_error / i_error (reads and clears) */
extern "C" int ramd_calc_wrote_images(void) {
  int w = g_calc_wrote_images;
  g_calc_wrote_images = 0;
  return w;
}

extern "C" void ramd_launch_calc_deltas(ramd_stream_t st_, const RamdShape *sh,
                                        const RamdBuffers *b, int row0, int nrows, int accumulate,
                                        const int *ranges, int range_stride,
                                        const unsigned char *active, unsigned flags,
                                        RamdPendingDelta *defer) {
  g_calc_wrote_images = 0;
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  // top layer
  size_t shm = (size_t)(sh->O + sh->H) * sizeof(float);
  if (!(flags & 0x40000000u)) { /* ramd_launch_text_top has already done the top backprop */
    if (ranges && env_int("RECUR_AMD_TOP_RANGED", 1)) {
      /* up to 16 workgroups per stream (their partial sums sit in the split-K workspace, which
       * nothing uses at this point) */
      /* (measured: 256 streams with 1 / 2 / 4 / 8 / 16 workgroups per stream = 552 / 548 / 538 /
       * 550 / 629 us per generation; 64 streams with 4 / 16: 347 / 358; 32 streams with 8 / 16: 306 / 312) */
      int nb = 256 / nrows;
      if (nb < 4) nb = nrows > 1024 ? 1 : 4;
      if (nb > 16) nb = 16;
      if ((size_t)nrows * nb > b->slab_floats) nb = 1;
      RAMD_LAUNCH(k_top_backprop_ranged, dim3(nrows, nb), dim3(1024), shm, st, v, row0, ranges, range_stride,
                  active, b->slab);
      if (nb > 1)
        RAMD_LAUNCH(k_top_backprop_scale, dim3(nrows), dim3(256), 0, st, v, row0, active, b->slab, nb);
    } else
      RAMD_LAUNCH(k_top_backprop, dim3(nrows), dim3(256), shm, st, v, row0, ranges, range_stride, active);
  }
  /* the weight-delta GEMM's path is decided here already: when it ends with the small GEMM
   * over the rows above the last whole 128-row tile, the top layer's equally small delta
   * GEMM can share that launch (nothing before the optimiser needs its result) */
  const bool dma = b->uniform_idx >= 0 && nrows % BK == 0 && sh->hidden_size % 128 == 0 &&
                   sh->I >= 128 && sh->activation != 5 && env_int("RECUR_AMD_DELTA_DMA", 1);
  const bool has_rest = dma && (sh->I / 128) * 128 < sh->I;
  bool ho_paired = false, ho_finalize_after = false, ho_in_final = false;
  ProbHoDelta ho_p = {};
  int ho_nkt = 0, ho_ks = 0;
  if (!(flags & 0x80000000u)) { /* the fused single-net path updates W_ho directly */
    int tm = (sh->H + BM - 1) / BM, tn = (sh->O + BN - 1) / BN;
    int nkt = (nrows + BK - 1) / BK;
    int ho = sh->H * sh->O;
    int ks = pick_ks(tm * tn, nkt, "RECUR_AMD_KS_HO", b->slab_floats, (size_t)ho);
    /* per-stream 1.0 / 0.0 participation flags as floats (b->coef plane 0 is free here:
     * k_bptt_control rewrites it later in this call) */
    const float *live = b->ones + row0;
    if (active) {
      RAMD_LAUNCH(k_live_mask, dim3((nrows + 255) / 256), dim3(256), 0, st, b->coef + row0,
                         active, nrows);
      live = b->coef + row0;
    }
    ProbHoDelta p = {v, row0, nrows, live};
    if (defer) defer->ho_slab = nullptr;
    if (defer && !accumulate && !ranges && b->ho_slab) {
      /* the optimiser launch that follows sums these slabs itself (and stores ho_delta) */
      if (ks > 8) ks = 8;
      if (has_rest && !active && env_int("RECUR_AMD_PAIR_HO", 1)) {
        ho_paired = true; /* launched together with the rest rows of the weight-delta GEMM */
        ho_p = p;
        ho_nkt = nkt;
        ho_ks = ks;
      } else {
        launch_gemm<true, true, ProbHoDelta>(st, p, b->ho_slab, sh->H, sh->O, nkt, ks, T_OTHER);
      }
      defer->ho_slab = b->ho_slab;
      defer->ho_n = (size_t)ho;
      defer->ho_ks = ks;
      defer->ho_delta_out = b->ho_delta;
    } else if (has_rest && !active && b->ho_slab && env_int("RECUR_AMD_PAIR_HO", 1)) {
      /* not deferred (the deltas are wanted as such: accumulation, an all-reduce between the
       * ranks): still one launch with the rest rows, summed right after it */
      if (ks > 8) ks = 8;
      ho_paired = true;
      ho_finalize_after = true;
      ho_p = p;
      ho_nkt = nkt;
      ho_ks = ks;
    } else {
      launch_gemm<true, true, ProbHoDelta>(st, p, b->slab, sh->H, sh->O, nkt, ks, T_OTHER);
      /* with one range list per stream the set of touched columns differs per stream; the
       * error is zero outside a stream's own ranges, so every column may take its sum */
      RAMD_LAUNCH(k_ho_delta_finalize, dim3((ho + 255) / 256), dim3(256), 0, st, v, b->slab,
                         ks, accumulate, range_stride ? nullptr : ranges);
    }
  }
  // BPTT chain: D dependent steps, one launch each, then the extras of all steps
  bool control_done = false;
  const int tn = (sh->hidden_size + CN - 1) / CN;
  int tn_parts = tn; /* partial sums of squares per (step, stream): one per column tile of the chain kernel used */
  const int nx = sh->I - sh->hidden_size; /* column 0 + the input columns */
  const int nxp = (nx + 3) & ~3;
  if (nrows == 1 && !active && row0 < sh->Scap && sh->H <= 256 &&
      env_int("RECUR_AMD_BPTT_SMALL", 1)) {
    /* one stream of a small net (the per-net calls): chain, extras, control and weight deltas
     * in one workgroup */
    /* h_size <= 128 and i_size <= 256 (text-predict's default 99 hidden units: 100 x 142):
     * the matrix lives in the workgroup's registers; larger nets take the launch-per-step route */
    const size_t shm = (size_t)(128 + 256 + 256 + 20 + ((sh->D + 3) & ~3) + (size_t)sh->D * sh->I) * sizeof(float);
    if (sh->H <= 128 && sh->I <= 256 && shm <= 150 * 1024) {
      static bool attr_set = false;
      if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_bptt_small,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
      }
      if (defer) defer->slab = nullptr; /* ih_delta is written here: nothing left for the optimiser to sum */
      int ev = timing_begin(st, T_CHAIN, 1);
      RAMD_LAUNCH(k_bptt_small, dim3(1), dim3(1024), shm, st, v, row0, accumulate, flags, nx, nxp);
      timing_end(st, ev);
      g_calc_wrote_images = 1; /* the error images are done: no k_err_writeback for this call */
      return;
    }
  }
  {
    int tm = (nrows + CM - 1) / CM;
    int nstages = (sh->hidden_size + CK - 1) / CK; /* K = the hidden columns 1..hidden_size */
    int blocks = ((tn + 7) / 8) * 8 * tm;
    /* one event pair around the D launches: the per-launch average then carries
     * 1/D of the event overhead instead of all of it */
    const View *d_view = device_view(st, v);
    /* a set that is not whole 16-row tiles runs over the rows above it (Scap is a multiple of 16:
     * they exist), which are multiplied along and never stored (PAD) -- a one-net trainer or a
     * per-net call then takes the one-launch chain with a single tile instead of D launches */
    int chain_rows = nrows;
    if (nrows % 16 != 0 && row0 + ((nrows + 15) & ~15) <= sh->Scap) chain_rows = (nrows + 15) & ~15;
    /* ... and a small set that does not start on a tile boundary (a per-net call on stream j):
     * the tiles from the boundary below it, one launch */
    const int span_base = row0 & ~15, span = ((row0 + nrows + 15) & ~15) - span_base;
    const bool windowed = span_base != row0 && span_base + span <= sh->Scap && chain_persist_ok(sh, b, span) &&
                          span / 16 <= chain_persist_seats(sh);
    bool windowed_done = false;
    if (windowed) {
      windowed_done = launch_chain_persist(st, d_view, sh, b, span_base, span, true, row0 - span_base + nrows, row0 - span_base);
    }
    bool persist = windowed_done || chain_persist_ok(sh, b, chain_rows);
    if (persist && !windowed_done) { /* as many row tiles per launch as there are seats; more streams: more launches */
      /* (an odd number of 16-stream tiles beyond one launch: 32-stream tiles, the last 16 streams alone) */
      for (int r = 0; r < chain_rows;) {
        const int left = chain_rows - r, real_left = nrows - r;
        bool one = chain_persist_one(sh, left);
        int n = 0;
        if (!one) { /* 32-stream tiles over whole, real tiles only */
          n = real_left & ~31;
          if (n > chain_persist_rows(sh, false)) n = chain_persist_rows(sh, false);
          if (n == 0) one = true;
        }
        if (one) {
          n = chain_persist_rows(sh, true);
          if (n > left) n = left;
        }
        if (!launch_chain_persist(st, d_view, sh, b, row0 + r, n, one, real_left < n ? real_left : n)) {
          persist = false; /* (only a process's first launch can fail here: r == 0, nothing done yet) */
          break;
        }
        r += n;
      }
    }
    if (persist) tn_parts = 0; /* the one-launch chain leaves no partial sums: the extras sum the rows themselves */
    /* big sets of a wide net: 64 x 64 tiles (k_chain_wide), one partial sum per 64 columns */
    const int wide_ns = sh->hidden_size / WK;
    const bool wide = !persist && b->uniform_idx >= 0 && nrows % WM == 0 && sh->hidden_size % WN == 0 &&
                      (wide_ns == 16 || wide_ns == 24 || wide_ns == 32) &&
                      (nrows / WM) * (sh->hidden_size / WN) >= 128 && env_int("RECUR_AMD_CHAIN_WIDE", 1);
    if (wide) {
      static bool attr_set = false;
      const size_t shm = (size_t)W_STAGES * W_STAGE_FLOATS * sizeof(float);
      if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_wide<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_wide<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_wide<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        attr_set = true;
      }
      const int wtm = nrows / WM, wtn = sh->hidden_size / WN;
      const int wblocks = ((wtn + 7) / 8) * 8 * wtm;
      tn_parts = wtn;
      int evw = timing_begin(st, T_CHAIN, sh->D);
      for (int t = 0; t < sh->D; t++) {
        if (wide_ns == 32)
          RAMD_LAUNCH(k_chain_wide<32>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, t, wtm, wtn);
        else if (wide_ns == 24)
          RAMD_LAUNCH(k_chain_wide<24>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, t, wtm, wtn);
        else
          RAMD_LAUNCH(k_chain_wide<16>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, t, wtm, wtn);
      }
      timing_end(st, evw);
    }
    int ev = (persist || wide) ? -1 : timing_begin(st, T_CHAIN, sh->D);
    for (int t = 0; t < ((persist || wide) ? 0 : sh->D); t++) {
#define CHAIN_NS(NS)                                                                               \
  RAMD_LAUNCH((k_chain_main<true, NS>), dim3(blocks), dim3(512), 0, st, d_view, b->uniform_idx, \
                     row0, nrows, t, tm, tn, nstages)
      const bool exact = sh->hidden_size % CK == 0;
      if (b->uniform_idx >= 0 && exact && nstages == 8) CHAIN_NS(8);
      else if (b->uniform_idx >= 0 && exact && nstages == 4) CHAIN_NS(4);
      else if (b->uniform_idx >= 0 && exact && nstages == 2) CHAIN_NS(2);
      else if (b->uniform_idx >= 0 && exact && nstages == 16) CHAIN_NS(16);
      else if (b->uniform_idx >= 0)
        RAMD_LAUNCH(k_chain_main<true>, dim3(blocks), dim3(512), 0, st, d_view, b->uniform_idx, row0,
                           nrows, t, tm, tn, nstages);
      else
        RAMD_LAUNCH(k_chain_main<false>, dim3(blocks), dim3(512), 0, st, d_view, b->uniform_idx,
                           row0, nrows, t, tm, tn, nstages);
#undef CHAIN_NS
    }
    timing_end(st, ev);
    int M = sh->D * nrows;
    int etm = (M + BM - 1) / BM, etn = (nx + BN - 1) / BN, nkt = (sh->H + BK - 1) / BK;
    int ks = pick_ks(etm * etn, nkt, "RECUR_AMD_KS_EXTRAS", b->slab_floats, (size_t)M * nxp);
    /* the gather over the non-zero input rows (one-hot symbols: two rows per step and stream) or,
     * for dense inputs with more than a handful of columns, the GEMM over all of them */
    if (sh->H <= 2304 && !env_int("RECUR_AMD_EXTRAS_GEMM", 0) && !(b->dense_inputs && nx > 8)) {
      const int nq = (sh->H / 4 + 63) / 64;
      if (env_int("RECUR_AMD_EXTRAS_SPLIT", 0)) {
        if (nq <= 5)
          RAMD_LAUNCH(k_extras_gather<5>, dim3((M + 3) / 4), dim3(256), 0, st, v, row0, nrows, nx,
                             nxp, tn_parts);
        else if (nq <= 8)
          RAMD_LAUNCH(k_extras_gather<8>, dim3((M + 3) / 4), dim3(256), 0, st, v, row0, nrows, nx,
                             nxp, tn_parts);
        else /* h_size 2052: hidden 2048 */
          RAMD_LAUNCH(k_extras_gather<9>, dim3((M + 3) / 4), dim3(256), 0, st, v, row0, nrows, nx,
                             nxp, tn_parts);
      } else {
        /* extras and control in one launch, one workgroup per stream */
        const size_t shm = (size_t)(2 * sh->D + 1) * sizeof(float);
        if (nq <= 5)
          RAMD_LAUNCH((k_extras_control<5, 1024>), dim3(nrows), dim3(1024), shm, st, v, row0, nrows,
                             nx, nxp, tn_parts, active, flags);
        else if (nq <= 8)
          RAMD_LAUNCH((k_extras_control<8, 512>), dim3(nrows), dim3(512), shm, st, v, row0, nrows,
                             nx, nxp, tn_parts, active, flags);
        else /* h_size 2052: hidden 2048 */
          RAMD_LAUNCH((k_extras_control<9, 512>), dim3(nrows), dim3(512), shm, st, v, row0, nrows,
                             nx, nxp, tn_parts, active, flags);
        control_done = true;
      }
    } else { /* very wide nets: the dense GEMM over all extra columns */
      ProbExtras p = {v, row0, nrows, nx};
      launch_gemm<false, false, ProbExtras>(st, p, b->slab, M, nxp, nkt, ks, T_OTHER);
      RAMD_LAUNCH(k_extras_finalize, dim3(M), dim3(64), 0, st, v, row0, nrows, nx, nxp, ks, tn_parts);
    }
  }
  if (!control_done)
    RAMD_LAUNCH(k_bptt_control, dim3((nrows + 3) / 4), dim3(256), 0, st, v, row0, nrows,
                       active, flags, tn);
  // weight deltas: one GEMM over (step, stream)
  {
    /* only columns 1..hidden_size of the delta can be non-zero (h_error[0] and the pad are
     * zero, recur-nn.c:334-337), so the column tiles start at 1: at hidden 1024 that is 16
     * exact tiles instead of 17 */
    const int ncol = sh->hidden_size + 1;
    int tm = (sh->I + BM - 1) / BM, tn = (ncol - 1 + BN - 1) / BN;
    int rtiles = (nrows + BK - 1) / BK;
    int nkt = sh->D * rtiles;
    size_t n = (size_t)sh->I * sh->H;
    int ks = pick_ks(tm * tn, nkt, "RECUR_AMD_KS_DELTA", b->slab_floats, n);
    const bool big = env_int("RECUR_AMD_DELTA_TILE", sh->I >= 256 && nkt >= 16 ? 128 : 64) == 128;
    if (big) {
      int tm2 = (sh->I + BM2 - 1) / BM2, tn2 = (ncol - 1 + BN2 - 1) / BN2;
      ks = pick_ks(tm2 * tn2, nkt, "RECUR_AMD_KS_DELTA", b->slab_floats, n);
    }
    int rows_core = sh->I, ks_rest = ks;
    float *rest_base = b->slab; /* planes of the rows from rows_core on: rest_base + z * rest_stride */
    size_t rest_stride = n;
    if (dma) {
      /* whole 128-row tiles by LDS-DMA, one workgroup per CU; the rows above them (the
       * input rows of a text net) by the generic kernel with its own K split */
      static bool attr_set = false;
      size_t shm = (size_t)DD_STAGES * DD_STAGE_FLOATS * sizeof(float);
      const size_t shm_rest = shm + (size_t)DD_REST_FLOATS * sizeof(float);
      if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_delta_dma<0>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_delta_dma<64>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_rest));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_delta_dma<128>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_rest));
        attr_set = true;
      }
      rows_core = (sh->I / 128) * 128;
      GemmOut o;
      o.slab = b->slab;
      o.M = sh->I;
      o.N = ncol;
      o.ldc = sh->H;
      o.zs = n;
      o.nkt = nkt;
      o.tm = rows_core / 128;
      o.tn = sh->hidden_size / 128;
      o.col0 = 1;
      o.row0m = 0;
      int tiles = o.tm * o.tn;
      int kd = env_int("RECUR_AMD_KS_DELTA", 0);
      if (kd != 1 && kd != 2 && kd != 4 && kd != 8) {
        kd = 8;
        while (kd > 1 && tiles * kd > 256) kd >>= 1;
      }
      while (kd > 1 && (kd > nkt || (size_t)kd * n > b->slab_floats)) kd >>= 1;
      o.ks = ks = kd;
      const int per = 8 / kd;
      int blocks = ((tiles + per - 1) / per) * 8;
      /* the rows above the last whole tile inside the same launch (see DeltaRest) when there are at
       * most 128 of them, at least four row tiles to share them out, and room for ks * tm planes */
      const int rest_rows = sh->I - rows_core;
      const size_t rest_plane = (size_t)rest_rows * sh->H;
      const bool rest_in = rest_rows > 0 && rest_rows <= 128 && o.tm >= 4 && kd * o.tm <= RAMD_MAX_REST_PLANES &&
                           (size_t)kd * n + (size_t)kd * o.tm * rest_plane <= b->slab_floats &&
                           env_int("RECUR_AMD_DELTA_REST_IN", 1);
      /* ---- the two-halves form (see g_delta_half_hook): rows [0, tm / 2 tiles) with twice the K split
       * (the same number of workgroups and of slab bytes), summed into ih_delta, hook; then the upper
       * tiles with the rest rows riding along, the top layer's deltas, summed, hook */
      if (g_delta_half_hook && !defer && rest_in && o.tm >= 8 && o.tm % 2 == 0 && ho_paired && ho_finalize_after &&
          !ranges && env_int("RECUR_AMD_DIST_OVERLAP", 0)) {
        /* OFF by default -- measured with ONE rank (bench.py --dist, round 3): 305 against 255 us per
         * generation.  Two launches of half the rows with twice the K split cost the GEMM class +24 us
         * (each workgroup's prologue, epilogue and ring fill amortise over 20 instead of 40 K tiles, the
         * finalize sums 8 planes twice), the two event hand-overs and RCCL calls another ~25 us: more
         * than the ~2.2 MB all-reduce it could hide is expected to take over xGMI.  And while a half's
         * GEMM holds every CU with 148 KB of LDS, RCCL's own workgroups can only become resident as that
         * launch drains.  Kept for the day a multi-GPU node says otherwise (RECUR_AMD_DIST_OVERLAP=1;
         * results equal the one-launch form: tests/test_gpu_dist.py). */
        const int tmh = o.tm / 2;
        int kd2 = 8;
        while (kd2 > 1 && (tmh * o.tn * kd2 > 256 || kd2 > nkt ||
                           (size_t)kd2 * n + (size_t)kd2 * tmh * rest_plane > b->slab_floats))
          kd2 >>= 1;
        const int per2 = 8 / kd2;
        const int blocks2 = ((tmh * o.tn + per2 - 1) / per2) * 8;
        const size_t half_floats = (size_t)tmh * 128 * sh->H, n4h = half_floats / 4;
        GemmOut oh = o;
        oh.tm = tmh;
        oh.ks = kd2;
        int evh = timing_begin(st, T_DELTA, 2);
        { /* lower half: no rest rows */
          DeltaRest none = {};
          oh.row0m = 0;
          RAMD_LAUNCH(k_delta_dma<0>, dim3(blocks2), dim3(512), shm, st, v, row0, nrows, oh, none);
          RAMD_LAUNCH(k_delta_finalize, dim3((unsigned)((n4h + 255) / 256)), dim3(256), 0, st, b->ih_delta, b->slab, n4h, n,
                      kd2, accumulate, sh->H, sh->hidden_size, tmh * 128, 0, b->slab, (size_t)0, b->ho_delta,
                      (const float *)nullptr, (size_t)0, 0);
          g_delta_half_hook(g_delta_half_ctx, 0, 0, half_floats);
        }
        { /* upper half + rest rows + the top layer */
          DeltaRest dr;
          dr.planes = b->slab + (size_t)kd2 * n;
          dr.stride = rest_plane;
          dr.rows = rest_rows;
          dr.col = rows_core;
          oh.row0m = tmh * 128;
          if (rest_rows <= 64)
            RAMD_LAUNCH(k_delta_dma<64>, dim3(blocks2), dim3(512), shm_rest, st, v, row0, nrows, oh, dr);
          else
            RAMD_LAUNCH(k_delta_dma<128>, dim3(blocks2), dim3(512), shm_rest, st, v, row0, nrows, oh, dr);
          timing_end(st, evh);
          launch_gemm<true, true, ProbHoDelta>(st, ho_p, b->ho_slab, sh->H, sh->O, ho_nkt, ho_ks, T_OTHER);
          const size_t up_floats = n - half_floats, n4u = up_floats / 4, ho_n = (size_t)sh->H * sh->O;
          const unsigned fin_blocks = (unsigned)((n4u + 255) / 256) + (unsigned)((ho_n / 4 + 255) / 256);
          RAMD_LAUNCH(k_delta_finalize, dim3(fin_blocks), dim3(256), 0, st, b->ih_delta + half_floats, b->slab + half_floats,
                      n4u, n, kd2, accumulate, sh->H, sh->hidden_size, rows_core - tmh * 128, kd2 * tmh, dr.planes,
                      rest_plane, b->ho_delta, b->ho_slab, ho_n, ho_ks);
          g_delta_half_hook(g_delta_half_ctx, 1, half_floats, up_floats + ho_n);
        }
        return;
      }
      int ev = timing_begin(st, T_DELTA);
      if (rest_in) {
        DeltaRest dr;
        dr.planes = b->slab + (size_t)kd * n;
        dr.stride = rest_plane;
        dr.rows = rest_rows;
        dr.col = rows_core;
        if (rest_rows <= 64)
          RAMD_LAUNCH(k_delta_dma<64>, dim3(blocks), dim3(512), shm_rest, st, v, row0, nrows, o, dr);
        else
          RAMD_LAUNCH(k_delta_dma<128>, dim3(blocks), dim3(512), shm_rest, st, v, row0, nrows, o, dr);
      } else {
        DeltaRest dr = {};
        RAMD_LAUNCH(k_delta_dma<0>, dim3(blocks), dim3(512), shm, st, v, row0, nrows, o, dr);
      }
      timing_end(st, ev);
      ks_rest = 0;
      if (rest_in) {
        ks_rest = kd * o.tm;
        rest_base = b->slab + (size_t)kd * n;
        rest_stride = rest_plane;
        if (ho_paired) { /* the top layer's delta GEMM had been waiting for the pair launch */
          launch_gemm<true, true, ProbHoDelta>(st, ho_p, b->ho_slab, sh->H, sh->O, ho_nkt, ho_ks, T_OTHER);
          ho_paired = false;
          if (ho_finalize_after && !ranges) {
            ho_in_final = true; /* summed by the k_delta_finalize launch below */
          } else if (ho_finalize_after) {
            RAMD_LAUNCH(k_ho_delta_finalize, dim3((sh->H * sh->O + 255) / 256), dim3(256), 0, st, v,
                               b->ho_slab, ho_ks, accumulate, range_stride ? nullptr : ranges);
          }
        }
      } else if (rows_core < sh->I) {
        /* The rest rows' planes are compact ([ks_rest][I - rows_core][H], behind the core planes).
         * This GEMM is a few rows tall and K = S * D deep; measured at the north star its time does
         * not fall below 17 us for any K split from 16 to 48 (one workgroup per CU and ten K tiles
         * each, or three per CU and three tiles each: 0.87 us per 64 x 64 x 32 tile step and CU
         * either way), while every further plane costs the optimiser's sum: 16 it is. */
        int tmr = (rest_rows + BM - 1) / BM, tnr = (ncol - 1 + BN - 1) / BN;
        ks_rest = pick_ks(tmr * tnr, nkt, "RECUR_AMD_KS_DELTA_REST", (size_t)RAMD_MAX_REST_PLANES, 1);
        if (ks_rest > RAMD_MAX_REST_PLANES) ks_rest = RAMD_MAX_REST_PLANES;
        if (ks_rest > nkt) ks_rest = nkt;
        while (ks_rest > 1 && (size_t)ks * n + (size_t)ks_rest * rest_plane > b->slab_floats) ks_rest--;
        if (ks_rest < 1) ks_rest = 1;
        rest_base = b->slab + (size_t)ks * n;
        rest_stride = rest_plane;
        ProbDelta<true> p = {v, row0, nrows, rtiles};
        if (ho_paired) {
          int blocks_a, blocks_b;
          GemmOut oa = make_gemm_out(b->ho_slab, sh->H, sh->O, ho_nkt, ho_ks, 0, 0, 0, &blocks_a);
          GemmOut ob = make_gemm_out(rest_base - (size_t)rows_core * sh->H, sh->I, ncol, nkt, ks_rest, 1,
                                     sh->H, rows_core, &blocks_b);
          ob.zs = rest_stride;
          int ev2 = timing_begin(st, T_DELTA);
          RAMD_LAUNCH((k_gemm_pair<ProbHoDelta, ProbDelta<true>>), dim3(blocks_a + blocks_b),
                             dim3(256), 0, st, ho_p, oa, blocks_a, p, ob);
          timing_end(st, ev2);
          ho_paired = false;
          if (ho_finalize_after && !ranges) {
            ho_in_final = true; /* summed by the k_delta_finalize launch below */
          } else if (ho_finalize_after) {
            RAMD_LAUNCH(k_ho_delta_finalize, dim3((sh->H * sh->O + 255) / 256), dim3(256), 0, st, v,
                               b->ho_slab, ho_ks, accumulate, range_stride ? nullptr : ranges);
          }
        } else {
          launch_gemm<true, true, ProbDelta<true>>(st, p, rest_base - (size_t)rows_core * sh->H, sh->I, ncol,
                                                   nkt, ks_rest, T_DELTA, 1, sh->H, rows_core, rest_stride);
        }
      }
    } else if (big && b->uniform_idx >= 0) {
      ProbDelta<true> p = {v, row0, nrows, rtiles};
      launch_gemm2<ProbDelta<true>>(st, p, b->slab, sh->I, ncol, nkt, ks, T_DELTA, 1, sh->H);
    } else if (big) {
      ProbDelta<false> p = {v, row0, nrows, rtiles};
      launch_gemm2<ProbDelta<false>>(st, p, b->slab, sh->I, ncol, nkt, ks, T_DELTA, 1, sh->H);
    } else if (b->uniform_idx >= 0) {
      ProbDelta<true> p = {v, row0, nrows, rtiles};
      launch_gemm<true, true, ProbDelta<true>>(st, p, b->slab, sh->I, ncol, nkt, ks, T_DELTA, 1, sh->H);
    } else {
      ProbDelta<false> p = {v, row0, nrows, rtiles};
      launch_gemm<true, true, ProbDelta<false>>(st, p, b->slab, sh->I, ncol, nkt, ks, T_DELTA, 1, sh->H);
    }
    size_t n4 = n / 4;
    if (defer && !accumulate) { /* the optimiser launch that follows sums the slabs itself */
      defer->slab = b->slab;
      defer->n = n;
      defer->ks = ks;
      defer->H = sh->H;
      defer->hidden_size = sh->hidden_size;
      defer->rows_core = rows_core;
      defer->ks_rest = ks_rest;
      defer->rest = rest_base == b->slab ? b->slab + (size_t)rows_core * sh->H : rest_base;
      defer->rest_stride = rest_stride;
      defer->delta_out = b->ih_delta;
      return;
    }
    if (defer) defer->slab = nullptr;
    const size_t ho_n = (size_t)sh->H * sh->O;
    const unsigned fin_blocks = (unsigned)((n4 + 255) / 256) + (ho_in_final ? (unsigned)((ho_n / 4 + 255) / 256) : 0u);
    RAMD_LAUNCH(k_delta_finalize, dim3(fin_blocks), dim3(256), 0, st, b->ih_delta, b->slab, n4, n, ks,
                       accumulate, sh->H, sh->hidden_size, rows_core, ks_rest,
                       rest_base == b->slab ? b->slab + (size_t)rows_core * sh->H : rest_base, rest_stride, b->ho_delta,
                       ho_in_final ? b->ho_slab : nullptr, ho_n, ho_ks);
  }
}

// Up to 12 small word-wise copies in one launch: the per-net calls' traffic between a pinned
// host mailbox and the device arrays (either side may be the host: the mailbox is mapped).
struct SegCopy {
  unsigned *dst[12];
  const unsigned *src[12];
  unsigned n[12];
  int nseg;
};
__global__ __launch_bounds__(256) void k_segcopy(SegCopy sc) {
  const int g = blockIdx.y;
  if (g >= sc.nseg) return;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < sc.n[g]; i += gridDim.x * 256) sc.dst[g][i] = sc.src[g][i];
}
extern "C" void ramd_launch_noise_speculate(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b,
                                            int row0, int nrows, float noise) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_noise_speculate, dim3((nrows + 63) / 64), dim3(64), 0, st, v, row0, nrows, noise, b->noise_spec,
              (DevRng *)b->rng_spec);
}

extern "C" void ramd_launch_segcopy(ramd_stream_t st_, int nseg, void *const *dst,
                                    const void *const *src, const unsigned *nwords) {
  hipStream_t st = (hipStream_t)st_;
  SegCopy sc = {};
  unsigned most = 1;
  for (int g = 0; g < nseg; g++) {
    sc.dst[g] = (unsigned *)dst[g];
    sc.src[g] = (const unsigned *)src[g];
    sc.n[g] = nwords[g];
    if (nwords[g] > most) most = nwords[g];
  }
  sc.nseg = nseg;
  unsigned bx = (most + 255) / 256;
  if (bx > 16) bx = 16;
  RAMD_LAUNCH(k_segcopy, dim3(bx, nseg), dim3(256), 0, st, sc);
}

extern "C" void ramd_launch_err_writeback(ramd_stream_t st_, const RamdShape *sh,
                                          const RamdBuffers *b, int row0, int nrows) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  int nxp = (sh->I - sh->hidden_size + 3) & ~3;
  RAMD_LAUNCH(k_err_writeback, dim3(nrows), dim3(256), 0, st, v, row0, nxp);
}

extern "C" void ramd_launch_apply_multi(ramd_stream_t st_, int method, int nseg, float *const *w,
                                        const float *const *delta, float *const *m,
                                        float *const *aux, const size_t *n, const float *rate,
                                        float momentum, float mw, const float *rs,
                                        const RamdPendingDelta *pend) {
  hipStream_t st = (hipStream_t)st_;
  ApplySegs sg = {};
  if (pend && nseg >= 2) sg.pend = *pend;
  unsigned blocks = 0;
  for (int g = 0; g < 3; g++) {
    sg.first[g] = blocks;
    if (g < nseg) {
      sg.w[g] = w[g];
      sg.delta[g] = delta[g];
      sg.m[g] = m[g];
      sg.aux[g] = aux[g];
      sg.n4[g] = n[g] / 4;
      sg.rate[g] = rate[g];
      blocks += (unsigned)((sg.n4[g] + 255) / 256);
      if (g == 1) sg.end1 = blocks;
    }
  }
  sg.first[3] = blocks;
  for (int g = nseg; g < 3; g++) sg.first[g] = 0xffffffffu; /* never selected */
  dim3 gr(blocks), bl(256);
  int ev = timing_begin(st, T_APPLY);
  switch (method) {
  case 1: RAMD_LAUNCH(k_apply<1>, gr, bl, 0, st, sg, momentum, mw, rs); break;
  case 4: RAMD_LAUNCH(k_apply<4>, gr, bl, 0, st, sg, momentum, mw, rs); break;
  case 5: RAMD_LAUNCH(k_apply<5>, gr, bl, 0, st, sg, momentum, mw, rs); break;
  case 6: RAMD_LAUNCH(k_apply<6>, gr, bl, 0, st, sg, momentum, mw, rs); break;
  default: RAMD_LAUNCH(k_apply<0>, gr, bl, 0, st, sg, momentum, mw, rs); break;
  }
  timing_end(st, ev);
}

extern "C" void ramd_launch_apply(ramd_stream_t st_, int method, float *w, const float *delta,
                                  float *m, float *aux, size_t n, float rate, float momentum,
                                  float mw, const float *rs) {
  ramd_launch_apply_multi(st_, method, 1, &w, &delta, &m, &aux, &n, &rate, momentum, mw, rs, nullptr);
}

extern "C" void ramd_launch_fused_updates(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b, int row,
                                          float rate, float momentum, float mw, int apply_ih, const float *rs) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  const unsigned top_blocks = (unsigned)((sh->H * sh->O + 255) / 256);
  const unsigned ih_blocks = apply_ih ? (unsigned)(((size_t)sh->I * sh->H / 4 + 255) / 256) : 0u;
  RAMD_LAUNCH(k_fused_updates, dim3(top_blocks + ih_blocks), dim3(256), 0, st, v, row, rate, momentum, mw, top_blocks,
              rs);
}

extern "C" void ramd_launch_scale(ramd_stream_t st, float *a, size_t n, float scale) {
  RAMD_LAUNCH(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)st, a, n, scale);
}
extern "C" void ramd_launch_zero_small(ramd_stream_t st, float *a, size_t n) {
  RAMD_LAUNCH(k_zero_small, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)st, a, n);
}
extern "C" void ramd_launch_clamp(ramd_stream_t st, float *a, size_t n, float lo, float hi) {
  RAMD_LAUNCH(k_clamp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)st, a, n, lo, hi);
}
extern "C" void ramd_launch_add_at(ramd_stream_t st, float *a, size_t index, float v) {
  RAMD_LAUNCH(k_add_at, dim3(1), dim3(1), 0, (hipStream_t)st, a, index, v);
}
extern "C" void ramd_launch_tall_poppy(ramd_stream_t st, float *a, size_t n, float threshold,
                                       float scale, void *scratch) {
  const int parts = 256; /* scratch holds 256 BestAbs */
  RAMD_LAUNCH(k_absmax_part, dim3(parts), dim3(256), 0, (hipStream_t)st, a, n, (BestAbs *)scratch);
  RAMD_LAUNCH(k_tall_poppy, dim3(1), dim3(1), 0, (hipStream_t)st, a, (const BestAbs *)scratch, parts,
                     threshold, scale);
}
