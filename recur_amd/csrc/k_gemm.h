// k_gemm.h -- the generic split-K fp32 MFMA GEMM (64 x 64 and 128 x 128 tiles), its problem
// descriptions (forward, output layer, top-layer delta, extras, weight delta) and launchers.  Included
// by the translation units that instantiate it (kernels_forward.hip, kernels_bptt.hip).
#pragma once
#include "k_common.h"

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
