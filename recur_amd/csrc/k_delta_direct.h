// k_delta_direct.h -- the weight-delta GEMM WITHOUT a K split over workgroups and without LDS staging:
//
//   ih_delta[m][n] = sum over (step t, stream r) of X_t[r][m] * coef[t][r] * E_t[r][n]      (recur-nn.c:343-358, 734-739)
//
// with, optionally, rnn_apply_learning's momentum update (recur-nn.c:482-487) of the tile in the epilogue.
//
// Why another form (round 5).  k_delta_dma's 128 x 128 tiles are 64 at hidden 1024, so it splits K four ways over
// workgroups and leaves four planes (16.8 MB) that the optimiser launch reads back: 94 MB of traffic for 46 MB, a
// launch boundary behind 16.8 MB of dirty lines, and an update launch that moves 44 MB for 22.9.  Here a workgroup owns
// a 64 x 64 tile of the delta for ALL of K -- 16 x 16 = 256 tiles at hidden 1024, one per CU -- and K is split over
// the workgroup's WAVES instead: wave w takes every NW-th quad of K (four consecutive streams of one step) and keeps
// the whole 64 x 64 tile as 16 accumulators of v_mfma_f32_16x16x4_f32.  The waves' tiles meet in LDS once, at the
// end, and the workgroup that holds the finished tile is the only one that ever touches it: it can apply the update
// (weights, momentum) right there and store the delta once.  No planes, no second launch for the core rows.
//
// Operands go from memory straight into registers.  One global_load_dwordx4 per operand and K quad: lane l reads
// floats 4 (l % 16) .. + 3 of row k0 + l / 16 -- which IS the A (or B) fragment layout of v_mfma_f32_16x16x4_f32
// (lane l: index l % 16, k = l / 16) for FOUR interleaved 16-row groups at once: register j of the load holds rows
// (columns) 4 c + j, c = 0 .. 15.  The tile's rows therefore sit in the accumulators in a permuted order, which only
// the epilogue's addresses see: accumulator (i, jn), register r, lane l = element (4 (4 (l / 16) + r) + i, 4 (l % 16) + jn).
// Two loads feed 16 MFMAs (512 matrix-pipe cycles); a ring of P quads per wave is in flight (counted vmcnt, every
// address from a scalar base + a launch-invariant per-lane offset: no vector-ALU instruction in the loop at all).
// No barrier, no LDS read and no loader wave stands between the matrix pipe and its operands.
//
// The rows above the last whole 64-row tile (the input rows of a text net: 44 at the north star) need no K split
// either: the error quad a workgroup has in registers anyway is four interleaved 16-column groups, and the 16 workgroups of a column tile (mt = 0 .. 15) each
// take ONE (quarter of the rest rows mt / 4, column group mt % 4) pair for all of K: one more dword load and one more MFMA per 16,
// the same 6 % as sharing the rest tile out by K would cost, and no partial planes to add up afterwards.  The
// workgroup updates its 16 x 16 piece like its tile.  The columns outside 1 .. hidden_size (delta 0 there:
// recur-nn.c:334-337) are the first and last column tiles' business, the top layer's update (45 k weights) is shared
// out over all workgroups: with the momentum rule the launch leaves NOTHING for an optimiser launch.
//
// Coefficients: a stream's ih_scale while the step counted, 0 for steps past its break (which may hold inf: the
// product is v_mul_legacy_f32's, 0 * anything = 0).  In steady state every coefficient is exactly 1.0 (census: round 3);
// the control logic leaves a word that says so, and the loop without the multiplies is chosen per launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

typedef float dd_f4 __attribute__((ext_vector_type(4)));
typedef float dd_f4u __attribute__((ext_vector_type(4), aligned(4)));

struct DdArgs {
  const float *x;      /* history ring: slot 0, the call's first stream, column 0        [D][Scap][I]  */
  const float *e;      /* error planes: plane 0, the call's first stream, column 1       [D+1][Scap][I] */
  const float *coef;   /* [D][Scap], the call's first stream                                             */
  const int *n_exec;   /* [Scap] executed steps and ...                                                  */
  const float *ih_scale; /* ... ih_scale of the call's streams (from its first): all D and all 1.0 <=> every coefficient is 1.0 */
  float *w, *m, *delta; /* [I][H] weights, momentum, ih_delta                                          */
  size_t plane;        /* Scap * I: floats between slots / planes                                       */
  int I, H, Scap;
  int nrows, D, uidx;  /* streams of the call (a multiple of 4 NW), steps, ring position of step 0     */
  int tm, tn;          /* 64-row and 64-column tiles (columns start at column 1)                       */
  int rest;            /* I - 64 tm (a multiple of 4, <= 64 NPW; tm >= 16), 0: none                     */
  int hidden_size;     /* 64 tn                                                                         */
  int mode;            /* 0: delta = sum, 1: delta += sum, 2: delta = sum and the update (`method`)    */
  int method;          /* mode 2: 0 the momentum rule (recur-nn.c:482-487), 4 ADAGRAD (518-524: the momentum array is its accumulator) */
  float rate, momentum, mw;
  /* mode 2: the top layer's update rides along (recur-nn.c:653-676): ho_n4 float4s shared out over the workgroups */
  float *ho_w, *ho_m;
  const float *ho_delta;
  float *ho_delta_out; /* where the top layer's sums are to be stored as well (they came as a plane), or NULL */
  unsigned ho_n4;
  float ho_rate;
  /* K split over workgroups, for shapes whose 64 x 64 tiles are a fraction of the chip (hidden 512: 8 x 8): the grid is
   * ksplit times the tiles, part z takes iterations [z, z + 1) n_it / ksplit of every wave and stores (mode 0) its sums
   * in plane z, `kplane` floats behind the last -- the planes the optimiser's launch or k_delta_finalize sum */
  int ksplit;
  size_t kplane;
  int rgroups; /* row groups of the rest rows' pieces (dd_body) */
  int ho_ks;   /* the top layer's sums as ho_ks planes, ho_plane floats apart (a split-K GEMM left them): added here */
  size_t ho_plane;
  /* iterations of the FIRST wave of each SIMD's pair (waves w and w + 4 share SIMD w: dd_body), of the 2 n_it the pair
   * has per K part; 0: half */
  int fast_its;
};

/* COLD: outside the loop, with five wait states in front.  An SGPR base that a vector-ALU instruction has just written
 * (v_readlane: hipcc keeps spilled SGPRs in VGPR lanes) may be read by a memory instruction five wait states later at the
 * earliest; hipcc pads its own memory instructions and does not look into an asm block.  The loop's bases come from the
 * scalar ALU (no hazard) and tools/isa_lint_async_loads.py checks every build for the pattern. */
template <bool COLD = false> __device__ __forceinline__ dd_f4 dd_load4(const void *sbase, unsigned voff) {
  dd_f4 r;
  if constexpr (COLD) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
template <bool COLD = false> __device__ __forceinline__ float dd_load1(const void *sbase, unsigned voff) {
  float r;
  if constexpr (COLD) asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
typedef int dd_i4 __attribute__((ext_vector_type(4)));
constexpr int DD_FLAG_LOADS = 8; /* 4 x 256 streams of n_exec, of ih_scale */
__device__ __forceinline__ dd_i4 dd_load4i(const void *sbase, unsigned voff) {
  dd_i4 r;
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
/* wait until at most N younger loads are outstanding: the flag loads (the wave's first) have landed */
template <int N> __device__ __forceinline__ void dd_flag_wait(dd_i4 (&n)[DD_FLAG_LOADS / 2], dd_f4 (&s)[DD_FLAG_LOADS / 2]) {
  static_assert(DD_FLAG_LOADS == 8, "operand list");
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]) : "n"(N));
}
template <int N> __device__ __forceinline__ void dd_wait3(dd_f4 &a, dd_f4 &b, float &c) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}
template <int N> __device__ __forceinline__ void dd_wait4(dd_f4 &a, dd_f4 &b, float &c, float &d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N> __device__ __forceinline__ void dd_wait5(dd_f4 &a, dd_f4 &b, float &c, float &d, float &e) {
  asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N));
}
/* Wait for EVERY load of the wave, with the ring's registers as operands: hipcc does not know that the loads of the
 * inline asm are asynchronous, and a register of the ring that it considers dead it hands to something else -- which
 * the load then overwrites when it lands (seen: the accumulators, moved into ring registers behind the loop, took
 * the values of the surplus loads of the last round).  As operands of this wait the registers are alive until it. */
template <int P> __device__ __forceinline__ void dd_drain(const dd_f4 (&a)[P], const dd_f4 (&b)[P], const float (&c)[2][P], const float (&d)[P]) {
  static_assert(P == 5, "operand list");
  /* (inputs only: as tied operands hipcc copied the registers -- loads outstanding -- in front of the wait) */
  asm volatile("s_waitcnt vmcnt(0)"
               :
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]),
                 "v"(c[0][0]), "v"(c[0][1]), "v"(c[0][2]), "v"(c[0][3]), "v"(c[0][4]), "v"(c[1][0]), "v"(c[1][1]), "v"(c[1][2]),
                 "v"(c[1][3]), "v"(c[1][4]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4])
               : "memory");
}
template <int... Is, class F>
__device__ __forceinline__ void dd_static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void dd_static_for(F &&f) {
  dd_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int DD_LD = 64; /* floats per row of a wave's tile in LDS */
/* the epilogue's stores of the tile: 0 plain; 1: the delta and the momentum (nobody reads them before the next
 * generation's weight-delta launch) as non-temporal stores, 2: the weights too -- an experiment about what the launch leaves
 * for the end-of-kernel write-back (profiles/NOTES_r06.md) */
#ifndef DD_NT_STORES
#define DD_NT_STORES 1
#endif
template <int LEVEL> __device__ __forceinline__ void dd_store4(float *p, const float __attribute__((ext_vector_type(4))) & v) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  if constexpr (DD_NT_STORES >= LEVEL) __builtin_nontemporal_store(v, reinterpret_cast<f4u *>(p));
  else *reinterpret_cast<f4u *>(p) = v;
}

/* NW: waves per workgroup (8: two per SIMD -- while one waits for operands or sits in its epilogue the other has the
 * matrix pipe; with 4 hipcc keeps the accumulators in AGPRs and shuffles ring registers through them between an
 * asynchronous load and its wait); P: K quads in flight per wave (the ring); NPW: pieces of the rest rows per
 * workgroup (1: up to 64 rest rows, 2: up to 128). */
#ifdef PC_STAMPS /* development builds only (tools/mkabl.sh -DPC_STAMPS, tools/gpu_delta_direct_stamps.py) */
__device__ unsigned long long g_ddir_stamps[4][8];
__device__ unsigned long long g_ddir_wave[4][8]; /* loop done, per wave */
__device__ unsigned long long g_ddir_clk[4][4];  /* the loop by both clocks: s_memrealtime, s_memtime at its start and end */
#define DDIR_CLOCKS(i) do { if ((blockIdx.x & 63) == 0 && threadIdx.x == 0) { g_ddir_clk[blockIdx.x >> 6][i] = __builtin_amdgcn_s_memrealtime(); g_ddir_clk[blockIdx.x >> 6][(i) + 1] = __builtin_amdgcn_s_memtime(); } } while (0)
#define DDIR_STAMP(i) do { if ((blockIdx.x & 63) == 0 && threadIdx.x == 0) g_ddir_stamps[blockIdx.x >> 6][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DDIR_STAMP(i) do { } while (0)
#define DDIR_CLOCKS(i) do { } while (0)
#endif
struct DdNoPre {
  __device__ __forceinline__ void operator()() const {}
};
/* PRE: work of the caller's that runs once the ring's first operands have been requested and before anything waits for
 * them (the fused text step's top-layer delta and update: kernels_bptt.hip) */
template <int NW, int P, int NPW, class PRE = DdNoPre>
__device__ __forceinline__ void dd_body(const DdArgs &a, float *lds, PRE pre = PRE{}) {
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  /* workgroup -> tile.  Workgroups are dealt to the XCDs in turn (speed only): XCD x takes a block of the tile
   * grid, so that its L2 sees tm / 4 x tn / 2 of the operands' columns */
  DDIR_STAMP(0);
  const int ksp = a.ksplit > 1 ? a.ksplit : 1, per_part = gridDim.x / ksp;
  const int kz = (int)blockIdx.x / per_part;
  const int L = (int)blockIdx.x - kz * per_part, xcd = L & 7, q = L >> 3;
  float *const dlt = a.delta + (size_t)kz * a.kplane;
  int mt, nt;
  if ((a.tm & 3) == 0 && (a.tn & 1) == 0) {
    const int bm = a.tm >> 2, bn = a.tn >> 1; /* the XCD's block: bm x bn tiles */
    mt = (xcd & 3) * bm + q % bm;
    nt = (xcd >> 2) * bn + q / bm;
  } else {
    mt = L % a.tm;
    nt = L / a.tm;
  }
  if (nt >= a.tn) return;
  const int m0 = 64 * mt, n0 = 64 * nt; /* (a.e, a.w ... already point at column 1) */
  const int I = a.I;
  const int QPS = a.nrows >> 2;          /* K quads per step */
  /* K over the waves: waves w and w + 4 share SIMD w (w < 4) and take every fourth quad of K from quad w on -- the pair's
   * sequence --, the first wave items [0, nf) of it, the second [nf, 2 n_it).  NOT half each where the loop is long: the
   * SIMD issues from its older wave first, so with equal shares wave w ran at 3.0 iterations per us and wave w + 4 at 1.6
   * beside it, the first was done at 59 us, and the second finished ALONE, at 70 % of the pipe's rate (its five quads in
   * flight do not cover the memory latency by themselves), at 83 us -- stamps of round 5, profiles/NOTES_r05.md section
   * 10.  With nf = 5/8 of the pair's items both are done together. */
  constexpr int ST = NW / 2;
  const int n_pair = 2 * (a.D * (QPS / NW) / ksp);           /* the pair's items in this workgroup's part of K */
  const int nf = a.fast_its > 0 ? a.fast_its : n_pair / 2;
  const bool first_of_pair = wv < ST;
  const int n_it = first_of_pair ? nf : n_pair - nf;         /* this wave's iterations */
  const int q0 = (wv & (ST - 1)) + ST * (kz * n_pair + (first_of_pair ? 0 : nf));
  const int g_t0 = q0 / QPS, g_w0 = q0 - g_t0 * QPS;
  const unsigned voff = (unsigned)(((size_t)(lane >> 4) * I + (lane & 15) * 4) * sizeof(float));
  /* the rest rows, for the first 16 row tiles of a column tile: NPW pieces each, piece pi = NPW mt + p = (row group
   * pi / 4, column group pi % 4) of 4 NPW row groups x 4 column groups.  A row group is rg = ceil(rest / 4 NPW)
   * CONSECUTIVE rows (64 tm + ri rg + c, c = lane % 16 -- one or two cache lines per history row; lanes past the
   * group's last row repeat it), a column group the columns n0 + 4 c + rj: register rj of the error quad.  The
   * load's base is the A operand's: the column distance is in the per-lane offset.  (Workgroups without a share
   * load and multiply all the same -- a branch would end the scheduling region --, their result is never stored.) */
  /* (a.rgroups row groups: 4 NPW where a column tile has 16 row tiles to share the 16 NPW pieces out over; 4 with two
   * pieces per workgroup where it has only 8 -- hidden 512 --, for up to 64 rest rows) */
  const int RG = a.rgroups;
  const bool has_rest = a.rest > 0 && NPW * mt < 4 * RG;
  const int rg = (a.rest + RG - 1) / RG;
  int ri[NPW], rj[NPW], rrows[NPW]; /* (rrows: the group's rows that exist) */
  unsigned voff_r[NPW];
  unsigned long long sel_lo[NPW], sel_hi[NPW]; /* register rj of a quad, by two wave-uniform masks (as ?: chains hipcc makes branches of it) */
#pragma unroll
  for (int p = 0; p < NPW; p++) {
    const int pi = has_rest ? NPW * mt + p : 0;
    ri[p] = pi >> 2;
    rj[p] = pi & 3;
    int left = a.rest - ri[p] * rg;
    rrows[p] = has_rest ? (left < 0 ? 0 : left < rg ? left : rg) : 0;
    const int c = lane & 15;
    const int rcol = rrows[p] > 0 ? 64 * a.tm - m0 + ri[p] * rg + (c < rrows[p] ? c : rrows[p] - 1) : 0;
    voff_r[p] = (unsigned)(((size_t)(lane >> 4) * I + rcol) * sizeof(float));
    sel_lo[p] = (rj[p] & 1) ? ~0ull : 0ull;
    sel_hi[p] = (rj[p] & 2) ? ~0ull : 0ull;
  }
  const unsigned voff_c = (unsigned)((lane >> 4) * sizeof(float));

  constexpr int CPT = 1024 / (64 * NW); /* float4 chunks of the tile per thread */
  const bool upd = a.mode == 2;
  /* (requested at the start of the launch and kept in registers instead, the epilogue's weights and momentum changed
   * nothing: 88.9 against 88.4 us per launch -- the epilogue's round trip hides behind the other workgroups' loops) */

  /* Is every coefficient of the call exactly 1.0?  They are when every stream ran all D steps unclipped: n_exec ==
   * D and ih_scale == 1.0 (k_extras.h: bptt_control_wave; an inactive stream has n_exec 0).  DD_FLAG_LOADS loads per
   * wave, four streams per lane each (a call of up to 1024 streams; lanes past the end repeat the last quad),
   * requested FIRST and waited for behind the ring's first operands: the answer costs no round trip of its own. */
  dd_f4 fl_s[DD_FLAG_LOADS / 2];
  dd_i4 fl_n[DD_FLAG_LOADS / 2];
#pragma unroll
  for (int u = 0; u < DD_FLAG_LOADS / 2; u++) {
    int r4 = 4 * (64 * u + lane);
    r4 = r4 < a.nrows ? r4 : a.nrows - 4;
    fl_n[u] = dd_load4i(a.n_exec, (unsigned)(r4 * sizeof(int)));
    fl_s[u] = dd_load4<true>(a.ih_scale, (unsigned)(r4 * sizeof(float)));
  }

  dd_f4 ra[P], re[P];
  float rr[2][P], rcf[P]; /* (rr[1]: the second piece's loads, NPW == 2) */
  /* The operands of this wave's iterations, in order: quad (step t, streams 4 within .. + 3), from quad q0 on in
   * steps of NW / 2.  The generator is scalar and branch-free (counters, selects and multiplies on the scalar ALU: it is
   * scheduled into the shadows of the MFMAs; a division would go through the vector ALU, a branch would end the
   * scheduling region).  Past the last iteration it keeps going over valid memory (step clamped): those loads only
   * keep the counted waits exact. */
  int g_t = g_t0, g_within = g_w0;
  const float *g_xb, *g_eb, *g_cb;
  auto advance = [&]() {
    const int t = g_t < a.D ? g_t : a.D - 1;
    int slot = a.uidx - t;
    slot = slot < 0 ? slot + a.D : slot;
    const size_t ro = (size_t)(g_within << 2) * I;
    g_xb = a.x + (size_t)slot * a.plane + ro + m0;
    g_eb = a.e + (size_t)t * a.plane + ro + n0;
    g_cb = a.coef + (size_t)t * a.Scap + (g_within << 2);
    g_within += ST;
    const int wrap = g_within >= QPS ? 1 : 0;
    g_within -= wrap ? QPS : 0;
    g_t += wrap;
  };
  dd_f4 acc[4][4], racc[NPW];
#pragma unroll
  for (int p = 0; p < NPW; p++) racc[p] = dd_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < P; j++) rr[1][j] = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = dd_f4{0.f, 0.f, 0.f, 0.f};
  /* the ring's first tenants, requested before the answer about the coefficients is there (their loop without
   * multiplies is the usual one; the other one starts over: below) */
  dd_static_for<P>([&](auto JC) {
    constexpr int j = decltype(JC)::value;
    advance();
    ra[j] = dd_load4<true>(g_xb, voff);
    re[j] = dd_load4<true>(g_eb, voff);
#pragma unroll
    for (int p = 0; p < NPW; p++) rr[p][j] = dd_load1<true>(g_xb, voff_r[p]);
  });
  pre(); /* (its own loads are younger than the ring's: every counted wait below then waits for them too, never for less) */
  int ones;
  {
    DDIR_STAMP(1);
    dd_flag_wait<(2 + NPW) * P>(fl_n, fl_s);
    DDIR_STAMP(2);
    DDIR_CLOCKS(0);
    bool ok = true;
#pragma unroll
    for (int u = 0; u < DD_FLAG_LOADS / 2; u++)
#pragma unroll
      for (int k = 0; k < 4; k++) ok = ok && fl_n[u][k] == a.D && fl_s[u][k] == 1.0f;
    ones = __builtin_amdgcn_readfirstlane(__all(ok) ? 1 : 0);
  }
  /* one round of P iterations (a wave-uniform choice of body per launch, not per iteration) */
  auto round = [&](auto ONESC) {
    constexpr bool ONES = decltype(ONESC)::value;
    dd_static_for<P>([&](auto JC) {
      constexpr int j = decltype(JC)::value;
      constexpr int PER = 2 + NPW + (ONES ? 0 : 1); /* loads per iteration */
      if constexpr (ONES && NPW == 1) dd_wait3<PER * (P - 1)>(ra[j], re[j], rr[0][j]);
      else if constexpr (ONES) dd_wait4<PER * (P - 1)>(ra[j], re[j], rr[0][j], rr[1][j]);
      else if constexpr (NPW == 1) dd_wait4<PER * (P - 1)>(ra[j], re[j], rr[0][j], rcf[j]);
      else dd_wait5<PER * (P - 1)>(ra[j], re[j], rr[0][j], rr[1][j], rcf[j]);
      dd_f4 fa = ra[j], fe = re[j];
      float fr[NPW];
#pragma unroll
      for (int p = 0; p < NPW; p++) fr[p] = rr[p][j];
      if constexpr (!ONES) {
        /* v_mul_legacy_f32: 0 * x is 0 for ANY x (a step past the break may hold inf), otherwise the IEEE product */
        asm volatile("v_mul_legacy_f32 %0, %4, %0\n\tv_mul_legacy_f32 %1, %4, %1\n\t"
                     "v_mul_legacy_f32 %2, %4, %2\n\tv_mul_legacy_f32 %3, %4, %3\n\ts_nop 1"
                     : "+v"(fe[0]), "+v"(fe[1]), "+v"(fe[2]), "+v"(fe[3])
                     : "v"(rcf[j]));
      }
      __builtin_amdgcn_sched_barrier(0);
      advance(); /* the slot's next tenant: its addresses, between the MFMAs */
      float fsel[NPW];
#pragma unroll
      for (int p = 0; p < NPW; p++) {
        float s01, s23;
        asm("v_cndmask_b32 %0, %2, %3, %6\n\tv_cndmask_b32 %1, %4, %5, %6"
            : "=&v"(s01), "=&v"(s23) : "v"(fe[0]), "v"(fe[1]), "v"(fe[2]), "v"(fe[3]), "s"(sel_lo[p]));
        /* (s_nop 1: two wait states between a vector-ALU write and the MFMA that reads the register -- hipcc puts ONE
         * behind an inline-asm output, and the second piece's MFMA directly behind its select then read the old value:
         * tools/isa_lint_async_loads.py checks it) */
        asm("v_cndmask_b32 %0, %1, %2, %3\n\ts_nop 1" : "=v"(fsel[p]) : "v"(s01), "v"(s23), "s"(sel_hi[p]));
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jn = 0; jn < 4; jn++)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fe[jn], acc[i][jn], 0, 0, 0);
#pragma unroll
      for (int p = 0; p < NPW; p++) racc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[p], fsel[p], racc[p], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 16 + NPW; g++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* one MFMA */
        __builtin_amdgcn_sched_group_barrier(0x004, 4, 0); /* up to four scalar-ALU instructions */
      }
      __builtin_amdgcn_sched_barrier(0);
      ra[j] = dd_load4(g_xb, voff);
      re[j] = dd_load4(g_eb, voff);
#pragma unroll
      for (int p = 0; p < NPW; p++) rr[p][j] = dd_load1(g_xb, voff_r[p]);
      if constexpr (!ONES) rcf[j] = dd_load1(g_cb, voff_c);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  if (ones) {
    for (int i0 = 0; i0 < n_it; i0 += P) round(std::true_type{});
    DDIR_STAMP(3);
    DDIR_CLOCKS(2);
#ifdef PC_STAMPS
    if ((blockIdx.x & 63) == 0 && lane == 0) g_ddir_wave[blockIdx.x >> 6][wv] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int j = 0; j < P; j++) rcf[j] = 0.0f;
    dd_drain<P>(ra, re, rr, rcf); /* the surplus loads (a wait per path: one behind the join costs copies of the ring) */
  } else {
    /* the rare launch (a stream was soft-clipped or stopped early): drop the ring and start over with the
     * coefficients as a fourth load per iteration -- one round trip, instead of a first round with waits of its
     * own, whose code hipcc gave register copies between a load and its wait */
#pragma unroll
    for (int j = 0; j < P; j++) rcf[j] = 0.0f;
    dd_drain<P>(ra, re, rr, rcf);
    g_t = g_t0;
    g_within = g_w0;
    dd_static_for<P>([&](auto JC) {
      constexpr int j = decltype(JC)::value;
      advance();
      ra[j] = dd_load4<true>(g_xb, voff);
      re[j] = dd_load4<true>(g_eb, voff);
#pragma unroll
      for (int p = 0; p < NPW; p++) rr[p][j] = dd_load1<true>(g_xb, voff_r[p]);
      rcf[j] = dd_load1<true>(g_cb, voff_c);
    });
    for (int i0 = 0; i0 < n_it; i0 += P) round(std::false_type{});
    dd_drain<P>(ra, re, rr, rcf);
  }

  // ---- the waves' tiles meet in LDS; every thread then owns float4s of the finished tile
  {
    float *base = lds + (size_t)wv * 64 * DD_LD;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * (lane >> 4) + 4 * r + i;
        *reinterpret_cast<dd_f4 *>(base + row * DD_LD + 4 * (lane & 15)) = dd_f4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      }
    /* the rest pieces: [NPW][NW][16 rows c][16 columns c'] behind the tiles; register r of lane l is (4 (l / 16) + r, l % 16) */
#pragma unroll
    for (int p = 0; p < NPW; p++) {
      float *rb = lds + (size_t)NW * 64 * DD_LD + (p * NW + wv) * 256;
#pragma unroll
      for (int r = 0; r < 4; r++) rb[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = racc[p][r];
    }
  }
  DDIR_STAMP(4);
  __syncthreads();
  DDIR_STAMP(5);
  const bool adagrad = a.method == 4; /* (a kernel argument: uniform) */
  auto update4 = [&](dd_f4 &W, dd_f4 &M, const dd_f4 &d, float rate) { /* recur-nn.c:482-487, or 518-524 (k_apply<4>'s arithmetic) */
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (adagrad) {
        const float acc2 = M[k] + d[k] * d[k];
        W[k] += d[k] * rate / sqrtf(acc2);
        M[k] = acc2;
      } else {
        const float t = d[k] * rate, mm = M[k];
        W[k] += t + mm * a.mw;
        M[k] = (mm + t) * a.momentum;
      }
    }
  };
#pragma unroll
  for (int u = 0; u < CPT; u++) {
    const int ch = u * 64 * NW + threadIdx.x; /* chunk ch: row ch / 16, columns 4 (ch % 16) .. + 3, the waves in order */
    const float *p = lds + (ch >> 4) * DD_LD + 4 * (ch & 15);
    dd_f4 s = *reinterpret_cast<const dd_f4 *>(p);
#pragma unroll
    for (int w2 = 1; w2 < NW; w2++) s += *reinterpret_cast<const dd_f4 *>(p + (size_t)w2 * 64 * DD_LD);
    const size_t off = (size_t)(m0 + (ch >> 4)) * a.H + n0 + 4 * (ch & 15);
    if (a.mode == 1) s += *reinterpret_cast<const dd_f4u *>(dlt + off);
    dd_store4<1>(dlt + off, s);
    if (upd) {
      dd_f4 W = *reinterpret_cast<const dd_f4u *>(a.w + off), M = *reinterpret_cast<const dd_f4u *>(a.m + off);
      update4(W, M, s, a.rate);
      dd_store4<2>(a.w + off, W);
      dd_store4<1>(a.m + off, M);
    }
  }
  auto update1 = [&](size_t off, float d, float rate, float *w, float *m) {
    if (adagrad) {
      const float acc2 = m[off] + d * d;
      w[off] += d * rate / sqrtf(acc2);
      m[off] = acc2;
      return;
    }
    const float t = d * rate, mm = m[off];
    w[off] += t + mm * a.mw;
    m[off] = (mm + t) * a.momentum;
  };
  /* the workgroup's 16 x 16 pieces of the rest rows: thread (piece, c, c') */
  if (has_rest && threadIdx.x < 256 * NPW) {
    const int p = threadIdx.x >> 8, t8 = threadIdx.x & 255;
    const int c = t8 >> 4, c2 = t8 & 15;
    const int pri = NPW == 1 ? ri[0] : (p ? ri[NPW - 1] : ri[0]), prj = NPW == 1 ? rj[0] : (p ? rj[NPW - 1] : rj[0]);
    const int prows = NPW == 1 ? rrows[0] : (p ? rrows[NPW - 1] : rrows[0]);
    if (c < prows) {
      const float *q = lds + (size_t)NW * 64 * DD_LD + (size_t)p * NW * 256 + t8;
      float d = q[0];
#pragma unroll
      for (int w2 = 1; w2 < NW; w2++) d += q[w2 * 256];
      const size_t off = (size_t)(64 * a.tm + pri * rg + c) * a.H + n0 + 4 * c2 + prj;
      if (a.mode == 1) d += dlt[off];
      dlt[off] = d;
      if (upd) update1(off, d, a.rate, a.w, a.m);
    }
  }
  /* The columns outside 1 .. hidden_size (a.w[-1] is column 0): the delta is 0 there, the update is the momentum's own
   * (recur-nn.c:482-487 with t = 0).  Column 0 by the first column tile's workgroups, the columns above hidden_size
   * by the last one's, for their 64 rows; row tile 0 also takes those columns of the rest rows. */
  const int hi_cols = a.H - a.hidden_size - 1;
  if (a.mode != 1 && (nt == 0 || nt == a.tn - 1)) {
    const int ncol = nt == 0 && nt == a.tn - 1 ? 1 + hi_cols : nt == 0 ? 1 : hi_cols;
    const int nrow = 64 + (mt == 0 ? a.rest : 0);
    for (int e = threadIdx.x; e < nrow * ncol; e += 64 * NW) {
      const int rr_ = e / ncol, cc = e - rr_ * ncol;
      const int row = rr_ < 64 ? m0 + rr_ : 64 * a.tm + rr_ - 64;
      const int col = (nt == 0 && cc == 0) ? -1 : a.hidden_size + (nt == 0 ? cc - 1 : cc);
      const ptrdiff_t off = (ptrdiff_t)row * a.H + col;
      dlt[off] = 0.0f;
      if (upd && adagrad) { /* the rule with a delta of 0, as the optimiser's launch computes it (0 / sqrt(0) where nothing was ever added) */
        a.w[off] += 0.0f * a.rate / sqrtf(a.m[off]);
      } else if (upd) {
        const float mm = a.m[off];
        a.w[off] += mm * a.mw;
        a.m[off] = mm * a.momentum;
      }
    }
  }
  /* the top layer's update, shared out over the workgroups */
  if (upd && a.ho_n4) {
    /* (a loop: a wide top layer -- H * O / 4 above 64 NW float4s per workgroup -- gives a workgroup more than one
     * float4 per thread; without it the tail of every share kept its old weights and fuse_done hid the omission) */
    const unsigned per = (a.ho_n4 + gridDim.x - 1) / gridDim.x;
    for (unsigned k = threadIdx.x; k < per; k += 64 * NW) {
      const unsigned idx = blockIdx.x * per + k;
      if (idx >= a.ho_n4) break;
      const size_t off = 4 * (size_t)idx;
      dd_f4 d = *reinterpret_cast<const dd_f4 *>(a.ho_delta + off);
      for (int z = 1; z < a.ho_ks; z++) {
        const dd_f4 t = *reinterpret_cast<const dd_f4 *>(a.ho_delta + (size_t)z * a.ho_plane + off);
        d[0] += t[0]; d[1] += t[1]; d[2] += t[2]; d[3] += t[3];
      }
      if (a.ho_delta_out) *reinterpret_cast<dd_f4 *>(a.ho_delta_out + off) = d;
      dd_f4 W = *reinterpret_cast<const dd_f4 *>(a.ho_w + off), M = *reinterpret_cast<const dd_f4 *>(a.ho_m + off);
      update4(W, M, d, a.ho_rate);
      *reinterpret_cast<dd_f4 *>(a.ho_w + off) = W;
      *reinterpret_cast<dd_f4 *>(a.ho_m + off) = M;
    }
  }
  DDIR_STAMP(6);
}

/* (DDIR_STAMP(6) at the end of dd_body) */
constexpr int dd_lds_bytes(int NW, int NPW = 1) { return (NW * 64 * DD_LD + NPW * NW * 256) * 4; }

/* (one workgroup per CU: NW / 4 waves per SIMD, with the registers that leaves each -- told to hipcc, which otherwise
 * aims at a higher occupancy and parks ring registers in AGPRs between an asynchronous load and its wait) */
#ifndef DD_BND_MARK /* (development builds, -DBND_STAMPS: kernels_bptt.hip marks this launch's workgroups too) */
#define DD_BND_MARK(which) do { } while (0)
#endif
template <int NW, int P, int NPW = 1>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void k_delta_direct(DdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dd_lds[];
  __builtin_amdgcn_s_setprio(2);
  DD_BND_MARK(0);
  dd_body<NW, P, NPW>(a, dd_lds);
  DD_BND_MARK(1);
}
