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
// The rows above the last whole 64-row tile (the input rows of a text net: 44 at the north star) ride along as in
// k_delta_dma: the tm workgroups of a column tile each take 1 / tm of K for them, after their own tile, into a second
// set of accumulators (the error quad is fetched again: it is in the L2, usually in the L1), and leave tm partial
// planes for the small launch that updates those rows, the columns outside 1 .. hidden_size and the top layer
// (k_apply_edges).
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
  float *rest_planes;  /* [tm][rest][H] partial sums of the rows from 64 tm on                          */
  size_t plane;        /* Scap * I: floats between slots / planes                                       */
  size_t rest_stride;  /* rest * H                                                                      */
  int I, H, Scap;
  int nrows, D, uidx;  /* streams of the call (a multiple of 4 NW), steps, ring position of step 0     */
  int tm, tn;          /* 64-row and 64-column tiles (columns start at column 1)                       */
  int rest;            /* I - 64 tm (a multiple of 4, <= 64), 0: none                                   */
  int mode;            /* 0: delta = sum, 1: delta += sum, 2: delta = sum and the update (method 0)     */
  float rate, momentum, mw;
};

__device__ __forceinline__ dd_f4 dd_load4(const void *sbase, unsigned voff) {
  dd_f4 r;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
__device__ __forceinline__ float dd_load1(const void *sbase, unsigned voff) {
  float r;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
typedef int dd_i4 __attribute__((ext_vector_type(4)));
constexpr int DD_FLAG_LOADS = 8; /* 4 x 256 streams of n_exec, of ih_scale */
__device__ __forceinline__ dd_i4 dd_load4i(const void *sbase, unsigned voff) {
  dd_i4 r;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
/* wait until at most N younger loads are outstanding: the flag loads (the wave's first) have landed */
template <int N> __device__ __forceinline__ void dd_flag_wait(dd_i4 (&n)[DD_FLAG_LOADS / 2], dd_f4 (&s)[DD_FLAG_LOADS / 2]) {
  static_assert(DD_FLAG_LOADS == 8, "operand list");
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]) : "n"(N));
}
template <int N> __device__ __forceinline__ void dd_wait2(dd_f4 &a, dd_f4 &b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N> __device__ __forceinline__ void dd_wait3(dd_f4 &a, dd_f4 &b, float &c) {
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}
template <int... Is, class F>
__device__ __forceinline__ void dd_static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void dd_static_for(F &&f) {
  dd_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

constexpr int DD_LD = 64; /* floats per row of a wave's tile in LDS */
constexpr int dd_lds_bytes(int NW) { return NW * 64 * DD_LD * 4; }

/* NW: waves per workgroup (8: two per SIMD -- while one waits for operands or sits in its epilogue the other has the
 * matrix pipe; with 4 hipcc keeps the accumulators in AGPRs and shuffles ring registers through them between an
 * asynchronous load and its wait); P: K quads in flight per wave (the ring). */
template <int NW, int P>
__device__ __forceinline__ void dd_body(const DdArgs &a, float *lds) {
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  /* workgroup -> tile.  Workgroups are dealt to the XCDs in turn (speed only): XCD x takes a block of the tile
   * grid, so that its L2 sees tm / 4 x tn / 2 of the operands' columns */
  const int L = blockIdx.x, xcd = L & 7, q = L >> 3;
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
  const int IPT = QPS / NW;              /* this wave's quads per step */
  const int n_it = a.D * IPT;            /* its iterations over its own tile */
  const int NQ = a.D * QPS;
  const int rq = a.rest ? NQ / a.tm : 0; /* quads of the rest tile that this workgroup takes */
  const int n_r = rq / NW;
  const int total = n_it + n_r;
  const unsigned voff = (unsigned)(((size_t)(lane >> 4) * I + (lane & 15) * 4) * sizeof(float));
  /* the rest rows: chunks past the last real row repeat the last one (nothing is read beyond a history row) */
  const int rc = a.rest ? ((lane & 15) * 4 < a.rest ? (lane & 15) * 4 : a.rest - 4) : 0;
  const unsigned voff_r = (unsigned)(((size_t)(lane >> 4) * I + rc) * sizeof(float));
  const unsigned voff_c = (unsigned)((lane >> 4) * sizeof(float));

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
    fl_s[u] = dd_load4(a.ih_scale, (unsigned)(r4 * sizeof(float)));
  }

  dd_f4 ra[P], re[P];
  float rcf[P];
  /* The operands of this wave's iterations, in order: quad (step t, streams 4 within .. + 3) advances by NW per
   * iteration -- its own tile: from quad wv over all of K; then the rest tile's share: from quad mt * rq + wv.  The
   * generator is scalar and branch-free (counters, selects and multiplies on the scalar ALU: it is scheduled into
   * the shadows of the MFMAs; a division would go through the vector ALU, a branch would end the scheduling
   * region).  Past the last iteration it keeps going over valid memory (step clamped): those loads only keep the
   * counted waits exact. */
  const int r_q0 = mt * rq + wv, r_t0 = r_q0 / QPS, r_w0 = r_q0 - r_t0 * QPS; /* the rest share's first quad */
  int g_t = 0, g_within = wv, g_left = n_it, g_rest = 0;
  const float *g_xb, *g_eb, *g_cb;
  unsigned g_vo;
  auto advance = [&]() {
    const int t = g_t < a.D ? g_t : a.D - 1;
    int slot = a.uidx - t;
    slot = slot < 0 ? slot + a.D : slot;
    const size_t ro = (size_t)(g_within << 2) * I;
    g_xb = a.x + (size_t)slot * a.plane + ro + (g_rest ? 64 * a.tm : m0);
    g_eb = a.e + (size_t)t * a.plane + ro + n0;
    g_cb = a.coef + (size_t)t * a.Scap + (g_within << 2);
    g_vo = g_rest ? voff_r : voff;
    g_within += NW;
    const int wrap = g_within >= QPS ? 1 : 0;
    g_within -= wrap ? QPS : 0;
    g_t += wrap;
    g_left -= 1;
    const int sw = (g_left == 0 && g_rest == 0 && n_r > 0) ? 1 : 0; /* on to the rest tile's share */
    g_t = sw ? r_t0 : g_t;
    g_within = sw ? r_w0 : g_within;
    g_left = sw ? n_r : g_left;
    g_rest = sw ? 1 : g_rest;
  };
  /* ONE set of accumulators: the tile's, then (spilled to the wave's own part of LDS in between) the rest share's */
  dd_f4 acc[4][4];
  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = dd_f4{0.f, 0.f, 0.f, 0.f};
  };
  clear();
  /* the ring's first tenants: the operands now, the coefficients (a third load per iteration) once they are known to
   * be needed -- the coefficient of ring slot j then sits behind ALL P operand pairs in the queue, which the first
   * round's waits account for (FIRST) */
  const float *cb0[P];
  dd_static_for<P>([&](auto JC) {
    constexpr int j = decltype(JC)::value;
    advance();
    ra[j] = dd_load4(g_xb, g_vo);
    re[j] = dd_load4(g_eb, voff);
    cb0[j] = g_cb;
  });
  int ones;
  {
    dd_flag_wait<2 * P>(fl_n, fl_s);
    bool ok = true;
#pragma unroll
    for (int u = 0; u < DD_FLAG_LOADS / 2; u++)
#pragma unroll
      for (int k = 0; k < 4; k++) ok = ok && fl_n[u][k] == a.D && fl_s[u][k] == 1.0f;
    ones = __builtin_amdgcn_readfirstlane(__all(ok) ? 1 : 0);
  }
  /* one round of P iterations (a wave-uniform choice of body per launch, not per iteration) */
  auto round = [&](auto ONESC, auto FIRSTC) {
    constexpr bool ONES = decltype(ONESC)::value, FIRST = decltype(FIRSTC)::value;
    dd_static_for<P>([&](auto JC) {
      constexpr int j = decltype(JC)::value;
      if constexpr (ONES) dd_wait2<2 * (P - 1)>(ra[j], re[j]);
      else dd_wait3<FIRST ? P - 1 + 2 * j : 3 * (P - 1)>(ra[j], re[j], rcf[j]);
      dd_f4 fa = ra[j], fe = re[j];
      if constexpr (!ONES) {
        /* v_mul_legacy_f32: 0 * x is 0 for ANY x (a step past the break may hold inf), otherwise the IEEE product */
        asm volatile("v_mul_legacy_f32 %0, %4, %0\n\tv_mul_legacy_f32 %1, %4, %1\n\t"
                     "v_mul_legacy_f32 %2, %4, %2\n\tv_mul_legacy_f32 %3, %4, %3\n\ts_nop 1"
                     : "+v"(fe[0]), "+v"(fe[1]), "+v"(fe[2]), "+v"(fe[3])
                     : "v"(rcf[j]));
      }
      __builtin_amdgcn_sched_barrier(0);
      advance(); /* the slot's next tenant: its addresses, between the MFMAs */
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jn = 0; jn < 4; jn++)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fe[jn], acc[i][jn], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 16; g++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* one MFMA */
        __builtin_amdgcn_sched_group_barrier(0x004, 4, 0); /* up to four scalar-ALU instructions */
      }
      __builtin_amdgcn_sched_barrier(0);
      ra[j] = dd_load4(g_xb, g_vo);
      re[j] = dd_load4(g_eb, voff);
      if constexpr (!ONES) rcf[j] = dd_load1(g_cb, voff_c);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // ---- the waves' tiles meet in LDS; every thread then owns float4s of the finished tile
  auto spill = [&]() {
    float *base = lds + (size_t)wv * 64 * DD_LD;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * (lane >> 4) + 4 * r + i;
        *reinterpret_cast<dd_f4 *>(base + row * DD_LD + 4 * (lane & 15)) = dd_f4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  if (ones) {
    for (int i0 = 0; i0 < n_it; i0 += P) round(T_{}, F_{});
    spill();
    if (n_r > 0) {
      clear();
      for (int i0 = 0; i0 < n_r; i0 += P) round(T_{}, F_{});
    }
  } else {
    dd_static_for<P>([&](auto JC) { rcf[decltype(JC)::value] = dd_load1(cb0[decltype(JC)::value], voff_c); });
    round(F_{}, T_{});
    for (int i0 = P; i0 < n_it; i0 += P) round(F_{}, F_{});
    spill();
    if (n_r > 0) {
      clear();
      for (int i0 = 0; i0 < n_r; i0 += P) round(F_{}, F_{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the surplus loads */
  auto total4 = [&](int ch) { /* chunk ch: row ch / 16, columns 4 (ch % 16) .. + 3, the waves in order */
    const float *p = lds + (ch >> 4) * DD_LD + 4 * (ch & 15);
    dd_f4 s = *reinterpret_cast<const dd_f4 *>(p);
#pragma unroll
    for (int w2 = 1; w2 < NW; w2++) s += *reinterpret_cast<const dd_f4 *>(p + (size_t)w2 * 64 * DD_LD);
    return s;
  };
  __syncthreads();
  constexpr int CPT = 1024 / (64 * NW); /* chunks per thread */
#pragma unroll
  for (int u = 0; u < CPT; u++) {
    const int ch = u * 64 * NW + threadIdx.x;
    dd_f4 s = total4(ch);
    const size_t off = (size_t)(m0 + (ch >> 4)) * a.H + n0 + 4 * (ch & 15);
    if (a.mode == 1) s += *reinterpret_cast<const dd_f4u *>(a.delta + off);
    *reinterpret_cast<dd_f4u *>(a.delta + off) = s;
    if (a.mode == 2) {
      dd_f4 W = *reinterpret_cast<const dd_f4u *>(a.w + off), M = *reinterpret_cast<const dd_f4u *>(a.m + off);
#pragma unroll
      for (int k = 0; k < 4; k++) { /* recur-nn.c:482-487 */
        const float t = s[k] * a.rate, mm = M[k];
        W[k] += t + mm * a.mw;
        M[k] = (mm + t) * a.momentum;
      }
      *reinterpret_cast<dd_f4u *>(a.w + off) = W;
      *reinterpret_cast<dd_f4u *>(a.m + off) = M;
    }
  }
  if (n_r > 0) {
    __syncthreads();
    spill();
    __syncthreads();
    float *rp = a.rest_planes + (size_t)mt * a.rest_stride;
#pragma unroll
    for (int u = 0; u < CPT; u++) {
      const int ch = u * 64 * NW + threadIdx.x;
      if ((ch >> 4) < a.rest) {
        const dd_f4 s = total4(ch);
        *reinterpret_cast<dd_f4u *>(rp + (size_t)(ch >> 4) * a.H + n0 + 4 * (ch & 15)) = s;
      }
    }
  }
}

/* (one workgroup per CU: NW / 4 waves per SIMD, with the registers that leaves each -- told to hipcc, which otherwise
 * aims at a higher occupancy and parks ring registers in AGPRs between an asynchronous load and its wait) */
template <int NW, int P>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void k_delta_direct(DdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dd_lds[];
  __builtin_amdgcn_s_setprio(2);
  dd_body<NW, P>(a, dd_lds);
}

/* What k_delta_direct leaves for the update: the rows from 64 tm on (sum of the tm partial planes), the columns
 * outside 1 .. hidden_size of every row (delta 0 there: recur-nn.c:334-337), and the top layer -- rnn_apply_learning's
 * momentum update (recur-nn.c:482-487, 653-676) for all of them, the rest rows' delta stored on the way.  One thread
 * per float4 of: [rest rows][H] | [core rows] x the edge float4s (column 0's and the last column's) | ho. */
struct DdEdgeArgs {
  float *w, *m, *delta;        /* [I][H] (column 0) */
  const float *rest_planes;
  size_t rest_stride;
  int tm_planes, rows_core, rest, H, hidden_size;
  float *ho_w, *ho_m;
  const float *ho_delta;
  float *ho_delta_out; /* where the top layer's sums are to be stored as well (they came as a plane), or NULL */
  size_t ho_n4;        /* 0: the top layer is not this launch's business */
  float rate, ho_rate, momentum, mw;
  int mode; /* as DdArgs.mode: the update only with 2; 1 adds to delta */
};
static inline size_t dd_edge_threads(const DdEdgeArgs &a) {
  const int h4 = a.H / 4, e4 = (a.hidden_size + 1) / 4;
  return (size_t)a.rest * a.H / 4 + (size_t)a.rows_core * (1 + (h4 - e4)) + a.ho_n4;
}
__global__ __launch_bounds__(256) void k_apply_edges(DdEdgeArgs a) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t rest4 = (size_t)a.rest * a.H / 4;
  const int h4 = a.H / 4;
  const int e4 = (a.hidden_size + 1) / 4; /* the float4 that holds column hidden_size + 1 (== h4 - 1 when H = hidden + 4) */
  const size_t edge4 = (size_t)a.rows_core * (1 + (h4 - e4));
  float *w, *m;
  dd_f4 d;
  float rate = a.rate;
  if (tid < rest4) {
    const size_t off = 4 * tid;
    const int c = (int)(off % (size_t)a.H);
    /* the tm planes in plane order, sixteen loads in flight */
    dd_f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < a.tm_planes; z0 += 16) {
      dd_f4 t[16];
#pragma unroll
      for (int z = 0; z < 16; z++)
        t[z] = *reinterpret_cast<const dd_f4 *>(a.rest_planes + (size_t)(z0 + z < a.tm_planes ? z0 + z : z0) * a.rest_stride + off);
#pragma unroll
      for (int z = 0; z < 16; z++)
        if (z0 + z < a.tm_planes) s += t[z];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) d[k] = (c + k >= 1 && c + k <= a.hidden_size) ? s[k] : 0.0f;
    const size_t g = (size_t)a.rows_core * a.H + off;
    if (a.mode == 1) d += *reinterpret_cast<const dd_f4 *>(a.delta + g);
    *reinterpret_cast<dd_f4 *>(a.delta + g) = d;
    w = a.w + g;
    m = a.m + g;
  } else if (tid < rest4 + edge4) {
    /* core rows: float4 0 (column 0 is outside, 1 .. 3 are the tile kernel's) and the float4s from e4 on */
    const size_t k = tid - rest4;
    const int per = 1 + (h4 - e4);
    const int row = (int)(k / per), which = (int)(k % per);
    const int c = which == 0 ? 0 : 4 * (e4 + which - 1);
    const size_t g = (size_t)row * a.H + c;
    /* only the columns outside 1 .. hidden_size: delta = 0 there */
    dd_f4 W = *reinterpret_cast<const dd_f4 *>(a.w + g), M = *reinterpret_cast<const dd_f4 *>(a.m + g);
    dd_f4 Dl = *reinterpret_cast<const dd_f4 *>(a.delta + g);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      if (c + kk >= 1 && c + kk <= a.hidden_size) continue;
      if (a.mode != 1) Dl[kk] = 0.0f;
      if (a.mode == 2) {
        const float mm = M[kk];
        W[kk] += mm * a.mw; /* t = 0 * rate */
        M[kk] = mm * a.momentum;
      }
    }
    *reinterpret_cast<dd_f4 *>(a.delta + g) = Dl;
    if (a.mode == 2) {
      *reinterpret_cast<dd_f4 *>(a.w + g) = W;
      *reinterpret_cast<dd_f4 *>(a.m + g) = M;
    }
    return;
  } else if (tid < rest4 + edge4 + a.ho_n4) {
    const size_t off = 4 * (tid - rest4 - edge4);
    d = *reinterpret_cast<const dd_f4 *>(a.ho_delta + off);
    if (a.ho_delta_out) *reinterpret_cast<dd_f4 *>(a.ho_delta_out + off) = d;
    w = a.ho_w + off;
    m = a.ho_m + off;
    rate = a.ho_rate;
  } else {
    return;
  }
  if (a.mode != 2) return;
  dd_f4 W = *reinterpret_cast<const dd_f4 *>(w), M = *reinterpret_cast<const dd_f4 *>(m);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float t = d[k] * rate, mm = M[k];
    W[k] += t + mm * a.mw;
    M[k] = (mm + t) * a.momentum;
  }
  *reinterpret_cast<dd_f4 *>(w) = W;
  *reinterpret_cast<dd_f4 *>(m) = M;
}
