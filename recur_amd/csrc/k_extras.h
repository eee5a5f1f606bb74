// k_extras.h -- the chain "extras" (column 0 and the input columns of every step's error, recur-nn.c:338-376)
// and the data-dependent control of bptt_and_accumulate_error (recur-nn.c:317-330, 383-413) as device
// functions: shared by k_extras_control / k_extras_gather / k_bptt_control (kernels_bptt.hip) and by the tail of
// the one-launch chain (kernels_chain.hip), which runs them for its own streams without a launch of their own.
#pragma once
#include "k_common.h"
#ifndef XC_STAMP
#define XC_STAMP(i) do { } while (0)
#endif

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
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
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

// ----------------------------------------------------- K9: BPTT control --

// The data-dependent part of bptt_and_accumulate_error (recur-nn.c:317-330,
// 383-413), one wave per stream: lane k holds the error sum of step k, a ballot finds
// the step at which the reference's loop would have stopped, lane 0 derives ih_scale
// and the adaptive min_error_factor, and the lanes publish coef[t][r] = ih_scale while
// the step counts, 0 afterwards.
/* es: the stream's error sums by step, es[k * stride] */
/* what the control logic reads about its stream besides the error sums: requested by the caller before it
 * starts the work whose results the logic waits for (one memory round trip less behind the barrier) */
struct ControlIn {
  float top, mef, lr;
  double depth_total; /* stat_depth so far */
  bool live;
};
__device__ __forceinline__ ControlIn bptt_control_load(const View &v, int r, int j, const unsigned char *active) {
  ControlIn ci;
  ci.top = as_global(v.b.top_scaled)[r];
  ci.mef = as_global(v.b.mef)[r];
  ci.lr = as_global(v.b.lr)[r];
  ci.depth_total = as_global(v.b.stat_depth)[r];
  ci.live = !active || as_global(active)[j];
  return ci;
}
__device__ __forceinline__ void bptt_control_wave(const View &v, int r, int j, int lane, const ControlIn &ci,
                                                  unsigned flags, const float *es_src, size_t es_stride) {
  const RamdShape &s = v.sh;
  const int D = s.D;
  if (!ci.live) {
    for (int k = lane; k < D; k += 64) as_global(v.b.coef)[(size_t)k * s.Scap + r] = 0.0f;
    if (lane == 0) as_global(v.b.n_exec)[r] = 0; /* no step ran: k_err_writeback leaves its images alone */
    return;
  }
  float top = ci.top;
  float max_error_sum = MAX_ERROR_GAIN_F * top + 1;
  float error_sum_ceiling = ERROR_GAIN_CEILING_F * top;
  float min_error_gain = MIN_ERROR_GAIN_F * top;
  float mef = ci.mef;
  /* MIN(a, b) of the reference is (a < b) ? a : b: keep NaN behaviour aligned */
  float mef_rate = mef / ci.lr;
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
    as_global(v.b.mef)[r] = mef;
    as_global(v.b.ih_scale)[r] = scale;
    as_global(v.b.bptt_err)[r] = error_sum;
    as_global(v.b.n_exec)[r] = n_exec;
    as_global(v.b.depth_log)[r] = D - t;
    as_global(v.b.stat_depth)[r] = ci.depth_total + (double)(D - t);
  }
  /* 0x20000000: rnn_bptt_calculate without batching leaves the UNSCALED sum in ih_delta and puts
   * ih_scale into the rate (recur-nn.c:966-975) */
  const float cf = (flags & 0x20000000u) ? 1.0f : scale;
  for (int k = lane; k < D; k += 64) as_global(v.b.coef)[(size_t)k * s.Scap + r] = (k < n_exec) ? cf : 0.0f;
}


// The extras of all steps of ONE stream (state row r, number j within the call's rows) and its control
// logic, by a workgroup of THREADS threads that all call this: the waves share out the stream's
// steps, leave each step's error sum in LDS (es_sh: 2 D + 1 floats), and wave 0 then runs the control
// logic on them (nothing else needs the sums of other streams).  Ends without a barrier: es_sh may
// be reused after the caller's next one.
template <int MAXQ, int THREADS>
__device__ __forceinline__ void extras_control_stream(const View &v, int r, int j, int nx, int nxp, int tn,
                                                      const unsigned char *active, unsigned flags,
                                                      float *es_sh) {
  const RamdShape &s = v.sh;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  /* tn == 0 (after the one-launch chain, which leaves no per-tile partial sums): the hidden columns'
   * part of step t's sum of squares (recur-nn.c:371) is the sum over the step's OUTPUT row, error
   * plane t + 1 -- the row that the item of step t + 1 holds in registers for its dot products.  So
   * every item also sums its own row, one more item (t = D) does only that, and the totals are put
   * together after the barrier. */
  const int items = tn == 0 ? s.D + 1 : s.D;
  float *hs_sh = es_sh + s.D;
  /* the next item's reads are requested before the current one is worked on */
  ExtrasIn<MAXQ> cur, nxt;
  ControlIn ci = {0.0f, 0.0f, 1.0f, 0.0, true};
  if (wave == 0) ci = bptt_control_load(v, r, j, active);
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
    bptt_control_wave(v, r, j, lane, ci, flags, es_sh, 1);
  }
}

// What a wave of the chain's tail (extras_control_tail below) can ask for when the launch STARTS: the input values
// of its first batch of items (they name the weight rows its dot products need: history slots that the chain itself
// does not read, 2-3 us away at the end of a launch that has streamed the whole history through the L2) and,
// wave 0, the control logic's inputs.  A handful of registers held across the chain.
constexpr int XT_B = 3, XT_BL = 2; /* items per batch of an early wave; items of a late wave (see extras_control_tail) */
struct TailPre {
  float xi[XT_B][2]; /* [item][columns lane, lane + 64] */
  ControlIn ci;
};
/* the first item, the stride and the end of wave `wave`'s items (NW waves, the upper half late) */
__device__ __forceinline__ void tail_items(int wave, int NW, int D, int &t0, int &stride, int &t_end) {
  const int NE = NW / 2;
  const int first_late = D > NE * XT_BL ? D - NE * XT_BL : 0;
  stride = NE;
  if (wave < NE) {
    t0 = wave;
    t_end = first_late;
  } else {
    t0 = first_late + (wave - NE);
    t_end = D;
  }
}
template <int THREADS>
__device__ __forceinline__ TailPre extras_tail_prefetch(const View &v, int r, int j, int nx, const unsigned char *active) {
  const RamdShape &s = v.sh;
  constexpr int NW = THREADS / 64;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int D = s.D;
  TailPre p;
  p.ci = ControlIn{0.0f, 0.0f, 1.0f, 0.0, true};
  if (wave == 0) p.ci = bptt_control_load(v, r, j, active);
  const int xcol[2] = {(lane == 0 || lane >= nx) ? 0 : s.hidden_size + lane, lane + 64 < nx ? s.hidden_size + lane + 64 : 0};
  int t0, stride, t_end;
  tail_items(wave, NW, D, t0, stride, t_end);
#pragma unroll
  for (int b = 0; b < XT_B; b++) {
    const int t = t0 + stride * b;
    const auto *x = as_global(input_row<true>(v, r, t < D ? t : D - 1));
#pragma unroll
    for (int h = 0; h < 2; h++) p.xi[b][h] = x[xcol[h]]; /* as loaded: a mask applied HERE is a wait for the load, a memory round trip in front of the chain's weight panel (0.9 us by stamps); the tail masks it */
  }
  return p;
}

// The same for the tail of the one-launch chain (tn == 0, nx <= 128, THREADS threads that all call it), written
// for latency: behind the chain every error row is in the XCD's L2 and the tail is a handful of dependent
// round trips, so a wave requests EVERYTHING its items need before it computes anything -- the input values
// of all its items first (they name the weight rows), the error rows (past the L1: other CUs wrote them
// during this launch), the bias row of W once per wave, each item's first other live row as soon as the
// input values are there -- B items per wave and batch.  The upper half of the waves (the chain's multiplying
// waves, which still finish and publish the last half-step when the others get here) take two items each, the
// last ones; the lower half share the rest.  The last plane's row (only its sum of squares is wanted) is the
// last wave's, which alone has to wait for the chain's last flags (wait_last(), before it requests anything).
// Written so that hipcc has no reason to branch around a load (it then waits for each at the join: see
// extras_load): every load is unconditional from a clamped address, and what lies outside a row is
// removed by a 0 / 1 factor on the error values, not by a select on what was loaded.
// Sums in extras_compute's order: the results equal k_extras_control's.
/* pre: extras_tail_prefetch's answer for this stream, or nullptr (everything is requested here) */
template <int MAXQ, int THREADS, class WaitLast>
__device__ __forceinline__ void extras_control_tail(const View &v, int r, int j, int nx, int nxp,
                                                    const unsigned char *active, unsigned flags, float *es_sh,
                                                    const TailPre *pre, WaitLast &&wait_last) {
  constexpr int B = XT_B;
  const RamdShape &s = v.sh;
  constexpr int NW = THREADS / 64;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int D = s.D;
  float *hs_sh = es_sh + D;
  const int nq = (s.H / 4 + 63) / 64, last4 = s.H / 4 - 1;
  ControlIn ci = {0.0f, 0.0f, 1.0f, 0.0, true};
  if (pre) ci = pre->ci;
  else if (wave == 0) ci = bptt_control_load(v, r, j, active);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)v.b.ehi, 0, 0x7fffffff, 0x00020000);
  /* column 0, then the input columns: a lane looks after columns lane and lane + 64 (nx <= 128) */
  const int xcol[2] = {(lane == 0 || lane >= nx) ? 0 : s.hidden_size + lane, lane + 64 < nx ? s.hidden_size + lane + 64 : 0};
  const unsigned xmask[2] = {lane < nx ? 0xffffffffu : 0u, lane + 64 < nx ? 0xffffffffu : 0u};
  unsigned koff[MAXQ]; /* byte offset of this lane's chunk i within a row (clamped) */
  float km[MAXQ];      /* 1 where the chunk is part of the row, else 0 */
#pragma unroll
  for (int i = 0; i < MAXQ; i++) {
    const int k4 = lane + 64 * i;
    koff[i] = 16u * (unsigned)min(k4, last4);
    km[i] = (i < nq && k4 <= last4) ? 1.0f : 0.0f;
  }
  auto load_row = [&](int plane, float4 (&e)[MAXQ]) {
    const unsigned row_off = (unsigned)((plane * s.Scap + r) * s.I) * 4u;
#pragma unroll
    for (int i = 0; i < MAXQ; i++) {
      const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(row_off + koff[i]), 0, 16 /* sc1 */);
      e[i] = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
    }
  };
  auto mask_row = [&](float4 (&e)[MAXQ]) {
#pragma unroll
    for (int i = 0; i < MAXQ; i++) { e[i].x *= km[i]; e[i].y *= km[i]; e[i].z *= km[i]; e[i].w *= km[i]; }
  };
  XC_STAMP(3);
  constexpr int NE = NW / 2; /* early waves */
  /* items tb, tb + t_stride, ... (BB of them, those below t_end) */
  auto batch = [&](auto BC, const int tb, const int t_stride, const int t_end, const float (*xpre)[2]) {
    constexpr int BB = decltype(BC)::value;
    float xi[BB][2];
    float4 ev[BB][MAXQ], wh[BB][MAXQ], w0[MAXQ];
    // 1. the input values
#pragma unroll
    for (int b = 0; b < BB; b++) {
      if (xpre) { /* (uniform) */
        xi[b][0] = __uint_as_float(__float_as_uint(xpre[b][0]) & xmask[0]);
        xi[b][1] = __uint_as_float(__float_as_uint(xpre[b][1]) & xmask[1]);
      } else {
        const int t = tb + t_stride * b;
        const auto *x = as_global(input_row<true>(v, r, t < D ? t : D - 1)); /* (the one-launch chain: one ring position) */
#pragma unroll
        for (int h = 0; h < 2; h++) xi[b][h] = __uint_as_float(__float_as_uint(x[xcol[h]]) & xmask[h]);
      }
    }
    // 2. the bias row of W and the error rows
#pragma unroll
    for (int i = 0; i < MAXQ; i++) w0[i] = ld4g(reinterpret_cast<const char *>(v.b.ih_w) + koff[i]);
#pragma unroll
    for (int b = 0; b < BB; b++) {
      const int t = tb + t_stride * b;
      load_row(t < D ? t : D - 1, ev[b]);
    }
    XC_STAMP(5);
    // 3. each item's first live input row (a one-hot stream has exactly one beside the bias)
    unsigned long long live[BB][2]; /* columns 0 .. 63, 64 .. 127 */
    int l1[BB];
    auto value_of = [&](const float (&x)[2], int col) { return __shfl(col < 64 ? x[0] : x[1], col & 63, 64); };
#pragma unroll
    for (int b = 0; b < BB; b++) {
#pragma unroll
      for (int h = 0; h < 2; h++) live[b][h] = __ballot(xi[b][h] != 0.0f && (s.activation != 5 || xi[b][h] < 20.0f));
      const unsigned long long rest = live[b][0] & ~1ull;
      l1[b] = rest ? __ffsll((long long)rest) - 1 : live[b][1] ? 64 + __ffsll((long long)live[b][1]) - 1 : -1;
      const char *wr = reinterpret_cast<const char *>(v.b.ih_w + (size_t)(l1[b] >= 0 ? s.hidden_size + l1[b] : 0) * s.H);
#pragma unroll
      for (int i = 0; i < MAXQ; i++) wh[b][i] = ld4g(wr + koff[i]);
    }
    XC_STAMP(4);
    // 4. the sums
    float hs[BB], a0[BB], a1[BB];
#pragma unroll
    for (int b = 0; b < BB; b++) {
      mask_row(ev[b]);
      hs[b] = a0[b] = a1[b] = 0.0f;
#pragma unroll
      for (int i = 0; i < MAXQ; i++) {
        const float4 e = ev[b][i];
        hs[b] += (e.x * e.x + e.y * e.y) + (e.z * e.z + e.w * e.w);
        a0[b] += e.x * w0[i].x + e.y * w0[i].y + e.z * w0[i].z + e.w * w0[i].w;
        a1[b] += e.x * wh[b][i].x + e.y * wh[b][i].y + e.z * wh[b][i].z + e.w * wh[b][i].w;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int b = 0; b < BB; b++) {
        hs[b] += __shfl_xor(hs[b], off, 64);
        a0[b] += __shfl_xor(a0[b], off, 64);
        a1[b] += __shfl_xor(a1[b], off, 64);
      }
    }
#pragma unroll
    for (int b = 0; b < BB; b++) {
      const int t = tb + t_stride * b;
      if (t >= t_end) continue;
      if (lane == 0) hs_sh[t] = hs[b];
      const bool has0 = (live[b][0] & 1ull) != 0;
      float e0 = has0 ? a0[b] : 0.0f, e1 = l1[b] >= 0 ? a1[b] : 0.0f;
      if (s.activation == 2) {
        e0 /= 2 * (value_of(xi[b], 0) + 1.0f);
        e1 /= 2 * (value_of(xi[b], l1[b] >= 0 ? l1[b] : 0) + 1.0f);
      }
      auto *dst = as_global(v.b.ex + ((size_t)(t + 1) * s.Scap + r) * nxp);
      if (lane < nx) dst[lane] = (lane == 0 && has0) ? e0 : (lane == l1[b]) ? e1 : 0.0f;
      if (lane + 64 < nx) dst[lane + 64] = (lane + 64 == l1[b]) ? e1 : 0.0f;
      float sq = 0.0f;
      if (has0) sq += e0 * e0;
      if (l1[b] >= 0) sq += e1 * e1;
      // further live columns (dense inputs): two at a time, as extras_compute does
      unsigned long long more[2] = {live[b][0] & ~1ull, live[b][1]};
      auto pop = [&]() { /* the lowest live column left, or -1 */
        if (more[0]) {
          const int c = __ffsll((long long)more[0]) - 1;
          more[0] &= more[0] - 1;
          return c;
        }
        if (more[1]) {
          const int c = 64 + __ffsll((long long)more[1]) - 1;
          more[1] &= more[1] - 1;
          return c;
        }
        return -1;
      };
      (void)pop(); /* l1: done above */
      while (more[0] | more[1]) {
        const int la = pop();
        const int lb = pop();
        const float xa = value_of(xi[b], la), xb = value_of(xi[b], lb >= 0 ? lb : la);
        const char *wa = reinterpret_cast<const char *>(v.b.ih_w + (size_t)(s.hidden_size + la) * s.H);
        const char *wb = reinterpret_cast<const char *>(v.b.ih_w + (size_t)(s.hidden_size + (lb >= 0 ? lb : la)) * s.H);
        float acca = 0.0f, accb = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXQ; i++) {
          const float4 ta = ld4g(wa + koff[i]), tb4 = ld4g(wb + koff[i]);
          const float4 e = ev[b][i]; /* zero outside the row */
          acca += e.x * ta.x + e.y * ta.y + e.z * ta.z + e.w * ta.w;
          accb += e.x * tb4.x + e.y * tb4.y + e.z * tb4.z + e.w * tb4.w;
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
          dst[la] = acca;
          if (lb >= 0) dst[lb] = accb;
        }
        sq += acca * acca;
        if (lb >= 0) sq += accb * accb;
      }
      if (lane == 0) es_sh[t] = sq;
    }
    };
  constexpr int BL = XT_BL;                             /* items of a late wave: the last NE * BL of them */
  const int first_late = D > NE * BL ? D - NE * BL : 0;
  if (wave < NE) {
    for (int tb = wave; tb < first_late; tb += NE * B)
      batch(std::integral_constant<int, B>{}, tb, NE, first_late, (pre && tb == wave) ? pre->xi : (const float (*)[2])nullptr);
  } else {
    float4 elast[MAXQ]; /* the last plane's row: the sum of squares of step D - 1's output row */
    if (wave == NW - 1) {
      wait_last();
      load_row(D, elast);
    }
    batch(std::integral_constant<int, BL>{}, first_late + (wave - NE), NE, D, pre ? pre->xi : (const float (*)[2])nullptr);
    if (wave == NW - 1) {
      mask_row(elast);
      float h = 0.0f;
#pragma unroll
      for (int i = 0; i < MAXQ; i++) h += (elast[i].x * elast[i].x + elast[i].y * elast[i].y) + (elast[i].z * elast[i].z + elast[i].w * elast[i].w);
      for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off, 64);
      if (lane == 0) hs_sh[D] = h;
    }
  }
  XC_STAMP(6);
  __syncthreads();
  XC_STAMP(7);
  if (wave == 0) {
    for (int k = lane; k < D; k += 64) {
      const float es = hs_sh[k + 1] + es_sh[k];
      es_sh[k] = es;
      as_global(v.b.esum)[(size_t)k * s.Scap + r] = es;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): the wave's own LDS writes before it reads them back */
    bptt_control_wave(v, r, j, lane, ci, flags, es_sh, 1);
    XC_STAMP(8);
  }
}

// The tail of the one-launch chain for nets with DENSE inputs (gstclassify's features, up to 47 of them): ONE stream's
// extras -- k_extras_dense's work for the rows (step t, stream r), t = 0 .. D - 1: the error rows times W_ih's bias and input
// rows on the matrix cores (v_mfma_f32_16x16x4_f32, operands straight from memory, the rows past the L1: other CUs of this
// XCD wrote them during the launch), the row rule, the sums of squares -- and the stream's control logic, by the workgroup
// whose column tile the stream belongs to.  Instead of a k_extras_dense launch and a k_bptt_control launch behind the
// chain (10.1 + 4.9 us at 512 / 128 / 30).  32 rows (steps) per pass, the eight waves every eighth 16-deep chunk of K.
// lds: THREADS / 64 * 32 * 49 + THREADS / 64 * 32 + 192 + D floats (52 KB: the chain's operand area at hidden >= 512).
template <int THREADS, class WaitLast>
__device__ __forceinline__ void extras_dense_tail(const View &v, int r, int j, int nx, int nxp, const unsigned char *active,
                                                  unsigned flags, float *lds, WaitLast &&wait_last) {
  constexpr int NW = THREADS / 64, NT3 = 3, PF = 4, LD = 16 * NT3 + 1;
  const RamdShape &s = v.sh;
  const int D = s.D;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = tid & 63;
  const int lm = lane & 15, kq = lane >> 4;
  float *red = lds;                  /* [NW][32][LD] */
  float *rsq = red + NW * 32 * LD;   /* [NW][32]: the waves' shares of the rows' sums of squares */
  float *rowsq = rsq + NW * 32;      /* [64]: sum of squares of plane t's row */
  float *sqx = rowsq + 64;           /* [64]: sum of squares of step t's extras */
  float *es_sh = sqx + 64;           /* [D] */
  ControlIn ci = {0.0f, 0.0f, 1.0f, 0.0, true};
  if (wave == 0) ci = bptt_control_load(v, r, j, active);
  XC_STAMP(3);
  if (wave == NW - 1) wait_last(); /* the last plane's row: its column tiles' flags (earlier planes were seen complete in the loop) */
  __syncthreads();
  XC_STAMP(5);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)v.b.ehi, 0, 0x7fffffff, 0x00020000);
  const float *brow[NT3];
#pragma unroll
  for (int jn = 0; jn < NT3; jn++) {
    const int c = 16 * jn + lm;
    brow[jn] = v.b.ih_w + (size_t)((c == 0 || c >= nx) ? 0 : s.hidden_size + c) * s.H; /* (columns >= nx: discarded below) */
  }
  const int nchunks = (s.H + 15) / 16;
  const int mine = (nchunks - wave + NW - 1) / NW; /* chunks wave, wave + NW, .. */
  const float *xin0 = input_row<true>(v, r, 0); /* (the one-launch chain: one ring position) */
  (void)xin0;
  for (int t0 = 0; t0 <= D; t0 += 32) {
    unsigned row_off[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int t = min(t0 + 16 * i + lm, D);
      row_off[i] = (unsigned)((t * s.Scap + r) * s.I) * 4u;
    }
    f32x4 acc[2][NT3];
    float sq2[2] = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int jn = 0; jn < NT3; jn++) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 fa[PF][2], fb[PF][NT3];
    float fm[PF];
    auto request = [&](int slot, int n) {
      const int k = 16 * (wave + NW * n) + 4 * kq;
      const bool in = n < mine && k < s.H;
      const int kc = in ? k : 0;
      fm[slot] = in ? 1.0f : 0.0f; /* (a factor on the values: a select on the loaded registers would wait for them here) */
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(row_off[i] + 4u * (unsigned)kc), 0, 16 /* sc1 */);
        fa[slot][i] = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
      }
#pragma unroll
      for (int jn = 0; jn < NT3; jn++) fb[slot][jn] = ld4(brow[jn] + kc);
    };
#pragma unroll
    for (int p = 0; p < PF; p++) request(p, p);
    /* the row rule's input values, requested with the first operands: thread (row, sub) of the epilogue */
    const int erow = (tid >> 3) & 31, sub = tid & 7;
    const int et = min(t0 + erow, D - 1);
    constexpr int CPT = (16 * NT3 + 7) / 8;
    float xv[CPT];
    if (tid < 8 * 32) {
      const float *x = input_row<true>(v, r, et);
#pragma unroll
      for (int q = 0; q < CPT; q++) {
        const int c = sub + 8 * q;
        xv[q] = x[(c == 0 || c >= nx) ? 0 : s.hidden_size + c];
      }
    }
    for (int n0 = 0; n0 < mine; n0 += PF) {
#pragma unroll
      for (int p = 0; p < PF; p++) {
        if (n0 + p < mine) { /* (wave-uniform) */
          float4 a[2], bb[NT3];
          const float m = fm[p];
#pragma unroll
          for (int i = 0; i < 2; i++) a[i] = make_float4(fa[p][i].x * m, fa[p][i].y * m, fa[p][i].z * m, fa[p][i].w * m);
#pragma unroll
          for (int jn = 0; jn < NT3; jn++) bb[jn] = fb[p][jn];
          request(p, n0 + p + PF);
#pragma unroll
          for (int i = 0; i < 2; i++) {
            sq2[i] += (a[i].x * a[i].x + a[i].y * a[i].y) + (a[i].z * a[i].z + a[i].w * a[i].w);
#pragma unroll
            for (int jn = 0; jn < NT3; jn++) {
              acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bb[jn].x, acc[i][jn], 0, 0, 0);
              acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bb[jn].y, acc[i][jn], 0, 0, 0);
              acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bb[jn].z, acc[i][jn], 0, 0, 0);
              acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bb[jn].w, acc[i][jn], 0, 0, 0);
            }
          }
        }
      }
    }
    XC_STAMP(4);
    // accumulator register rr of tile (i, jn): row 16 i + 4 kq + rr, column 16 jn + lm
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int jn = 0; jn < NT3; jn++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) red[(wave * 32 + 16 * i + 4 * kq + rr) * LD + 16 * jn + lm] = acc[i][jn][rr];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      sq2[i] += __shfl_xor(sq2[i], 16, 64);
      sq2[i] += __shfl_xor(sq2[i], 32, 64);
      if (kq == 0) rsq[wave * 32 + 16 * i + lm] = sq2[i];
    }
    __syncthreads();
    if (tid < 8 * 32) {
      const int t = t0 + erow;
      const bool live = t < D;
      float sq = 0.0f;
      float *dst = v.b.ex + ((size_t)(et + 1) * s.Scap + r) * nxp;
#pragma unroll
      for (int q = 0; q < CPT; q++) {
        const int c = sub + 8 * q;
        if (c < nx) {
          float e = 0.0f;
#pragma unroll
          for (int w = 0; w < NW; w++) e += red[(w * 32 + erow) * LD + c];
          const float xi = xv[q];
          const bool on = xi != 0.0f && (s.activation != 5 || xi < 20.0f);
          e = on ? e : 0.0f;
          if (on && s.activation == 2) e /= 2 * (xi + 1.0f);
          if (live) dst[c] = e;
          sq += e * e;
        }
      }
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) sq += __shfl_xor(sq, off, 64);
      if (sub == 0 && live) sqx[t] = sq;
    } else if (tid < 8 * 32 + 32) {
      const int row = tid - 8 * 32, t = t0 + row;
      float tot = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; w++) tot += rsq[w * 32 + row];
      if (t <= D) rowsq[t] = tot;
    }
    __syncthreads();
    XC_STAMP(6);
  }
  XC_STAMP(7);
  if (wave == 0) {
    for (int k = lane; k < D; k += 64) {
      const float es = rowsq[k + 1] + sqx[k];
      es_sh[k] = es;
      as_global(v.b.esum)[(size_t)k * s.Scap + r] = es;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): the wave's own LDS writes before it reads them back */
    bptt_control_wave(v, r, j, lane, ci, flags, es_sh, 1);
    XC_STAMP(8);
  }
}
