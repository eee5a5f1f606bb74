// kernels_apply.hip -- rnn_apply_learning (seven update rules), the fused two-update launch, conditioning helpers
// (scale, abs-max, zero) and the small segmented copy.
#include "k_common.h"

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
// one float4 of rnn_apply_learning's update (recur-nn.c:454-593): weights wv, deltas dv, momentum / accumulator mv
// and, for ADADELTA and RPROP, the second accumulator av
template <int METHOD>
__device__ __forceinline__ void apply_update4(float (&wv)[4], float (&dv)[4], float (&mv)[4], float (&av)[4], float rate,
                                              float momentum, float mw) {
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
}

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
    /* (the rest rows' many planes with 32 loads in flight instead of 8: no difference, 250.7 / 250.2 us) */
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
  apply_update4<METHOD>(wv, dv, mv, av, rate, momentum, mw);
  *reinterpret_cast<float4 *>(w + 4 * q) = make_float4(wv[0], wv[1], wv[2], wv[3]);
  *reinterpret_cast<float4 *>(m + 4 * q) = make_float4(mv[0], mv[1], mv[2], mv[3]);
  if (METHOD == 5 || METHOD == 6)
    *reinterpret_cast<float4 *>(aux + 4 * q) = make_float4(av[0], av[1], av[2], av[3]);
}

// --------------------------------- the exchange step as kernel-issued peer traffic --
//
// The sum of the weight deltas over the ranks (recur-nn.c:734-739 distributed) and the update (recur-nn.c:601-678) as
// reduce-scatter -> sharded optimiser -> all-gather IN ONE KERNEL, with no collective library in between (SURVEY.md
// section 8e's alternative to the all-reduce).  Every rank owns 1 / world of each weight array (a contiguous range of
// float4): for its range it adds the `world` ranks' local delta sums in RANK ORDER through peer pointers (every replica
// therefore sees bit-identical sums), applies the update rule to its own copy of weights and momentum -- the
// momentum / accumulator arrays are only ever touched in the owned range: optimiser state and traffic / world --, and
// writes the new weights into EVERY rank's weight array.  The summed deltas are left in the owner's delta array
// (owned range only).  Ordering between ranks comes from k_xchg_barrier launches around it (or, for ranks that one
// host thread drives in lock step -- the one-GPU emulation of the tests --, from that thread's own order).
constexpr int XCHG_MAX = 8;
struct XchgSeg {
  float *w[XCHG_MAX];            /* every rank's weight array (own included, at index `rank`) */
  const float *delta[XCHG_MAX];  /* every rank's local delta sums                              */
  float *m, *aux, *delta_out;    /* own                                                       */
  size_t lo4, hi4;               /* owned range in float4                                     */
  float rate;
};
struct XchgArgs {
  XchgSeg seg[2]; /* top layer, recurrent layer */
  int world, rank;
  unsigned first1; /* first block of segment 1 */
};
/* "a barrier of this process's exchange has given up": the DEVICE's copy of what k_xchg_barrier tells the host through
 * the host-mapped abort word.  Round 6: k_apply_xchg read that host word itself, every thread of it, over PCIe -- 50 us
 * of a launch that takes 6 (profiles/NOTES_r06.md section 9; the emulated exchange read +34 .. +86 us per rank). */
__device__ unsigned g_xchg_gave_up;
template <int METHOD>
__global__ __launch_bounds__(256) void k_apply_xchg(XchgArgs xa, float momentum, float mw, const unsigned *abort_word) {
  /* a barrier in front of this launch gave up (a rank missing): the ranks' sums are not all there -- touch nothing;
   * the host aborts at its next synchronisation (rnn_core.c: dsync), until then no weights are made from half a sum */
  if (abort_word && __hip_atomic_load(&g_xchg_gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  const int g = blockIdx.x >= xa.first1 ? 1 : 0;
  const XchgSeg &sg = xa.seg[g];
  const size_t q = sg.lo4 + (size_t)(blockIdx.x - (g ? xa.first1 : 0u)) * 256 + threadIdx.x;
  if (q >= sg.hi4) return;
  float4 t[XCHG_MAX];
#pragma unroll
  for (int p = 0; p < XCHG_MAX; p++) t[p] = ld4(sg.delta[p < xa.world ? p : 0] + 4 * q); /* all in flight */
  float4 Dl = t[0];
#pragma unroll
  for (int p = 1; p < XCHG_MAX; p++)
    if (p < xa.world) { Dl.x += t[p].x; Dl.y += t[p].y; Dl.z += t[p].z; Dl.w += t[p].w; }
  float *w = sg.w[xa.rank];
  const float4 W = ld4(w + 4 * q), M = ld4(sg.m + 4 * q);
  const float4 A = (METHOD == 5 || METHOD == 6) ? ld4(sg.aux + 4 * q) : zero4();
  float wv[4] = {W.x, W.y, W.z, W.w}, dv[4] = {Dl.x, Dl.y, Dl.z, Dl.w};
  float mv[4] = {M.x, M.y, M.z, M.w}, av[4] = {A.x, A.y, A.z, A.w};
  apply_update4<METHOD>(wv, dv, mv, av, sg.rate, momentum, mw);
  const float4 Wn = make_float4(wv[0], wv[1], wv[2], wv[3]);
#pragma unroll
  for (int p = 0; p < XCHG_MAX; p++)
    if (p < xa.world) *reinterpret_cast<float4 *>(sg.w[p] + 4 * q) = Wn; /* the all-gather, by stores */
  *reinterpret_cast<float4 *>(sg.m + 4 * q) = make_float4(mv[0], mv[1], mv[2], mv[3]);
  if (METHOD == 5 || METHOD == 6) *reinterpret_cast<float4 *>(sg.aux + 4 * q) = make_float4(av[0], av[1], av[2], av[3]);
  *reinterpret_cast<float4 *>(sg.delta_out + 4 * q) = Dl;
}

// Arrival barrier between the ranks' streams: this rank's slot of the shared counter array (host memory that every
// rank has mapped: visible across processes and GPUs inside a kernel) takes `seq`, then the launch waits until every
// slot has reached it.  What the previous kernels of this stream wrote is visible to the peers' later kernels through
// the kernel boundaries (release at the end of a launch, acquire at the start of the next, both at system scope).
// Bounded: 20 s of polling by the device's own clock (RECUR_AMD_XCHG_BARRIER_TIMEOUT_S) raise *abort_word (see dsync in
// rnn_core.c) and the device's copy of it.
__global__ void k_xchg_barrier(unsigned *flags, int rank, int world, unsigned seq, unsigned *abort_word,
                               unsigned long long timeout_ticks) {
  const int p = threadIdx.x;
  if (p == 0) {
    __threadfence_system();
    __hip_atomic_store(&flags[rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  /* (a barrier of this process has given up already: the ranks are out of step for good -- later barriers do not wait
   * their time out again, the host aborts at its next synchronisation) */
  if (__hip_atomic_load(&g_xchg_gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  if (p < world) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); /* 100 MHz */
    for (;;) {
      /* (seq - got) as a signed number: counters that have wrapped still compare */
      if ((int)(__hip_atomic_load(&flags[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
        if (abort_word) __hip_atomic_store(abort_word, 6u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&g_xchg_gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  }
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

// ================================================================ launchers ==

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

extern "C" void ramd_launch_xchg_barrier(ramd_stream_t st, unsigned *flags_dev, int rank, int world, unsigned seq,
                                         unsigned *abort_word_dev) {
  const unsigned long long ticks = 100000000ull * (unsigned long long)env_int("RECUR_AMD_XCHG_BARRIER_TIMEOUT_S", 20);
  RAMD_LAUNCH(k_xchg_barrier, dim3(1), dim3(64), 0, (hipStream_t)st, flags_dev, rank, world, seq, abort_word_dev, ticks);
}

/* the sharded update (k_apply_xchg): w / delta: [2][world] device pointers of every rank's top-layer (0) and recurrent
 * (1) arrays; m, aux, delta_out: own arrays [2]; n: floats per array [2] */
extern "C" void ramd_launch_apply_xchg(ramd_stream_t st_, int method, int rank, int world, float *const *w,
                                       const float *const *delta, float *const *m, float *const *aux,
                                       float *const *delta_out, const size_t *n, const float *rate, float momentum,
                                       float mw) {
  hipStream_t st = (hipStream_t)st_;
  if (world < 1 || world > XCHG_MAX || rank < 0 || rank >= world) {
    fprintf(stderr, "librecur_amd: the kernel-issued exchange takes 1..%d ranks (rank %d of %d)\n", XCHG_MAX, rank, world);
    abort();
  }
  XchgArgs xa = {};
  xa.world = world;
  xa.rank = rank;
  unsigned blocks = 0;
  for (int g = 0; g < 2; g++) {
    XchgSeg &sg = xa.seg[g];
    for (int p = 0; p < world; p++) {
      sg.w[p] = w[g * world + p];
      sg.delta[p] = delta[g * world + p];
    }
    sg.m = m[g];
    sg.aux = aux[g];
    sg.delta_out = delta_out[g];
    const size_t n4 = n[g] / 4;
    sg.lo4 = n4 * (size_t)rank / world;
    sg.hi4 = n4 * (size_t)(rank + 1) / world;
    sg.rate = rate[g];
    if (g == 1) xa.first1 = blocks;
    blocks += (unsigned)((sg.hi4 - sg.lo4 + 255) / 256);
  }
  if (!blocks) return;
  dim3 gr(blocks), bl(256);
  int ev = timing_begin(st, T_APPLY);
  switch (method) {
  case 1: RAMD_LAUNCH(k_apply_xchg<1>, gr, bl, 0, st, xa, momentum, mw, ramd_abort_word_dev()); break;
  case 4: RAMD_LAUNCH(k_apply_xchg<4>, gr, bl, 0, st, xa, momentum, mw, ramd_abort_word_dev()); break;
  case 5: RAMD_LAUNCH(k_apply_xchg<5>, gr, bl, 0, st, xa, momentum, mw, ramd_abort_word_dev()); break;
  case 6: RAMD_LAUNCH(k_apply_xchg<6>, gr, bl, 0, st, xa, momentum, mw, ramd_abort_word_dev()); break;
  default: RAMD_LAUNCH(k_apply_xchg<0>, gr, bl, 0, st, xa, momentum, mw, ramd_abort_word_dev()); break;
  }
  timing_end(st, ev);
}

// A replica's checksum, formed ON THE DEVICE through the loads every kernel of the path uses (L2, the per-XCD
// caches): sum over the 32-bit words of `n_arrays` arrays in a row of word_i * (2 i + 1) mod 2^64, i counted through
// the row -- independent of the order of summation (atomics), sensitive to position.  Beside the same sum over a
// device-to-host COPY of the arrays (rnn_core.c: rnn_amd_set_replica_checksum) it is what tells a launcher that what
// peers stored into this rank's arrays is what this rank's kernels read.
struct CksumArgs {
  const unsigned *a[4];
  unsigned long long n[4], first[4]; /* words, index of the array's first word in the row */
  int n_arrays;
};
__global__ __launch_bounds__(256) void k_replica_checksum(CksumArgs c, unsigned long long *out) {
  unsigned long long acc = 0;
  for (int k = 0; k < c.n_arrays; k++) {
    const unsigned *a = c.a[k];
    const unsigned long long n = c.n[k], f = c.first[k];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256)
      acc += (unsigned long long)a[i] * (2 * (f + i) + 1);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
extern "C" void ramd_launch_replica_checksum(ramd_stream_t st, int n_arrays, const float *const *arrays, const size_t *n_floats,
                                             unsigned long long *out_dev) {
  CksumArgs c = {};
  unsigned long long first = 0;
  c.n_arrays = n_arrays < 4 ? n_arrays : 4;
  for (int k = 0; k < c.n_arrays; k++) {
    c.a[k] = reinterpret_cast<const unsigned *>(arrays[k]);
    c.n[k] = n_floats[k];
    c.first[k] = first;
    first += n_floats[k];
  }
  HIP_CHECK(hipMemsetAsync(out_dev, 0, sizeof(unsigned long long), (hipStream_t)st));
  RAMD_LAUNCH(k_replica_checksum, dim3(512), dim3(256), 0, (hipStream_t)st, c, out_dev);
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
