// kernels_chain.hip -- the BPTT chain E_i = E_h . W_ih^T over the D steps (recur-nn.c:338-376): a launch per step
// (k_chain_main, k_chain_wide) or all steps in one launch (k_chain_persist), and their launch logic.
#include "k_common.h"
#ifdef PC_STAMPS /* development builds only: stamps of the chain's tail (tools/gpu_chain_stamps.py) */
__device__ unsigned long long g_xc_stamps[32];
#define XC_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == ((i) == 10 ? 256 : 0)) g_xc_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PRO_STAMP(i, tid) do { if (blockIdx.x == PRO_WG && threadIdx.x == (tid)) g_xc_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PRO_STAMP_DEP(i, tid, dep) do { if (blockIdx.x == PRO_WG && threadIdx.x == (tid) && (dep) != 0x7ffffff1) g_xc_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
/* both clocks: slot i the 100 MHz one, slot i + 1 the shader clock (s_memtime): the clock the chip ran at between two of these */
#define PRO_CLOCKS(i, tid) do { if (blockIdx.x == PRO_WG && threadIdx.x == (tid)) { g_xc_stamps[i] = __builtin_amdgcn_s_memrealtime(); g_xc_stamps[(i) + 1] = __builtin_amdgcn_s_memtime(); } } while (0)
#ifndef PRO_WG
#define PRO_WG 0
#endif
#else
#define PRO_STAMP(i, tid) do { } while (0)
#define PRO_STAMP_DEP(i, tid, dep) do { } while (0)
#define PRO_CLOCKS(i, tid) do { } while (0)
#endif
#include "k_extras.h"
BND_DECL(g_bnd_chain, ramd_bnd_chain_stamps)

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
// MT = 32 (round 5): tiles of 32 streams for sets that leave the 64-stream grid short of the chip -- hidden 2048 with
// 256 streams is 4 x 32 = 128 tiles of 64 x 64 on 256 CUs, 0.48 of peak; as 32 x 64 tiles it is 256 -- with the K of
// every stage split over the two wave pairs instead (waves 0, 1: chunks 0 .. 7 of a stage's 16; waves 2, 3: 8 .. 15;
// both pairs a 32 x 32 quadrant each) and the two partial tiles added in the epilogue.  A stage is then 8 KB of A
// and 16 KB of B: six LDS-DMA instructions per loader wave.
template <int NS, int MT = 64>
__global__ __launch_bounds__(512) void k_chain_wide(const View *__restrict__ vp, int uniform_idx, int row0,
                                                    int nrows, int t, int tm, int tn) {
  constexpr bool HALF = MT == 32;
  constexpr int LPW = HALF ? 6 : 8; /* LDS-DMA instructions per loader wave and stage */
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  View v = *vp;
  v.b.uniform_idx = uniform_idx;
  const RamdShape &s = v.sh;
  const int L = blockIdx.x;
  const int xcd = L & 7, q = L >> 3;
  const int mt = q % tm, nt = (q / tm) * 8 + xcd; /* the m tiles of one W panel share an XCD */
  if (nt >= tn) return;
  const int m0 = mt * MT, n0 = 1 + nt * WN; /* output columns start at 1 */
  const int wave8 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int lm = lane & 31, kh = lane >> 5;
  const float *ehi_t = v.b.ehi + ((size_t)t * s.Scap + row0) * s.I;
  const uint32_t lds0 = lds_byte_addr(wsm);

  if (loader) {
    // instruction i (0 .. 4 LPW - 1 over the four loader waves) fills four rows of the stage image -- the MT rows of A,
    // then the 64 of B: image row R = 4 i + (l >> 4) --: lane l brings chunk (l & 15) ^ (R & 15) of it
    const float *src[LPW];
#pragma unroll
    for (int j = 0; j < LPW; j++) {
      const int i = wave * LPW + j;
      const int R = 4 * i + (lane >> 4);
      const int c = (lane & 15) ^ (R & 15);
      const float *base = R < MT ? ehi_t + (size_t)(m0 + R) * s.I : v.b.ih_w + (size_t)(n0 + R - MT) * s.H;
      src[j] = base + 1 + 4 * c; /* K runs over the hidden columns 1..hidden_size */
    }
    auto issue = [&](int stage) {
      float *dst = wsm + (stage % W_STAGES) * W_STAGE_FLOATS + wave * LPW * 256;
#pragma unroll
      for (int j = 0; j < LPW; j++)
        __builtin_amdgcn_global_load_lds((glb_void_t *)(src[j] + stage * WK), (lds_void_t *)(dst + j * 256), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < W_STAGES - 1; p++)
      if (p < NS) issue(p);
#pragma unroll
    for (int st = 0; st < NS; st++) {
      const int ahead = (NS - 1 - st) < (W_STAGES - 2) ? (NS - 1 - st) : (W_STAGES - 2);
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); /* stage st has landed; stage st - 1's buffer is free */
      if (st + W_STAGES - 1 < NS) issue(st + W_STAGES - 1);
    }
    __syncthreads();
    return;
  }

  // ------------------------------------------------------------------ multiply
  const int wm = wave >> 1, wn = wave & 1; /* HALF: wm is the wave pair's half of each stage's K */
  // the gate values of this thread's RPT x 4 outputs in the epilogue, requested now
  constexpr int RPT = MT / 16, NU = HALF ? 4 : 8; /* rows per thread; fragment pairs per stage and wave */
  const int etid = threadIdx.x; /* 0..255 */
  const int rq = etid >> 4, c4 = (etid & 15) * 4;
  float xin[RPT][4];
#pragma unroll
  for (int rr = 0; rr < RPT; rr++) {
    const float *xrow = input_row<true>(v, row0 + m0 + RPT * rq + rr, t) + n0 + c4;
#pragma unroll
    for (int i = 0; i < 4; i++) xin[rr][i] = xrow[i];
  }
  f32x16 acc; /* (a second accumulator taking turns with this one measured no difference) */
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;
  const uint32_t arow = (uint32_t)((HALF ? 0 : wm * 32) + lm) * (WK * 4u), brow = (uint32_t)(MT + wn * 32 + lm) * (WK * 4u);
  auto rd = [&](int st, f32x4 (&a)[NU], f32x4 (&b)[NU]) {
    const uint32_t base = lds0 + (uint32_t)((st % W_STAGES) * W_STAGE_FLOATS) * 4u;
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const uint32_t off = (uint32_t)(((2 * (u + (HALF ? NU * wm : 0)) + kh) ^ (lm & 15)) * 16);
      a[u] = lds_read_b128(base + arow + off);
      b[u] = lds_read_b128(base + brow + off);
    }
  };
  auto step = [&](int st, f32x4 (&a)[NU], f32x4 (&b)[NU], f32x4 (&an)[NU], f32x4 (&bn)[NU]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this stage's fragments have arrived */
    if (st + 1 < NS) {
      __builtin_amdgcn_s_barrier(); /* stage st + 1 has landed */
      rd(st + 1, an, bn);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    f32x4 a0[NU], b0[NU], a1[NU], b1[NU];
    __builtin_amdgcn_s_barrier(); /* stage 0 has landed */
    rd(0, a0, b0);
#pragma unroll
    for (int st = 0; st < NS; st += 2) {
      step(st, a0, b0, a1, b1);
      if (st + 1 < NS) step(st + 1, a1, b1, a0, b0);
    }
  }
  // the tile through LDS (the ring buffer stage NS would have used was read four barriers ago); HALF: the two K
  // halves' tiles, [2][32][64], as rows 0 .. 31 and 32 .. 63 of the same image
  float *red = wsm + (NS % W_STAGES) * W_STAGE_FLOATS; /* [64][64] */
#pragma unroll
  for (int g = 0; g < 16; g++) {
    const int row = wm * 32 + (g & 3) + 8 * (g >> 2) + 4 * kh;
    red[row * WN + wn * 32 + lm] = acc[g];
  }
  __syncthreads();
  float *dst0 = v.b.ehi + ((size_t)(t + 1) * s.Scap + row0 + m0) * s.I + n0 + c4;
#pragma unroll
  for (int rr = 0; rr < RPT; rr++) {
    const int row = RPT * rq + rr;
    float4 e4 = ld4(red + row * WN + c4);
    if (HALF) {
      const float4 f4 = ld4(red + (32 + row) * WN + c4);
      e4.x += f4.x;
      e4.y += f4.y;
      e4.z += f4.z;
      e4.w += f4.w;
    }
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

struct ChainSync {
  unsigned tickets[8];       /* per XCD, monotonic over launches                     */
  unsigned pad0[24];
  unsigned flags[32][2][4][32]; /* [row tile][sub-chain][fetching wave][column tile]   */
  unsigned abort;            /* raised by any workgroup that gives up                */
};
#define PC_SYNC_CLEAR_BYTES sizeof(ChainSync)
/* The seat of the CU a workgroup finds itself on, by XCD and the CU's hardware number (HW_REG_HW_ID's se / sh / cu
 * bits): made once from the residency probe (chain_validate), 0xff: no such CU.  In device memory, never written
 * while a chain runs, and read through a CONSTANT-address-space pointer as the kernel's first instructions: one
 * scalar load beside the loads of the arguments and of the View.  (As a plain global pointer, and as a by-value
 * argument, hipcc made it a vector load behind everything else: +1.3 us per generation.) */
struct SeatTable {
  unsigned seat[8][256];
};
typedef const __attribute__((address_space(4))) unsigned *seat_tbl_p;
/* HW_REG_HW_ID (register 4) bits 8 .. 15: cu_id[3:0], sh_id, se_id[2:0] -- the CU's number within its XCD */
#define PC_HW_CU_KEY() (__builtin_amdgcn_s_getreg(((8 - 1) << 11) | (8 << 6) | 4) & 0xffu)

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

/* The top layer's weight delta in the chain launch's first microseconds (HoWork, k_common.h): 22 MFLOP that as
 * a launch of their own (k_gemm<ProbHoDelta>) cost 6 us of launch, ramp and three dependent memory round trips
 * on the generation's critical path.  Here the launch's workgroups share the rows of the delta, HR =
 * ceil(H / workers) each: all 256 of them before they request their weight panel / while their first operand
 * rows are in flight, or -- when at least half of the launch has no chain work, a small set -- only those
 * (hw.idle_only; the others start their chain at once).  Lane (output quad q4, group gl) of wave w sums
 * streams g = 4 w + gl, g + 32, ... for ALL HR rows: one float4 of o_error and HR hidden values per stream (a
 * thread per row instead had five lanes load every float4 again, and the texture path is paid by the lane:
 * 3.6 us, now 2.5).  The four groups of a wave are added by shuffles, the eight waves in order through LDS
 * (`lds`: 8 x HO_HR x 64 floats the caller does not need yet: the partial-tile area, which is unused until the
 * first multiply has finished, or the operand buffers of a workgroup without chain work).  A workgroup's 512
 * threads all call it, once (two barriers). */
#ifndef PC_ONE_SHARED_FETCH
#define PC_ONE_SHARED_FETCH 1
#endif
/* (chain_ho_sum / chain_ho_delta: k_common.h -- the weight-delta launch of the fused text step runs them too) */
/* ONE: row tiles of 16 streams, i.e. only sub-chain a exists and every other half-step is empty
 * (the workgroup multiplies, then finishes and publishes, then waits for the 32 producers of its
 * next operand): 4.5 instead of 6.4 us per step for HALF the streams per workgroup -- worse per
 * stream, but a small set (64 streams at hidden 1024: a GPU's share of 512 on eight) then runs
 * on twice as many CUs.  The launcher picks it when the 16-stream tiles still fit one launch. */
/* PAD (with ONE): the set is not whole row tiles -- only rows [vlo, nvalid) of the launch are its
 * own; the others belong to other streams or to nobody: they are multiplied like the rest (rows
 * do not mix) and never stored. */
template <int ACT, int K, bool ONE = false, bool PAD = false, bool XD = false> /* rnn_activation; hidden size: 1024, 512 or 256; XD: the tail for DENSE inputs (an instantiation of its own: the text step's kernel stays what it is) */
__global__ __launch_bounds__(512) void k_chain_persist(const View *__restrict__ vp, int uniform_idx,
                                                       int row0, int nrows, int depth, unsigned seq,
                                                       ChainSync *sy, unsigned *host_abort, int nvalid, int vlo,
                                                       HoWork hw, XcWork xc, int static_map, unsigned tseq, const unsigned *seats) {
  /* above the noise generator's waves (priority 0), which share four SIMDs with workgroups of this launch while the set
   * has presynaptic noise: the launch runs at the pace of its slowest workgroup (multi-head step, 256 / 32 streams:
   * 369 -> 360 / 221 -> 218 us per generation; nothing else changes) */
  __builtin_amdgcn_s_setprio(2);
  /* (first: see SeatTable) */
  const unsigned hw_xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; /* HW_REG_XCC_ID */
  const unsigned hw_seat = static_map == 2 ? ((seat_tbl_p)seats)[hw_xcc * 256u + PC_HW_CU_KEY()] : 0u;
  extern __shared__ __attribute__((aligned(16))) float psm[];
  constexpr int BUF = PC_SUB * K;             /* one sub-chain's operand (64 KB at K = 1024) */
  constexpr int NT = K / 32;                  /* column tiles of a row tile            */
  constexpr int KB = K / 64;                  /* 16-k blocks of a wave's K quarter     */
  constexpr int PPR = K / 256;                /* 1 KB LDS-DMA pieces per operand row   */
  float *abuf = psm;                          /* [2][16][K], swizzled chunks           */
  float *red = psm + 2 * BUF;                 /* [2][4 waves][16 rows][32 cols]        */
  unsigned *wg_info = reinterpret_cast<unsigned *>(red + 2 * PC_RED_FLOATS);
  XC_STAMP(0);
  BND_MARK(g_bnd_chain, 0);
  PRO_STAMP_DEP(16, 256, nrows);        /* the launch's arguments are here */
  PRO_STAMP_DEP(17, 256, (int)hw_seat); /* the seat */
  View v = *vp;
  PRO_STAMP_DEP(18, 256, v.sh.H);       /* the view */
  v.b.uniform_idx = uniform_idx;
  const RamdShape &s = v.sh;
  const int wave8 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int TR = ONE ? PC_SUB : 2 * PC_SUB; /* streams of a row tile */
  const int mtiles = nrows / TR;

  // --- which XCD am I on, and which of its 32 seats do I get.  static_map (the process's residency probe found the
  // launch's workgroups dealt to the XCDs in turn, workgroup i on XCD i % 8: chain_validate): XCD and seat follow
  // from the workgroup number, no ticket, no barrier -- each wave checks its XCD against the register, and a
  // mismatch gives the launch up (never silently wrong).  Otherwise: a ticket on the XCD the workgroup finds itself on.
  unsigned seat, my_xcc;
  if (static_map == 2) {
    /* the seat of the CU this workgroup runs on (one workgroup per CU: 137 KB of LDS): whatever order the dispatcher
     * dealt the workgroups in, and whatever else it dealt between them -- no atomic, no barrier, one scalar load */
    my_xcc = hw_xcc;
    seat = hw_seat;
  } else if (static_map) {
    my_xcc = blockIdx.x & 7u;
    seat = blockIdx.x >> 3;
    const unsigned real = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; /* HW_REG_XCC_ID */
    if (real != my_xcc) seat = 32u;
  } else {
    if (threadIdx.x == 0) {
      const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; /* HW_REG_XCC_ID */
      const unsigned t = __hip_atomic_fetch_add(&sy->tickets[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                         (tseq - 1u) * 32u; /* (tseq: this launch's number among those that draw tickets) */
      wg_info[0] = xcc;
      wg_info[1] = t;
    }
    __syncthreads();
    /* (wave-uniform by construction: say so, or every address built from them goes through the vector ALU) */
    seat = __builtin_amdgcn_readfirstlane(wg_info[1]);
    my_xcc = __builtin_amdgcn_readfirstlane(wg_info[0]);
  }
  /* row tile = XCD + 8 x (seat / column tiles): the first 8 row tiles spread over the 8 XCDs
   * before any XCD takes a second one; column tile = seat % column tiles */
  const int g = (int)my_xcc + 8 * (int)(seat / NT);
  if (seat >= 32u) { /* cannot happen with one workgroup per CU on a 256-CU part */
    if (threadIdx.x == 0) {
      __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(host_abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return; /* (the launcher repeats the call's work another way, the top layer's delta included) */
  }
  if (g >= mtiles) { /* fewer row tiles than seats: no chain work here, only a share of the top layer's delta */
    if (hw.dst && hw.idle_only) {
      /* a small set: the workgroups without chain work share the rows among themselves (the others start their
       * chain at once).  Rank among them from (XCD, seat): XCD x' has its row-tile slots r0(x') .. SP - 1 idle. */
      constexpr int SP = 32 / NT; /* row-tile slots of an XCD */
      const int x = (int)my_xcc;
      auto r0 = [&](int xx) { const int r = (mtiles - xx + 7) / 8; return r < 0 ? 0 : r > SP ? SP : r; };
      int rank = (int)seat - r0(x) * NT;
      for (int xx = 0; xx < x; xx++) rank += (SP - r0(xx)) * NT;
      chain_ho_delta<9, 4>(v, hw, psm, rank);
    } else if (hw.dst) {
      chain_ho_delta<5, HO_BATCH>(v, hw, red, (int)blockIdx.x);
    }
    return;
  }
  const bool ho_here = hw.dst && !hw.idle_only; /* (kernel arguments: the same for every thread) */
  const int j = (int)(seat % NT);
  const int m0 = TR * g, n0 = 1 + 32 * j;     /* output columns start at 1 */
  const unsigned epoch0 = seq * PC_EPOCH;
  const int halfsteps = 2 * depth;
  const int tn = NT;
  const size_t plane_stride = (size_t)s.Scap * s.I;
  /* the tail's first stream (row tile's stream j): what it can ask for now (extras_tail_prefetch) */
  const bool xc_first = !XD && xc.on && j < TR && !(PAD && (m0 + j < vlo || m0 + j >= nvalid));
  TailPre tpre;
  PRO_STAMP(19, 256);
  if (xc_first) tpre = extras_tail_prefetch<512>(v, row0 + m0 + j, row0 + m0 + j - xc.row0, xc.nx, xc.active);
  PRO_STAMP(20, 256);

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
    if (ho_here) chain_ho_delta<5, HO_BATCH>(v, hw, red, (int)blockIdx.x); /* before the panel's loads: its own loads need the registers, and return first */
    PRO_STAMP(11, 256);
    float wreg[KB][4][2];
    {
      const float *wb = v.b.ih_w + (size_t)(n0 + m) * s.H + 1 + (K / 4) * wv + 4 * kq;
#pragma unroll
      for (int u = 0; u < KB; u++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int h = 0; h < 2; h++) wreg[u][i][h] = wb[(size_t)16 * h * s.H + 16 * u + i];
      PRO_STAMP(12, 256);
      /* The panel has to have LANDED before the loop, as far as hipcc can tell: otherwise it puts the
       * `s_waitcnt vmcnt(0)` for these loads in front of the loop's first MFMA, where it waits in EVERY
       * half-step for the gate loads issued just before (inline asm, not on its scoreboard).  An empty
       * asm that reads the registers makes it wait here. */
#pragma unroll
      for (int u = 0; u < KB; u++)
        asm volatile("" : : "v"(wreg[u][0][0]), "v"(wreg[u][0][1]), "v"(wreg[u][1][0]), "v"(wreg[u][1][1]),
                     "v"(wreg[u][2][0]), "v"(wreg[u][2][1]), "v"(wreg[u][3][0]), "v"(wreg[u][3][1]));
      PRO_STAMP(13, 256);
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
    /* ONE: in the half-step that multiplies nothing this wave fetches the second half of its row group's
     * pieces of the next operand itself (rows 4 wv + 2, + 3 at K = 1024), the fetching wave the first half:
     * sixteen LDS-DMA instructions per wave took 0.4 us to issue, on the critical path of every step */
    unsigned hvoff[ONE ? 2 * PPR : 1];
    if constexpr (ONE) {
#pragma unroll
      for (int i = 0; i < 2 * PPR; i++) {
        const int ii = 2 * PPR + i, r = 4 * wv + ii / PPR;
        const int c = (64 * (ii % PPR) + lane) ^ r;
        hvoff[i] = (unsigned)(((size_t)r * s.I + 1 + 4 * c) * sizeof(float));
      }
    }
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
    bool mdead = false; /* somebody gave up: no more polling, only the barriers */
    __syncthreads(); /* barrier 0: both operands of the first two half-steps have landed */
    PRO_STAMP(14, 256);
    PRO_CLOCKS(24, 256);

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
        if constexpr (ONE) {
          if (k >= 1 && k + 1 < halfsteps && PC_ONE_SHARED_FETCH) {
            /* the flags of this wave's row group (the same ones its fetching wave polls), then its pieces */
            const unsigned want = epoch0 + (unsigned)((k - 1) >> 1) + 1u;
            const char *fbase = uniform_ptr(&sy->flags[g][0][wv][0]);
            const unsigned poff = (unsigned)((lane % NT) * sizeof(unsigned));
            for (unsigned spins = 0; !mdead; spins++) { /* (the fetching waves raise the alarm; this wave only heeds it) */
              unsigned got;
              asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(poff), "s"(fbase) : "memory");
              if (__all(got >= want)) break;
              if ((spins & 1023u) == 1023u) {
                const unsigned ab = __hip_atomic_load((gu32 *)&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__any(ab != 0u) || spins > (1u << 21)) mdead = true;
              }
              __builtin_amdgcn_s_sleep(PC_SLEEP1);
            }
            const char *base = uniform_ptr(ehi_sub + (size_t)((k + 1) >> 1) * plane_stride);
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr(abuf + 4 * wv * K));
#pragma unroll
            for (int i = 0; i < 2 * PPR; i++) {
              const int ii = 2 * PPR + i;
              lds_dma16_sc1(base, hvoff[i], dst + (uint32_t)(((ii / PPR) * K + 256 * (ii % PPR)) * sizeof(float)));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
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
  } else {
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
  auto fetch = [&](int x, int plane, bool first_half_only = false) {
    const char *base = reinterpret_cast<const char *>(sub_base + (size_t)plane * plane_stride +
                                                      (size_t)x * PC_SUB * s.I);
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr(abuf + x * BUF + 4 * lw * K));
#pragma unroll
    for (int i = 0; i < 4 * PPR; i++)
      if (i < 2 * PPR || !first_half_only) /* (ONE: the multiplying wave of the row group fetches the other half) */
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
  if (ho_here) chain_ho_delta<5, HO_BATCH>(v, hw, red, (int)blockIdx.x); /* while the first operand rows are on their way */
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PRO_STAMP(15, 0);
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
      if (!dead) fetch((k + 1) & 1, (k + 1) >> 1, ONE && PC_ONE_SHARED_FETCH);
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

  // ================================================ tail: the extras and the control logic
  // (XcWork, k_common.h.)  Stream i of the row tile belongs to column tile i % NT: at hidden 1024 with
  // 32-stream tiles every workgroup has exactly one.  A stream's rows of ALL error planes must have been
  // published by all NT column tiles: the flags of its row group at the last step (earlier ones were
  // raised before it, in program order behind a drained store queue) -- and only the item of the last
  // plane has to wait for them: every earlier plane was seen complete by this workgroup's own fetching
  // waves before a barrier that all its waves have passed.  The rows are read past the L1 (sc1 loads) from
  // this XCD's L2, where the row tile's planes were written.
  if (!xc.on) return;
  XC_STAMP(1);
  XC_STAMP(10);
  PRO_CLOCKS(26, 256);
  /* (the sums' scratch is the operand area: nobody reads it behind the loop's last barrier, while the
   * multiplying waves' last finish still reads `red` -- no barrier in front of the tail) */
  for (int i = j; i < TR; i += NT) {
    const int sr = m0 + i; /* row within the launch */
    if (PAD && (sr < vlo || sr >= nvalid)) continue; /* (kernel arguments and the seat: uniform over the workgroup) */
    const int x = i / PC_SUB, rg = (i % PC_SUB) >> 2;
    auto wait_last = [&]() {
      const unsigned want = epoch0 + (unsigned)depth;
      const char *fbase = uniform_ptr(&sy->flags[g][x][rg][0]);
      const unsigned poff = (unsigned)((lane % NT) * sizeof(unsigned));
      for (unsigned spins = 0;; spins++) {
        unsigned got;
        asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(poff), "s"(fbase) : "memory");
        if (__all(got >= want)) break;
        if ((spins & 1023u) == 1023u) {
          const unsigned ab = __hip_atomic_load((gu32 *)&sy->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__any(ab != 0u) || spins > (1u << 21)) { /* (the launch's results are discarded: see dsync) */
            if (lane == 0) {
              __hip_atomic_store((gu32 *)&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(host_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            break;
          }
        }
        __builtin_amdgcn_s_sleep(PC_SLEEP1);
      }
      XC_STAMP(2);
    };
    constexpr int XQ = (K / 4 + 1 + 63) / 64; /* float4 per lane of an error row (h_size = K + 4) */
    const int r = row0 + sr;
    if constexpr (XD) /* (the operand area holds its 52 KB from hidden 512 on: launcher) */
      extras_dense_tail<512>(v, r, r - xc.row0, xc.nx, xc.nxp, xc.active, xc.flags, abuf, wait_last);
    else
      extras_control_tail<XQ, 512>(v, r, r - xc.row0, xc.nx, xc.nxp, xc.active, xc.flags, abuf, (i == j && xc_first) ? &tpre : nullptr, wait_last);
    XC_STAMP(9);
    __syncthreads();
  }
  BND_MARK(g_bnd_chain, 1);
}

/* the device copy of the View for the kernels that take it by pointer, rewritten only when
 * it changes (the ring position is not part of it: those kernels get it as an argument) */
const View *device_view(hipStream_t st, const View &v) {
  static View *d_view = nullptr;
  static View h_view;
  static bool have = false;
  View cur = v;
  cur.b.uniform_idx = 0;
  /* what changes from call to call and none of those kernels reads (the speculated noise's buffers change places every
   * generation; the flags are per pass): left out, or every generation of a noisy net would rewrite the copy twice
   * (5 us each, seen in the multi-head generation's timeline) */
  cur.b.noise_spec = nullptr;
  cur.b.rng_spec = nullptr;
  cur.b.noise_spec_use = 0;
  cur.b.dense_inputs = 0;
  if (!d_view) HIP_CHECK(hipMalloc(&d_view, sizeof(View)));
  if (!have || memcmp(&cur, &h_view, sizeof(View)) != 0) {
    RAMD_LAUNCH(k_store_view, dim3(1), dim3(1), 0, st, cur, d_view);
    h_view = cur;
    have = true;
  }
  return d_view;
}

/* ---- the one-launch chain (k_chain_persist): its device state and the abort word (above) ---- */
/* A launch whose workgroups wait for one another (the one-launch chain) needs all 256 of them resident together,
 * one per CU.  Whether this process's device and queue grant that -- no CU mask, no
 * partition mode that still reports 256 CUs, no co-tenant holding CUs -- is found out ONCE, by a probe launch of the
 * same footprint (512 threads and the chain's LDS per workgroup, so one per CU) whose workgroups count themselves
 * and wait, bounded, for the count to reach 256 (chain_validate).  It also records the XCD each workgroup found
 * itself on: where workgroup i sits on XCD i % 8 -- the dispatcher deals workgroups to the XCDs in turn -- the
 * kernels take XCD and seat from the workgroup number (g_xcd_static; every wave still checks its XCD against the
 * register) instead of drawing a ticket.  A probe that fails switches the kernel off for the process; the
 * launch-per-step chain takes its place from the first call on.  A give-up in
 * mid-run -- a co-tenant that arrives later -- is caught at the next synchronisation (rnn_core.c: dsync). */
static bool g_chain_validated = false, g_chain_broken = false, g_xcd_static = false, g_seat_table = false;
static unsigned *g_seats = nullptr; /* SeatTable, device */

struct ProbeOut {
  unsigned arrived, fail;
  unsigned xcc[256];
  unsigned cu_key[256];
};
__global__ __launch_bounds__(512) void k_residency_probe(ProbeOut *p) {
  extern __shared__ float probe_lds[]; /* (sized by the launch: what makes it one workgroup per CU) */
  if (threadIdx.x != 0) return;
  p->xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; /* HW_REG_XCC_ID */
  p->cu_key[blockIdx.x] = PC_HW_CU_KEY();
  __hip_atomic_fetch_add(&p->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned spins = 0; __hip_atomic_load(&p->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; spins++) {
    if (spins > (1u << 20) || __hip_atomic_load(&p->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { /* ~0.5 s */
      __hip_atomic_store(&p->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

#ifdef PC_STAMPS
extern "C" void ramd_chain_stamps(unsigned long long *out) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_stamps), sizeof(unsigned long long) * 2 * 64 * 8));
}
extern "C" void ramd_chain_tail_stamps(unsigned long long *out) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xc_stamps), sizeof(unsigned long long) * 32));
}
#endif

static ChainSync *g_chain_sync = nullptr;
static unsigned *g_chain_abort_host = nullptr, *g_chain_abort_dev = nullptr;
static unsigned g_chain_seq = 0;
static int g_chain_cus = -1;
/* the probe (see above), once per process */
static void chain_validate(hipStream_t st) {
  if (g_chain_validated || g_chain_broken) return;
  if (g_chain_cus < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    g_chain_cus = prop.multiProcessorCount;
  }
  if (g_chain_cus != 256) { /* 8 XCDs x 32 CUs: one workgroup per CU, 32 seats per XCD */
    g_chain_broken = true;
    return;
  }
  if (!g_chain_sync) {
    HIP_CHECK(hipMalloc(&g_chain_sync, sizeof(ChainSync)));
    HIP_CHECK(hipMemset(g_chain_sync, 0, PC_SYNC_CLEAR_BYTES));
    ramd_abort_word_dev();
  }
  ProbeOut *d_probe = nullptr, h_probe;
  HIP_CHECK(hipMalloc(&d_probe, sizeof(ProbeOut)));
  HIP_CHECK(hipMemsetAsync(d_probe, 0, sizeof(ProbeOut), st));
  HIP_CHECK(hipFuncSetAttribute((const void *)k_residency_probe, hipFuncAttributeMaxDynamicSharedMemorySize, pc_lds_bytes(1024)));
  RAMD_LAUNCH(k_residency_probe, dim3(256), dim3(512), pc_lds_bytes(1024), st, d_probe);
  HIP_CHECK(hipMemcpyAsync(&h_probe, d_probe, sizeof(ProbeOut), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  HIP_CHECK(hipFree(d_probe));
  /* (RECUR_AMD_CHAIN_TEST_GIVEUP=1: the tests' way of taking the failure branch on a healthy device) */
  if (h_probe.fail || h_probe.arrived != 256u || env_int("RECUR_AMD_CHAIN_TEST_GIVEUP", 0) == 1) {
    fprintf(stderr, "librecur_amd: the residency probe failed (%u of 256 workgroups resident together, one per CU); "
                    "using the launch-per-step chain from here on\n", h_probe.arrived);
    g_chain_broken = true;
    return;
  }
  bool in_turn = true;
  for (int i = 0; i < 256; i++) in_turn = in_turn && h_probe.xcc[i] == (unsigned)(i & 7);
  /* XCD and seat from the workgroup NUMBER: only on request (RECUR_AMD_XCD_STATIC=1) -- anything else that runs on the
   * GPU beside the chain, in this process or another, makes the dispatcher interleave the launches, and a chain whose
   * workgroup is not where its number says gives the launch up (abort code 2) */
  g_xcd_static = in_turn && env_int("RECUR_AMD_XCD_STATIC", 0);
  /* XCD and seat from the CU the workgroup runs on (the default): the 256 workgroups of the probe, one per CU, named
   * 32 different CUs on each XCD -- their hardware numbers in ascending order are the seats */
  {
    static SeatTable h_seats;
    unsigned (*tbl)[256] = h_seats.seat;
    memset(&h_seats, 0xff, sizeof(h_seats));
    bool ok = true;
    for (int x = 0; x < 8 && ok; x++) {
      unsigned keys[256];
      int n = 0;
      for (int i = 0; i < 256; i++)
        if (h_probe.xcc[i] == (unsigned)x) keys[n++] = h_probe.cu_key[i] & 0xffu;
      if (n != 32) ok = false;
      for (int a = 1; a < n; a++) /* insertion sort */
        for (int c = a; c > 0 && keys[c - 1] > keys[c]; c--) {
          const unsigned t = keys[c];
          keys[c] = keys[c - 1];
          keys[c - 1] = t;
        }
      for (int a = 0; a < n && ok; a++) {
        if (a > 0 && keys[a] == keys[a - 1]) ok = false; /* two workgroups on one CU number: not what this relies on */
        tbl[x][keys[a]] = (unsigned)a;
      }
    }
    g_seat_table = ok && env_int("RECUR_AMD_XCD_TABLE", 1);
    if (g_seat_table) {
      if (!g_seats) HIP_CHECK(hipMalloc(&g_seats, sizeof(SeatTable)));
      HIP_CHECK(hipMemcpy(g_seats, &h_seats, sizeof(SeatTable), hipMemcpyHostToDevice));
    }
  }
  g_chain_validated = true;
}

/* Another queue of this process may have work on the device beside the chain (the noise generated ahead on a side
 * stream, an exchange overlapped on a communication stream): the dispatcher then deals the workgroups of BOTH
 * launches to the XCDs in turn, and workgroup i of the chain is not on XCD i % 8 (found by the configs[3] test, whose
 * every wave's check raised the abort word).  rnn_core.c says so when it creates such a stream; from then on the
 * chain draws tickets. */
static bool g_side_streams = false;
static unsigned g_ticket_launches = 0; /* launches that drew tickets: 32 per XCD each */
extern "C" void ramd_note_side_stream(void) { g_side_streams = true; }

/* the host-mapped abort word's device address, for other bounded device-side waits (the exchange's barriers) */
extern "C" unsigned *ramd_abort_word_dev(void) {
  if (!g_chain_abort_host) {
    HIP_CHECK(hipHostMalloc((void **)&g_chain_abort_host, 64, hipHostMallocMapped));
    *g_chain_abort_host = 0;
    HIP_CHECK(hipHostGetDevicePointer((void **)&g_chain_abort_dev, g_chain_abort_host, 0));
  }
  return g_chain_abort_dev;
}

extern "C" unsigned ramd_chain_abort_word(void) {
  return g_chain_abort_host ? *(volatile unsigned *)g_chain_abort_host : 0u;
}

static bool chain_persist_ok(hipStream_t st, const RamdShape *sh, const RamdBuffers *b, int nrows) {
  const int hs = sh->hidden_size;
  if (b->uniform_idx < 0 || (hs != 1024 && hs != 512 && hs != 256) || nrows < 1 || nrows % 16 != 0 ||
      sh->D > 60 || g_chain_broken || !env_int("RECUR_AMD_CHAIN_PERSIST", 1))
    return false;
  chain_validate(st);
  return g_chain_validated;
}

template <int ACT, int K>
static void launch_chain_persist_k(hipStream_t st, const View *d_view, const RamdShape *sh,
                                   const RamdBuffers *b, int row0, int nrows, unsigned seq, bool one, int nvalid,
                                   int vlo, const HoWork &hw, const XcWork &xc) {
  /* 2: seat from the CU's hardware number (robust beside other work); 1: from the workgroup number (on request, and
   * not beside this process's own side streams); 0: a ticket per workgroup */
  const int use_static = g_seat_table ? 2 : (g_xcd_static && !g_side_streams) ? 1 : 0;
  const unsigned tseq = use_static ? 0u : ++g_ticket_launches;
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
  if constexpr (K >= 512) {
    if (xc.on && xc.dense) { /* the tail for dense inputs */
      static bool attr_xd = false;
      if (!attr_xd) {
        HIP_CHECK(hipFuncSetAttribute((const void *)(k_chain_persist<ACT, K, false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, pc_lds_bytes(K)));
        HIP_CHECK(hipFuncSetAttribute((const void *)(k_chain_persist<ACT, K, true, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, pc_lds_bytes(K)));
        HIP_CHECK(hipFuncSetAttribute((const void *)(k_chain_persist<ACT, K, true, true, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, pc_lds_bytes(K)));
        attr_xd = true;
      }
      if (one && (nvalid < nrows || vlo > 0))
        RAMD_LAUNCH((k_chain_persist<ACT, K, true, true, true>), dim3(256), dim3(512), pc_lds_bytes(K), st, d_view,
                    b->uniform_idx, row0, nrows, sh->D, seq, g_chain_sync, g_chain_abort_dev, nvalid, vlo, hw, xc, (int)use_static, tseq, g_seats);
      else if (one)
        RAMD_LAUNCH((k_chain_persist<ACT, K, true, false, true>), dim3(256), dim3(512), pc_lds_bytes(K), st, d_view,
                    b->uniform_idx, row0, nrows, sh->D, seq, g_chain_sync, g_chain_abort_dev, nrows, 0, hw, xc, (int)use_static, tseq, g_seats);
      else
        RAMD_LAUNCH((k_chain_persist<ACT, K, false, false, true>), dim3(256), dim3(512), pc_lds_bytes(K), st, d_view,
                    b->uniform_idx, row0, nrows, sh->D, seq, g_chain_sync, g_chain_abort_dev, nrows, 0, hw, xc, (int)use_static, tseq, g_seats);
      return;
    }
  }
  if (one && (nvalid < nrows || vlo > 0))
    RAMD_LAUNCH((k_chain_persist<ACT, K, true, true>), dim3(256), dim3(512), pc_lds_bytes(K), st, d_view,
                b->uniform_idx, row0, nrows, sh->D, seq, g_chain_sync, g_chain_abort_dev, nvalid, vlo, hw, xc, (int)use_static, tseq, g_seats);
  else if (one)
    RAMD_LAUNCH((k_chain_persist<ACT, K, true>), dim3(256), dim3(512), pc_lds_bytes(K), st, d_view,
                b->uniform_idx, row0, nrows, sh->D, seq, g_chain_sync, g_chain_abort_dev, nrows, 0, hw, xc, (int)use_static, tseq, g_seats);
  else
    RAMD_LAUNCH((k_chain_persist<ACT, K, false>), dim3(256), dim3(512), pc_lds_bytes(K), st, d_view,
                b->uniform_idx, row0, nrows, sh->D, seq, g_chain_sync, g_chain_abort_dev, nrows, 0, hw, xc, (int)use_static, tseq, g_seats);
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

/* `ho`: a pending request for the top layer's delta rides in this launch (and is marked done when the launch stands) */
static bool launch_chain_persist(hipStream_t st, const View *d_view, const RamdShape *sh,
                                 const RamdBuffers *b, int row0, int nrows, bool one, int nvalid, int vlo, HoWork *ho,
                                 XcWork *xcp) {
  HoWork hw = {};
  XcWork xc = {};
  /* RECUR_AMD_CHAIN_CHECK=1 -- for a GPU that is shared, where a co-tenant may take CUs in mid-run: every chain
   * launch is followed by a synchronisation and a look at the abort word.  A launch that gave up is then not fatal:
   * the word is reset, the one-launch chain is switched off for the process and THIS call's chain runs again a
   * launch per step (the one-launch chain reads error plane 0 and writes planes >= 1 only: its input is intact).
   * The extras and the control logic do not ride in a launch that may be discarded (they update per-stream state --
   * min_error_factor, the depth statistics -- that a repeat would update twice).  Costs the host-side pipelining:
   * ~15 us per generation.  RECUR_AMD_CHAIN_TEST_GIVEUP=n (n > 1) implies it and pretends that the n-th launch
   * gave up (tests). */
  const int test_giveup = env_int("RECUR_AMD_CHAIN_TEST_GIVEUP", 0);
  const bool checked = env_int("RECUR_AMD_CHAIN_CHECK", 0) || test_giveup > 1;
  if (xcp && !checked) xc = *xcp;
  if (xcp && !xc.on) xcp->on = 0; /* (the caller then runs the extras as a launch of their own) */
  if (ho && !ho->done) {
    hw = *ho;
    /* the launch's workgroups without chain work, if they are at least half of it, else all 256 */
    const int busy = (nrows / (one ? 16 : 32)) * (sh->hidden_size / 32);
    hw.idle_only = 256 - busy >= 128;
    hw.workers = hw.idle_only ? 256 - busy : 256;
  }
  if (g_chain_seq >= (1u << 25)) { /* flags are seq * 64 + step and compare as unsigned numbers: start over */
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipMemset(g_chain_sync, 0, PC_SYNC_CLEAR_BYTES));
    g_chain_seq = 0;
    g_ticket_launches = 0;
  }
  const unsigned seq = ++g_chain_seq;
  int ev = timing_begin(st, T_CHAIN, 1);
#define CHAIN_PERSIST(ACT)                                                                  \
  do {                                                                                      \
    if (sh->hidden_size == 1024) launch_chain_persist_k<ACT, 1024>(st, d_view, sh, b, row0, nrows, seq, one, nvalid, vlo, hw, xc); \
    else if (sh->hidden_size == 512) launch_chain_persist_k<ACT, 512>(st, d_view, sh, b, row0, nrows, seq, one, nvalid, vlo, hw, xc); \
    else launch_chain_persist_k<ACT, 256>(st, d_view, sh, b, row0, nrows, seq, one, nvalid, vlo, hw, xc);         \
  } while (0)
  if (sh->activation == 2) CHAIN_PERSIST(2);
  else if (sh->activation == 5) CHAIN_PERSIST(5);
  else CHAIN_PERSIST(1);
#undef CHAIN_PERSIST
  timing_end(st, ev);
  if (checked) {
    static int checked_launches = 0;
    HIP_CHECK(hipStreamSynchronize(st));
    const unsigned word = *(volatile unsigned *)g_chain_abort_host;
    if (word || ++checked_launches == test_giveup) {
      fprintf(stderr, "librecur_amd: a one-launch BPTT chain gave up in mid-run (code %u: its 256 workgroups were no longer "
                      "all resident, or a hand-off timed out); this call's chain runs again a launch per step, which is "
                      "what the process uses from here on\n", word);
      *(volatile unsigned *)g_chain_abort_host = 0;
      HIP_CHECK(hipMemset(g_chain_sync, 0, PC_SYNC_CLEAR_BYTES));
      g_chain_seq = 0;
      g_ticket_launches = 0;
      g_chain_broken = true;
      return false;
    }
  }
  if (ho && hw.dst) ho->done = 1;
  return true;
}

/* All D steps of the chain for streams [row0, row0 + nrows) (the part of rnn_bptt_calc_deltas between the
 * top layer's backprop and the extras): the one-launch chain where it applies, otherwise a launch per
 * step with 64 x 64 or 32 x 32 tiles.  Returns the partial sums of squares per (step, stream) it left. */
int ramd_chain_steps(hipStream_t st, const View &v, const RamdShape *sh, const RamdBuffers *b, int row0,
                     int nrows, HoWork *ho, XcWork *xc) {
  const int tn = (sh->hidden_size + CN - 1) / CN;
  int tn_parts = tn; /* one per column tile of the chain kernel used */
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
  const bool windowed = span_base != row0 && span_base + span <= sh->Scap && chain_persist_ok(st, sh, b, span) &&
                        span / 16 <= chain_persist_seats(sh);
  bool windowed_done = false;
  if (windowed) {
    windowed_done = launch_chain_persist(st, d_view, sh, b, span_base, span, true, row0 - span_base + nrows, row0 - span_base, ho, xc);
  }
  bool persist = windowed_done || chain_persist_ok(st, sh, b, chain_rows);
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
      if (!launch_chain_persist(st, d_view, sh, b, row0 + r, n, one, real_left < n ? real_left : n, 0, ho, xc)) {
        persist = false; /* (RECUR_AMD_CHAIN_CHECK: the launch gave up; all of the call's rows go through the steps below) */
        break;
      }
      r += n;
    }
  }
  if (persist) tn_parts = 0; /* the one-launch chain leaves no partial sums: the extras sum the rows themselves */
  if (persist && xc && xc->on) xc->done = 1; /* every launch of the chain carried its rows' extras and control */
  /* big sets of a wide net: 64 x 64 tiles (k_chain_wide), one partial sum per 64 columns */
  const int wide_ns = sh->hidden_size / WK;
  /* ... as 32 x 64 tiles where that fills more of the chip: fewer than 192 tiles of 64 streams, and streams a multiple of 32 */
  const bool wide_half = !persist && b->uniform_idx >= 0 && nrows % 32 == 0 && sh->hidden_size % WN == 0 &&
                         (nrows / WM) * (sh->hidden_size / WN) < 192 && (nrows / 32) * (sh->hidden_size / WN) >= 128 &&
                         env_int("RECUR_AMD_CHAIN_WIDE_HALF", 1);
  const bool wide = !persist && b->uniform_idx >= 0 && (nrows % WM == 0 || wide_half) && sh->hidden_size % WN == 0 &&
                    (wide_ns == 16 || wide_ns == 24 || wide_ns == 32) &&
                    ((nrows / WM) * (sh->hidden_size / WN) >= 128 || wide_half) && env_int("RECUR_AMD_CHAIN_WIDE", 1);
  if (wide) {
    static bool attr_set = false;
    const size_t shm = (size_t)W_STAGES * W_STAGE_FLOATS * sizeof(float);
    if (!attr_set) {
      HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_wide<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_wide<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_wide<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      HIP_CHECK(hipFuncSetAttribute((const void *)(k_chain_wide<16, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      HIP_CHECK(hipFuncSetAttribute((const void *)(k_chain_wide<24, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      HIP_CHECK(hipFuncSetAttribute((const void *)(k_chain_wide<32, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      attr_set = true;
    }
    const int wtm = nrows / (wide_half ? 32 : WM), wtn = sh->hidden_size / WN;
    const int wblocks = ((wtn + 7) / 8) * 8 * wtm;
    tn_parts = wtn;
    int evw = timing_begin(st, T_CHAIN, sh->D);
    for (int t = 0; t < sh->D; t++) {
      if (wide_half && wide_ns == 32)
        RAMD_LAUNCH((k_chain_wide<32, 32>), dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, t, wtm, wtn);
      else if (wide_half && wide_ns == 24)
        RAMD_LAUNCH((k_chain_wide<24, 32>), dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, t, wtm, wtn);
      else if (wide_half)
        RAMD_LAUNCH((k_chain_wide<16, 32>), dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, t, wtm, wtn);
      else if (wide_ns == 32)
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
  return tn_parts;
}
