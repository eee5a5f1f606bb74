// kernels_bptt.hip -- the rest of rnn_bptt_calc_deltas: top-layer backprop, extras + control, the weight-delta GEMM,
// the one-workgroup small-net form, finalizes, and ramd_launch_calc_deltas which strings them together.
#include "k_common.h"
#include "k_gemm.h"

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

// backprop_single_layer_sparse (recur-nn.c:156-196) for a multi-head output layer as ONE fp32 MFMA GEMM over all
// streams: E_h[s][y] = sum_k o_error[s][k] W_ho[y][k] with K = the whole output row (3652 columns at 73 symbols x 50
// heads), instead of k_top_backprop_ranged's per-(stream, row, range) gathers (93 us, 480 MB of W_ho columns at
// 256 streams).  What makes that legitimate for the multi-head loss and for nothing else (launcher flag
// RAMD_RANGES_ARE_HEADS): the loss has cleared the error row outside the trained heads, so the columns between a
// stream's ranges contribute exact zeros, and its ranges -- runs of whole heads of >= 24 columns -- lie >= 16 columns
// apart.  The reference's error sum is NOT sum |e|: it adds |running value| after every range (recur-nn.c:178-191),
// so the running values at the range ends are needed, per (stream, row).  K therefore runs IN ORDER inside one
// accumulator chain per element (no split over waves or workgroups), in blocks of 16 columns, and after a block in
// which a stream's range ends -- the zeros up to the block boundary change nothing -- the lanes that hold that
// stream's rows add |accumulator| to their sums (a 32-bit mask per block says which streams: built in LDS from the
// range lists).  Workgroup = 32 streams x 32 rows: four multiplying waves with a 16 x 16 tile each
// (v_mfma_f32_16x16x4_f32) and four loader waves that stage both operands through an LDS ring in k_chain_main's
// manner, 64 columns at a time (as direct per-wave loads every wave fetched its own 16 + 16 rows: 87 us, bound by
// the traffic out of the L2); 264 workgroups at 256 streams, two per CU.  Rows whose hidden value is zero keep the stale entry of the last
// BPTT run and add nothing; the unscaled values go to error plane 0 and the per-tile sums to
// `part` for k_top_backprop_scale, as with k_top_backprop_ranged's shared-out form.
constexpr int TBH_K = 64, TBH_STAGES = 4;              /* columns per stage, stages of the ring: 64 KB, two workgroups per CU */
constexpr int TBH_STAGE_FLOATS = (32 + 32) * TBH_K;     /* 16 KB */
__global__ __launch_bounds__(512, 2) void k_top_backprop_heads(View v, int row0, int nrows, const int *ranges,
                                                               int range_stride, const unsigned char *active, float *part,
                                                               int nb, int tm) {
  __shared__ __attribute__((aligned(16))) float smem[TBH_STAGES * TBH_STAGE_FLOATS];
  extern __shared__ unsigned endmask[]; /* [KB] bit s: stream s0 + s has a range ending in this block */
  const RamdShape &s = v.sh;
  const int KB = (s.O + 15) / 16, nstages = (s.O + TBH_K - 1) / TBH_K;
  const int mt = blockIdx.x % tm, nt = blockIdx.x / tm;
  const int s0 = mt * 32, y0 = nt * 32;
  const int tid = threadIdx.x, wave8 = tid >> 6, lane = tid & 63;
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int wr = wave >> 1, wc = wave & 1, m = lane & 15, kq = lane >> 4;
  // --- loader waves: both operands are K-contiguous rows (32 error rows, 32 rows of W_ho), staged 64 columns at a
  // time by LDS-DMA the way k_chain_main stages its operands (kernels_chain.hip): instruction i of a stage fills rows
  // 4 i .. 4 i + 3 of A (i < 8) or of B, the 16-byte chunk c of row r at position c ^ (r & 15) (conflict-free
  // ds_read_b128 of sixteen rows at once); four instructions per loader wave and stage
  const float *src[4];
  int kcol[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int i = wave * 4 + j;
    const int row = 4 * (i & 7) + (lane >> 4);
    const int c = (lane & 15) ^ (row & 15);
    const float *base = (i < 8) ? v.b.o_error + (size_t)(row0 + min(s0 + row, nrows - 1)) * s.O
                                : v.b.ho_w + (size_t)min(y0 + row, s.H - 1) * s.O;
    src[j] = base + 4 * c;
    kcol[j] = 4 * c;
  }
  auto issue = [&](int stage) {
    float *dst = smem + (stage % TBH_STAGES) * TBH_STAGE_FLOATS + wave * 4 * 256;
    const int k0 = stage * TBH_K;
#pragma unroll
    for (int j = 0; j < 4; j++) { /* chunks past the row's end come from a zero line */
      const float *g = (k0 + kcol[j] + 4 <= s.O) ? src[j] + k0 : v.b.zeros;
      __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)(dst + j * 256), 16, 0, 0);
    }
  };
  if (loader) {
#pragma unroll
    for (int p = 0; p < TBH_STAGES - 1; p++)
      if (p < nstages) issue(p);
  }
  for (int kb = tid; kb < KB; kb += 512) endmask[kb] = 0u;
  __syncthreads();
  if (tid < 32 && s0 + tid < nrows) {
    const int *rg = ranges + (size_t)(s0 + tid) * range_stride;
    for (int i = 0; i < 64 && rg[2 * i] >= 0; i++) {
      const int end = (rg[2 * i] & ~3) + ((rg[2 * i + 1] + 3) & ~3);
      int kbe = (end + 15) / 16 - 1;
      kbe = kbe < 0 ? 0 : kbe >= KB ? KB - 1 : kbe;
      atomicOr(&endmask[kbe], 1u << tid);
    }
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float sabs[4] = {0.f, 0.f, 0.f, 0.f};
  if (loader) {
    for (int st = 0; st < nstages; st++) {
      const int ahead = min(TBH_STAGES - 2, nstages - 1 - st); /* stages that may stay in flight: 4 DMAs each */
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); /* stage st has landed; stage st - 1's buffer is free */
      if (st + TBH_STAGES - 1 < nstages) issue(st + TBH_STAGES - 1);
    }
    return;
  }
  const unsigned msh = 16u * (unsigned)wr + 4u * (unsigned)kq; /* this lane's four streams: bits msh .. msh + 3 */
  const uint32_t lds0 = lds_byte_addr(smem);
  const uint32_t a_off = (uint32_t)(16 * wr + m) * (TBH_K * 4u);
  const uint32_t b_off = (uint32_t)(32 * TBH_K) * 4u + (uint32_t)(16 * wc + m) * (TBH_K * 4u);
  for (int st = 0; st < nstages; st++) {
    __builtin_amdgcn_s_barrier(); /* stage st has landed */
    const uint32_t base = lds0 + (uint32_t)((st % TBH_STAGES) * TBH_STAGE_FLOATS) * 4u;
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t pos = (uint32_t)(((4 * u + kq) ^ m) * 16);
      a[u] = lds_read_b128(base + a_off + pos);
      b[u] = lds_read_b128(base + b_off + pos);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 :
                 : "memory");
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int kb = 4 * st + u;
      if (kb < KB) { /* (wave-uniform) */
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
        const unsigned em = (endmask[kb] >> (16u * (unsigned)wr)) & 0xffffu;
        if (em) { /* a range of one of this wave's streams ends here: |running value| joins its sum */
          const unsigned mine = (endmask[kb] >> msh) & 15u;
#pragma unroll
          for (int r = 0; r < 4; r++)
            if ((mine >> r) & 1u) sabs[r] += fabsf(acc[r]);
        }
      }
    }
  }
  // accumulator register r: stream 4 kq + r of the wave's sixteen, row (column of the tile) m
  const int y = y0 + 16 * wc + m;
  float psum[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int sj = s0 + 16 * wr + 4 * kq + r; /* within the call's rows */
    const bool valid = sj < nrows && y < s.H && !(active && !active[sj < nrows ? sj : 0]);
    const size_t rr = (size_t)(row0 + (sj < nrows ? sj : 0));
    const float hv = v.b.hidden[rr * s.H + (y < s.H ? y : 0)];
    const bool act = valid && y > 0 && hv != 0.0f;
    if (valid) {
      const float stale = v.b.err_a[rr * s.I + y];
      const float val = (y == 0) ? 0.0f : act ? acc[r] : stale;
      v.b.ehi[rr * s.I + y] = (y == 0 || y > s.hidden_size) ? 0.0f : val;
    }
    psum[r] = act ? sabs[r] : 0.0f;
  }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1)
#pragma unroll
    for (int r = 0; r < 4; r++) psum[r] += __shfl_xor(psum[r], off, 64);
  if (m == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int sj = s0 + 16 * wr + 4 * kq + r;
      if (sj < nrows) part[(size_t)sj * nb + 2 * nt + wc] = psum[r];
    }
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

// backprop_single_layer_sparse for the multi-head loss WITHOUT the zeros, in two launches.  A stream's error row is
// non-zero only in the heads it trained (its own and the few that leaked: ~6 of 50), so of k_top_backprop_heads' one
// GEMM over the whole output row (1.9 GFLOP, all of W_ho through every row tile's workgroups: 52 us at 1024 / 256 /
// 73 x 50) seven eighths are products with zeros.  Here
//   1. k_top_heads_partial: a workgroup takes head c and 128 rows of W_ho.  It lists the streams that trained the head
//      (the bit per head the loss left behind each range list, as k_ho_delta_heads does), stages W_ho[rows][head c]
//      once and the streams' error segments 32 streams at a time, and forms P[stream][c][row] = sum over the head's
//      columns on the matrix cores (wave w: rows 32 w .., v_mfma_f32_32x32x2_f32 with the streams as M).  W_ho is
//      read from memory exactly once.
//   2. k_top_heads_combine: a workgroup per stream walks its trained heads in ascending order, adds P[stream][c][.]
//      to the running values and, after a head that is not followed by the next one -- the end of one of the
//      reference's merged ranges -- adds |running value| to the row's share of the error sum (recur-nn.c:178-191);
//      then the stale entries, the sum, the soft clip (recur-nn.c:719-721) and the row, which is
//      what k_top_backprop_heads + k_top_backprop_scale left behind.
constexpr int THP_ROWS = 128, THP_U = 10;
__global__ __launch_bounds__(256) void k_top_heads_partial(View v, int row0, int nrows, const int *ranges, int range_stride,
                                                           const unsigned char *active, int alen, int ncls, float *P) {
  extern __shared__ float thp_sh[]; /* [THP_ROWS][ld] rows of W_ho, [32][ld] error segments */
  __shared__ short list[256];
  __shared__ int wcount[4];
  const RamdShape &s = v.sh;
  const int c = blockIdx.x, y0 = blockIdx.y * THP_ROWS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 31, kh = lane >> 5;
  const int col0 = c * alen, a0 = col0 & ~3, lead = col0 - a0; /* the aligned span starts `lead` columns before the head */
  const int span4 = (lead + alen + 3) >> 2;                     /* float4s of the span */
  const int ld = (4 * span4) | 1;
  float *w_sh = thp_sh, *e_sh = thp_sh + THP_ROWS * ld;
  /* the rows of W_ho: float4 t of the [THP_ROWS][span4] block (the span may reach past the head into the next one and
   * up to the padded row end: o_size is a multiple of 4).  Ten loads in flight per thread (all of the block at 73 symbols), no branch around a load
   * (see k_ho_delta_heads): what lies outside the block lands in a spare slot */
  const int dump = (THP_ROWS + 32) * ld;
  /* (the first 256 streams' head bits are asked for before the rows: one round trip less in the workgroup's chain) */
  const unsigned long long bits0 =
      *reinterpret_cast<const unsigned long long *>(ranges + (size_t)min(tid, nrows - 1) * range_stride + RAMD_HEADBITS_AT);
  const unsigned char live0 = active ? active[min(tid, nrows - 1)] : 1;
  for (int i0 = tid; i0 < THP_ROWS * span4; i0 += THP_U * 256) {
    float4 w[THP_U];
    int at[THP_U];
#pragma unroll
    for (int u = 0; u < THP_U; u++) {
      const int i = i0 + 256 * u;
      const bool in = i < THP_ROWS * span4;
      const int row = in ? i / span4 : 0, q = in ? i - row * span4 : 0;
      const int y = min(y0 + row, s.H - 1);
      w[u] = *reinterpret_cast<const float4 *>(v.b.ho_w + (size_t)y * s.O + a0 + 4 * q);
      at[u] = in ? row * ld + 4 * q : dump;
    }
#pragma unroll
    for (int u = 0; u < THP_U; u++) {
      float *d = w_sh + at[u];
      d[0] = w[u].x;
      d[1] = w[u].y;
      d[2] = w[u].z;
      d[3] = w[u].w;
    }
  }
  for (int base = 0; base < nrows; base += 256) {
    const int sj = base + tid, sjc = min(sj, nrows - 1);
    unsigned long long bits = bits0;
    unsigned char live = live0;
    if (base) {
      bits = *reinterpret_cast<const unsigned long long *>(ranges + (size_t)sjc * range_stride + RAMD_HEADBITS_AT);
      live = active ? active[sjc] : 1;
    }
    const bool mine = sj < nrows && live && ((bits >> c) & 1ull);
    const unsigned long long bal = __ballot(mine);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      if (w < wave) off += wcount[w];
      total += wcount[w];
    }
    if (mine) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (short)tid;
    __syncthreads();
    for (int k0 = 0; k0 < total; k0 += 32) {
      const int nk = min(32, total - k0);
      /* error segments: float4 q of stream k; what lies outside the head (the neighbours' columns) is cleared */
      {
        float4 e[5]; /* (32 streams of up to 34 float4s: 128 symbols) */
        int at[5], xs[5];
#pragma unroll
        for (int u = 0; u < 5; u++) {
          const int i = tid + 256 * u;
          const bool in = i < 32 * span4;
          const int k = in ? i / span4 : 0, q = in ? i - k * span4 : 0;
          const bool have = in && k < nk;
          e[u] = *reinterpret_cast<const float4 *>(v.b.o_error + (size_t)(row0 + base + list[k0 + (have ? k : 0)]) * s.O + a0 + 4 * q);
          at[u] = in ? k * ld + 4 * q : dump;
          xs[u] = have ? 4 * q - lead : -8; /* column of e.x within the head (-8: nothing of it) */
        }
#pragma unroll
        for (int u = 0; u < 5; u++) {
          float *d = e_sh + at[u] - (at[u] == dump ? THP_ROWS * ld : 0);
          const int x = xs[u];
          d[0] = (x >= 0 && x < alen) ? e[u].x : 0.0f;
          d[1] = (x + 1 >= 0 && x + 1 < alen) ? e[u].y : 0.0f;
          d[2] = (x + 2 >= 0 && x + 2 < alen) ? e[u].z : 0.0f;
          d[3] = (x + 3 >= 0 && x + 3 < alen) ? e[u].w : 0.0f;
        }
      }
      __syncthreads();
      f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const float *ea = e_sh + lm * ld + kh, *wb = w_sh + (32 * wave + lm) * ld + kh;
      const int kend = 4 * span4; /* even */
#pragma unroll 4
      for (int k = 0; k < kend; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[k], wb[k], acc, 0, 0, 0);
      const int y = y0 + 32 * wave + lm;
      if (y < s.H) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
          const int sr = (g & 3) + 8 * (g >> 2) + 4 * kh;
          if (sr < nk) P[((size_t)(row0 + base + list[k0 + sr]) * ncls + c) * s.H + y] = acc[g];
        }
      }
      __syncthreads(); /* before the next streams are staged (or the list rebuilt) */
    }
  }
}

constexpr int THC_THREADS = 320; /* a float4 of the row per thread: h_size <= 1280 in one pass (more: the loop) */
__global__ __launch_bounds__(THC_THREADS) void k_top_heads_combine(View v, int row0, const int *ranges, int range_stride,
                                                                   const unsigned char *active, int ncls, const float *P,
                                                                   int images_pending) {
  __shared__ float red[THC_THREADS / 64];
  __shared__ signed char heads[64];
  __shared__ unsigned char ends[64];
  const RamdShape &s = v.sh;
  const int j = blockIdx.x, r = row0 + j, tid = threadIdx.x;
  if (active && !active[j]) return;
  const unsigned long long bits =
      *reinterpret_cast<const unsigned long long *>(ranges + (size_t)j * range_stride + RAMD_HEADBITS_AT);
  const int nh = __popcll(bits);
  if (tid < 64 && ((bits >> tid) & 1ull)) {
    const int k = __popcll(bits & ((1ull << tid) - 1ull));
    heads[k] = (signed char)tid;
    ends[k] = tid == 63 || !((bits >> (tid + 1)) & 1ull); /* not followed by the next head: one of the merged ranges ends */
  }
  __syncthreads();
  const float *Pj = P + (size_t)r * ncls * s.H;
  float *dst = v.b.ehi + (size_t)r * s.I;
  /* the stale entries: err_a as the last BPTT run left it -- which, while k_err_writeback has not rebuilt the images,
   * is that run's error plane n (n its executed steps, even) or n - 1 (odd) in columns 1 .. hidden_size, the only ones
   * used here (k_err_writeback's rule; plane 0 is the row this thread is about to overwrite: read first).  A launch
   * less per generation of the multi-head step. */
  const float *stale_row = v.b.err_a + (size_t)r * s.I;
  if (images_pending) {
    const int n = v.b.n_exec[r];
    if (n > 0) stale_row = v.b.ehi + ((size_t)((n & 1) ? n - 1 : n) * s.Scap + r) * s.I;
  }
  float psum = 0.0f;
  for (int y4 = 4 * tid; y4 < s.H; y4 += 4 * THC_THREADS) { /* (h_size is a multiple of 4) */
    float e[4] = {0.f, 0.f, 0.f, 0.f}, sabs[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nh; k0 += 8) { /* eight heads' values requested together (a stream trains ~6) */
      float4 pv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) pv[u] = *reinterpret_cast<const float4 *>(Pj + (size_t)heads[min(k0 + u, nh - 1)] * s.H + y4);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (k0 + u < nh) {
          const float p[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w};
          const bool end = ends[k0 + u];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            e[i] += p[i];
            if (end) sabs[i] += fabsf(e[i]);
          }
        }
      }
    }
    const float4 hv4 = *reinterpret_cast<const float4 *>(v.b.hidden + (size_t)r * s.H + y4);
    const float4 st4 = *reinterpret_cast<const float4 *>(stale_row + y4);
    const float hv[4] = {hv4.x, hv4.y, hv4.z, hv4.w}, stale[4] = {st4.x, st4.y, st4.z, st4.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int y = y4 + i;
      const bool act = y > 0 && hv[i] != 0.0f;
      const float val = (y == 0) ? 0.0f : act ? e[i] : stale[i];
      o[i] = (y == 0 || y > s.hidden_size) ? 0.0f : val;
      if (act) psum += sabs[i];
    }
    *reinterpret_cast<float4 *>(dst + y4) = make_float4(o[0], o[1], o[2], o[3]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) psum += __shfl_xor(psum, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = psum;
  __syncthreads();
  float sum = 0.0f;
#pragma unroll
  for (int w = 0; w < THC_THREADS / 64; w++) sum += red[w];
  const float halfmax = s.H * MAX_TOP_ERROR_FACTOR_F;
  float scaled = sum;
  if (sum > halfmax) {
    const float scale = soft_clip_dev(sum, halfmax);
    scaled = scale * sum;
    for (int y4 = 4 * tid; y4 < s.H; y4 += 4 * THC_THREADS) { /* (this thread's own entries; 0 stays 0) */
      float4 o = *reinterpret_cast<float4 *>(dst + y4);
      o.x *= scale;
      o.y *= scale;
      o.z *= scale;
      o.w *= scale;
      *reinterpret_cast<float4 *>(dst + y4) = o;
    }
  }
  if (tid == 0) {
    v.b.top_raw[r] = sum;
    v.b.top_scaled[r] = scaled;
  }
}

__global__ void k_live_mask(float *dst, const unsigned char *active, int n) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = (!active || active[j]) ? 1.0f : 0.0f;
}

// The top layer's weight deltas under the multi-head loss (recur-nn.c:256-301 with per-stream range lists): of a
// stream's error row only the heads it trained are non-zero (a handful of, say, 50), so hidden^T . o_error as a dense
// GEMM over all streams is mostly products with zero.  Here a workgroup owns head c's columns of 32 rows of ho_delta:
// it lists the streams that trained head c (the bit per head the loss left behind each range list), in ascending order
// -- the order in which the reference adds the streams' products -- stages their error segments and their 32 hidden
// values in LDS, up to 48 streams at a time with every load requested at once, and sums from there on the matrix
// cores: wave w the 32 columns 32 w .. of the head.  The untrained heads' columns get their zeros (or keep what they
// had, accumulating) from the same store.  Heads of up to 128 symbols, any number of streams.
constexpr int HDH_U = 12, HDH_K = 48; /* (20 KB of LDS at 73 symbols: seven workgroups per CU, the whole grid of 50 x 32 resident at once) */
__global__ __launch_bounds__(256) void k_ho_delta_heads(View v, int row0, int nrows, const int *ranges, int range_stride,
                                                        const unsigned char *active, int alen, int ncls, int accumulate) {
  extern __shared__ float hdh_sh[]; /* [HDH_K][alenp] error segments + 4, [HDH_K][32] hidden values + 4 */
  __shared__ short list[256];
  __shared__ int wcount[4];
  const RamdShape &s = v.sh;
  const int c = blockIdx.x, h0 = blockIdx.y * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col0 = c * alen, alenp = alen | 1; /* (an odd row stride: the staging writes spread over the banks) */
  float *e_sh = hdh_sh, *h_sh = hdh_sh + HDH_K * alenp + 4; /* (e_sh[dump], four spare floats behind h_sh) */
  const int dump = HDH_K * alenp;
  const int lm = lane & 31, kh = lane >> 5;
  const bool tile_live = 32 * wave < alen, col_live = 32 * wave + lm < alen;
  const int ecol = col_live ? 32 * wave + lm : 0;
  const int kstep = 256 / alen, xstep = 256 - kstep * alen; /* a thread's next element of an [nk][alen] block */
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int base = 0; base < nrows; base += 256) {
    const int sj = base + tid;
    const int sjc = min(sj, nrows - 1);
    const unsigned long long bits =
        *reinterpret_cast<const unsigned long long *>(ranges + (size_t)sjc * range_stride + RAMD_HEADBITS_AT);
    const unsigned char live = active ? active[sjc] : 1;
    const bool mine = sj < nrows && live && ((bits >> c) & 1ull);
    const unsigned long long bal = __ballot(mine);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      if (w < wave) off += wcount[w];
      total += wcount[w];
    }
    if (mine) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (short)tid;
    __syncthreads();
    for (int k0 = 0; k0 < total; k0 += HDH_K) {
      const int nk = min(HDH_K, total - k0);
      /* Staging without a branch: every load is made (of a clamped, valid address) and every value stored, those
       * outside the block into a spare slot behind the arrays -- under a condition the compiler sinks each load into
       * its store's branch and waits there, a round trip to the L2 per element (30 us for the kernel).
       * hidden: thread t the float4 t % 8 of streams t / 8 and 32 + t / 8; the error segments: elements
       * t + 256 u of the [nk][alen] block, HDH_U of them in flight per thread */
      float4 hq[2];
      int hat[2];
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int k = (tid >> 3) + 32 * half, q = tid & 7;
        const bool in = k < nk && h0 + 4 * q < s.H;
        const size_t rr = (size_t)(row0 + base + list[k0 + (in ? k : 0)]);
        hq[half] = *reinterpret_cast<const float4 *>(v.b.hidden + rr * s.H + (in ? h0 + 4 * q : 0));
        hat[half] = in ? k * 32 + 4 * q : HDH_K * 32;
      }
      const int n = nk * alen;
      int ek = tid / alen, ex = tid - ek * alen;
      float t[HDH_U];
      int at[HDH_U];
      auto request = [&](int i0) {
#pragma unroll
        for (int u = 0; u < HDH_U; u++) {
          const bool in = i0 + 256 * u < n;
          at[u] = in ? ek * alenp + ex : dump;
          t[u] = v.b.o_error[(size_t)(row0 + base + list[k0 + (in ? ek : 0)]) * s.O + col0 + (in ? ex : 0)];
          ek += kstep;
          ex += xstep;
          if (ex >= alen) {
            ex -= alen;
            ek++;
          }
        }
      };
      /* (the first batch by every thread, the hidden values stored with it: left behind the loop the compiler moves
       * their loads there too, one more round trip) */
      request(tid);
#pragma unroll
      for (int u = 0; u < HDH_U; u++) e_sh[at[u]] = t[u];
#pragma unroll
      for (int half = 0; half < 2; half++) *reinterpret_cast<float4 *>(h_sh + hat[half]) = hq[half];
      for (int i0 = tid + HDH_U * 256; i0 < n; i0 += HDH_U * 256) {
        request(i0);
#pragma unroll
        for (int u = 0; u < HDH_U; u++) e_sh[at[u]] = t[u];
      }
      __syncthreads();
      /* wave w: the 32 x 32 tile of columns 32 w .. of the head, two streams per v_mfma_f32_32x32x2_f32 (operand lane
       * l: stream l / 32, row or column l % 32).  As FMAs with the hidden values broadcast from LDS the sum was bound by
       * the LDS (a broadcast ds_read_b128 costs what a full one does): 10 of the kernel's 27 us. */
      if (tile_live) {
#pragma unroll 4
        for (int k = 0; k < nk; k += 2) {
          const bool in = k + kh < nk;
          const int kk = in ? k + kh : 0;
          float a = h_sh[kk * 32 + lm], bb = e_sh[kk * alenp + ecol];
          a = in ? a : 0.0f;
          bb = (in && col_live) ? bb : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
        }
      }
      __syncthreads(); /* before the next streams are staged (or the list rebuilt) */
    }
  }
  if (tile_live && col_live) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      const int h = h0 + (g & 3) + 8 * (g >> 2) + 4 * kh;
      if (h >= s.H) continue;
      float *d = v.b.ho_delta + (size_t)h * s.O + col0 + 32 * wave + lm;
      *d = (accumulate ? *d : 0.0f) + acc[g];
    }
  }
  /* the columns behind the last head (o_size is padded to a multiple of 4): no error ever, zero unless accumulating */
  if (c == ncls - 1 && !accumulate && tid < 32 && h0 + tid < s.H)
    for (int x = ncls * alen; x < s.O; x++) v.b.ho_delta[(size_t)(h0 + tid) * s.O + x] = 0.0f;
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

#include "k_extras.h"


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

// The extras of nets with DENSE inputs (audio features, pixels: every input row counts, so the gather form has
// nothing to skip) as one launch: the GEMM [D x streams rows of the error planes] x [W_ih's bias row and input rows]^T
// with K = h_size, N = 1 + the inputs (33 for gstclassify, 36 for rnnca), and k_extras_finalize's work on its result.
// As k_gemm<ProbExtras> (64 x 64 tiles: one column tile, nine tenths of it padding, a K split and its slabs) + the
// finalize launch this was 47 + 13 us at 2048 / 512 / 10 and 13 + 6 us at 512 / 128 / 30.
// Workgroup = 16 or 32 (step, stream) rows x up to 48 columns, eight waves that each take every eighth 16-deep K chunk
// (v_mfma_f32_16x16x4_f32: 1 or 2 row tiles x 3 column tiles per wave), operands straight from memory as float4s -- a lane's
// four k of a chunk go to the chunk's four MFMAs, the same permutation of k on both sides -- four chunks in flight;
// the waves' sums meet in LDS, then 8 threads per row apply the row rule, store ex, and add the squares to the
// row's total (with the chain's partial sums, or the row's own sum of squares where it left none).
constexpr int XD_NT = 3, XD_PF = 8, XD_WAVES = 8;
/* (round 5: eight chunks in flight instead of four, and nothing in the epilogue waits for memory on its own -- the input
 * values of the row rule are requested before the K loop, the step's total apart from the extras (the chain's partial sums,
 * or the output row's sum of squares) is formed by ALL threads behind the loop, 32 or 16 per row, every load of a thread in
 * flight at once: as eight threads per row with two loads per trip that sum was 8 dependent round trips, half of the
 * launch's 11 us at 512 / 128 / 30; 27 -> ... us at 2048 / 512 / 10) */
template <int XD_MT> /* row tiles of 16 per workgroup: 2 where that still makes 128 workgroups (W_ih's rows are fetched by half as many), else 1 */
__global__ __launch_bounds__(64 * XD_WAVES) void k_extras_dense(View v, int row0, int nrows, int nx, int nxp, int tn, int t0, int nt) {
  constexpr int XD_ROWS = 16 * XD_MT;
  constexpr int TPR = 64 * XD_WAVES / XD_ROWS; /* threads per row for the step's total */
  __shared__ float red[XD_WAVES][XD_ROWS][16 * XD_NT + 1];
  __shared__ float oth_sh[XD_ROWS];
  const RamdShape &s = v.sh;
  /* rows: steps t0 .. t0 + nt - 1 of the call's streams (one launch for all steps, or one per step beside the chain) */
  const int M = nt * nrows;
  const int m0 = blockIdx.x * XD_ROWS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, kq = lane >> 4;
  // operand rows of this lane: two error rows, three rows of W_ih
  const float *arow[XD_MT], *brow[XD_NT];
#pragma unroll
  for (int i = 0; i < XD_MT; i++) {
    const int m = min(m0 + 16 * i + lm, M - 1);
    const int t = t0 + m / nrows, j = m % nrows;
    arow[i] = v.b.ehi + ((size_t)t * s.Scap + row0 + j) * s.I;
  }
#pragma unroll
  for (int jn = 0; jn < XD_NT; jn++) {
    const int c = 16 * jn + lm;
    brow[jn] = v.b.ih_w + (size_t)((c == 0 || c >= nx) ? 0 : s.hidden_size + c) * s.H; /* (columns >= nx: discarded below) */
  }
  f32x4 acc[XD_MT][XD_NT];
#pragma unroll
  for (int i = 0; i < XD_MT; i++)
#pragma unroll
    for (int jn = 0; jn < XD_NT; jn++) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nchunks = (s.H + 15) / 16; /* h_size is a multiple of 4: a lane's float4 is inside the row or wholly past it */
  const int mine = (nchunks - wave + XD_WAVES - 1) / XD_WAVES; /* chunks wave, wave + XD_WAVES, .. */
  float4 fa[XD_PF][XD_MT], fb[XD_PF][XD_NT];
  auto request = [&](int slot, int n) {
    const int k = 16 * (wave + XD_WAVES * n) + 4 * kq;
    const bool in = n < mine && k < s.H;
    const int kc = in ? k : 0;
#pragma unroll
    for (int i = 0; i < XD_MT; i++) fa[slot][i] = ld4(in ? arow[i] + k : v.b.zeros); /* (a select on the address: one on the value would wait for it here) */
#pragma unroll
    for (int jn = 0; jn < XD_NT; jn++) fb[slot][jn] = ld4(brow[jn] + kc);
  };
#pragma unroll
  for (int p = 0; p < XD_PF; p++) request(p, p);
  // the epilogue's row (eight threads per row: columns sub, sub + 8, ..) and the input values its rule asks for
  const int row = (tid >> 3) & (XD_ROWS - 1), sub = tid & 7;
  const int m = m0 + row;
  const bool live = m < M;
  const int mm = live ? m : M - 1;
  const int t = t0 + mm / nrows, j = mm % nrows, r = row0 + j;
  constexpr int XD_CPT = (16 * XD_NT + 7) / 8; /* columns per thread */
  float xv[XD_CPT];
  if (tid < 8 * XD_ROWS) {
    const float *x = input_row_auto(v, r, t);
#pragma unroll
    for (int q = 0; q < XD_CPT; q++) {
      const int c = sub + 8 * q;
      xv[q] = x[(c == 0 || c >= nx) ? 0 : s.hidden_size + c];
    }
  }
  for (int n0 = 0; n0 < mine; n0 += XD_PF) {
#pragma unroll
    for (int p = 0; p < XD_PF; p++) {
      if (n0 + p < mine) { /* (wave-uniform) */
        float4 a[XD_MT], bb[XD_NT];
#pragma unroll
        for (int i = 0; i < XD_MT; i++) a[i] = fa[p][i];
#pragma unroll
        for (int jn = 0; jn < XD_NT; jn++) bb[jn] = fb[p][jn];
        request(p, n0 + p + XD_PF);
#pragma unroll
        for (int i = 0; i < XD_MT; i++)
#pragma unroll
          for (int jn = 0; jn < XD_NT; jn++) {
            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bb[jn].x, acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bb[jn].y, acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bb[jn].z, acc[i][jn], 0, 0, 0);
            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bb[jn].w, acc[i][jn], 0, 0, 0);
          }
      }
    }
  }
  // accumulator register r of tile (i, jn): row 16 i + 4 kq + r, column 16 jn + lm
#pragma unroll
  for (int i = 0; i < XD_MT; i++)
#pragma unroll
    for (int jn = 0; jn < XD_NT; jn++)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) red[wave][16 * i + 4 * kq + rr][16 * jn + lm] = acc[i][jn][rr];
  /* the step's total apart from the extras: the chain's column-tile partials, or (tn == 0: the one-launch chain leaves
   * none) the sum of squares of the step's OUTPUT row, error plane t + 1 -- TPR threads per row, a fixed tree */
  {
    const int orow = tid / TPR, osub = tid % TPR;
    const int om = min(m0 + orow, M - 1);
    const int ot = t0 + om / nrows, orr = row0 + om % nrows;
    float o = 0.0f;
    if (tn == 0) {
      const float *erow = v.b.ehi + ((size_t)(ot + 1) * s.Scap + orr) * s.I;
      const int n4 = s.H / 4;
      constexpr int NB = 9; /* float4 per thread and batch: h_size <= 1152 in one batch of 32 threads */
      for (int b0 = osub; b0 < n4; b0 += TPR * NB) {
        float4 e[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
          const int k4 = b0 + i * TPR;
          e[i] = ld4(k4 < n4 ? erow + 4 * k4 : v.b.zeros);
        }
#pragma unroll
        for (int i = 0; i < NB; i++) o += (e[i].x * e[i].x + e[i].y * e[i].y) + (e[i].z * e[i].z + e[i].w * e[i].w);
      }
    } else {
      float pv[3]; /* (tn <= 33 column tiles, TPR >= 16) */
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int p = osub + i * TPR;
        pv[i] = p < tn ? v.b.esum_part[((size_t)ot * (tn + 1) + p) * s.Scap + orr] : 0.0f;
      }
      o = (pv[0] + pv[1]) + pv[2];
      for (int p = osub + 3 * TPR; p < tn; p += TPR) o += v.b.esum_part[((size_t)ot * (tn + 1) + p) * s.Scap + orr];
    }
#pragma unroll
    for (int off = 1; off < TPR; off <<= 1) o += __shfl_xor(o, off, 64);
    if (osub == 0) oth_sh[orow] = o;
  }
  __syncthreads();
  if (tid >= 8 * XD_ROWS) return;
  float *dst = v.b.ex + ((size_t)(t + 1) * s.Scap + r) * nxp;
  float sq = 0.0f;
#pragma unroll
  for (int q = 0; q < XD_CPT; q++) {
    const int c = sub + 8 * q;
    if (c < nx) {
      float e = 0.0f;
#pragma unroll
      for (int w = 0; w < XD_WAVES; w++) e += red[w][row][c];
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
  if (sub == 0 && live) v.b.esum[(size_t)t * s.Scap + r] = oth_sh[row] + sq;
}

/* (t0, nt: steps t0 .. t0 + nt - 1.  One launch per step on a second stream beside a launch-per-step chain was tried --
 * hidden 2048: 10 steps of 40 us -- and lost: the chain's workgroups fill every CU's LDS, the extras' wait for a step to
 * end and delay the next: 890 -> 1004 us per generation at 2048 / 512 / 10.  profiles/NOTES_r05.md) */
static void ramd_launch_extras_dense(hipStream_t st, const View &v, const RamdShape *sh, int row0, int nrows, int nx, int nxp,
                                     int tn_parts, int t0, int nt) {
  const int M = nt * nrows;
  if (M / 32 >= 128)
    RAMD_LAUNCH(k_extras_dense<2>, dim3((M + 31) / 32), dim3(64 * XD_WAVES), 0, st, v, row0, nrows, nx, nxp, tn_parts, t0, nt);
  else
    RAMD_LAUNCH(k_extras_dense<1>, dim3((M + 15) / 16), dim3(64 * XD_WAVES), 0, st, v, row0, nrows, nx, nxp, tn_parts, t0, nt);
}

BND_DECL(g_bnd_delta, ramd_bnd_delta_stamps)
#ifdef BND_STAMPS
#define DD_BND_MARK(which) BND_MARK(g_bnd_delta, which)
#endif
#include "k_delta_direct.h" /* the weight-delta GEMM without a K split over workgroups (hidden 1024 and up) */

#ifdef PC_STAMPS
extern "C" void ramd_ddir_stamps(unsigned long long *out) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ddir_stamps), sizeof(unsigned long long) * 32));
  HIP_CHECK(hipMemcpyFromSymbol(out + 32, HIP_SYMBOL(g_ddir_wave), sizeof(unsigned long long) * 32));
}
extern "C" void ramd_ddir_clocks(unsigned long long *out) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ddir_clk), sizeof(unsigned long long) * 16));
}
#endif
/* k_delta_direct with the top layer's weight delta AND its update in the launch's first microseconds (the fused text step):
 * workgroup i forms rows 5 i .. of ho_delta = hidden^T . o_error over the call's streams (chain_ho_delta, k_common.h: what
 * the chain launch otherwise does in ITS first 2.5 us, on the generation's critical path), stores them and updates those
 * rows of W_ho and its momentum (recur-nn.c:482-487 with the top layer's rate, 653-676) -- between the request for the
 * GEMM's first operands and the first wait for them (dd_body's PRE): the waves of this launch wait ~2 us there anyway. */
template <int NW, int P, int NPW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void k_delta_direct_ho(DdArgs a, View v,
                                                                                                                HoWork hw, HoApply ap) {
  extern __shared__ __attribute__((aligned(16))) float ddh_lds[];
  __builtin_amdgcn_s_setprio(2);
  BND_MARK(g_bnd_delta, 0);
#ifndef DD_HO_AT_END
#define DD_HO_AT_END 1
#endif
#if DD_HO_AT_END /* round 6 (profiles/NOTES_r06.md): the hook behind the GEMM's epilogue instead of in front of its first wait */
  dd_body<NW, P, NPW>(a, ddh_lds);
  __syncthreads();
  chain_ho_delta<5, 8>(v, hw, ddh_lds, (int)blockIdx.x, ap);
#else
  dd_body<NW, P, NPW>(a, ddh_lds, [&]() { chain_ho_delta<5, 8>(v, hw, ddh_lds, (int)blockIdx.x, ap); });
#endif
  BND_MARK(g_bnd_delta, 1);
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

#ifdef PC_STAMPS /* development builds only (tools/mkabl.sh -DPC_STAMPS, tools/gpu_delta_stamps.py) */
__device__ unsigned long long g_dd_stamps[64][8];
extern "C" void ramd_delta_stamps(unsigned long long *out) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dd_stamps), sizeof(unsigned long long) * 64 * 8));
}
#endif
/* RR: rows of the rest tile: 0 (none), 64 or 128 (a wave then has 32 or 64 rest rows x its 64 columns) */
template <int RR>
__global__ __launch_bounds__(512) void k_delta_dma(View v, int row0, int nrows, GemmOut o, DeltaRest dr) {
  /* above the noise generator's waves (priority 0), which share four SIMDs with workgroups of this launch while the set
   * has presynaptic noise: the launch runs at the pace of its slowest workgroup (multi-head step, 256 / 32 streams:
   * 369 -> 360 / 221 -> 218 us per generation; nothing else changes) */
  __builtin_amdgcn_s_setprio(2);
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
  // The fragments of K group u + 1 (8 k: four MFMA steps of two k each) are read from LDS while the 16
  // MFMAs of group u are issued, ONE READ BEHIND EACH OF THE FIRST NINE MFMAs, and used after them.
  // This wave is the only one that feeds its SIMD's matrix pipe, and it issues in order: whatever stands
  // between the last MFMA of a group and the first of the next beyond the ~60 cycles of that MFMA's own
  // shadow is time the pipe idles.  Measured (round 3, ablation builds): the nine reads bunched in front of
  // a group cost 3.8 us of a 100 us launch, the eight coefficient multiplies 4.4 us, and a run-time test
  // that skipped the multiplies cost what they did.  So: reads spread one per MFMA shadow; inline asm
  // (`ds_read2st64_b32`: two k rows, 128 floats apart, per instruction; immediate offsets in units of 64
  // floats reach every k of a stage; four address registers per STAGE) because as plain loads hipcc waited
  // for them twice per K tile; the multiplies only for a stage whose coefficients are not all exactly 1.0
  // (out of line); the stage's rest-row bookkeeping as two counters instead of a division per stage.
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
  /* per-lane byte addresses within a stage: A rows i = 0, 1; E columns jn = 0, 1; the coefficients (four of the
   * group for the multiplies; the lane's one of the stage's 32 for the all-ones test); the rest rows */
  const uint32_t dsm0 = lds_byte_addr(dsm);
  const uint32_t la_lane = dsm0 + 4u * (uint32_t)(4 * kh * 128 + wm * 64 + lm);
  const uint32_t le_lane = dsm0 + 4u * (uint32_t)(BK * 128 + 4 * kh * 128 + wn * 64 + lm);
  const uint32_t lc_lane = dsm0 + 4u * (uint32_t)(2 * BK * 128 + 4 * kh);
  const uint32_t lf_lane = dsm0 + 4u * (uint32_t)(2 * BK * 128 + (lane & 31));
  const uint32_t lr_lane = lds_byte_addr(rest_ring) + 4u * (uint32_t)(4 * kh * RR + wm * (RR / 2) + lm);
  uint32_t ad_a0 = 0, ad_a1 = 0, ad_e0 = 0, ad_e1 = 0, ad_c = 0, ad_r = 0; /* of the stage being READ */
  float cfl = 0.0f; /* the lane's coefficient of that stage */
  /* `slot`: the stage's ring buffer (st % DD_STAGES); `par`: which half of the rest ring ((st / tm) & 1) */
  auto stage_addr = [&](int slot, bool rest, int par) {
    const uint32_t sb = (uint32_t)(slot * DD_STAGE_FLOATS * (int)sizeof(float));
    ad_a0 = la_lane + sb;
    ad_a1 = ad_a0 + 128u;
    ad_e0 = le_lane + sb;
    ad_e1 = ad_e0 + 128u;
    ad_c = lc_lane + sb;
    if (REST && rest) ad_r = lr_lane + (uint32_t)((RR == 64 ? par * (BK * 64) : 0) * (int)sizeof(float));
    cfl = lds_read_b32_off<0>(lf_lane + sb);
  };
  /* read k (0 .. 8) of group G of the stage whose addresses are set */
  auto rd1 = [&](auto GC, auto KC, Frag &f) {
    constexpr int G = decltype(GC)::value, k = decltype(KC)::value;
    if constexpr (k == 0) {
      f.cf = lds_read_b128_off<32 * G>(ad_c);
    } else {
      constexpr int pair = (k - 1) >> 2, which = (k - 1) & 3;
      constexpr int o0 = 2 * (8 * G + 2 * pair), o1 = o0 + 2;
      if constexpr (which == 0) f.a[0][pair] = lds_read2st64<o0, o1>(ad_a0);
      if constexpr (which == 1) f.e[0][pair] = lds_read2st64<o0, o1>(ad_e0);
      if constexpr (which == 2) f.a[1][pair] = lds_read2st64<o0, o1>(ad_a1);
      if constexpr (which == 3) f.e[1][pair] = lds_read2st64<o0, o1>(ad_e1);
    }
  };
  /* ... and the rest rows' 2 RI reads of that group (rest stages only) */
  auto rd_rest = [&](auto GC, Frag &f) {
    constexpr int G = decltype(GC)::value;
    constexpr int RU = RR ? RR / 64 : 1; /* k rows of the rest tile are RR floats apart: RR / 64 offset units */
#pragma unroll
    for (int i = 0; i < RI; i++) {
      f.ar[i][0] = lds_read2st64<(8 * G + 0) * RU, (8 * G + 1) * RU>(ad_r + 128u * (uint32_t)i);
      f.ar[i][1] = lds_read2st64<(8 * G + 2) * RU, (8 * G + 3) * RU>(ad_r + 128u * (uint32_t)i);
    }
  };
  /* One group: the MFMAs on `cur` (whose reads were issued during the group before: all landed), the reads
   * of group GN into `nxt` behind the first nine of them.
   * `ones`: every coefficient of the stage is exactly 1.0 (the usual case: no stream was soft-clipped, none
   * broke off early) -- the error fragments go into the MFMAs as they are. */
  /* (the flags are ints in scalar registers: as bools that live across blocks hipcc keeps them as vector
   * masks and rebuilds each with two vector-ALU instructions per use) */
  auto group = [&](auto GN, Frag &cur, Frag &nxt, int rest, int rest_next, int ones) {
    frag_wait<0>(cur.cf, cur.a, cur.e);
    /* v_mul_legacy_f32: 0 * x is 0 for ANY x (a step past the break may hold inf), otherwise the
     * IEEE product: select and multiply in one instruction of the MFMAs' own ALU.  In place, all eight
     * in ONE statement: as separate statements with outputs of their own hipcc recycled two registers
     * between the MFMAs, and results went wrong.  (One MFMA sequence behind a conditional multiply, not
     * two sequences: with the MFMAs in both arms hipcc copies accumulators at the join.) */
    if (__builtin_expect(ones == 0, 0))
      asm volatile("v_mul_legacy_f32 %0, %8, %0\n\tv_mul_legacy_f32 %1, %9, %1\n\t"
                   "v_mul_legacy_f32 %2, %10, %2\n\tv_mul_legacy_f32 %3, %11, %3\n\t"
                   "v_mul_legacy_f32 %4, %8, %4\n\tv_mul_legacy_f32 %5, %9, %5\n\t"
                   "v_mul_legacy_f32 %6, %10, %6\n\tv_mul_legacy_f32 %7, %11, %7\n\ts_nop 1"
                   : "+v"(cur.e[0][0][0]), "+v"(cur.e[0][0][1]), "+v"(cur.e[0][1][0]), "+v"(cur.e[0][1][1]),
                     "+v"(cur.e[1][0][0]), "+v"(cur.e[1][0][1]), "+v"(cur.e[1][1][0]), "+v"(cur.e[1][1][1])
                   : "v"(cur.cf[0]), "v"(cur.cf[1]), "v"(cur.cf[2]), "v"(cur.cf[3]));
    if (REST && __builtin_expect(rest_next != 0, 0)) rd_rest(GN, nxt); /* wave-uniform */
    __builtin_amdgcn_sched_barrier(0);
    static_for<16>([&](auto MC) {
      constexpr int m = decltype(MC)::value, jj = m >> 2, i = (m >> 1) & 1, jn = m & 1;
      acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[i][jj >> 1][jj & 1], cur.e[jn][jj >> 1][jj & 1], acc[i][jn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (m < 9) {
        rd1(GN, std::integral_constant<int, m>{}, nxt);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    if (REST && __builtin_expect(rest != 0, 0)) { /* wave-uniform: this wave's 32 or 64 rest rows x its 64 columns */
#pragma unroll
      for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int i = 0; i < RI; i++)
#pragma unroll
          for (int jn = 0; jn < 2; jn++)
            racc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.ar[i][jj >> 1][jj & 1], cur.e[jn][jj >> 1][jj & 1], racc[i][jn], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  /* (the lane's coefficient was requested before the nine reads of the stage's group 0) */
  auto all_ones = [&]() {
    asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(cfl));
    return __builtin_amdgcn_readfirstlane(__ballot(cfl == 1.0f) == ~0ull ? 1 : 0);
  };
  Frag f0, f1;
  /* stage st carries rest work when st % tm == mt (ph), in half (st / tm) & 1 of the rest ring (par) */
  int ph = 0, par = 0;
#ifdef PC_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_dd_stamps[0][4] = __builtin_amdgcn_s_memrealtime();
    g_dd_stamps[0][7] = __builtin_readcyclecounter(); /* shader clocks: with [1][7], the clock the loop ran at */
  }
#endif
  if (nst > 0) {
    __builtin_amdgcn_s_barrier(); /* stage 0 has landed */
    asm volatile("" ::: "memory");
    stage_addr(0, REST && mt == 0, 0);
    if (REST && mt == 0) rd_rest(std::integral_constant<int, 0>{}, f0);
    static_for<9>([&](auto KC) { rd1(std::integral_constant<int, 0>{}, KC, f0); });
  }
  int ones = all_ones();
  for (int st = 0; st < nst; st++) {
    const int r = __builtin_amdgcn_readfirstlane(REST && ph == mt ? 1 : 0);
    int phn = ph + 1, parn = par;
    if (phn == o.tm) {
      phn = 0;
      parn ^= 1;
    }
    const bool more = st + 1 < nst;
    const int rn = __builtin_amdgcn_readfirstlane(REST && more && phn == mt ? 1 : 0);
    group(std::integral_constant<int, 1>{}, f0, f1, r, r, ones);
    group(std::integral_constant<int, 2>{}, f1, f0, r, r, ones);
    group(std::integral_constant<int, 3>{}, f0, f1, r, r, ones);
    /* group 3 runs across the stage boundary: its reads are group 0 of stage st + 1 (after the last stage
     * group 0 of the same stage once more, into the idle fragment: every group then looks the same) */
    if (more) {
#ifdef PC_STAMPS
      if (blockIdx.x == 0 && threadIdx.x == 0 && st < 63) g_dd_stamps[st + 1][4] = __builtin_amdgcn_s_memrealtime();
#endif
      __builtin_amdgcn_s_barrier(); /* stage st + 1 has landed; stage st - 1's buffer is free */
      asm volatile("" ::: "memory");
#ifdef PC_STAMPS
      if (blockIdx.x == 0 && threadIdx.x == 0 && st < 63) g_dd_stamps[st + 1][5] = __builtin_amdgcn_s_memrealtime();
#endif
      stage_addr((st + 1) % DD_STAGES, rn != 0, parn);
    }
    group(std::integral_constant<int, 0>{}, f1, f0, r, rn, ones);
#ifdef PC_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0 && st < 63) g_dd_stamps[st][3] = ones ? 1ull : 0ull;
#endif
    if (more) ones = all_ones();
    ph = phn;
    par = parn;
  }
#ifdef PC_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_dd_stamps[0][6] = __builtin_amdgcn_s_memrealtime();
    g_dd_stamps[1][7] = __builtin_readcyclecounter();
  }
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


__global__ __launch_bounds__(256) void k_bptt_control(View v, int row0, int nrows,
                                                      const unsigned char *active, unsigned flags,
                                                      int tn) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= nrows) return;
  const int r = row0 + j;
  bptt_control_wave(v, r, j, lane, bptt_control_load(v, r, j, active), flags, v.b.esum + r, (size_t)v.sh.Scap);
}

// k_extras_gather and k_bptt_control in one launch, one workgroup per stream: the waves
// share out the stream's steps, leave each step's error sum in LDS, and wave 0 then runs
// the control logic on them (nothing else needs the sums of other streams).  The body is
// extras_control_stream (k_extras.h), which the one-launch chain also runs in its tail.
template <int MAXQ, int THREADS>
__global__ __launch_bounds__(THREADS) void k_extras_control(View v, int row0, int nrows, int nx,
                                                            int nxp, int tn,
                                                            const unsigned char *active,
                                                            unsigned flags) {
  extern __shared__ float es_sh[]; /* [D] the steps' totals; tn == 0: then [D + 1] the rows' own sums of squares */
  const int j = blockIdx.x;
  extras_control_stream<MAXQ, THREADS>(v, row0 + j, j, nx, nxp, tn, active, flags, es_sh);
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
    bptt_control_wave(v, r, 0, lane, bptt_control_load(v, r, 0, nullptr), flags, es_sh, 1);
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

extern "C" void ramd_launch_pending_finalize(ramd_stream_t st_, const RamdPendingDelta *p) {
  hipStream_t st = (hipStream_t)st_;
  const size_t n4 = p->slab ? p->n / 4 : 0;
  const unsigned blocks = (unsigned)((n4 + 255) / 256) + (p->ho_slab ? (unsigned)((p->ho_n / 4 + 255) / 256) : 0u);
  if (!blocks) return;
  RAMD_LAUNCH(k_delta_finalize, dim3(blocks), dim3(256), 0, st, p->delta_out, p->slab, n4, p->n, p->ks, 0, p->H,
              p->hidden_size, p->rows_core, p->ks_rest, p->rest, p->rest_stride, p->ho_delta_out, p->ho_slab, p->ho_n,
              p->ho_ks);
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

__global__ void k_zero_f4(float *a, size_t n4) {
  size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n4) *reinterpret_cast<float4 *>(a + 4 * q) = zero4();
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
  bool images_done = !(flags & RAMD_IMAGES_PENDING);
  const int h_alen0 = b->mheads_alen, h_ncls0 = h_alen0 > 0 ? sh->output_size / h_alen0 : 0;
  const bool top_sparse = !(flags & 0x40000000u) && ranges && range_stride && (flags & RAMD_RANGES_ARE_HEADS) &&
                          h_alen0 >= 24 && h_alen0 <= 128 && h_ncls0 <= 64 && sh->output_size == h_ncls0 * h_alen0 &&
                          b->mheads_part && (size_t)sh->Scap * h_ncls0 * sh->H <= b->mheads_part_floats &&
                          row0 + nrows <= sh->Scap && env_int("RECUR_AMD_TOP_SPARSE", 1);
  if (!images_done && !(top_sparse && env_int("RECUR_AMD_STALE_FROM_PLANES", 1))) {
    ramd_launch_err_writeback(st_, sh, b, row0, nrows); /* the images, before anything below overwrites the planes */
    images_done = true;
    flags &= ~RAMD_IMAGES_PENDING;
  }
  if (!(flags & 0x40000000u)) { /* ramd_launch_text_top has already done the top backprop */
    const int h_alen = h_alen0, h_ncls = h_ncls0;
    if (top_sparse) {
      /* the multi-head loss's ranges, only the heads a stream trained: partial products per (stream, head), then the
       * ordered sums and the clip */
      const int span4 = (3 + h_alen + 3) / 4, ld = (4 * span4) | 1; /* (the widest span: a head that starts 3 columns into its float4) */
      static bool thp_attr = false;
      if (!thp_attr) { /* (85 KB at 128 symbols) */
        HIP_CHECK(hipFuncSetAttribute((const void *)k_top_heads_partial, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        thp_attr = true;
      }
      RAMD_LAUNCH(k_top_heads_partial, dim3(h_ncls, (sh->H + THP_ROWS - 1) / THP_ROWS), dim3(256),
                  (size_t)((THP_ROWS + 32) * ld + 4) * sizeof(float), st, v, row0, nrows, ranges, range_stride, active, h_alen,
                  h_ncls, b->mheads_part);
      RAMD_LAUNCH(k_top_heads_combine, dim3(nrows), dim3(THC_THREADS), 0, st, v, row0, ranges, range_stride, active, h_ncls,
                  b->mheads_part, (flags & RAMD_IMAGES_PENDING) ? 1 : 0);
      images_done = true; /* (it took the stale entries from the planes) */
    } else if (ranges && range_stride && (flags & RAMD_RANGES_ARE_HEADS) && sh->O % 4 == 0 && sh->O >= 64 &&
        (size_t)nrows * 2 * ((sh->H + 31) / 32) <= b->slab_floats && env_int("RECUR_AMD_TOP_HEADS", 1)) {
      /* the multi-head loss's ranges: one GEMM over all streams (k_top_backprop_heads), then the sums and the clip */
      const int tm = (nrows + 31) / 32, tn = (sh->H + 31) / 32, nb = 2 * tn;
      RAMD_LAUNCH(k_top_backprop_heads, dim3(tm * tn), dim3(512), (size_t)((sh->O + 15) / 16) * sizeof(unsigned), st, v,
                  row0, nrows, ranges, range_stride, active, b->slab, nb, tm);
      RAMD_LAUNCH(k_top_backprop_scale, dim3(nrows), dim3(256), 0, st, v, row0, active, b->slab, nb);
    } else if (ranges && env_int("RECUR_AMD_TOP_RANGED", 1)) {
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
  /* ---- the weight-delta GEMM's form is decided here already (k_delta_direct below): when it carries the update, the top
   * layer's delta and update ride in ITS first microseconds instead of the chain launch's (ho_in_delta) */
  constexpr int DNW = 8, DP = 5;
  /* row tiles: as many whole ones as make whole rounds of 256 workgroups, or nearly -- a multi-head net's 1100
   * input rows are 17 x 16 = 272 tiles, a round of 256 and a round of 16: twice the time; as 16 row tiles they
   * are one round with 76 rest rows, two pieces of them per workgroup (NPW) */
  const int dtn = sh->hidden_size / 64;
  int dtm = sh->I / 64, drest = sh->I - 64 * dtm;
  auto rounds_ok = [&](int tm_) {
    const int tiles = tm_ * dtn, rounds = (tiles + 255) / 256;
    return tiles >= 192 && 10 * tiles >= 9 * 256 * rounds;
  };
  if (!rounds_ok(dtm) && dtm > 16 && sh->I - 64 * (dtm - 1) <= 128 && rounds_ok(dtm - 1)) {
    dtm--;
    drest = sh->I - 64 * dtm;
  }
  /* the rest rows' pieces: 16 per column tile (up to 64 rest rows) or 32, one or two per workgroup of the tile's first
   * 16 row tiles -- or two per workgroup where there are only 8 row tiles (hidden 512) */
  const int npw = dtm >= 16 ? (drest > 64 ? 2 : 1) : 2;
  const int drg = dtm >= 16 ? 4 * npw : 4;
  const int dQPS = nrows / 4, dn_it = dQPS % DNW == 0 ? sh->D * (dQPS / DNW) : -1;
  /* fewer tiles than that (hidden 512: 8 x 8): K split two or four ways over workgroups, the parts' sums as planes for the
   * optimiser's launch (or k_delta_finalize) to add -- what k_delta_dma leaves, from 64 x 64 tiles without LDS staging:
   * 43.5 -> ... us at 512 / 128 / 30 */
  int dks = 1;
  if (!rounds_ok(dtm) && dn_it > 0 && env_int("RECUR_AMD_DELTA_DIRECT_SPLIT", 1))
    for (int k = 4; k >= 2 && dks == 1; k -= 2)
      if (dtm * dtn * k >= 192 && dtm * dtn * k <= 256 && dn_it % (k * DP) == 0 && dn_it / k >= DP) dks = k;
  const bool direct = b->uniform_idx >= 0 && sh->hidden_size % 64 == 0 && dtn > 0 && (rounds_ok(dtm) || dks > 1) &&
                      (drest == 0 || (dtm >= 16 && drest <= 128) || (dtm >= 8 && drest <= 64)) &&
                      nrows % (4 * DNW) == 0 && nrows <= 256 * (DD_FLAG_LOADS / 2) && row0 + nrows <= sh->Scap &&
                      sh->activation != 5 && dn_it >= DP && dn_it % DP == 0 &&
                      !(g_delta_half_hook && env_int("RECUR_AMD_DIST_OVERLAP", 0)) && env_int("RECUR_AMD_DELTA_DIRECT", 1);
  const bool direct_fuse = direct && dks == 1 && defer && defer->fuse_want && !accumulate && !(flags & 0xa0000000u);
  /* (HoWork's preconditions: up to 256 streams, o_size <= 48, five rows of ho_delta per workgroup at most) */
  /* (not where the chain launch has workgroups without chain work -- half of it or more, ramd_chain_steps: there the
   * request costs the chain nothing, here it costs 2.4 us: the 48 loads per wave queue behind the ring's at the CU's
   * 64 bytes per clock.  256 streams at hidden 1024: chain 107.6 -> 104.1 us, this launch 91.9 -> 94.3, generation 221.0 -> 219.9) */
  const bool ho_in_delta = direct_fuse && defer->fuse_method == 0 && !ranges && !active && nrows >= 16 && nrows <= 256 && sh->O <= 48 && sh->O % 4 == 0 &&
                           sh->H <= 5 * dtm * dtn && (nrows / 32) * (sh->hidden_size / 32) > 128 &&
                           env_int("RECUR_AMD_HO_IN_DELTA", 1);
  bool ho_paired = false, ho_finalize_after = false, ho_in_final = false;
  ProbHoDelta ho_p = {};
  int ho_nkt = 0, ho_ks = 0;
  /* The top layer's delta.  Where the one-launch chain will run it is handed to that launch as a request
   * (HoWork: formed while the chain's weight panels are on their way, no launch of its own); otherwise, and
   * when the chain declines, the GEMM below. */
  HoWork ho_req = {};
  /* (up to 256 streams: the request costs the chain launch 0.014 us per stream -- 3.6 us at 256 against the GEMM's
   * 6.2 us launch -- and nothing where the set leaves workgroups of that launch without chain work: 32 streams
   * 145.3 -> 141.0 us per generation, 64: 155.7 -> 151.2, 256: 244.8 -> 242.4) */
  const bool ho_asked = !ho_in_delta && !(flags & 0x80000000u) && !ranges && !accumulate && nrows >= 16 && nrows <= 256 && sh->O <= 48 &&
                        env_int("RECUR_AMD_HO_IN_CHAIN", 1);
  auto ho_classic = [&]() { /* (the fused single-net path, flag 0x80000000, updates W_ho directly) */
    if (ranges && range_stride && (flags & RAMD_RANGES_ARE_HEADS) && b->mheads_alen >= 24 && b->mheads_alen <= 128 &&
        sh->output_size % b->mheads_alen == 0 && sh->output_size / b->mheads_alen <= 64 && env_int("RECUR_AMD_HO_HEADS", 1)) {
      /* the multi-head loss: only the heads a stream trained carry error (k_ho_delta_heads) */
      const int alen = b->mheads_alen, ncls = sh->output_size / alen;
      if (defer) defer->ho_slab = nullptr;
      RAMD_LAUNCH(k_ho_delta_heads, dim3(ncls, (sh->H + 31) / 32), dim3(256),
                  (size_t)(HDH_K * ((alen | 1) + 32) + 8) * sizeof(float), st, v, row0, nrows, ranges, range_stride, active, alen,
                  ncls, accumulate);
      return;
    }
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
  };
  if (ho_asked) {
    ho_req.dst = (defer && b->ho_slab) ? b->ho_slab : b->ho_delta;
    ho_req.active = active;
    ho_req.row0 = row0;
    ho_req.nrows = nrows;
  } else if (!(flags & 0x80000000u) && !ho_in_delta) {
    ho_classic();
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
    /* the extras and the control logic ride in the one-launch chain's tail where they are the gather form
     * (XcWork, k_common.h); where the chain declines, the launch below */
    const bool extras_gather = sh->H <= 2304 && !env_int("RECUR_AMD_EXTRAS_GEMM", 0) && !(b->dense_inputs && nx > 8);
    XcWork xc_req = {};
    if (extras_gather && nx <= 128 && !env_int("RECUR_AMD_EXTRAS_SPLIT", 0) && env_int("RECUR_AMD_XC_IN_CHAIN", 1) &&
        (size_t)(sh->D + 1) * sh->Scap * sh->I * sizeof(float) < ((size_t)1 << 31)) { /* (32-bit byte offsets into the planes) */
      xc_req.on = 1;
      xc_req.row0 = row0;
      xc_req.nx = nx;
      xc_req.nxp = nxp;
      xc_req.active = active;
      xc_req.flags = flags;
    } else if (b->dense_inputs && nx > 8 && nx <= 16 * XD_NT && (sh->hidden_size == 512 || sh->hidden_size == 1024) &&
               sh->D <= 63 && !env_int("RECUR_AMD_EXTRAS_GEMM", 0) && env_int("RECUR_AMD_XC_IN_CHAIN", 1) &&
               env_int("RECUR_AMD_XC_DENSE_IN_CHAIN", 1) &&
               (size_t)(sh->D + 1) * sh->Scap * sh->I * sizeof(float) < ((size_t)1 << 31)) {
      /* dense inputs (gstclassify's features): the extras as a small GEMM in the one-launch chain's tail (extras_dense_tail) */
      xc_req.on = 1;
      xc_req.dense = 1;
      xc_req.row0 = row0;
      xc_req.nx = nx;
      xc_req.nxp = nxp;
      xc_req.active = active;
      xc_req.flags = flags;
    }
    tn_parts = ramd_chain_steps(st, v, sh, b, row0, nrows, ho_asked ? &ho_req : nullptr, xc_req.on ? &xc_req : nullptr);
    if (ho_asked && ho_req.done) {
      if (defer) { /* one plane for the optimiser launch to take (or ho_delta is complete already) */
        defer->ho_slab = ho_req.dst == b->ho_slab ? b->ho_slab : nullptr;
        defer->ho_n = (size_t)sh->H * sh->O;
        defer->ho_ks = 1;
        defer->ho_delta_out = b->ho_delta;
      }
    } else if (ho_asked) {
      ho_classic();
    }
    int M = sh->D * nrows;
    int etm = (M + BM - 1) / BM, etn = (nx + BN - 1) / BN, nkt = (sh->H + BK - 1) / BK;
    int ks = pick_ks(etm * etn, nkt, "RECUR_AMD_KS_EXTRAS", b->slab_floats, (size_t)M * nxp);
    /* the gather over the non-zero input rows (one-hot symbols: two rows per step and stream) or,
     * for dense inputs with more than a handful of columns, the GEMM over all of them */
    if (xc_req.done) {
      control_done = true; /* extras and control: done in the chain launch */
    } else if (extras_gather) {
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
    } else if (nx <= 16 * XD_NT && env_int("RECUR_AMD_EXTRAS_DENSE", 1)) {
      /* dense inputs, up to 47 of them: GEMM and finalize in one launch */
      ramd_launch_extras_dense(st, v, sh, row0, nrows, nx, nxp, tn_parts, 0, sh->D);
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
  RamdBuffers own_ws;
  if (defer && defer->own_slab && !accumulate) {
    /* planes that are to outlive this call go to the caller's workspace (the kernels' View, made above, is not
     * concerned: the planes reach them as arguments) */
    own_ws = *b;
    own_ws.slab = defer->own_slab;
    own_ws.slab_floats = defer->own_slab_floats;
    b = &own_ws;
  }
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
    /* ---- k_delta_direct (k_delta_direct.h): 64 x 64 tiles that own ALL of K, where those fill the chip (hidden 1024:
     * 256 tiles); the sum goes straight into ih_delta -- and, when the caller's update is the momentum rule and
     * nothing else wants the sums first, weights and momentum are updated in the same epilogue (fuse_want) */
    bool gemm_done = false;
    {
      constexpr int NW = DNW, P = DP;
      if (direct && (dks == 1 || (size_t)dks * n <= b->slab_floats)) {
        static bool attr_set = false;
        if (!attr_set) {
          HIP_CHECK(hipFuncSetAttribute((const void *)(k_delta_direct<NW, P, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        dd_lds_bytes(NW, 1)));
          HIP_CHECK(hipFuncSetAttribute((const void *)(k_delta_direct<NW, P, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        dd_lds_bytes(NW, 2)));
          HIP_CHECK(hipFuncSetAttribute((const void *)(k_delta_direct_ho<NW, P, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        dd_lds_bytes(NW, 1)));
          HIP_CHECK(hipFuncSetAttribute((const void *)(k_delta_direct_ho<NW, P, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        dd_lds_bytes(NW, 2)));
          attr_set = true;
        }
        if (ho_paired) { /* the top layer's delta GEMM had been waiting for a pair launch */
          launch_gemm<true, true, ProbHoDelta>(st, ho_p, b->ho_slab, sh->H, sh->O, ho_nkt, ho_ks, T_OTHER);
          ho_paired = false;
          if (ho_finalize_after)
            RAMD_LAUNCH(k_ho_delta_finalize, dim3((sh->H * sh->O + 255) / 256), dim3(256), 0, st, v, b->ho_slab, ho_ks,
                        accumulate, range_stride ? nullptr : ranges);
        }
        /* the top layer's sum as ONE array by now?  (the chain launch formed it, or a finalize has run) */
        const float *ho_src = nullptr;
        int ho_src_ks = 1;
        if (defer && defer->ho_slab) { /* (planes of a split-K GEMM: the epilogue adds them) */
          ho_src = defer->ho_slab;
          ho_src_ks = defer->ho_ks;
        } else
          ho_src = b->ho_delta;
        const bool fuse = direct_fuse && (ho_src || ho_in_delta);
        DdArgs a = {};
        a.x = b->arena + (size_t)row0 * sh->I;
        a.e = b->ehi + (size_t)row0 * sh->I + 1;
        a.coef = b->coef + row0;
        a.n_exec = b->n_exec + row0;
        a.ih_scale = b->ih_scale + row0;
        a.w = b->ih_w + 1;
        a.m = b->ih_m + 1;
        a.delta = b->ih_delta + 1;
        a.plane = (size_t)sh->Scap * sh->I;
        a.I = sh->I;
        a.H = sh->H;
        a.Scap = sh->Scap;
        a.nrows = nrows;
        a.D = sh->D;
        a.uidx = b->uniform_idx;
        a.tm = dtm;
        a.tn = dtn;
        a.rest = drest;
        a.rgroups = drg;
        { /* the shares of a SIMD's two waves (dd_body): 5 / 8 to the first where the loop is long, rounded to whole rings */
          const int pct = env_int("RECUR_AMD_DELTA_FAST_PCT", 66), n_pair = 2 * (dn_it / dks);
          if (pct > 0 && dn_it / dks >= env_int("RECUR_AMD_DELTA_FAST_MIN", 40)) {
            int nf = (n_pair * pct / 100 + P / 2) / P * P;
            if (nf < P) nf = P;
            if (nf > n_pair - P) nf = n_pair - P;
            a.fast_its = nf;
          }
        }
        a.hidden_size = sh->hidden_size;
        a.mode = fuse ? 2 : accumulate ? 1 : 0;
        if (dks > 1) { /* planes: stored, summed (and added to ih_delta where the call accumulates) by what follows */
          a.delta = b->slab + 1;
          a.ksplit = dks;
          a.kplane = n;
          a.mode = 0;
        }
        if (fuse) {
          a.rate = defer->fuse_rate;
          a.momentum = defer->fuse_momentum;
          a.mw = defer->fuse_mw;
          a.method = defer->fuse_method;
          if (!ho_in_delta) { /* the top layer's sums are there (the chain launch formed them): its update, shared out */
            a.ho_w = b->ho_w;
            a.ho_m = b->ho_m;
            a.ho_delta = ho_src;
            a.ho_delta_out = ho_src == b->ho_delta ? nullptr : b->ho_delta;
            a.ho_ks = ho_src_ks;
            a.ho_plane = defer->ho_n;
            a.ho_n4 = (unsigned)((size_t)sh->H * sh->O / 4);
            a.ho_rate = defer->fuse_ho_rate;
          }
        }
        int ev = timing_begin(st, T_DELTA);
        if (fuse && ho_in_delta) {
          /* ... or formed HERE, in the launch's first microseconds (its waves wait ~2 us for their first operands anyway):
           * workgroup i sums and updates rows 5 i .. of ho_delta / W_ho / its momentum -- 2.5 us less in the chain launch */
          HoWork hw = {};
          hw.dst = b->ho_delta;
          hw.row0 = row0;
          hw.nrows = nrows;
          hw.workers = dtm * dtn;
          HoApply ap = {b->ho_w, b->ho_m, defer->fuse_ho_rate, defer->fuse_momentum, defer->fuse_mw};
          if (npw == 2)
            RAMD_LAUNCH((k_delta_direct_ho<NW, P, 2>), dim3(dtm * dtn), dim3(64 * NW), dd_lds_bytes(NW, 2), st, a, v, hw, ap);
          else
            RAMD_LAUNCH((k_delta_direct_ho<NW, P, 1>), dim3(dtm * dtn), dim3(64 * NW), dd_lds_bytes(NW, 1), st, a, v, hw, ap);
        } else if (npw == 2)
          RAMD_LAUNCH((k_delta_direct<NW, P, 2>), dim3(dks * dtm * dtn), dim3(64 * NW), dd_lds_bytes(NW, 2), st, a);
        else
          RAMD_LAUNCH((k_delta_direct<NW, P, 1>), dim3(dks * dtm * dtn), dim3(64 * NW), dd_lds_bytes(NW, 1), st, a);
        timing_end(st, ev);
        if (dks == 1) {
          if (defer) {
            defer->slab = nullptr; /* ih_delta is complete */
            if (fuse) {
              defer->ho_slab = nullptr;
              defer->fuse_done = 1;
            }
          }
          return;
        }
        ks = dks; /* whole planes: every row and column of ih_delta in each */
        gemm_done = true;
      }
    }
    if (gemm_done) {
    } else if (dma) {
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
      const bool ho_direct = ho_asked && ho_req.done && ho_req.dst == b->ho_delta; /* complete since the chain launch */
      if (g_delta_half_hook && !defer && rest_in && o.tm >= 8 && o.tm % 2 == 0 &&
          ((ho_paired && ho_finalize_after) || ho_direct) && !ranges && env_int("RECUR_AMD_DIST_OVERLAP", 0)) {
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
          if (!ho_direct) launch_gemm<true, true, ProbHoDelta>(st, ho_p, b->ho_slab, sh->H, sh->O, ho_nkt, ho_ks, T_OTHER);
          const size_t up_floats = n - half_floats, n4u = up_floats / 4, ho_n = (size_t)sh->H * sh->O;
          const unsigned fin_blocks = (unsigned)((n4u + 255) / 256) + (ho_direct ? 0u : (unsigned)((ho_n / 4 + 255) / 256));
          RAMD_LAUNCH(k_delta_finalize, dim3(fin_blocks), dim3(256), 0, st, b->ih_delta + half_floats, b->slab + half_floats,
                      n4u, n, kd2, accumulate, sh->H, sh->hidden_size, rows_core - tmh * 128, kd2 * tmh, dr.planes,
                      rest_plane, b->ho_delta, ho_direct ? (const float *)nullptr : b->ho_slab, ho_direct ? (size_t)0 : ho_n,
                      ho_direct ? 0 : ho_ks);
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

extern "C" void ramd_launch_err_writeback(ramd_stream_t st_, const RamdShape *sh,
                                          const RamdBuffers *b, int row0, int nrows) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  int nxp = (sh->I - sh->hidden_size + 3) & ~3;
  RAMD_LAUNCH(k_err_writeback, dim3(nrows), dim3(256), 0, st, v, row0, nxp);
}
