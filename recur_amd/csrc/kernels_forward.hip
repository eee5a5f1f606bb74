// kernels_forward.hip -- rnn_bptt_advance and rnn_opinion on the device: history advance, input assembly,
// presynaptic noise, the bottom layer, the hidden-layer GEMM in its four forms and the output layer.
#include "k_common.h"
#include "k_gemm.h"

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
/* src / tclass / skip: the forms for the multi-head step -- start from the states an earlier pass left (`src`: the ones
 * the pass before adopted, or the ones the speculation before this one wrote) and first make the draws the multi-head
 * loss makes from the stream's generator in between, one per head other than the stream's own
 * (k_multi_softmax_error): `ncls` heads with the classes in `tclass`, or `skip` draws for every stream where the classes
 * are not known yet -- so that the values are those of the pass AFTER that loss although the loss has not run */
__global__ void k_noise_speculate(View v, int row0, int nrows, float deviation, float *out, DevRng *state,
                                  const DevRng *src, const int *tclass, int ncls, int skip) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nrows) return;
  const RamdShape &s = v.sh;
  DevRng g = src ? src[j] : reinterpret_cast<DevRng *>(v.b.rng)[row0 + j];
  if (tclass) {
    const int own = tclass[j];
    skip = ncls - ((own >= 0 && own < ncls) ? 1 : 0);
  }
  for (int i = 0; i < skip; i++) (void)dev_rand64(g);
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
/* (Round 6, measured and removed: a WAVE per stream with the recurrence on the scalar unit -- four 64-bit scalar
 * registers, s_lshl_b64 / s_or_b64 / s_add_u32 + s_addc_u32, the twelve 16-bit fields of a value added two at a time in
 * the halves of a 64-bit word, lane k keeping value k of 64: bit-exact, and SLOWER -- 283 against 218 us per generation
 * of 32 streams, 384 against 338 of 256: a dependent scalar instruction takes ~7 cycles here, and the form has 85 of them
 * per value against the vector form's 60.  profiles/NOTES_r06.md.) */
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

// the same for k_fwd_fused's result (one plane of sums for the columns inside its tiles, per-tile partial sums
// [tn][nrows][4] for the last four columns of h_size): the set calls that go on to a generic output layer, where
// k_text_top is not there to do it.  with_spec: the presynaptic noise generated ahead (noise_speculate) is added here
// and its generator states adopted (k_noise_apply's work).
__global__ __launch_bounds__(256) void k_fwd_finalize_fused(View v, int row0, int nrows, int tn, int with_spec, int nmain) {
  const RamdShape &s = v.sh;
  const int per_row = (s.H >> 2) - 1; /* float4s of a row inside the tiles */
  int j, c;
  float4 a;
  if ((int)blockIdx.x < nmain) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nrows * per_row) return;
    j = q / per_row;
    c = (q - j * per_row) * 4;
    a = ld4(v.b.slab + (size_t)j * s.H + c);
  } else {
    /* the last four columns: 32 lanes per row, lane t the partial sums of tiles t, t + 32, ..; eight rows per workgroup */
    const int t = threadIdx.x & 31;
    j = ((int)blockIdx.x - nmain) * 8 + (threadIdx.x >> 5);
    c = s.H - 4;
    const int jc = min(j, nrows - 1);
    a = zero4();
    for (int tt = t; tt < tn; tt += 32) {
      const float4 p = ld4(v.b.slab + (size_t)nrows * s.H + ((size_t)tt * nrows + jc) * 4);
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      a.x += __shfl_xor(a.x, off, 64);
      a.y += __shfl_xor(a.y, off, 64);
      a.z += __shfl_xor(a.z, off, 64);
      a.w += __shfl_xor(a.w, off, 64);
    }
    if (t != 0 || j >= nrows) return;
  }
  if (with_spec) {
    const float4 n4 = ld4(v.b.noise_spec + (size_t)j * s.H + c);
    a.x += n4.x; a.y += n4.y; a.z += n4.z; a.w += n4.w;
    if (c == 0) reinterpret_cast<DevRng *>(v.b.rng)[row0 + j] = reinterpret_cast<const DevRng *>(v.b.rng_spec)[j];
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
  *reinterpret_cast<float4 *>(v.b.hidden + (size_t)(row0 + j) * s.H + c) = make_float4(h[0], h[1], h[2], h[3]);
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
// The same plan serves the OUTPUT layer of a wide net (the multi-head nets' 3652 columns: out = hidden . W_ho, 1.9 GFLOP
// at 256 streams, recur-nn.c:150-151): the operands come as a WideOp -- A rows K-contiguous with stride lda, B K-major
// with stride ldb, K and N their extents, C with stride ldc -- and the full K in every workgroup means the tile is the
// result (no K slabs to add up).  a == nullptr: the hidden layer's operands, from the View.
struct WideOp {
  const float *a; /* [rows][lda], K contiguous */
  const float *b; /* [K][ldb]                  */
  float *c;       /* [rows][ldc]               */
  int lda, ldb, ldc, K, N;
};
template <int NS>
__global__ __launch_bounds__(512) void k_fwd_wide(const View *__restrict__ vp, int uniform_idx, int row0,
                                                  int nrows, int tm, int tn, WideOp op) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  View v = *vp;
  v.b.uniform_idx = uniform_idx;
  const RamdShape &s = v.sh;
  const bool hidden_layer = op.a == nullptr;
  const int opK = hidden_layer ? s.I : op.K, opN = hidden_layer ? s.H : op.N;
  const int ldb = hidden_layer ? s.H : op.ldb, ldc = hidden_layer ? s.H : op.ldc;
  const float *opB = hidden_layer ? v.b.ih_w : op.b;
  float *opC = hidden_layer ? v.b.slab : op.c;
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
        src[j] = (hidden_layer ? input_row<false>(v, row0 + m0 + row, 0) : op.a + (size_t)(m0 + row) * op.lda) + 4 * c;
        kofs[j] = 4 * c;
      } else {
        const int k = 4 * (i - 16) + (lane >> 4);
        src[j] = opB + (size_t)k * ldb + n0 + 4 * (lane & 15);
        kofs[j] = k;
      }
    }
    auto issue = [&](int stage) {
      float *dst = wsm + (stage % W_STAGES) * W_STAGE_FLOATS + wave * 8 * 256;
      const int k0 = stage * WK;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool a_side = wave * 8 + j < 16;
        const float *g = a_side ? src[j] + k0 : src[j] + (size_t)k0 * ldb;
        if ((k0 + WK > opK && k0 + kofs[j] >= opK) || (!a_side && n0 + 4 * (lane & 15) >= opN))
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
  if (n0 + c4 < opN) { /* h_size % 4 == 0, o_size % 4 == 0: whole float4s */
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int row = 4 * rq + rr;
      *reinterpret_cast<float4 *>(opC + (size_t)(m0 + row) * ldc + n0 + c4) = ld4(red + row * WN + c4);
    }
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

#ifdef PC_STAMPS /* development builds only (tools/mkabl.sh -DPC_STAMPS, tools/gpu_fwd_stamps.py) */
__device__ unsigned long long g_ff_stamps[8];
extern "C" void ramd_fwd_stamps(unsigned long long *out) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ff_stamps), sizeof(unsigned long long) * 8));
}
#define FF_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ff_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FF_STAMP(i) do { } while (0)
#endif
BND_DECL(g_bnd_fwd, ramd_bnd_fwd_stamps)
constexpr int FF_MAXIN = 64;
template <int NS = 0>
__global__ __launch_bounds__(512) void k_fwd_fused(const View *__restrict__ vp, int new_idx, int row0,
                                                   int nrows, int tm, int tn, int nstages_arg,
                                                   int mode, int text_i, int global_first,
                                                   int n_set, const float *__restrict__ dense, int ld) {
  FF_STAMP(0);
  BND_MARK(g_bnd_fwd, 0);
  View v = *vp;
  const int nstages = NS > 0 ? NS : nstages_arg;
  __shared__ __attribute__((aligned(16))) float smem[C_STAGES * C_STAGE_FLOATS];
  __shared__ float rs_sh[4][CM];
  __shared__ float4 wt_sh[CN];
  /* dense inputs (mode RAMD_IN_DENSE: the feature vectors of gstclassify, rnnca's neighbourhoods -- up to FF_MAXIN
   * columns): the tile's 32 input rows and the W rows of the input columns under its 32 columns, for the epilogue */
  __shared__ float xin_sh[CM][FF_MAXIN + 1];
  __shared__ __attribute__((aligned(16))) float win_sh[FF_MAXIN][CN];
  __shared__ float4 wint_sh[FF_MAXIN]; /* ... and those rows' tail columns (column tile 0) */
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
  /* column tile 0 also covers what no tile does (one row in eight lanes): the hidden values from `tail` on and the
   * one-hot input, times their W rows' tail columns -- requested with the rest, not in the epilogue (round 3: those
   * eight workgroups' two extra round trips were the launch's last microsecond) */
  constexpr int TAILK = 4; /* hidden values tail .. hidden_size: at most h_size - tail */
  float xt[TAILK] = {0.f, 0.f, 0.f, 0.f};
  float4 wtl[TAILK] = {zero4(), zero4(), zero4(), zero4()}, wth = zero4();
  const bool tail_row = nt == 0 && (etid & 7) == 0;
  const bool dns = mode == RAMD_IN_DENSE;
  const int insz = dns ? s.input_size : 0;
  float xin_r[CM * FF_MAXIN / 256]; /* requested here, stored to LDS behind the K loop: nobody waits for them */
  float4 win_r[FF_MAXIN * (CN / 4) / 256], wint_r = zero4();
  if (!loader && dns) {
#pragma unroll
    for (int u = 0; u < CM * FF_MAXIN / 256; u++) {
      const int idx = etid + 256 * u, rr = idx / insz, k = idx - rr * insz;
      const int rc = m0 + rr < nrows ? m0 + rr : nrows - 1;
      xin_r[u] = idx < CM * insz ? dense[(size_t)rc * ld + k] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < FF_MAXIN * (CN / 4) / 256; u++) {
      const int idx = etid + 256 * u, k = idx >> 3;
      win_r[u] = k < insz ? ld4(v.b.ih_w + (size_t)(s.hidden_size + 1 + k) * s.H + n0 + 4 * (idx & 7)) : zero4();
    }
    if (nt == 0 && etid < insz) wint_r = ld4(v.b.ih_w + (size_t)(s.hidden_size + 1 + etid) * s.H + tail);
  }
  if (!loader) {
    if (mode == RAMD_IN_TEXT) { /* charmodel-predict.c:273, 295-298 */
      int len = v.b.text_len;
      int spacing = (len - 1) / n_set;
      text_o = text_i + (global_first + er) * spacing;
      if (text_o >= len - 1) text_o -= len - 1;
      hot = v.b.text[text_o];
    } else if (!dns) {
      hot = v.b.hot[grow];
    }
    a4 = ld4(hid0 + (size_t)er * s.H + n0 + ec4);
    wb = ld4(v.b.ih_w + n0 + ec4);
    /* the tail columns of W in this tile's 32 input rows: one row per thread of the first half
     * wave, shared through LDS at the end */
    if (etid < CN) wt_mine = ld4(v.b.ih_w + (size_t)(n0 + etid) * s.H + tail);
    if (tail_row) {
#pragma unroll
      for (int i = 0; i < TAILK; i++) {
        const int k = tail + i <= s.hidden_size ? tail + i : tail;
        xt[i] = hid0[(size_t)er * s.H + k];
        wtl[i] = ld4(v.b.ih_w + (size_t)k * s.H + tail);
      }
    }
  }
  if (!loader) { /* ... and what depends on the symbol (a load behind a load) */
    if (hot < 0 || hot >= s.input_size) hot = -1;
    if (n0 + ec4 == 0) a4.x = 1.0f;
    ws = ld4(v.b.ih_w + (size_t)(hot >= 0 ? s.hidden_size + 1 + hot : 0) * s.H + n0 + ec4);
    if (tail_row) wth = ld4(v.b.ih_w + (size_t)(hot >= 0 ? s.hidden_size + 1 + hot : 0) * s.H + tail);
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
      FF_STAMP(1);
      __builtin_amdgcn_s_barrier();
      FF_STAMP(2);
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
  if (!loader) FF_STAMP(3);
  if (!loader) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      int row = (g & 3) + 8 * (g >> 2) + 4 * kh;
      red[(wave * CM + row) * CN + lm] = acc[g];
    }
    rsum += __shfl_xor(rsum, 32, 64);
    if (kh == 0) rs_sh[wave][lm] = rsum;
    if (etid < CN) wt_sh[etid] = wt_mine;
    if (dns) {
#pragma unroll
      for (int u = 0; u < CM * FF_MAXIN / 256; u++) {
        const int idx = etid + 256 * u, rr = idx / insz, k = idx - rr * insz;
        if (idx < CM * insz) xin_sh[rr][k] = xin_r[u];
      }
#pragma unroll
      for (int u = 0; u < FF_MAXIN * (CN / 4) / 256; u++) {
        const int idx = etid + 256 * u, k = idx >> 3;
        if (k < insz) *reinterpret_cast<float4 *>(&win_sh[k][4 * (idx & 7)]) = win_r[u];
      }
      if (nt == 0 && etid < insz) wint_sh[etid] = wint_r;
    }
  }
  __syncthreads();
  if (loader) return;
  FF_STAMP(4);
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
  // dense inputs: their products under this thread's four columns (the input rows of W_ih lie behind the hidden
  // values', recur-nn.c:104-112), and their sum
  float4 din = zero4();
  float sum_in = 0.0f;
  for (int k = 0; k < insz; k++) {
    const float x = xin_sh[row][k];
    const float4 w = *reinterpret_cast<const float4 *>(&win_sh[k][c4]);
    din.x += x * w.x; din.y += x * w.y; din.z += x * w.z; din.w += x * w.w;
    sum_in += x;
  }
  // the row's input sum: bias + previous hidden values + the one-hot input (or the dense ones)
  float sum = ((rs_sh[0][row] + rs_sh[1][row]) + (rs_sh[2][row] + rs_sh[3][row])) + 1.0f +
              (dns ? sum_in : hot >= 0 ? 1.0f : 0.0f);
  const float softclip = s.I * INPUT_MEAN_SOFT_TOP_F;
  const float scale = (sum > softclip) ? soft_clip_dev(sum, softclip) : 1.0f;
  const bool live = m0 + row < nrows;
  // this workgroup's share of the four tail columns: its 32 inputs x W[k][tail .. tail + 3]
  float4 pp;
  pp.x = ((a4.x * wt[0].x + a4.y * wt[1].x) + (a4.z * wt[2].x + a4.w * wt[3].x));
  pp.y = ((a4.x * wt[0].y + a4.y * wt[1].y) + (a4.z * wt[2].y + a4.w * wt[3].y));
  pp.z = ((a4.x * wt[0].z + a4.y * wt[1].z) + (a4.z * wt[2].z + a4.w * wt[3].z));
  pp.w = ((a4.x * wt[0].w + a4.y * wt[1].w) + (a4.z * wt[2].w + a4.w * wt[3].w));
  /* the eight lanes of a row: two quad permutes and a half-row mirror (DPP: a few cycles each; as __shfl_xor twelve
   * dependent ds_bpermute round trips in the launch's last microsecond) */
#define FF_DPP_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
  FF_DPP_ADD(pp.x, 0xB1); FF_DPP_ADD(pp.y, 0xB1); FF_DPP_ADD(pp.z, 0xB1); FF_DPP_ADD(pp.w, 0xB1);
  FF_DPP_ADD(pp.x, 0x4E); FF_DPP_ADD(pp.y, 0x4E); FF_DPP_ADD(pp.z, 0x4E); FF_DPP_ADD(pp.w, 0x4E);
  FF_DPP_ADD(pp.x, 0x141); FF_DPP_ADD(pp.y, 0x141); FF_DPP_ADD(pp.z, 0x141); FF_DPP_ADD(pp.w, 0x141);
#undef FF_DPP_ADD
  if (!live) return;
  float *out = v.b.slab + (size_t)er * s.H;
  float *slot = v.b.arena + ((size_t)new_idx * s.Scap + grow) * s.I;
  {
    float4 o;
    o.x = ((e.x + wb.x) + (dns ? din.x : hot >= 0 ? ws.x : 0.0f)) * scale;
    o.y = ((e.y + wb.y) + (dns ? din.y : hot >= 0 ? ws.y : 0.0f)) * scale;
    o.z = ((e.z + wb.z) + (dns ? din.z : hot >= 0 ? ws.z : 0.0f)) * scale;
    o.w = ((e.w + wb.w) + (dns ? din.w : hot >= 0 ? ws.w : 0.0f)) * scale;
    *reinterpret_cast<float4 *>(out + n0 + c4) = o;
    *reinterpret_cast<float4 *>(slot + n0 + c4) = make_float4(a4.x * scale, a4.y * scale, a4.z * scale, a4.w * scale);
  }
  if ((etid & 7) == 0) {
    if (nt == 0) {
      /* once per row: the inputs the tiles do not cover -- hidden values tail .. hidden_size and
       * the one-hot input -- times their W rows */
#pragma unroll
      for (int i = 0; i < TAILK; i++) {
        if (tail + i <= s.hidden_size) {
          const float x = xt[i];
          const float4 w = wtl[i];
          pp.x += x * w.x; pp.y += x * w.y; pp.z += x * w.z; pp.w += x * w.w;
          slot[tail + i] = x * scale;
        }
      }
      if (hot >= 0) {
        pp.x += wth.x; pp.y += wth.y; pp.z += wth.z; pp.w += wth.w;
      }
      for (int k = 0; k < insz; k++) {
        const float x = xin_sh[row][k];
        const float4 w = wint_sh[k];
        pp.x += x * w.x; pp.y += x * w.y; pp.z += x * w.z; pp.w += x * w.w;
      }
    }
    float *pd = v.b.slab + (size_t)nrows * s.H + ((size_t)nt * nrows + er) * 4;
    *reinterpret_cast<float4 *>(pd) = make_float4(pp.x * scale, pp.y * scale, pp.z * scale, pp.w * scale);
  }
  /* (by column tile 1's workgroups where there is one: column tile 0's have the tail columns' extra loads and sums above,
   * and were the launch's last to end by 0.9 us -- round 6, the workgroups' own marks) */
  if (nt == (tn > 1 ? 1 : 0)) { /* the rest of k_assemble's row: input columns, ring index, target */
    const int sub = etid & 7;
    if (sub == 0) {
      v.b.idx[grow] = new_idx;
      if (mode == RAMD_IN_TEXT) v.b.target[grow] = v.b.text[text_o + 1];
    }
    if (dns)
      for (int k = sub; k < s.input_size; k += 8) slot[s.hidden_size + 1 + k] = xin_sh[row][k] * scale;
    else
      for (int k = sub; k < s.input_size; k += 8) slot[s.hidden_size + 1 + k] = (k == hot) ? scale : 0.0f;
  }
  FF_STAMP(5);
  BND_MARK(g_bnd_fwd, 1);
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
  /* Round 6: this thread's weights of BOTH layers are requested before anything else -- they depend on nothing the launch
   * computes (the ring index, the input row and the hidden row were dependent round trips in front of them: 9.2 -> 8.2 us
   * per launch).  A row whose input is zero is skipped as before (a select on the input value: the weight is not
   * multiplied). */
  constexpr int FS_WI = 32, FS_WO = 16; /* rows per thread at most */
  const int HCp = H <= 128 ? 128 : 256, Gp = 1024 / HCp;
  const int np = tid & (HCp - 1), gp = tid / HCp;
  const bool pre_ok = (I + Gp - 1) / Gp <= FS_WI && (H + 15) / 16 <= FS_WO; /* (uniform) */
  float wi[FS_WI], wo[FS_WO];
  if (pre_ok) {
    const float *w = v.b.ih_w + (np < H ? np : 0);
#pragma unroll
    for (int k = 0; k < FS_WI; k++) {
      const int y = gp + Gp * k;
      wi[k] = w[(size_t)(y < I ? y : 0) * H];
    }
    const int o_ = tid & 63, g2_ = tid >> 6;
    const float *wq = v.b.ho_w + (o_ < O ? o_ : 0);
#pragma unroll
    for (int k = 0; k < FS_WO; k++) {
      const int y = g2_ + 16 * k;
      wo[k] = wq[(size_t)(y < H ? y : 0) * O];
    }
  }
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
  if (n < H && pre_ok) {
#pragma unroll
    for (int k = 0; k < FS_WI; k++) {
      const int y = g + G * k;
      const float x = y < I ? xs[y] : 0.0f;
      acc = (x != 0.0f) ? acc + x * wi[k] : acc; /* the rows in the same order; a zero row is not multiplied */
    }
  } else if (n < H) {
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
    if (o < O && pre_ok) {
#pragma unroll
      for (int k = 0; k < FS_WO; k++) {
        const int y = g2 + 16 * k;
        if (y < H) a += hsh[y] * wo[k];
      }
    } else if (o < O) {
      for (int y = g2; y < H; y += 16) a += hsh[y] * v.b.ho_w[(size_t)y * O + o];
    }
    part[tid] = a;
    __syncthreads();
    if (tid < O) {
      float x = part[tid];
      for (int k = 1; k < 16; k++) x += part[k * 64 + tid];
      v.b.out[(size_t)r * O + tid] = x;
    }
  }
}

#pragma clang fp contract(off)
// The output layer of rnn_opinion (recur-nn.c:150-151): out = hidden . W_ho for one state
// row per workgroup.  O is small (tens to a few hundred columns) against H, so this is
// not worth an MFMA launch plus a slab pass: thread (seg, col) walks one sixteenth of the
// hidden units down one column of W_ho (a wave reads whole contiguous rows), sixteen
// waves keep enough rows in flight to cover the L2 latency, and the segments are added
// in order.
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
#pragma clang fp contract(fast)

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

/* assemble + hidden layer in one launch for the text step (k_fwd_fused); returns what
 * ramd_launch_text_top wants as fwd_ks (negative: one plane of sums + per-tile padding
 * partials), or 0 when the preconditions do not hold and nothing was launched */
extern "C" int ramd_launch_forward_fused(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b,
                                         int row0, int nrows, int mode, int text_i,
                                         int global_first, int n_set, int for_top, const float *dense, int ld) {
  /* for_top: the text step, which stops after the hidden layer's sums (k_text_top takes them from there); otherwise a
   * one-hot or text pass that goes on to ramd_launch_forward_finish */
  const bool dense_ok = mode == RAMD_IN_DENSE && dense && sh->input_size <= FF_MAXIN && env_int("RECUR_AMD_FWD_FUSED_DENSE", 1);
  if (b->uniform_idx < 0 || sh->bI || sh->hidden_size % CN != 0 || row0 + nrows > sh->Scap ||
      (for_top ? ((mode != RAMD_IN_TEXT && !dense_ok) || !ramd_text_top_ok(sh))
               : ((mode != RAMD_IN_TEXT && mode != RAMD_IN_ONE_HOT && !dense_ok) || !env_int("RECUR_AMD_FWD_FUSED_ANY", 1))) ||
      env_int("RECUR_AMD_NO_FWD_FUSED", 0)) {
    if (env_int("RECUR_AMD_TRACE_FWD", 0))
      fprintf(stderr, "librecur_amd: forward not fused: uniform_idx %d, bottom %d, hidden %d, rows %d + %d of %d, mode %d, "
                      "for_top %d, dense %p, inputs %d\n", b->uniform_idx, sh->bI, sh->hidden_size, row0, nrows, sh->Scap, mode,
              for_top, (const void *)dense, sh->input_size);
    return 0;
  }
  const int tm = (nrows + CM - 1) / CM, tn = sh->hidden_size / CN;
  /* dense inputs: where the 32 x 32 tiles are one round of workgroups (gstclassify's 512 / 128: 64 tiles; one launch less,
   * the time of assemble + GEMM).  Beyond that the tiles' operand traffic decides -- 8 flop per byte from L2: 67.7 us at
   * 2048 / 512 (1024 tiles) against 58.7 + 6.3 us for k_assemble and the 128 x 128 tiles of k_gemm */
  if (mode == RAMD_IN_DENSE && tm * tn > 256 && !env_int("RECUR_AMD_FWD_FUSED_DENSE_ANY", 0)) return 0;
  /* plane 0: sums; plane 1: [tn][nrows][4] padding partials */
  if ((size_t)nrows * sh->H + (size_t)tn * nrows * 4 > b->slab_floats || tn * 4 > sh->H) {
    if (env_int("RECUR_AMD_TRACE_FWD", 0))
      fprintf(stderr, "librecur_amd: forward not fused: workspace of %zu floats, %zu wanted\n", b->slab_floats,
              (size_t)nrows * sh->H + (size_t)tn * nrows * 4);
    return 0;
  }
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  const View *d_view = device_view(st, v);
  const int nstages = (sh->hidden_size + CK - 1) / CK;
  const int blocks = ((tn + 7) / 8) * 8 * tm;
  int ev = timing_begin(st, T_FWD);
  const bool exact = sh->hidden_size % CK == 0 && !env_int("RECUR_AMD_FWD_NS0", 0);
#define FWD_FUSED(NS)                                                                              \
  RAMD_LAUNCH((k_fwd_fused<NS>), dim3(blocks), dim3(512), 0, st, d_view, b->uniform_idx, row0, \
                     nrows, tm, tn, nstages, mode, text_i, global_first, n_set, dense, ld)
  if (exact && nstages == 8) FWD_FUSED(8);
  else if (exact && nstages == 4) FWD_FUSED(4);
  else if (exact && nstages == 2) FWD_FUSED(2);
  else if (exact && nstages == 16) FWD_FUSED(16);
  else FWD_FUSED(0);
#undef FWD_FUSED
  timing_end(st, ev);
  return -tn;
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

static void launch_output_layer(hipStream_t st, const RamdShape *sh, const RamdBuffers *b, int row0, int nrows);

extern "C" void ramd_launch_forward(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b,
                                    int row0, int nrows, float noise) {
  ramd_launch_forward_hidden(st_, sh, b, row0, nrows, noise, 0);
  launch_output_layer((hipStream_t)st_, sh, b, row0, nrows);
}

/* what follows ramd_launch_forward_fused(for_top = 0): the sums' tail columns, the noise generated ahead
 * (b->noise_spec_use), the activation; then the output layer */
extern "C" void ramd_launch_forward_finish(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b, int row0,
                                           int nrows, int fused, int part) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  if (part != 2) { /* the hidden layer's end */
    const int nmain = (nrows * (sh->H / 4 - 1) + 255) / 256;
    RAMD_LAUNCH(k_fwd_finalize_fused, dim3(nmain + (nrows + 7) / 8), dim3(256), 0, st, v, row0, nrows, -fused,
                b->noise_spec_use, nmain);
  }
  if (part != 1) launch_output_layer(st, sh, b, row0, nrows);
}

static void launch_output_layer(hipStream_t st, const RamdShape *sh, const RamdBuffers *b, int row0, int nrows) {
  View v = make_view(sh, b);
  if (sh->O == 4 && nrows >= 64) {
    RAMD_LAUNCH(k_out_layer_o4, dim3((nrows + 3) / 4), dim3(256), 0, st, v, row0, nrows);
  } else if (sh->O <= 256 && !env_int("RECUR_AMD_OUT_GEMM", 0)) {
    RAMD_LAUNCH(k_out_layer, dim3(nrows), dim3(1024),
                       (size_t)(sh->H + OUT_SEGS * 64) * sizeof(float), st, v, row0);
  } else if (nrows % WM == 0 && sh->O >= 1024 && ((sh->H + WK - 1) / WK == 9 || (sh->H + WK - 1) / WK == 17 || (sh->H + WK - 1) / WK == 33) &&
             (nrows / WM) * ((sh->O + WN - 1) / WN) >= 128 && env_int("RECUR_AMD_OUT_WIDE", 1)) {
    /* wide output layers with enough tiles to fill the device (the multi-head nets: 4 x 58 at 256 streams): k_fwd_wide's
     * 64 x 64 tiles with the full K in every workgroup write `out` directly (33 + 6 us as k_gemm + k_sum_slabs) */
    static bool attr_set2 = false;
    const size_t shm = (size_t)W_STAGES * W_STAGE_FLOATS * sizeof(float);
    if (!attr_set2) {
      HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_wide<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_wide<17>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_wide<33>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
      attr_set2 = true;
    }
    const View *d_view = device_view(st, v);
    const int ns = (sh->H + WK - 1) / WK;
    const int wtm = nrows / WM, wtn = (sh->O + WN - 1) / WN;
    const int supertiles = ((wtm + 3) / 4) * ((wtn + 7) / 8);
    const int wblocks = ((supertiles + 7) / 8) * 8 * 32;
    WideOp op = {b->hidden + (size_t)row0 * sh->H, b->ho_w, b->out + (size_t)row0 * sh->O, sh->H, sh->O, sh->O, sh->H, sh->O};
    int ev = timing_begin(st, T_OTHER);
    if (ns == 33) RAMD_LAUNCH(k_fwd_wide<33>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, wtm, wtn, op);
    else if (ns == 17) RAMD_LAUNCH(k_fwd_wide<17>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, wtm, wtn, op);
    else RAMD_LAUNCH(k_fwd_wide<9>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, wtm, wtn, op);
    timing_end(st, ev);
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
        RAMD_LAUNCH(k_fwd_wide<33>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, wtm, wtn, WideOp{});
      else if (wide_ns == 17)
        RAMD_LAUNCH(k_fwd_wide<17>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, wtm, wtn, WideOp{});
      else
        RAMD_LAUNCH(k_fwd_wide<9>, dim3(wblocks), dim3(512), shm, st, d_view, b->uniform_idx, row0, nrows, wtm, wtn, WideOp{});
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

extern "C" void ramd_launch_noise_speculate(ramd_stream_t st_, const RamdShape *sh, const RamdBuffers *b,
                                            int row0, int nrows, float noise, float *out_noise, void *out_states,
                                            const void *src_states, const int *loss_classes, int n_classes, int skip) {
  hipStream_t st = (hipStream_t)st_;
  View v = make_view(sh, b);
  RAMD_LAUNCH(k_noise_speculate, dim3((nrows + 63) / 64), dim3(64), 0, st, v, row0, nrows, noise, out_noise,
              (DevRng *)out_states, (const DevRng *)src_states, loss_classes, n_classes, skip);
}

