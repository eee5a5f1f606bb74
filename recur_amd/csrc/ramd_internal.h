/* ramd_internal.h -- private interface between the gnu11 C host code
 * (rnn_core.c, rnn_init.c, rnn_io.c) and the HIP kernels (kernels_*.hip).
 *
 * The HIP side is a thin shim: every ramd_launch_* function enqueues one
 * kernel (or a fixed short sequence) on the given stream and returns.  All
 * policy -- what lives where, when to copy, the order of launches -- is in the
 * C code.
 */
#ifndef RAMD_INTERNAL_H
#define RAMD_INTERNAL_H 1
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Shape of one engine: a set of weights and the streams that share it. */
typedef struct RamdShape {
  int input_size, hidden_size, output_size;
  int I, H, O;    /* padded sizes (multiples of 4), recur-nn-init.c:87-91 */
  int D;          /* BPTT ring depth                                      */
  int Scap;       /* capacity in training streams: row stride of a slot    */
  int Fcap;       /* capacity in forward-only streams                      */
  int activation; /* rnn_activation                                        */
  /* optional bottom layer (recur-nn.h:211-227): b_in real inputs (+ bias) feed
   * b_out = input_size rectified outputs; bI, bO are its padded sizes, 0 without one */
  int b_in, b_out, bI, bO;
} RamdShape;

/* Device arrays.  "state row" r addresses hidden/out: r < Scap is training
 * stream r, r >= Scap is forward-only stream r - Scap.
 *
 * HBM layout (all fp32, row major):
 *   arena   [D][Scap][I] history ring, slot major, then [Fcap][I] input rows
 *           of the forward-only streams
 *   ehi     [D+1][Scap][I] back-propagated error per BPTT step (columns
 *           1..hidden_size; column 0 and the columns above stay zero): row 0 is the
 *           (soft-clipped) top-layer error, row t+1 the input error of step t
 *   esum    [D][Scap]  sum of squares of each step's input error
 *   coef    [D][Scap]  ih_scale of the stream while the step was executed, else 0
 */
typedef struct RamdBuffers {
  float *ih_w, *ho_w, *ih_m, *ho_m, *ih_aux, *ho_aux, *ih_delta, *ho_delta;
  float *arena, *hidden, *out, *o_error, *err_a, *err_b, *ehi, *esum, *coef;
  float *ex;        /* [D+1][Scap][nxp] column 0 and the input columns of each step's error */
  float *esum_part; /* [D][tn+1][Scap] per-column-tile partial sums of squares         */
  float *zeros;     /* 256 bytes of zeros: source of out-of-range LDS-DMA chunks       */
  unsigned long long *rng; /* [Scap+Fcap][4] each stream's generator (recur-rng.h:15-20)   */
  float *ones;      /* [Scap] of 1.0: the "every stream takes part" mask               */
  float *slab;
  float *ho_slab; /* 8 planes of H * O for the deferred top-layer delta sum, or NULL (very wide O) */
  size_t slab_floats;
  int *idx;       /* [Scap] ring position                           */
  float *lr;      /* [Scap] each stream's own learn_rate copy        */
  float *mef;     /* [Scap] min_error_factor                         */
  float *ih_scale, *top_raw, *top_scaled, *bptt_err; /* [Scap]       */
  int *n_exec;    /* [Scap] executed BPTT steps                      */
  int *depth_log; /* [Scap] "depth - t" as the reference logs it     */
  int *target;    /* [Scap+Fcap] class the loss kernels score against */
  double *xent;   /* [Scap+Fcap] running sum of log2 p(target), k_xent_accumulate */
  int *hot;       /* [Scap+Fcap] one-hot input index scratch         */
  unsigned char *active; /* [Scap] scratch for the active mask       */
  /* per-stream loss statistics (single writer per entry: deterministic) */
  double *stat_err, *stat_ent, *stat_zero, *stat_depth; /* [Scap]    */
  long long *stat_correct, *stat_count;                 /* [Scap]    */
  unsigned char *text;   /* encoded text for the host-free epoch loop */
  int text_len;
  /* bottom layer: weights and their optimiser arrays [bI][bO]; each stream's input row
   * binp [Scap+Fcap][bI] and raw output row bout [Scap+Fcap][bO]; berr [Scap][bO] the
   * stream's input error summed over its executed BPTT steps; bcarry [2][bO] the
   * reference's never-cleared bottom->o_error accumulator (ping-pong, bcarry_cur = the
   * current one) */
  float *bw, *bm, *baux, *bdelta, *binp, *bout, *berr, *bcarry;
  /* the layer has ONE input buffer for all clones and one_hot_opinion never clears its last entry
   * (charmodel-helpers.h:20-31): that entry, as the last dense / kept forward pass left it */
  float *blast;
  int bcarry_cur;
  /* ring position shared by every training stream of the current call, or -1
   * when they differ (set by the host before each launch) */
  int uniform_idx;
  /* presynaptic noise generated ahead of its forward pass (see noise_speculate in rnn_core.c):
   * [n][H] values and the generator states after them; noise_spec_use: the next forward adds
   * these and adopts the states instead of running the generators */
  float *noise_spec;
  void *rng_spec;
  int noise_spec_use;
  /* hint: the set's last forward pass took dense input rows (audio features, pixels) rather than
   * one-hot symbols -- the extras of the BPTT chain are then a GEMM, not a gather over the few
   * non-zero input rows (either is correct for any input) */
  int dense_inputs;
  /* symbols per head of the last multi-head loss on these rows (0: none): with RAMD_RANGES_ARE_HEADS the range lists
   * are runs of whole heads of this many columns */
  int mheads_alen;
  /* [Scap][heads][H] partial products of the sparse top backprop (k_top_heads_partial), or NULL */
  float *mheads_part;
  size_t mheads_part_floats;
} RamdBuffers;

/* ih_delta left as un-summed K slabs by ramd_launch_calc_deltas, for the optimiser launch
 * that follows immediately to sum (and store) itself; slab == NULL: nothing pending */
typedef struct RamdPendingDelta {
  const float *slab;   /* [ks][n] planes of the rows below rows_core                  */
  size_t n;            /* plane stride = I * H                                        */
  int ks, H, hidden_size, rows_core, ks_rest;
  const float *rest;   /* planes of the rows from rows_core on: rest + z * rest_stride */
  size_t rest_stride;
  float *delta_out;    /* ih_delta                                                    */
  /* the same for ho_delta (the top layer's GEMM keeps its slabs in b->ho_slab, which nothing
   * else uses); ho_slab == NULL: ho_delta has been summed already */
  const float *ho_slab;
  size_t ho_n;         /* plane stride = H * O */
  int ho_ks;
  float *ho_delta_out;
  /* in: a workspace of the caller's own for the planes (own_slab_floats of it) instead of the shared split-K
   * workspace -- for sums that are to outlive the call (the engine's kept deltas, rnn_core.c) */
  float *own_slab;
  size_t own_slab_floats;
  /* in: the update that follows, should the weight-delta GEMM be able to carry it out itself (k_delta_direct's
   * epilogue + k_apply_edges: rnn_apply_learning's momentum rule, recur-nn.c:482-487, with these rates) --
   * fuse_want != 0 asks; out: fuse_done != 0 says that weights and momentum ARE updated (both layers) and the
   * delta arrays hold the sums: nothing is left pending */
  int fuse_want, fuse_done;
  float fuse_rate, fuse_ho_rate, fuse_momentum, fuse_mw;
  int fuse_method; /* 0: the momentum rule (recur-nn.c:482-487); 4: ADAGRAD (recur-nn.c:518-524; round 6) */
} RamdPendingDelta;

enum { RAMD_IN_KEEP = 0, RAMD_IN_ONE_HOT = 1, RAMD_IN_DENSE = 2, RAMD_IN_TEXT = 3 };

typedef void *ramd_stream_t;

/* sums what ramd_launch_calc_deltas left pending into the delta arrays after all (nobody took it along) */
void ramd_launch_pending_finalize(ramd_stream_t st, const RamdPendingDelta *p);

/* ---- forward ---- */
void ramd_launch_advance(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                         int row0, int nrows);
/* builds the input rows (recur-nn.c:104-115 + 68-81).  mode selects where the
 * real inputs come from; dense is a device pointer with leading dimension ld;
 * text_i is the text position for RAMD_IN_TEXT (also fills b->target); the
 * row's global stream number is global_first + (row - row0) of global_count.
 * advance != 0 steps each stream's ring index first (rnn_bptt_advance). */
void ramd_launch_assemble(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                          int row0, int nrows, int mode, const float *dense, int ld,
                          int text_i, int global_first, int global_count, int advance);
/* the bottom layer of rnn_opinion (recur-nn.c:88-103): fills each stream's bottom input
 * row as `mode` says (as ramd_launch_assemble), multiplies it through the bottom
 * weights, adds the presynaptic noise and writes the rectified result as the stream's
 * real inputs.  The ring index must already be the current one. */
void ramd_launch_bottom_forward(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                                int row0, int nrows, int mode, const float *dense, int ld,
                                int text_i, int global_first, int global_count,
                                float presynaptic_noise);
/* the bottom layer's share of rnn_bptt_calc_deltas (recur-nn.c:377-382, 750-757) for
 * rows that ramd_launch_calc_deltas has just processed; flips b->bcarry_cur */
void ramd_launch_bottom_deltas(ramd_stream_t st, const RamdShape *sh, RamdBuffers *b, int row0,
                               int nrows, int accumulate, const unsigned char *active);
/* hidden = act(X . W_ih), out = hidden . W_ho (recur-nn.c:117-151) */
void ramd_launch_forward(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                         int row0, int nrows, float presynaptic_noise);
/* the hidden layer only (the first half of ramd_launch_forward) */
int ramd_launch_forward_hidden(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                               int row0, int nrows, float presynaptic_noise, int leave_slabs);
/* output layer + softmax error against b->target + dense top backprop in one launch (what
 * ramd_launch_forward's second half, ramd_launch_softmax_error and the first kernel of
 * ramd_launch_calc_deltas do); follow with ramd_launch_calc_deltas(flags | RAMD_TOP_DONE).
 * ramd_text_top_ok says whether the shape allows it. */
#define RAMD_TOP_DONE 0x40000000u
/* flag of ramd_launch_calc_deltas: the per-stream range lists are the multi-head loss's (runs of whole heads of at
 * least 24 columns, the error row zero outside them): the top backprop may run as one GEMM (k_top_backprop_heads) */
#define RAMD_RANGES_ARE_HEADS 0x10000000u
/* flag of ramd_launch_calc_deltas: the error images (err_a / err_b) of these very rows have not been rebuilt since
 * their last BPTT run (k_err_writeback pending).  The ranged top backprops keep the images' stale entries for rows
 * whose hidden value is zero: the launcher rebuilds the images first, unless its top-layer form
 * takes the stale entries from the error planes itself (k_top_heads_combine) */
#define RAMD_IMAGES_PENDING 0x08000000u
/* a stream's range list as the multi-head loss leaves it: up to 64 + 1 (start, len) pairs, then one bit per head the
 * stream trained (an unsigned long long at this int offset; the stride keeps it 8-byte aligned) */
#define RAMD_HEADBITS_AT 130
#define RAMD_MULTI_RANGE_STRIDE 132
int ramd_text_top_ok(const RamdShape *sh);
int ramd_dense_top_ok(const RamdShape *sh); /* ramd_launch_dense_top will take the shape (O <= 64, RECUR_AMD_DENSE_TOP) */
int ramd_launch_forward_fused(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b, int row0,
                              int nrows, int mode, int text_i, int global_first, int n_set, int for_top,
                              const float *dense, int ld); /* (dense, ld: the inputs of mode RAMD_IN_DENSE, [nrows][ld]) */
/* after ramd_launch_forward_fused(for_top = 0) returned `fused` != 0: tail columns, noise generated ahead, activation,
 * output layer */
void ramd_launch_forward_finish(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b, int row0, int nrows,
                                int fused, int part); /* part 1: the hidden layer's end, 2: the output layer, 0: both */
void ramd_launch_text_top(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b, int row0,
                          int nrows, int fwd_ks);
/* the same launch with rnnca's loss (targets [nrows][ld] on the device, first n outputs; ngroups == 0) or gstclassify's
 * class groups between output layer and backprop; 0: not this kernel's shape, nothing launched */
int ramd_launch_dense_top(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b, int row0, int nrows, int fwd_ks,
                          const float *targets, int ld, int n, int ngroups, const int *goff, const int *gsize, const int *gt,
                          const float *weight);
/* o_error = onehot(target) - softmax(out) and statistics
 * (charmodel-predict.c:18-27, 299-304) */
void ramd_launch_softmax_error(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                               int row0, int nrows);

/* one step of get_cross_entropy (charmodel-predict.c:71-76) for state row `row`: adds
 * capped_log2f(softmax(out)[target]) to b->xent[row] when count_it */
void ramd_launch_xent_accumulate(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                                 int row, int count_it);

/* ---- backward ---- */
/* ranges: device array of (start,len) pairs ending with start < 0, or NULL;
 * active: device mask per row or NULL.  accumulate == 0 zeroes the deltas
 * first.  This is rnn_bptt_calc_deltas (recur-nn.c:707-772) for the rows. */
void ramd_launch_calc_deltas(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                             int row0, int nrows, int accumulate, const int *ranges,
                             int range_stride, const unsigned char *active, unsigned flags,
                             RamdPendingDelta *defer);
/* train_channel's loss (gstclassify.c:2070-2119): softmax error per class group against
 * gt[row][group] (< 0: the group is not trained), then the per-output error weights; all
 * arrays are device pointers, `largest` the largest group size */
void ramd_launch_grouped_softmax_error(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                                       int row0, int nrows, int ngroups, int largest,
                                       const int *goff, const int *gsize, const int *gt,
                                       const float *weight);
/* multi_softmax_error (charmodel-multi-predict.c:17-58) for rows whose opinion has been
 * formed: b->target holds each stream's next symbol, tclass[j] its own class head;
 * writes o_error and, per stream, the merged (start, len) range list (terminated by
 * start < 0) at ranges + j * range_stride */
void ramd_launch_multi_softmax_error(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                                     int row0, int nrows, int alphabet_len, int n_classes,
                                     unsigned long long threshold, const int *tclass, int *ranges,
                                     int range_stride);
/* one step of rnn_char_multi_cross_entropy (charmodel-multi-predict.c:395-403) for state row
 * `row`: per head c, capped log2 of softmax(head c)[target] added to acc[c] (device doubles) */
void ramd_launch_multi_xent_accumulate(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                                       int row, int alphabet_len, int n_classes, double *acc,
                                       int count_it);
/* rnnca's loss (gstrnnca.c:701-714): sigmoid in place on the first n outputs, slope * (target -
 * a) into o_error; targets is a device array [nrows][ld] */
void ramd_launch_sigmoid_mse_error(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                                   int row0, int nrows, int n, const float *targets, int ld);
/* fast_sigmoid_array in place on the first n outputs of state rows r0 .. r0 + nrows */
void ramd_launch_sigmoid_outputs(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b, int r0,
                                 int nrows, int n);
/* rnn_opinion's device work for one stream of a small net in one launch; 0: not its kind of shape */
int ramd_launch_forward_small(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b, int r);
/* The next ramd_launch_calc_deltas calls may run the weight-delta GEMM in two row halves and call `hook`
 * (ctx, half 0 / 1, first float from ih_delta, floats) after each half's deltas are complete; NULL: off. */
void ramd_set_delta_half_hook(void (*hook)(void *ctx, int half, size_t first_float, size_t n_floats), void *ctx);
/* whether the last ramd_launch_calc_deltas also rebuilt the h_error / i_error images (reads and clears) */
int ramd_calc_wrote_images(void);
/* up to 12 word-wise copies (nwords[g] 32-bit words from src[g] to dst[g]) in one launch */
/* the noise of the next forward pass of rows [row0, row0 + nrows), from the generators' current
 * states, into b->noise_spec / b->rng_spec; the generators themselves are not touched */
void ramd_launch_noise_speculate(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b, int row0, int nrows,
                                 float noise, float *out_noise, void *out_states, const void *src_states,
                                 const int *loss_classes, int n_classes, int skip);
void ramd_launch_segcopy(ramd_stream_t st, int nseg, void *const *dst, const void *const *src,
                         const unsigned *nwords);
/* rebuilds err_a / err_b (bptt->h_error, i_error) from ehi after a calc_deltas */
void ramd_launch_err_writeback(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b,
                               int row0, int nrows);
void ramd_launch_clear_deltas(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b);

/* ---- optimiser (recur-nn.c:454-678): one array at a time ---- */
void ramd_launch_apply(ramd_stream_t st, int method, float *w, const float *delta, float *m,
                       float *aux, size_t n, float rate, float momentum,
                       float momentum_weight, const float *rate_scale_dev);
/* the same for up to three arrays in one launch (top, recurrent, bottom) */
void ramd_launch_apply_multi(ramd_stream_t st, int method, int nseg, float *const *w,
                             const float *const *delta, float *const *m, float *const *aux,
                             const size_t *n, const float *rate, float momentum,
                             float momentum_weight, const float *rate_scale_dev,
                             const RamdPendingDelta *pend);
/* the exchange step as kernel-issued peer traffic (kernels_apply.hip: k_apply_xchg, k_xchg_barrier) */
/* sum of word_i * (2 i + 1) mod 2^64 over the arrays in a row, on the device (kernels_apply.hip: k_replica_checksum) */
void ramd_launch_replica_checksum(ramd_stream_t st, int n_arrays, const float *const *arrays, const size_t *n_floats,
                                  unsigned long long *out_dev);
void ramd_launch_xchg_barrier(ramd_stream_t st, unsigned *flags_dev, int rank, int world, unsigned seq,
                              unsigned *abort_word_dev);
void ramd_launch_apply_xchg(ramd_stream_t st, int method, int rank, int world, float *const *w,
                            const float *const *delta, float *const *m, float *const *aux, float *const *delta_out,
                            const size_t *n, const float *rate, float momentum, float mw);
/* conditioning pieces (recur-nn.c:782-855) */
void ramd_launch_scale(ramd_stream_t st, float *a, size_t n, float scale);
void ramd_launch_zero_small(ramd_stream_t st, float *a, size_t n);
void ramd_launch_clamp(ramd_stream_t st, float *a, size_t n, float lo, float hi);
void ramd_launch_tall_poppy(ramd_stream_t st, float *a, size_t n, float threshold,
                            float scale, void *scratch);
void ramd_launch_add_at(ramd_stream_t st, float *a, size_t index, float v);
/* the immediate top-layer update of rnn_bptt_calculate (recur-nn.c:941-964) */
void ramd_launch_fused_updates(ramd_stream_t st, const RamdShape *sh, const RamdBuffers *b, int row, float rate,
                               float momentum, float mw, int apply_ih, const float *rs);
/* non-zero once the one-launch BPTT chain has given up (its workgroups were not all
 * resident, or a poll timed out): the results of that launch are not valid */
unsigned ramd_chain_abort_word(void);
unsigned *ramd_abort_word_dev(void);
/* the library has created a second stream with work that may run beside the BPTT chain: the chain then stops
 * deriving its workgroups' XCDs from their numbers (kernels_chain.hip) */
void ramd_note_side_stream(void);

/* ---- timing hooks ---- */
void ramd_timing_enable(int enable);
double ramd_timing_ms(int which, long *launches, int reset);
#define RAMD_T_XCHG 5 /* rnn_amd_kernel_time_ms(5, ..): the exchange step between ranks, per generation */
int ramd_timing_begin(ramd_stream_t st, int cls); /* HIP events on st around what follows, when timing is on (else -1) */
void ramd_timing_end(ramd_stream_t st, int i);

#ifdef __cplusplus
}
#endif
#endif
