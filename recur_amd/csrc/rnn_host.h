/* rnn_host.h -- private declarations shared by the gnu11 C host files. */
#ifndef RNN_HOST_H
#define RNN_HOST_H 1
#define _GNU_SOURCE 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "recur_amd.h"
#include "ramd_internal.h"

#define RAMD_MAGIC 0x444d4152u /* "RAMD" */
#define RAMD_HDR_FLOATS 16     /* 64-byte private header in front of net->mem */

#define RAMD_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define RAMD_MAX(a, b) (((a) >= (b)) ? (a) : (b))

struct RamdEngine;

/* Lives in the first 64 bytes of the block net->mem points at.  The public
 * struct has no spare field (its layout is ABI), and no caller of the
 * reference touches net->mem (it only exists for free()). */
typedef struct RamdPriv {
  uint32_t magic;
  int stream; /* training-stream row in the engine, or -1 */
  int fwd;    /* forward-only row, or -1                   */
  int host_valid, dev_valid; /* this stream's state (RNN_AMD_STREAM) */
  struct RamdEngine *eng;
} RamdPriv;

/* One engine per set of weights: the device image of everything that the
 * nets sharing those weights own between them. */
typedef struct RamdEngine {
  struct RamdEngine *next;
  RecurNN *owner; /* the net whose ih_weights/ho_weights these are */
  RamdShape sh;
  RamdBuffers b;
  size_t ih_size, ho_size;
  int n_streams, cap_streams;
  RecurNN **streams;
  int n_fwd, cap_fwd;
  RecurNN **fwd;
  int dev_ready;
  /* coherence of the engine-level array classes (RNN_AMD_WEIGHTS, _MOMENTUMS,
   * _DELTAS): bit set = that copy is current */
  int host_valid, dev_valid;
  int has_momentum, has_aux, has_delta;
  /* device scratch */
  void *d_scratch;      /* 256 (value,index) pairs for tall poppy + ranges */
  int *d_ranges;        /* up to 64 (start,len) pairs                       */
  int *d_mranges;       /* [Scap][65] pairs: one range list per stream (multi-head loss) */
  int *d_mclass;        /* [Scap] each stream's own class head                */
  void *d_group;        /* staging of the class-group loss: offsets, sizes, targets, weights */
  size_t d_group_bytes;
  float *d_dense;       /* staging for dense inputs, [Scap+Fcap][input_size] */
  float *delta_own;     /* library-owned ih_delta||ho_delta                */
  int delta_external;
  /* The delta sums of the last set call, kept as the un-summed planes the GEMM left (in a workspace of their own)
   * for the rnn_apply_learning that normally follows to sum on its way -- the separate calls then cost what the
   * one-call text step costs, a k_delta_finalize launch less.  Whoever else wants the delta arrays gets them summed
   * first (deltas_materialize). */
  u8 *active_host; /* what b.active holds (the last active mask sent), or NULL */
  int active_host_n;
  /* the top layer's backprop of these rows has been done with the loss (rnn_amd_set_opinion_sigmoid_mse /
   * _grouped_softmax): the next rnn_amd_set_calc_deltas over them skips it -- unless anything touched the weights, the
   * hidden rows or the output error in between (top_done_clear); masked: only for the active flags in active_host */
  int top_done, top_done_row0, top_done_n, top_done_masked;
  u8 *top_done_mask; /* masked: the streams whose backprop was done (the loss's `trained` flags) */
  RamdPendingDelta kept;
  int kept_live;
  float *d_kept_slab;
  size_t kept_floats;
  /* h_error/i_error images are rebuilt lazily from ehi */
  int err_pending, err_row0, err_nrows;
  /* last per-stream scalars pushed to the device */
  float *lr_pushed;
  int *idx_pushed;
  /* rnn_bptt_clear_deltas was called and nothing has needed the zeros yet: an accumulating
   * calc_deltas that follows simply does not accumulate (deltas_materialize otherwise) */
  int deltas_zero_pending;
  /* noise generated ahead (noise_speculate): valid while nothing else has moved the device's
   * generators since (rng_version) */
  unsigned long rng_version;
  /* Two passes' worth of buffers: sp[k] holds (or is being filled with) the noise of one coming pass and the generator
   * states after it; sp_head is the one the NEXT forward pass takes.  `version`: rng_version as it must stand when that
   * pass begins; `assumed_classes` != 0: generated before the multi-head loss in front of it had its classes -- on the
   * assumption that every stream's own class is one of that many heads, checked when the loss comes (multi_loss). */
  struct {
    float *noise;
    void *states;
    void *done; /* hipEvent_t */
    int pending, row0, n, assumed_classes;
    unsigned long version;
    float dev;
  } sp[2];
  int sp_head;
  int sp_adopted;        /* the buffer the set's last forward pass took its noise and states from, or -1 */
  int mclass_in_range;   /* every stream's class of the last multi-head upload is one of the heads */
  void *spec_go;         /* hipEvent_t */
  int scalars_dev_valid; /* device mef/ih_scale newer than the host structs */
  /* this engine hosts ONE SHARD of a training set spread over the ranks of the process group
   * (rnn_amd_new_training_set_shard, rnn_amd_set_shard, or a training set opened after
   * rnn_amd_dist_init): only then are the replicated host draws (weight noise, perforation,
   * random damage) a collective that takes rank 0's generator (ramd_shared_rng) */
  int sharded;
  int sharded_sets, sharded_sticky; /* open sets that shard the engine's streams; sharded for good (rnn_amd_set_shard, the shard constructor): sharded = sticky || sets > 0 */
  int mheads_alen; /* symbols per head of the last multi-head loss (0: none yet) */
  /* the exchange step as kernel-issued peer traffic (rnn_amd_set_exchange_join): every rank's delta and weight
   * arrays as device pointers valid HERE (own ones at index xchg_rank), the shared arrival counters */
  int xchg_world, xchg_rank, xchg_lockstep;
  float *xchg_delta[8], *xchg_ihw[8], *xchg_how[8];
  void *xchg_opened[8][3];   /* what hipIpcOpenMemHandle returned (to close), NULL for same-process peers */
  unsigned *xchg_flags_dev;  /* the callers' shared host counters, mapped                                  */
  void *xchg_flags_host;
  unsigned xchg_seq;
} RamdEngine;

struct RnnAmdSet {
  RamdEngine *eng;
  RecurNN **nets;
  int n;
  int row0;     /* first training-stream row, or first forward-only index when fwd_only */
  int fwd_only; /* the set is made of forward-only clones (no bptt): opinion calls only */
  int global_first, global_count;
  int counts_shard;   /* this set made the engine's streams a shard when it was opened (RamdEngine.sharded_sets counts it) */
  /* > 0 during the multi-head step's forward pass: the number of heads whose leak decisions follow the pass, for the
   * early noise speculation (noise_speculate_from); -1 once that has been launched */
  int early_spec_classes;
};

static inline RamdPriv *ramd_priv(const RecurNN *net) {
  RamdPriv *p = (RamdPriv *)net->mem;
  if (!p || p->magic != RAMD_MAGIC) {
    fprintf(stderr, "librecur_amd: RecurNN %p was not created by this library\n", (void *)net);
    abort();
  }
  return p;
}

/* rnn_core.c */
void *ramd_zalloc(size_t bytes);
RamdEngine *ramd_engine_of(RecurNN *net);
void ramd_need_host(RecurNN *net, int what);
void ramd_host_wrote(RecurNN *net, int what);
void ramd_require_device(const char *what);
void ramd_rng_to_host(RecurNN *net);
void ramd_rng_from_host(RecurNN *net);

/* dist.c */
int ramd_dist_active(void);
void ramd_dist_bcast(void *host, size_t bytes, int root);
void ramd_dist_all_reduce_on(void *device_buffer, size_t n_floats, void *stream);
rand_ctx *ramd_shared_rng(RecurNN *net, rand_ctx *tmp);

/* rnn_init.c: Jenkins PRNG (recur-rng.h) */
uint64_t ramd_rand64(rand_ctx *x);
void ramd_init_rand64(rand_ctx *x, uint64_t seed);
void ramd_init_rand64_maybe_randomly(rand_ctx *x, uint64_t seed);
double ramd_rand_double(rand_ctx *x);
int ramd_rand_small_int(rand_ctx *x, int cap);
float ramd_cheap_gaussian_noise(rand_ctx *x);
float ramd_fast_expf(float x);

#endif
