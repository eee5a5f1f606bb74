/* rnn_core.c -- host side of the RNN core (gnu11 C).
 *
 * Implements the reference's RecurNN API (recur-nn.h:269-334; behaviour of
 * recur-nn.c and recur-nn-init.c) on top of the HIP kernels in kernels_*.hip.
 * The structs the caller sees are host memory, as the ABI demands; the device
 * holds the authoritative copy of anything that has been computed on, and this
 * file is the coherence protocol between the two plus the order of launches.
 *
 * There is no CPU implementation of the arithmetic here: without a HIP device
 * every compute entry point aborts (ramd_require_device).
 */
#include "rnn_host.h"
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <unistd.h>
#include <time.h>

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "librecur_amd: HIP error \"%s\" at %s:%d\n", hipGetErrorString(e_), \
              __FILE__, __LINE__);                                                 \
      abort();                                                                     \
    }                                                                              \
  } while (0)

static void dsync(void);
static int g_device = -1;      /* -1: not initialised */
static int g_device_count = -1;
static hipStream_t g_stream = NULL;
static RamdEngine *g_engines = NULL;

/* ------------------------------------------------------------------ device -- */

int rnn_amd_device_count(void) {
  if (g_device_count < 0) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
      n = 0;
    }
    g_device_count = n;
  }
  return g_device_count;
}

void ramd_require_device(const char *what) {
  if (rnn_amd_device_count() <= 0) {
    fprintf(stderr,
            "librecur_amd: %s needs a HIP device (gfx950) and none is visible; "
            "this library has no CPU fallback\n",
            what);
    abort();
  }
  if (g_device < 0) {
    int dev = 0;
    const char *lr = getenv("LOCAL_RANK");
    if (lr && *lr) {
      dev = atoi(lr) % rnn_amd_device_count();
    }
    HIP_OK(hipSetDevice(dev));
    g_device = dev;
  }
}

void rnn_amd_use_device(int device, void *hip_stream) {
  if (rnn_amd_device_count() <= 0) {
    ramd_require_device("rnn_amd_use_device");
  }
  HIP_OK(hipSetDevice(device));
  g_device = device;
  g_stream = (hipStream_t)hip_stream;
}

void *rnn_amd_current_stream(void) { return (void *)g_stream; }
const char *rnn_amd_version(void) { return "recur_amd 0.1 (gfx950, fp32 MFMA)"; }
void rnn_amd_synchronize(void) {
  if (g_device >= 0) {
    dsync();
    HIP_OK(hipDeviceSynchronize());
  }
}
void rnn_amd_kernel_time_enable(int enable) { ramd_timing_enable(enable); }
double rnn_amd_kernel_time_ms(int which, long *launches, int reset) {
  return ramd_timing_ms(which, launches, reset);
}

static void *dev_alloc(size_t bytes) {
  void *p = NULL;
  if (bytes == 0) {
    bytes = 16;
  }
  HIP_OK(hipMalloc(&p, bytes));
  HIP_OK(hipMemsetAsync(p, 0, bytes, g_stream));
  return p;
}
/* The arrays that PEERS read and write in the kernel-issued exchange (weights, delta sums): RECUR_AMD_XCHG_FINEGRAINED=1
 * makes them fine-grained device allocations (hipExtMallocWithFlags: coherent between devices inside a kernel, not only
 * at kernel boundaries) -- for a node on which bench.py's start-up cross-check finds the coarse-grained default's
 * replicas differing (nothing of the kind has been seen: no multi-GPU node has run this yet).
 * Costs the owner's own kernels their L2 for these arrays (measured on one GPU: profiles/NOTES_r06.md). */
static void *dev_alloc_exchanged(size_t bytes) {
  static int fine = -1;
  if (fine < 0) {
    const char *v = getenv("RECUR_AMD_XCHG_FINEGRAINED");
    fine = v && *v == '1';
  }
  if (!fine) {
    return dev_alloc(bytes);
  }
  void *p = NULL;
  HIP_OK(hipExtMallocWithFlags(&p, bytes ? bytes : 16, hipDeviceMallocFinegrained));
  HIP_OK(hipMemsetAsync(p, 0, bytes ? bytes : 16, g_stream));
  return p;
}
static void dev_free(void *p) {
  if (p) {
    HIP_OK(hipFree(p));
  }
}
static void h2d(void *d, const void *h, size_t bytes) {
  if (bytes) {
    HIP_OK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, g_stream));
  }
}
static void d2h(void *h, const void *d, size_t bytes) {
  if (bytes) {
    HIP_OK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, g_stream));
  }
}
static unsigned long g_syncs = 0; /* completed stream synchronisations */
static void dsync(void) {
  HIP_OK(hipStreamSynchronize(g_stream));
  g_syncs++;
  if (ramd_chain_abort_word()) {
    const unsigned code = ramd_chain_abort_word();
    if (code == 6) {
      fprintf(stderr, "librecur_amd: a rank waited its time out (20 s; RECUR_AMD_XCHG_BARRIER_TIMEOUT_S) at a barrier of the kernel-issued exchange: not every rank "
                      "reached it (did one die, or drive fewer generations?); the replicas are out of step.\n");
    } else if (code == 2) {
      fprintf(stderr, "librecur_amd: a workgroup of the one-launch BPTT chain was not on the XCD its number implies (code 2): "
                      "something else ran on this GPU beside the chain and the dispatcher interleaved the two launches; "
                      "its results are invalid.  That form is only taken with RECUR_AMD_XCD_STATIC=1: drop it (the default "
                      "takes the seat from the CU a workgroup runs on, whatever runs beside it).\n");
    } else {
      fprintf(stderr, "librecur_amd: the one-launch BPTT chain gave up (code %u: its 256 workgroups were "
                      "not all resident, one per CU, or a hand-off timed out); its results are invalid.  "
                      "Set RECUR_AMD_CHAIN_CHECK=1 (every launch checked, a chain that gives up is redone) or "
                      "RECUR_AMD_CHAIN_PERSIST=0 (the launch-per-step chain).\n", code);
    }
    abort();
  }
}

/* ---------------------------------------------------------------- mailbox -- */

/* The per-net calls move a handful of small vectors each way (the caller's inputs or output
 * error in; the layers, the error images and two scalars out).  As separate hipMemcpyAsync
 * calls on pageable memory each of them costs a staging copy and an implicit wait; here they
 * are queued, carried by ONE kernel launch per direction through a pinned, device-mapped
 * mailbox, and a call ends with one stream synchronisation.  mail_in snapshots the host
 * data at once (the caller may change it as soon as we return), mail_out delivers after
 * the synchronisation in mail_out_flush. */
#define MAIL_SEGS 12
#define MAIL_WORDS (1u << 18) /* per direction: 1 MB */
static struct {
  unsigned *buf; /* [2][MAIL_WORDS] pinned: inbound half, outbound half */
  void *dst[MAIL_SEGS];
  const void *src[MAIL_SEGS];
  unsigned n[MAIL_SEGS];
  int nseg;
  unsigned cur;
  unsigned long in_sync; /* g_syncs when the inbound half was last handed to the device */
  int in_busy;
} g_mail;

static void mail_ready(void) {
  if (!g_mail.buf) {
    HIP_OK(hipHostMalloc((void **)&g_mail.buf, 2 * (size_t)MAIL_WORDS * sizeof(unsigned),
                         hipHostMallocPortable | hipHostMallocMapped));
  }
}
static void mail_in_flush(void) {
  if (g_mail.nseg) {
    ramd_launch_segcopy(g_stream, g_mail.nseg, g_mail.dst, g_mail.src, g_mail.n);
    g_mail.nseg = 0;
    g_mail.in_busy = 1;
    g_mail.in_sync = g_syncs;
  }
}
static void mail_in(void *dev, const void *host, size_t bytes) {
  unsigned words = (unsigned)(bytes / 4);
  if (!bytes) {
    return;
  }
  mail_ready();
  if ((bytes & 3) || words > MAIL_WORDS) { /* not this route's kind of copy */
    mail_in_flush();
    h2d(dev, host, bytes);
    dsync(); /* the caller's buffer is its own again when we return */
    return;
  }
  if (g_mail.nseg == 0 && g_mail.in_busy) {
    /* the inbound half may still be being read by the launch that carried the last batch */
    if (g_mail.in_sync == g_syncs && g_mail.cur + words > MAIL_WORDS) {
      dsync();
    }
    if (g_mail.in_sync != g_syncs) {
      g_mail.in_busy = 0;
      g_mail.cur = 0;
    }
  }
  if (g_mail.nseg == MAIL_SEGS || g_mail.cur + words > MAIL_WORDS) {
    mail_in_flush();
    dsync();
    g_mail.in_busy = 0;
    g_mail.cur = 0;
  }
  unsigned *slot = g_mail.buf + g_mail.cur;
  memcpy(slot, host, bytes);
  g_mail.dst[g_mail.nseg] = dev;
  g_mail.src[g_mail.nseg] = slot;
  g_mail.n[g_mail.nseg] = words;
  g_mail.nseg++;
  g_mail.cur += words;
}
/* outbound: queue, then one launch + one synchronisation + the deliveries */
static struct {
  void *dst[MAIL_SEGS];
  const void *src[MAIL_SEGS];
  unsigned n[MAIL_SEGS];
  void *host[MAIL_SEGS];
  int nseg;
  unsigned cur;
} g_mail_out;
static void mail_out_flush(void) {
  mail_in_flush();
  if (g_mail_out.nseg) {
    ramd_launch_segcopy(g_stream, g_mail_out.nseg, g_mail_out.dst, g_mail_out.src, g_mail_out.n);
  }
  dsync();
  for (int g = 0; g < g_mail_out.nseg; g++) {
    memcpy(g_mail_out.host[g], g_mail_out.dst[g], (size_t)g_mail_out.n[g] * 4);
  }
  g_mail_out.nseg = 0;
  g_mail_out.cur = 0;
}
static void mail_out(void *host, const void *dev, size_t bytes) {
  unsigned words = (unsigned)(bytes / 4);
  if (!bytes) {
    return;
  }
  mail_ready();
  if ((bytes & 3) || words > MAIL_WORDS) {
    d2h(host, dev, bytes); /* completes with the synchronisation of the flush */
    return;
  }
  if (g_mail_out.nseg == MAIL_SEGS || g_mail_out.cur + words > MAIL_WORDS) {
    mail_out_flush();
  }
  unsigned *slot = g_mail.buf + MAIL_WORDS + g_mail_out.cur;
  g_mail_out.dst[g_mail_out.nseg] = slot;
  g_mail_out.src[g_mail_out.nseg] = dev;
  g_mail_out.n[g_mail_out.nseg] = words;
  g_mail_out.host[g_mail_out.nseg] = host;
  g_mail_out.nseg++;
  g_mail_out.cur += words;
}

/* The set calls' small per-generation uploads (symbols, targets, dense input rows): snapshot into
 * the pinned mailbox and carried by a launch on the stream -- the caller's buffer is free when we
 * return and nobody waits for the device; what does not fit or is not word-sized goes the old way
 * (copy, then wait).  rows of `width` bytes, `host_pitch` apart, land contiguously at `dev`. */
static void upload_rows_q(void *dev, const void *host, size_t host_pitch, size_t width, size_t rows, int flush) {
  size_t bytes = width * rows;
  if (!bytes) {
    return;
  }
  mail_ready();
  if ((bytes & 3) || bytes / 4 > MAIL_WORDS) {
    mail_in_flush();
    if (host_pitch == width) {
      h2d(dev, host, bytes);
    } else {
      HIP_OK(hipMemcpy2DAsync(dev, width, host, host_pitch, width, rows, hipMemcpyHostToDevice, g_stream));
    }
    dsync();
    return;
  }
  if (host_pitch == width) {
    mail_in(dev, host, bytes);
  } else {
    /* pack the rows behind one another in the mailbox: one segment */
    unsigned words = (unsigned)(bytes / 4);
    if (g_mail.nseg == 0 && g_mail.in_busy && g_mail.in_sync != g_syncs) {
      g_mail.in_busy = 0;
      g_mail.cur = 0;
    }
    if (g_mail.nseg == MAIL_SEGS || g_mail.cur + words > MAIL_WORDS) {
      mail_in_flush();
      dsync();
      g_mail.in_busy = 0;
      g_mail.cur = 0;
    }
    unsigned *slot = g_mail.buf + g_mail.cur;
    for (size_t r = 0; r < rows; r++) {
      memcpy((char *)slot + r * width, (const char *)host + r * host_pitch, width);
    }
    g_mail.dst[g_mail.nseg] = dev;
    g_mail.src[g_mail.nseg] = slot;
    g_mail.n[g_mail.nseg] = words;
    g_mail.nseg++;
    g_mail.cur += words;
  }
  if (flush) {
    mail_in_flush();
  }
}
static void upload_rows(void *dev, const void *host, size_t host_pitch, size_t width, size_t rows) {
  upload_rows_q(dev, host, host_pitch, width, rows, 1);
}
static void upload(void *dev, const void *host, size_t bytes) { upload_rows_q(dev, host, bytes, bytes, 1, 1); }
/* queued: leaves with the next upload() / mail_in_flush() */
static void upload_q(void *dev, const void *host, size_t bytes) { upload_rows_q(dev, host, bytes, bytes, 1, 0); }

void *ramd_zalloc(size_t bytes) {
  void *p = NULL;
  if (posix_memalign(&p, 64, bytes ? bytes : 64)) {
    fprintf(stderr, "librecur_amd: cannot allocate %zu bytes\n", bytes);
    abort(); /* recur-common.h:95-132: allocation failure aborts */
  }
  memset(p, 0, bytes ? bytes : 64);
  return p;
}

/* ----------------------------------------------------------------- engines -- */

static RamdEngine *engine_new(RecurNN *owner) {
  RamdEngine *e = ramd_zalloc(sizeof(RamdEngine));
  e->owner = owner;
  e->sh.input_size = owner->input_size;
  e->sh.hidden_size = owner->hidden_size;
  e->sh.output_size = owner->output_size;
  e->sh.I = owner->i_size;
  e->sh.H = owner->h_size;
  e->sh.O = owner->o_size;
  e->sh.activation = owner->activation;
  e->ih_size = (size_t)owner->ih_size;
  e->ho_size = (size_t)owner->ho_size;
  e->host_valid = RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS;
  e->sp_adopted = -1;
  e->next = g_engines;
  g_engines = e;
  return e;
}

static hipStream_t g_side = NULL; /* noise_speculate's stream */
static void engine_free_device(RamdEngine *e) {
  if (!e->dev_ready) {
    return;
  }
  RamdBuffers *b = &e->b;
  if ((e->sp[0].pending || e->sp[1].pending) && g_side) {
    HIP_OK(hipStreamSynchronize(g_side)); /* it writes into the buffers freed below */
  }
  for (int k = 0; k < 2; k++) {
    e->sp[k].pending = 0;
    dev_free(e->sp[k].noise);
    dev_free(e->sp[k].states);
    e->sp[k].noise = NULL;
    e->sp[k].states = NULL;
  }
  e->sp_adopted = -1;
  b->noise_spec = NULL;
  b->rng_spec = NULL;
  dev_free(b->ih_w); dev_free(b->ho_w); dev_free(b->ih_m); dev_free(b->ho_m);
  dev_free(b->ih_aux); dev_free(b->ho_aux); dev_free(e->delta_own);
  dev_free(b->arena); dev_free(b->hidden); dev_free(b->out); dev_free(b->o_error);
  dev_free(b->err_a); dev_free(b->err_b); dev_free(b->ehi); dev_free(b->esum);
  dev_free(b->coef); dev_free(b->ex); dev_free(b->esum_part); dev_free(b->zeros); dev_free(b->ones); dev_free(b->rng); dev_free(b->slab); dev_free(b->ho_slab); dev_free(b->idx); dev_free(b->lr); dev_free(b->mef);
  dev_free(b->ih_scale); dev_free(b->top_raw); dev_free(b->top_scaled); dev_free(b->bptt_err);
  dev_free(b->n_exec); dev_free(b->depth_log); dev_free(b->target); dev_free(b->hot);
  dev_free(b->xent);
  dev_free(b->active); dev_free(b->stat_err); dev_free(b->stat_ent); dev_free(b->stat_zero);
  dev_free(b->stat_depth); dev_free(b->stat_correct); dev_free(b->stat_count);
  dev_free(b->text);
  dev_free(b->bw); dev_free(b->bm); dev_free(b->baux); dev_free(b->bdelta);
  dev_free(b->binp); dev_free(b->bout); dev_free(b->berr); dev_free(b->bcarry); dev_free(b->blast);
  dev_free(e->d_scratch); dev_free(e->d_ranges); dev_free(e->d_dense);
  dev_free(e->d_mranges); dev_free(e->d_mclass); dev_free(e->d_group);
  dev_free(b->mheads_part);
  b->mheads_part = NULL;
  b->mheads_part_floats = 0;
  free(e->active_host); /* (b.active goes with the device arrays) */
  e->active_host = NULL;
  e->active_host_n = 0;
  free(e->top_done_mask); /* (sized by Scap, which may grow) */
  e->top_done_mask = NULL;
  e->top_done = 0;
  dev_free(e->d_kept_slab);
  e->d_kept_slab = NULL;
  e->kept_floats = 0;
  e->kept_live = 0;
  e->d_group = NULL;
  e->d_group_bytes = 0;
  e->d_mranges = NULL;
  e->d_mclass = NULL;
  free(e->lr_pushed);
  e->lr_pushed = NULL;
  free(e->idx_pushed);
  e->idx_pushed = NULL;
  memset(b, 0, sizeof(*b));
  e->delta_own = NULL;
  e->delta_external = 0;
  e->dev_ready = 0;
}

static void engine_delete(RamdEngine *e) {
  if (g_device >= 0) {
    dsync();
  }
  engine_free_device(e);
  if (e->spec_go) { /* noise_speculate's hand-over events */
    HIP_OK(hipEventDestroy((hipEvent_t)e->spec_go));
    HIP_OK(hipEventDestroy((hipEvent_t)e->sp[0].done));
    HIP_OK(hipEventDestroy((hipEvent_t)e->sp[1].done));
    e->spec_go = e->sp[0].done = e->sp[1].done = NULL;
  }
  /* clones may outlive the net that owns the weights (text-predict.c:654-656 deletes the
   * training set, then its confab and validation clones): detach them so that their own
   * rnn_delete_net finds no engine instead of a freed one */
  for (int i = 0; i < e->n_streams; i++) {
    if (e->streams[i] && e->streams[i] != e->owner) {
      RamdPriv *q = ramd_priv(e->streams[i]);
      q->eng = NULL;
      q->stream = q->fwd = -1;
    }
  }
  for (int i = 0; i < e->n_fwd; i++) {
    if (e->fwd[i] && e->fwd[i] != e->owner) {
      RamdPriv *q = ramd_priv(e->fwd[i]);
      q->eng = NULL;
      q->stream = q->fwd = -1;
    }
  }
  RamdEngine **pp = &g_engines;
  while (*pp && *pp != e) {
    pp = &(*pp)->next;
  }
  if (*pp) {
    *pp = e->next;
  }
  free(e->streams);
  free(e->fwd);
  free(e);
}

static void engine_attach(RamdEngine *e, RecurNN *net) {
  RamdPriv *p = ramd_priv(net);
  p->eng = e;
  p->host_valid = 1;
  p->dev_valid = 0;
  if (net->bptt) {
    if (e->n_streams == 0) {
      e->sh.D = net->bptt->depth;
    } else if (net->bptt->depth != e->sh.D) {
      fprintf(stderr, "librecur_amd: nets sharing weights must share the BPTT depth (%d vs %d)\n",
              net->bptt->depth, e->sh.D);
      abort();
    }
    if (e->n_streams == e->cap_streams) {
      e->cap_streams = e->cap_streams ? e->cap_streams * 2 : 4;
      e->streams = realloc(e->streams, e->cap_streams * sizeof(RecurNN *));
    }
    p->stream = e->n_streams;
    p->fwd = -1;
    e->streams[e->n_streams++] = net;
  } else {
    if (e->n_fwd == e->cap_fwd) {
      e->cap_fwd = e->cap_fwd ? e->cap_fwd * 2 : 4;
      e->fwd = realloc(e->fwd, e->cap_fwd * sizeof(RecurNN *));
    }
    p->fwd = e->n_fwd;
    p->stream = -1;
    e->fwd[e->n_fwd++] = net;
  }
}

RamdEngine *ramd_engine_of(RecurNN *net) {
  RamdPriv *p = ramd_priv(net);
  if (!p->eng) {
    /* a net made by rnn_new without OWN_WEIGHTS whose caller pointed it at
     * somebody's weights by hand: find that somebody */
    for (RamdEngine *e = g_engines; e; e = e->next) {
      if (e->owner->ih_weights == net->ih_weights) {
        engine_attach(e, net);
        break;
      }
    }
    if (!p->eng) {
      fprintf(stderr, "librecur_amd: net %p has no weights known to the library\n", (void *)net);
      abort();
    }
  }
  return p->eng;
}

static int state_row(const RamdEngine *e, const RamdPriv *p) {
  return p->stream >= 0 ? p->stream : e->sh.Scap + p->fwd;
}

/* -------------------------------------------- host <-> device: one stream -- */

/* A generator of this engine has been written from the host (a caller's draw, a per-net call's upload): nothing that was
 * generated ahead from the device's states can be right any more.  (The version numbers of the two-ahead buffers are
 * PREDICTED -- one pass and one loss on -- so a different sequence of the same length, an upload instead of the loss,
 * would have been accepted: ADVICE.md round 4.) */
static void rng_written_from_host(RamdEngine *e) {
  e->rng_version++;
  e->sp[0].pending = e->sp[1].pending = 0;
}

static void stream_copy(RamdEngine *e, RecurNN *net, int to_device) {
  RamdPriv *p = ramd_priv(net);
  const RamdShape *s = &e->sh;
  RamdBuffers *b = &e->b;
  int r = state_row(e, p);
  size_t I = s->I, H = s->H, O = s->O;
#define COPY(dptr, hptr, n)                                                        \
  do {                                                                             \
    if (to_device) h2d((dptr), (hptr), (n) * sizeof(float));                       \
    else d2h((hptr), (dptr), (n) * sizeof(float));                                 \
  } while (0)
  COPY(b->hidden + (size_t)r * H, net->hidden_layer, H);
  COPY(b->out + (size_t)r * O, net->output_layer, O);
  if (to_device) {
    h2d((char *)b->rng + (size_t)r * sizeof(rand_ctx), &net->rng, sizeof(rand_ctx));
    rng_written_from_host(e);
  } else {
    d2h(&net->rng, (char *)b->rng + (size_t)r * sizeof(rand_ctx), sizeof(rand_ctx));
  }
  if (p->stream >= 0) {
    RecurNNBPTT *bp = net->bptt;
    int j = p->stream;
    /* history: host [D][I] per net <-> device [D][Scap][I] */
    if (to_device) {
      HIP_OK(hipMemcpy2DAsync(b->arena + (size_t)j * I, (size_t)s->Scap * I * sizeof(float),
                              bp->history, I * sizeof(float), I * sizeof(float), s->D,
                              hipMemcpyHostToDevice, g_stream));
    } else {
      HIP_OK(hipMemcpy2DAsync(bp->history, I * sizeof(float), b->arena + (size_t)j * I,
                              (size_t)s->Scap * I * sizeof(float), I * sizeof(float), s->D,
                              hipMemcpyDeviceToHost, g_stream));
    }
    COPY(b->o_error + (size_t)j * O, bp->o_error, O);
    COPY(b->err_a + (size_t)j * I, bp->h_error, I);
    COPY(b->err_b + (size_t)j * I, bp->i_error, I);
    COPY(b->mef + j, &bp->min_error_factor, 1);
    COPY(b->ih_scale + j, &bp->ih_scale, 1);
    if (to_device) {
      h2d(b->idx + j, &bp->index, sizeof(int));
      h2d(b->lr + j, &bp->learn_rate, sizeof(float));
      e->lr_pushed[j] = bp->learn_rate;
      e->idx_pushed[j] = bp->index;
    }
  } else {
    COPY(b->arena + ((size_t)s->D * s->Scap + p->fwd) * I, net->input_layer, I);
  }
#undef COPY
}

static void top_done_clear(RamdEngine *e) { e->top_done = 0; }

static void err_flush(RamdEngine *e) {
  if (e->err_pending) {
    ramd_launch_err_writeback(g_stream, &e->sh, &e->b, e->err_row0, e->err_nrows);
    e->err_pending = 0;
  }
}

static void stream_need_host(RamdEngine *e, RecurNN *net) {
  RamdPriv *p = ramd_priv(net);
  if (!p->host_valid && e->dev_ready) {
    err_flush(e);
    stream_copy(e, net, 0);
    dsync();
  }
  p->host_valid = 1;
}

static void stream_need_dev(RamdEngine *e, RecurNN *net) {
  RamdPriv *p = ramd_priv(net);
  if (!p->dev_valid) {
    stream_copy(e, net, 1);
    dsync(); /* the host arrays may change as soon as we return */
    top_done_clear(e);
  }
  p->dev_valid = 1;
}

/* ------------------------------------------- host <-> device: the big arrays -- */

/* rnn_bptt_clear_deltas is lazy: the callers that clear per generation (gstclassify, rnnca) go on
 * with an accumulating rnn_bptt_calc_deltas, which can just as well not accumulate -- two memset
 * launches less.  Whoever else needs the delta arrays gets the zeros first. */
static void deltas_materialize(RamdEngine *e) {
  if (e->kept_live) { /* (never together with a pending clear: whichever came later dropped the other) */
    e->kept_live = 0;
    if (e->dev_ready) {
      ramd_launch_pending_finalize(g_stream, &e->kept);
    }
  }
  if (e->deltas_zero_pending) {
    e->deltas_zero_pending = 0;
    if (e->dev_ready) {
      ramd_launch_clear_deltas(g_stream, &e->sh, &e->b);
    }
  }
}

static void engine_need_host(RamdEngine *e, int what) {
  if (what & RNN_AMD_DELTAS) {
    deltas_materialize(e);
  }
  RecurNN *o = e->owner;
  RecurExtraLayer *bl = e->sh.bI ? o->bottom_layer : NULL;
  size_t bn = (size_t)e->sh.bI * e->sh.bO;
  int fetch = what & ~e->host_valid & (RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  if (fetch && e->dev_ready) {
    if (fetch & RNN_AMD_WEIGHTS) {
      d2h(o->ih_weights, e->b.ih_w, e->ih_size * sizeof(float));
      d2h(o->ho_weights, e->b.ho_w, e->ho_size * sizeof(float));
      if (bl) d2h(bl->weights, e->b.bw, bn * sizeof(float));
    }
    if ((fetch & RNN_AMD_MOMENTUMS) && bl) {
      d2h(bl->momentums, e->b.bm, bn * sizeof(float));
      if (bl->aux) d2h(bl->aux, e->b.baux, bn * sizeof(float));
    }
    if ((fetch & RNN_AMD_DELTAS) && bl) {
      d2h(bl->delta, e->b.bdelta, bn * sizeof(float));
      d2h(bl->o_error, e->b.bcarry + (size_t)e->b.bcarry_cur * e->sh.bO, e->sh.bO * sizeof(float));
    }
    if ((fetch & RNN_AMD_MOMENTUMS) && o->bptt) {
      if (e->has_momentum) {
        d2h(o->bptt->ih_momentum, e->b.ih_m, e->ih_size * sizeof(float));
        d2h(o->bptt->ho_momentum, e->b.ho_m, e->ho_size * sizeof(float));
      }
      if (e->has_aux) {
        d2h(o->bptt->ih_aux, e->b.ih_aux, e->ih_size * sizeof(float));
        d2h(o->bptt->ho_aux, e->b.ho_aux, e->ho_size * sizeof(float));
      }
    }
    if ((fetch & RNN_AMD_DELTAS) && o->bptt && e->has_delta) {
      d2h(o->bptt->ih_delta, e->b.ih_delta, e->ih_size * sizeof(float));
      d2h(o->bptt->ho_delta, e->b.ho_delta, e->ho_size * sizeof(float));
    }
    dsync();
  }
  e->host_valid |= fetch;
}

static void engine_need_dev(RamdEngine *e, int what) {
  if (what & RNN_AMD_DELTAS) {
    deltas_materialize(e);
  }
  RecurNN *o = e->owner;
  RecurExtraLayer *bl = e->sh.bI ? o->bottom_layer : NULL;
  size_t bn = (size_t)e->sh.bI * e->sh.bO;
  int push = what & ~e->dev_valid & (RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  if (!push) {
    return;
  }
  top_done_clear(e);
  if (push & RNN_AMD_WEIGHTS) {
    h2d(e->b.ih_w, o->ih_weights, e->ih_size * sizeof(float));
    h2d(e->b.ho_w, o->ho_weights, e->ho_size * sizeof(float));
    if (bl) h2d(e->b.bw, bl->weights, bn * sizeof(float));
  }
  if ((push & RNN_AMD_MOMENTUMS) && bl) {
    h2d(e->b.bm, bl->momentums, bn * sizeof(float));
    if (bl->aux) h2d(e->b.baux, bl->aux, bn * sizeof(float));
  }
  if ((push & RNN_AMD_DELTAS) && bl) {
    h2d(e->b.bdelta, bl->delta, bn * sizeof(float));
    h2d(e->b.bcarry + (size_t)e->b.bcarry_cur * e->sh.bO, bl->o_error, e->sh.bO * sizeof(float));
  }
  if ((push & RNN_AMD_MOMENTUMS) && o->bptt) {
    if (e->has_momentum) {
      h2d(e->b.ih_m, o->bptt->ih_momentum, e->ih_size * sizeof(float));
      h2d(e->b.ho_m, o->bptt->ho_momentum, e->ho_size * sizeof(float));
    }
    if (e->has_aux) {
      h2d(e->b.ih_aux, o->bptt->ih_aux, e->ih_size * sizeof(float));
      h2d(e->b.ho_aux, o->bptt->ho_aux, e->ho_size * sizeof(float));
    }
  }
  if ((push & RNN_AMD_DELTAS) && o->bptt && e->has_delta) {
    h2d(e->b.ih_delta, o->bptt->ih_delta, e->ih_size * sizeof(float));
    h2d(e->b.ho_delta, o->bptt->ho_delta, e->ho_size * sizeof(float));
  }
  dsync();
  e->dev_valid |= push;
}

static void engine_dev_wrote(RamdEngine *e, int what) {
  e->dev_valid |= what;
  e->host_valid &= ~what;
  if (what & RNN_AMD_WEIGHTS) {
    top_done_clear(e);
  }
}

/* public: bring host copies up to date */
void ramd_need_host(RecurNN *net, int what) {
  RamdEngine *e = ramd_engine_of(net);
  engine_need_host(e, what);
  if (what & RNN_AMD_ALL_STREAMS) {
    for (int i = 0; i < e->n_streams; i++) {
      stream_need_host(e, e->streams[i]);
    }
    for (int i = 0; i < e->n_fwd; i++) {
      stream_need_host(e, e->fwd[i]);
    }
  } else if (what & RNN_AMD_STREAM) {
    stream_need_host(e, net);
  }
}

void ramd_host_wrote(RecurNN *net, int what) {
  RamdEngine *e = ramd_engine_of(net);
  int big = what & (RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  /* a partial host write on top of a stale host copy would lose the rest */
  engine_need_host(e, big);
  e->host_valid |= big;
  e->dev_valid &= ~big;
  if (what & RNN_AMD_ALL_STREAMS) {
    for (int i = 0; i < e->n_streams; i++) {
      stream_need_host(e, e->streams[i]);
      ramd_priv(e->streams[i])->dev_valid = 0;
    }
    for (int i = 0; i < e->n_fwd; i++) {
      stream_need_host(e, e->fwd[i]);
      ramd_priv(e->fwd[i])->dev_valid = 0;
    }
  } else if (what & RNN_AMD_STREAM) {
    stream_need_host(e, net);
    ramd_priv(net)->dev_valid = 0;
  }
}

/* A host-side draw from net->rng (charmodel-predict.c:52, 93) while the stream's state
 * lives on the device: fetch only the generator before, push only it back after. */
void ramd_rng_to_host(RecurNN *net) {
  RamdPriv *p = ramd_priv(net);
  RamdEngine *e = p->eng;
  if (e && e->dev_ready && p->dev_valid && !p->host_valid) {
    d2h(&net->rng, (char *)e->b.rng + (size_t)state_row(e, p) * sizeof(rand_ctx), sizeof(rand_ctx));
    dsync();
  }
}
void ramd_rng_from_host(RecurNN *net) {
  RamdPriv *p = ramd_priv(net);
  RamdEngine *e = p->eng;
  if (e && e->dev_ready && p->dev_valid) {
    h2d((char *)e->b.rng + (size_t)state_row(e, p) * sizeof(rand_ctx), &net->rng, sizeof(rand_ctx));
    rng_written_from_host(e);
    dsync();
  }
}

void rnn_amd_sync_host(RecurNN *net, int what) { ramd_need_host(net, what); }
void rnn_amd_host_written(RecurNN *net, int what) { ramd_host_wrote(net, what); }

/* Allocate (or grow) the device image.  Growing evicts everything to the host
 * first and starts over; it only happens when nets are cloned after the first
 * device call. */
static void engine_ensure_device(RamdEngine *e) {
  ramd_require_device("this call");
  /* the bottom layer is attached after rnn_new returns (recur-nn-init.c:215) */
  RecurExtraLayer *obl = e->owner->bottom_layer;
  int bI = obl ? obl->i_size : 0, bO = obl ? obl->o_size : 0;
  if (obl && (obl->output_size != e->sh.input_size || bI < 4 || bO < 4)) {
    fprintf(stderr, "librecur_amd: bottom layer of %d outputs under a net of %d inputs\n",
            obl->output_size, e->sh.input_size);
    abort();
  }
  int same_bottom = bI == e->sh.bI && bO == e->sh.bO;
  if (e->dev_ready && e->sh.Scap >= e->n_streams && e->sh.Fcap >= e->n_fwd && same_bottom) {
    return;
  }
  if (e->dev_ready && e->xchg_world) {
    fprintf(stderr, "librecur_amd: the device image has to grow (a clone made after rnn_amd_set_exchange_join?) while "
                    "other ranks hold pointers into it; leave the exchange first\n");
    abort();
  }
  unsigned char *keep_text = NULL;
  int keep_text_len = 0;
  float *keep_ext = NULL;
  if (e->dev_ready) {
    engine_need_host(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
    for (int i = 0; i < e->n_streams; i++) {
      RamdPriv *p = ramd_priv(e->streams[i]);
      if (p->dev_valid && !p->host_valid) {
        stream_need_host(e, e->streams[i]);
      }
    }
    for (int i = 0; i < e->n_fwd; i++) {
      RamdPriv *p = ramd_priv(e->fwd[i]);
      if (p->dev_valid && !p->host_valid) {
        stream_need_host(e, e->fwd[i]);
      }
    }
    dsync();
    /* what the caller registered with the engine survives the regrow: the text of the
     * host-free epoch loop and the external delta buffer of the multi-GPU path */
    if (e->b.text && e->b.text_len > 0) {
      keep_text_len = e->b.text_len;
      keep_text = malloc((size_t)keep_text_len);
      d2h(keep_text, e->b.text, (size_t)keep_text_len);
      dsync();
    }
    if (e->delta_external) {
      keep_ext = e->b.ih_delta;
    }
    engine_free_device(e);
  }
  RamdShape *s = &e->sh;
  RamdBuffers *b = &e->b;
  RecurNN *o = e->owner;
  s->bI = bI;
  s->bO = bO;
  s->b_in = obl ? obl->input_size : 0;
  s->b_out = obl ? obl->output_size : 0;
  s->Scap = (RAMD_MAX(e->n_streams, 1) + 15) / 16 * 16; /* whole 16-row tiles (k_chain_persist's PAD launches run over the rows above a set) */
  s->Fcap = RAMD_MAX(e->n_fwd, 1);
  if (s->D < 1) {
    s->D = 1;
  }
  size_t I = s->I, H = s->H, O = s->O, S = s->Scap, F = s->Fcap, D = s->D;
  size_t fl = sizeof(float);
  if ((D + 1) * (S + F) * I >= ((size_t)1 << 31) || e->ih_size >= ((size_t)1 << 31)) {
    fprintf(stderr, "librecur_amd: %zu streams x depth %zu x %zu inputs exceeds the 2^31-element "
                    "offsets of the kernels\n", S, D, I);
    abort();
  }
  e->has_momentum = o->bptt && o->bptt->ih_momentum;
  e->has_aux = o->bptt && o->bptt->ih_aux;
  e->has_delta = o->bptt && o->bptt->ih_delta;
  b->ih_w = dev_alloc_exchanged(e->ih_size * fl);
  b->ho_w = dev_alloc_exchanged(e->ho_size * fl);
  b->ih_m = dev_alloc(e->ih_size * fl);
  b->ho_m = dev_alloc(e->ho_size * fl);
  if (e->has_aux) {
    b->ih_aux = dev_alloc(e->ih_size * fl);
    b->ho_aux = dev_alloc(e->ho_size * fl);
  }
  e->delta_own = dev_alloc_exchanged((e->ih_size + e->ho_size) * fl);
  b->ih_delta = e->delta_own;
  b->ho_delta = e->delta_own + e->ih_size;
  b->arena = dev_alloc(((D * S + F) * I + 128) * fl); /* + slack: k_delta_dma's rest tile reads up to 128 floats from column rows_core of the last row */
  b->hidden = dev_alloc((S + F) * H * fl);
  b->out = dev_alloc((S + F) * O * fl);
  b->o_error = dev_alloc(S * O * fl);
  b->err_a = dev_alloc(S * I * fl);
  b->err_b = dev_alloc(S * I * fl);
  b->ehi = dev_alloc((D + 1) * S * I * fl);
  b->esum = dev_alloc(D * S * fl);
  b->coef = dev_alloc(D * S * fl);
  {
    size_t nxp = ((size_t)(s->I - s->hidden_size) + 3) & ~(size_t)3;
    size_t tn = ((size_t)s->hidden_size + 31) / 32;
    b->ex = dev_alloc((D + 1) * S * nxp * fl);
    b->esum_part = dev_alloc(D * (tn + 1) * S * fl);
    b->zeros = dev_alloc(256);
    b->rng = dev_alloc((S + F) * sizeof(rand_ctx));
    b->ones = dev_alloc(S * fl);
    {
      float *one = malloc(S * fl);
      for (size_t i = 0; i < S; i++) one[i] = 1.0f;
      h2d(b->ones, one, S * fl);
      dsync();
      free(one);
    }
  }
  /* split-K workspace: up to 16 slabs of the largest GEMM output */
  {
    size_t per = RAMD_MAX(e->ih_size, RAMD_MAX((S + F) * I, (S + F) * H));
    per = RAMD_MAX(per, D * S * ((I - s->hidden_size + 3) & ~(size_t)3));
    size_t slabs = 16;
    const char *env = getenv("RECUR_AMD_MAX_SLABS");
    if (env && atoi(env) > 0) {
      slabs = (size_t)atoi(env);
    }
    b->slab_floats = per * slabs;
    /* ... and at least ONE slab of the two outputs that scale with o_size (the top layer's delta
     * GEMM, the output layer of all rows): a narrow net with a very wide output layer (40 hidden
     * units under 56 heads of 35 symbols) has them larger than 16 of everything else */
    b->slab_floats = RAMD_MAX(b->slab_floats, RAMD_MAX(e->ho_size, (S + F) * O));
    b->slab = dev_alloc(b->slab_floats * fl);
    b->ho_slab = e->ho_size <= ((size_t)1 << 20) ? dev_alloc(8 * e->ho_size * fl) : NULL;
  }
  b->idx = dev_alloc(S * sizeof(int));
  b->lr = dev_alloc(S * fl);
  b->mef = dev_alloc(S * fl);
  b->ih_scale = dev_alloc(S * fl);
  b->top_raw = dev_alloc(S * fl);
  b->top_scaled = dev_alloc(S * fl);
  b->bptt_err = dev_alloc(S * fl);
  b->n_exec = dev_alloc(S * sizeof(int));
  b->depth_log = dev_alloc(S * sizeof(int));
  b->target = dev_alloc((S + F) * sizeof(int));
  b->xent = dev_alloc((S + F) * sizeof(double));
  b->hot = dev_alloc((S + F) * sizeof(int));
  b->active = dev_alloc(S);
  b->stat_err = dev_alloc(S * sizeof(double));
  b->stat_ent = dev_alloc(S * sizeof(double));
  b->stat_zero = dev_alloc(S * sizeof(double));
  b->stat_depth = dev_alloc(S * sizeof(double));
  b->stat_correct = dev_alloc(S * sizeof(long long));
  b->stat_count = dev_alloc(S * sizeof(long long));
  e->d_scratch = dev_alloc(256 * 16);
  e->d_ranges = dev_alloc(130 * sizeof(int));
  e->d_dense = dev_alloc((S + F) * (size_t)RAMD_MAX(s->input_size, s->b_in) * fl);
  if (bI) {
    size_t bn = (size_t)bI * bO;
    b->bw = dev_alloc(bn * fl);
    b->bm = dev_alloc(bn * fl);
    b->baux = obl->aux ? dev_alloc(bn * fl) : NULL;
    b->bdelta = dev_alloc(bn * fl);
    b->binp = dev_alloc((S + F) * bI * fl);
    b->bout = dev_alloc((S + F) * bO * fl);
    b->berr = dev_alloc(S * bO * fl);
    b->bcarry = dev_alloc(2 * (size_t)bO * fl);
    b->bcarry_cur = 0;
    b->blast = dev_alloc(16);
    h2d(b->blast, obl->inputs + obl->input_size, fl); /* the host's buffer is kept current (set_opinion) */
    dsync();
  }
  e->lr_pushed = ramd_zalloc(S * sizeof(float));
  e->idx_pushed = ramd_zalloc(S * sizeof(int));
  if (keep_text) {
    b->text = dev_alloc((size_t)keep_text_len + 1);
    h2d(b->text, keep_text, (size_t)keep_text_len);
    b->text_len = keep_text_len;
    dsync();
    free(keep_text);
  }
  if (keep_ext) {
    b->ih_delta = keep_ext;
    b->ho_delta = keep_ext + e->ih_size;
    e->delta_external = 1;
  }
  e->dev_ready = 1;
  e->dev_valid = 0;
  e->err_pending = 0;
  for (int i = 0; i < e->n_streams; i++) {
    ramd_priv(e->streams[i])->dev_valid = 0;
  }
  for (int i = 0; i < e->n_fwd; i++) {
    ramd_priv(e->fwd[i])->dev_valid = 0;
  }
  dsync();
}

/* ------------------------------------------------- construction (host only) -- */

static size_t round4(size_t x) { return (x + 3) & ~(size_t)3; }

/* rnn_bptt_advance (recur-nn.c:696-704), host side */
static void host_advance(RecurNN *net) {
  RecurNNBPTT *bptt = net->bptt;
  bptt->index++;
  if (bptt->index == bptt->depth) {
    bptt->index -= bptt->depth;
  }
  net->input_layer = bptt->history + (size_t)bptt->index * net->i_size;
  net->real_inputs = net->input_layer + net->hidden_size + 1;
}

/* new_bptt (recur-nn-init.c:6-78): which arrays exist depends on the flags;
 * the order inside the block is ours. */
static RecurNNBPTT *bptt_new(RecurNN *net, int depth, float learn_rate, float momentum,
                             u32 flags) {
  RecurNNBPTT *bptt = ramd_zalloc(sizeof(RecurNNBPTT));
  int own_momentums = !(flags & RNN_NET_FLAG_NO_MOMENTUMS);
  int own_deltas = !(flags & RNN_NET_FLAG_NO_DELTAS);
  int aux_arrays = !!(flags & RNN_NET_FLAG_AUX_ARRAYS);
  size_t ih = (size_t)net->ih_size, ho = (size_t)net->ho_size;
  size_t I = net->i_size, O = net->o_size;
  size_t n = O + 2 * I + (size_t)depth * I;
  if (own_momentums) n += ih + ho;
  if (own_deltas) n += 2 * ih + ho;
  if (aux_arrays) n += ih + ho;
  float *fm = ramd_zalloc(n * sizeof(float));
  bptt->mem = fm;
  bptt->depth = depth;
  bptt->learn_rate = learn_rate;
  bptt->momentum = momentum;
  bptt->momentum_weight = RNN_MOMENTUM_WEIGHT;
#define TAKE(field, count) do { bptt->field = fm; fm += (count); } while (0)
  TAKE(o_error, O);
  TAKE(i_error, I);
  TAKE(h_error, I); /* i_size long so the two can be swapped, recur-nn-init.c:39-41 */
  TAKE(history, (size_t)depth * I);
  if (own_momentums) {
    TAKE(ih_momentum, ih);
    TAKE(ho_momentum, ho);
  }
  if (own_deltas) {
    TAKE(ih_delta, ih);
    TAKE(ho_delta, ho);
    TAKE(ih_delta_tmp, ih);
  }
  if (aux_arrays) {
    TAKE(ih_aux, ih);
    TAKE(ho_aux, ho);
  }
#undef TAKE
  bptt->index = 0;
  bptt->ho_scale = 1.0f;
  bptt->ih_scale = 1.0f;
  bptt->min_error_factor = BASE_MIN_ERROR_FACTOR * net->h_size;
  return bptt;
}

static RecurNN *net_new(uint input_size, uint hidden_size, uint output_size, u32 flags,
                        u64 rng_seed, const char *log_file, int bptt_depth, float learn_rate,
                        float momentum, float presynaptic_noise, rnn_activation activation,
                        RamdEngine *borrow) {
  RecurNN *net = ramd_zalloc(sizeof(RecurNN));
  /* padded sizes, recur-nn-init.c:87-91 */
  size_t i_size = round4((size_t)hidden_size + input_size + 1);
  size_t h_size = round4((size_t)hidden_size + 1);
  size_t o_size = round4(output_size);
  size_t ih_size = i_size * h_size, ho_size = h_size * o_size;
  net->i_size = (int)i_size;
  net->h_size = (int)h_size;
  net->o_size = (int)o_size;
  net->input_size = (int)input_size;
  net->hidden_size = (int)hidden_size;
  net->output_size = (int)output_size;
  net->ih_size = (int)ih_size;
  net->ho_size = (int)ho_size;
  net->generation = 0;
  net->flags = flags;
  net->presynaptic_noise = presynaptic_noise;
  if (activation >= RNN_ACTIVATION_LAST) {
    activation = RNN_RELU;
  }
  net->activation = activation;
  ramd_init_rand64_maybe_randomly(&net->rng, rng_seed);

  size_t n = RAMD_HDR_FLOATS + i_size + h_size + o_size;
  if (flags & RNN_NET_FLAG_OWN_WEIGHTS) {
    n += ih_size + ho_size;
  }
  float *fm = ramd_zalloc(n * sizeof(float));
  net->mem = fm;
  RamdPriv *priv = (RamdPriv *)fm;
  priv->magic = RAMD_MAGIC;
  priv->stream = priv->fwd = -1;
  priv->host_valid = 1;
  fm += RAMD_HDR_FLOATS;
  net->input_layer = fm; fm += i_size;
  net->hidden_layer = fm; fm += h_size;
  net->output_layer = fm; fm += o_size;
  if (flags & RNN_NET_FLAG_OWN_WEIGHTS) {
    net->ih_weights = fm; fm += ih_size;
    net->ho_weights = fm; fm += ho_size;
  }
  if (flags & RNN_NET_FLAG_OWN_BPTT) {
    net->bptt = bptt_new(net, bptt_depth, learn_rate, momentum, flags);
    host_advance(net); /* recur-nn-init.c:133: the first slot in use is 1 */
  } else {
    net->real_inputs = net->input_layer + net->hidden_size + 1;
  }
  if (flags & RNN_NET_FLAG_OWN_WEIGHTS) {
    engine_attach(engine_new(net), net);
  } else if (borrow) {
    engine_attach(borrow, net);
  }
  if (log_file) {
    rnn_set_log_file(net, log_file, flags & RNN_NET_FLAG_LOG_APPEND);
  }
  return net;
}

/* recur-nn.h:269-271 / recur-nn-init.c:80-143 */
RecurNN *rnn_new(uint input_size, uint hidden_size, uint output_size, u32 flags, u64 rng_seed,
                 const char *log_file, int bptt_depth, float learn_rate, float momentum,
                 float presynaptic_noise, rnn_activation activation) {
  return net_new(input_size, hidden_size, output_size, flags, rng_seed, log_file, bptt_depth,
                 learn_rate, momentum, presynaptic_noise, activation, NULL);
}

/* recur-nn-init.c:158-192.  Like the reference, nothing ever frees a layer: every
 * clone borrows the pointer (recur-nn-init.c:345-346). */
RecurExtraLayer *rnn_new_extra_layer(int input_size, int output_size, int overlap, u32 flags) {
  RecurExtraLayer *layer = ramd_zalloc(sizeof(RecurExtraLayer));
  layer->input_size = input_size;
  layer->output_size = output_size;
  layer->overlap = overlap;
  layer->learn_rate_scale = 1.0;
  layer->i_size = (int)round4(input_size + 1);
  layer->o_size = (int)round4(output_size);
  size_t m = (size_t)layer->i_size * layer->o_size;
  int aux = !!(flags & RNN_NET_FLAG_AUX_ARRAYS);
  size_t n = m * (3 + aux) + 2 * (size_t)(layer->i_size + layer->o_size);
  float *fm = ramd_zalloc(n * sizeof(float));
  layer->mem = fm;
  layer->momentums = fm; fm += m;
  layer->inputs = fm; fm += layer->i_size;
  layer->weights = fm; fm += m;
  layer->outputs = fm; fm += layer->o_size;
  layer->delta = fm; fm += m;
  layer->i_error = fm; fm += layer->i_size;
  layer->o_error = fm; fm += layer->o_size;
  if (aux) {
    layer->aux = fm;
  }
  return layer;
}

/* recur-nn-init.c:194-219 */
RecurNN *rnn_new_with_bottom_layer(int n_inputs, int r_input_size, int hidden_size,
                                   int output_size, u32 flags, u64 rng_seed,
                                   const char *log_file, int bptt_depth, float learn_rate,
                                   float momentum, float presynaptic_noise,
                                   rnn_activation activation, int convolutional_overlap) {
  if (r_input_size == 0) {
    flags &= ~RNN_NET_FLAG_BOTTOM_LAYER;
    return rnn_new(n_inputs, hidden_size, output_size, flags, rng_seed, log_file, bptt_depth,
                   learn_rate, momentum, presynaptic_noise, activation);
  }
  flags |= RNN_NET_FLAG_BOTTOM_LAYER;
  RecurNN *net = rnn_new(r_input_size, hidden_size, output_size, flags, rng_seed, log_file,
                         bptt_depth, learn_rate, momentum, presynaptic_noise, activation);
  net->bottom_layer = rnn_new_extra_layer(n_inputs, r_input_size, convolutional_overlap,
                                          net->flags);
  return net;
}

/* recur-nn-init.c:145-155 */
void rnn_delete_net(RecurNN *net) {
  RamdPriv *p = ramd_priv(net);
  RamdEngine *e = p->eng;
  if (e) {
    if (e->owner == net) {
      engine_delete(e);
    } else {
      /* the row stays reserved; forget the pointer */
      if (p->stream >= 0 && p->stream < e->n_streams) e->streams[p->stream] = e->owner;
      if (p->fwd >= 0 && p->fwd < e->n_fwd) e->fwd[p->fwd] = e->owner;
      /* trailing rows can be given back */
      while (e->n_streams > 0 && e->streams[e->n_streams - 1] == e->owner &&
             ramd_priv(e->owner)->stream != e->n_streams - 1) {
        e->n_streams--;
      }
      while (e->n_fwd > 0 && e->fwd[e->n_fwd - 1] == e->owner &&
             ramd_priv(e->owner)->fwd != e->n_fwd - 1) {
        e->n_fwd--;
      }
    }
  }
  if (net->bptt && (net->flags & RNN_NET_FLAG_OWN_BPTT)) {
    free(net->bptt->mem);
    free(net->bptt);
  }
  if (net->log) {
    fclose(net->log);
  }
  free(net->mem);
  free(net);
}

/* recur-nn-init.c:268-283 */
void rnn_set_log_file(RecurNN *net, const char *log_file, int append_dont_truncate) {
  if (net->log) {
    fclose(net->log);
  }
  if (log_file) {
    net->log = fopen(log_file, append_dont_truncate ? "a" : "w");
    if (!append_dont_truncate) {
      rnn_log_int(net, "generation", net->generation);
    }
  } else {
    net->log = NULL;
  }
}

/* recur-nn-init.c:296-350 */
RecurNN *rnn_clone(RecurNN *parent, u32 flags, u64 rng_seed, const char *log_file) {
  if (rng_seed == RECUR_RNG_SUBSEED) {
    if (ramd_priv(parent)->eng && !ramd_priv(parent)->host_valid) {
      ramd_need_host(parent, RNN_AMD_STREAM);
    }
    ramd_priv(parent)->dev_valid = 0;
    do {
      rng_seed = ramd_rand64(&parent->rng);
    } while (rng_seed == RECUR_RNG_RANDOM_SEED);
  }
  float learn_rate = 0, momentum = 0;
  int bptt_depth = 0;
  if (parent->bptt && (flags & RNN_NET_FLAG_OWN_BPTT)) {
    learn_rate = parent->bptt->learn_rate;
    bptt_depth = parent->bptt->depth;
    momentum = parent->bptt->momentum;
  }
  if (!(parent->bptt && (flags & RNN_NET_FLAG_OWN_BPTT))) {
    flags &= ~RNN_NET_FLAG_OWN_BPTT; /* no parent bptt to model it on */
  }
  RamdEngine *pe = ramd_engine_of(parent);
  RecurNN *net = net_new(parent->input_size, parent->hidden_size, parent->output_size, flags,
                         rng_seed, log_file, bptt_depth, learn_rate, momentum,
                         parent->presynaptic_noise, parent->activation,
                         (flags & RNN_NET_FLAG_OWN_WEIGHTS) ? NULL : pe);
  if (net->bptt) {
    net->bptt->momentum_weight = parent->bptt->momentum_weight;
    if (flags & RNN_NET_FLAG_NO_MOMENTUMS) {
      net->bptt->ih_momentum = parent->bptt->ih_momentum;
      net->bptt->ho_momentum = parent->bptt->ho_momentum;
    }
    if (flags & RNN_NET_FLAG_NO_DELTAS) {
      net->bptt->ih_delta = parent->bptt->ih_delta;
      net->bptt->ho_delta = parent->bptt->ho_delta;
    }
  }
  if (flags & RNN_NET_FLAG_OWN_WEIGHTS) {
    ramd_need_host(parent, RNN_AMD_WEIGHTS);
    memcpy(net->ih_weights, parent->ih_weights, (size_t)net->ih_size * sizeof(float));
    memcpy(net->ho_weights, parent->ho_weights, (size_t)net->ho_size * sizeof(float));
  } else {
    net->ih_weights = parent->ih_weights;
    net->ho_weights = parent->ho_weights;
  }
  net->bottom_layer = parent->bottom_layer;
  net->generation = parent->generation;
  net->presynaptic_noise = parent->presynaptic_noise;
  return net;
}

/* recur-nn-init.c:221-243 */
RecurNN **rnn_new_training_set(RecurNN *prototype, int n_nets) {
  if (n_nets < 1) {
    fprintf(stderr, "A training set of size %d is not possible\n", n_nets);
    return NULL;
  }
  RecurNN **nets = ramd_zalloc(n_nets * sizeof(RecurNN *));
  nets[0] = prototype;
  u32 flags = prototype->flags;
  flags &= ~RNN_NET_FLAG_OWN_WEIGHTS;
  flags |= RNN_NET_FLAG_NO_MOMENTUMS;
  flags |= RNN_NET_FLAG_NO_DELTAS;
  for (int i = 1; i < n_nets; i++) {
    nets[i] = rnn_clone(prototype, flags, RECUR_RNG_SUBSEED, NULL);
    nets[i]->bptt->ih_delta = prototype->bptt->ih_delta;
    nets[i]->bptt->ih_delta_tmp = prototype->bptt->ih_delta_tmp;
    nets[i]->bptt->ho_delta = prototype->bptt->ho_delta;
  }
  return nets;
}

/* One shard of a training set whose streams are spread over several processes (one per
 * GPU).  The reference seeds clone g from the g-th draw of the prototype's generator
 * (recur-nn-init.c:232-241, 300-305); every rank replays ALL the draws so that global
 * stream g gets the reference's generator wherever it lives, and keeps its own range. */
RecurNN **rnn_amd_new_training_set_shard(RecurNN *prototype, int n_local, int global_first,
                                         int global_count) {
  if (n_local < 1 || global_first < 0 || global_first + n_local > global_count) {
    fprintf(stderr, "A training set shard of %d streams at %d of %d is not possible\n", n_local,
            global_first, global_count);
    return NULL;
  }
  RecurNN **nets = ramd_zalloc(n_local * sizeof(RecurNN *));
  nets[0] = prototype;
  u32 flags = prototype->flags;
  flags &= ~RNN_NET_FLAG_OWN_WEIGHTS;
  flags |= RNN_NET_FLAG_NO_MOMENTUMS;
  flags |= RNN_NET_FLAG_NO_DELTAS;
  if (ramd_priv(prototype)->eng && !ramd_priv(prototype)->host_valid) {
    ramd_need_host(prototype, RNN_AMD_STREAM);
  }
  ramd_priv(prototype)->dev_valid = 0;
  u64 first_seed = 0;
  for (int g = 1; g < global_count; g++) {
    u64 seed;
    do {
      seed = ramd_rand64(&prototype->rng);
    } while (seed == RECUR_RNG_RANDOM_SEED);
    int j = g - global_first;
    if (j == 0) {
      first_seed = seed;
    } else if (j > 0 && j < n_local) {
      nets[j] = rnn_clone(prototype, flags, seed, NULL);
      nets[j]->bptt->ih_delta = prototype->bptt->ih_delta;
      nets[j]->bptt->ih_delta_tmp = prototype->bptt->ih_delta_tmp;
      nets[j]->bptt->ho_delta = prototype->bptt->ho_delta;
    }
  }
  if (global_first > 0) {
    /* this rank's first stream is global stream global_first: its generator, not the
     * prototype's (which belongs to global stream 0 on rank 0; see ramd_shared_rng) */
    ramd_init_rand64_maybe_randomly(&prototype->rng, first_seed);
  }
  if (ramd_priv(prototype)->eng && n_local != global_count) {
    ramd_priv(prototype)->eng->sharded_sticky = 1;
    ramd_priv(prototype)->eng->sharded = 1;
  }
  return nets;
}

/* recur-nn-init.c:245-257 */
void rnn_delete_training_set(RecurNN **nets, int n_nets, int leave_prototype) {
  /* clones first: the prototype owns the weights and the device image */
  for (int i = n_nets - 1; i >= 1; i--) {
    if (nets[i]) {
      rnn_delete_net(nets[i]);
    }
  }
  if (!leave_prototype && nets[0]) {
    rnn_delete_net(nets[0]);
  }
  free(nets);
}

/* The ring position every stream of [row0, row0 + nrows) shares, or -1.  The
 * host mirrors the indices exactly (they only ever change by rnn_bptt_advance). */
static void push_indices(RamdEngine *e, int row0, int nrows);
static void set_uniform_idx(RamdEngine *e, int row0, int nrows) {
  int u = -1;
  if (e->dev_ready && row0 < e->n_streams && nrows > 0) {
    push_indices(e, row0, nrows);
  }
  if (row0 < e->n_streams && nrows > 0) {
    u = e->streams[row0]->bptt->index;
    for (int j = row0 + 1; j < row0 + nrows && j < e->n_streams; j++) {
      if (e->streams[j]->bptt->index != u) {
        u = -1;
        break;
      }
    }
  }
  e->b.uniform_idx = u;
  mail_in_flush(); /* whatever the caller queued for the launches that follow */
}

/* ------------------------------------------------------------ scalars push -- */

/* learn_rate is host-authoritative (callers write bptt->learn_rate, e.g.
 * charmodel-predict.c:107, and clones keep their stale copy: SURVEY quirk 4). */
static void push_learn_rates(RamdEngine *e, int row0, int nrows) {
  int dirty = 0;
  for (int j = row0; j < row0 + nrows; j++) {
    float lr = e->streams[j]->bptt->learn_rate;
    if (lr != e->lr_pushed[j]) {
      e->lr_pushed[j] = lr;
      dirty = 1;
    }
  }
  if (dirty) {
    mail_in(e->b.lr + row0, e->lr_pushed + row0, nrows * sizeof(float)); /* leaves with the next flush */
  }
}

/* The ring indices are host-authoritative too: rnn_bptt_advance only steps the host's copy,
 * and the device's is brought up to date here, before the next launch that reads it (the set
 * calls that advance on the device record the new value in the mirror themselves). */
static void push_indices(RamdEngine *e, int row0, int nrows) {
  int dirty = 0;
  for (int j = row0; j < row0 + nrows && j < e->n_streams; j++) {
    int idx = e->streams[j]->bptt->index;
    if (idx != e->idx_pushed[j]) {
      e->idx_pushed[j] = idx;
      dirty = 1;
    }
  }
  if (dirty) {
    mail_in(e->b.idx + row0, e->idx_pushed + row0, nrows * sizeof(int));
  }
}

/* device mef / ih_scale -> host structs, for a range of streams */
static void pull_scalars(RamdEngine *e, int row0, int nrows) {
  float *tmp = malloc(2 * nrows * sizeof(float));
  d2h(tmp, e->b.mef + row0, nrows * sizeof(float));
  d2h(tmp + nrows, e->b.ih_scale + row0, nrows * sizeof(float));
  dsync();
  for (int j = 0; j < nrows; j++) {
    RecurNNBPTT *bp = e->streams[row0 + j]->bptt;
    bp->min_error_factor = tmp[j];
    bp->ih_scale = tmp[nrows + j];
  }
  free(tmp);
}

/* ----------------------------------------------------------------- logging -- */

/* What bptt_and_accumulate_error and rnn_bptt_calc_deltas write to net->log
 * (recur-nn.c:415-448, 766-771), rebuilt from the device's per-stream results. */
static void log_bptt(RamdEngine *e, RecurNN *net, float mef_before) {
  if (!net->log) {
    return;
  }
  RamdPriv *p = ramd_priv(net);
  int j = p->stream, D = e->sh.D, S = e->sh.Scap;
  float top_raw, top_scaled, bptt_err, scale, mef;
  int depth, n_exec;
  float *es = malloc(D * sizeof(float));
  d2h(&top_raw, e->b.top_raw + j, 4);
  d2h(&top_scaled, e->b.top_scaled + j, 4);
  d2h(&bptt_err, e->b.bptt_err + j, 4);
  d2h(&scale, e->b.ih_scale + j, 4);
  d2h(&mef, e->b.mef + j, 4);
  d2h(&depth, e->b.depth_log + j, 4);
  d2h(&n_exec, e->b.n_exec + j, 4);
  HIP_OK(hipMemcpy2DAsync(es, sizeof(float), e->b.esum + j, S * sizeof(float), sizeof(float), D,
                          hipMemcpyDeviceToHost, g_stream));
  d2h(net->hidden_layer, e->b.hidden + (size_t)j * e->sh.H, e->sh.H * sizeof(float));
  dsync();
  float cum_error = 0.0f;
  for (int k = 0; k < n_exec; k++) {
    cum_error += sqrtf(es[k]);
  }
  free(es);
  float min_gain = MIN_ERROR_GAIN * top_scaled;
  float thr = RAMD_MIN(mef_before / net->bptt->learn_rate, min_gain);
  rnn_log_int(net, "depth", depth);
  rnn_log_float(net, "scaled_error", scale * bptt_err);
  rnn_log_float(net, "ih_scale", scale);
  rnn_log_float(net, "min_error_threshold", thr);
  rnn_log_float(net, "min_error_factor", mef);
  rnn_log_float(net, "cum_error", cum_error);
  if (net->flags & RNN_NET_FLAG_LOG_HIDDEN_SUM) {
    float hidden_sum = 0, hidden_magnitude = 0;
    int hidden_zeros = 0;
    for (int i = 0; i < net->h_size; i++) {
      float h = net->hidden_layer[i];
      hidden_sum += h;
      hidden_magnitude += h * h;
      hidden_zeros += (h == 0.0f);
    }
    rnn_log_float(net, "hidden_sum", hidden_sum);
    rnn_log_float(net, "hidden_magnitude", sqrtf(hidden_magnitude));
    rnn_log_float(net, "hidden_zeros", hidden_zeros / (float)net->hidden_size);
  }
  if (net->flags & RNN_NET_FLAG_LOG_WEIGHT_SUM) {
    engine_need_host(e, RNN_AMD_WEIGHTS);
    float weight_sum = 0.0f;
    for (int i = 0; i < net->ih_size; i++) {
      weight_sum += fabsf(net->ih_weights[i]);
    }
    rnn_log_float(net, "weight_sum", weight_sum);
  }
  rnn_log_float(net, "error_gain", bptt_err / (top_scaled + 1e-6));
  rnn_log_float(net, "top_error_scaled", top_scaled);
  rnn_log_float(net, "top_error_raw", top_raw);
}

/* ---------------------------------------------------------- per-net hot path -- */

/* recur-nn.h:310 */
void rnn_bptt_advance(RecurNN *net) {
  host_advance(net); /* the device's copy follows before the next launch that reads it: push_indices */
}

/* recur-nn.h:302 / recur-nn.c:83-154 for one stream */
float *rnn_opinion(RecurNN *net, const float *inputs, float presynaptic_noise) {
  RamdEngine *e = ramd_engine_of(net);
  RamdPriv *p = ramd_priv(net);
  engine_ensure_device(e);
  top_done_clear(e);
  engine_need_dev(e, RNN_AMD_WEIGHTS);
  const RamdShape *s = &e->sh;
  RecurExtraLayer *bl = s->bI ? net->bottom_layer : NULL;
  /* the caller's real inputs are authoritative: keep them across a refresh */
  float *keep = malloc(sizeof(float) * s->input_size);
  memcpy(keep, (inputs && !bl) ? inputs : net->real_inputs, sizeof(float) * s->input_size);
  stream_need_host(e, net);
  stream_need_dev(e, net);
  memcpy(net->real_inputs, keep, sizeof(float) * s->input_size);
  free(keep);
  int r = state_row(e, p);
  float *d_slot;
  /* everything that goes in travels in one mailbox launch, everything that comes back in
   * another, and the call ends with one synchronisation */
  if (p->stream >= 0) {
    d_slot = e->b.arena + ((size_t)net->bptt->index * s->Scap + p->stream) * s->I;
  } else {
    d_slot = e->b.arena + ((size_t)s->D * s->Scap + p->fwd) * s->I;
  }
  if (presynaptic_noise != 0.0f) { /* the host generator is the one the caller may have used */
    mail_in((char *)e->b.rng + (size_t)r * sizeof(rand_ctx), &net->rng, sizeof(rand_ctx));
    rng_written_from_host(e);
  }
  if (bl) { /* recur-nn.c:88-103: the layer's one input buffer is shared by every clone */
    bl->inputs[0] = 1.0f;
    if (inputs) {
      memcpy(bl->inputs + 1, inputs, sizeof(float) * bl->input_size);
    }
    mail_in(e->b.binp + (size_t)r * s->bI, bl->inputs, sizeof(float) * s->bI);
  } else {
    mail_in(d_slot + s->hidden_size + 1, net->real_inputs, sizeof(float) * s->input_size);
  }
  set_uniform_idx(e, p->stream >= 0 ? p->stream : e->n_streams, p->stream >= 0 ? 1 : 0);
  mail_in_flush();
  if (bl) {
    ramd_launch_bottom_forward(g_stream, s, &e->b, r, 1, RAMD_IN_KEEP, NULL, 0, 0, 0, 1,
                               presynaptic_noise);
    mail_out(bl->outputs, e->b.bout + (size_t)r * s->bO, sizeof(float) * s->bO);
  }
  if (bl || presynaptic_noise != 0.0f || !ramd_launch_forward_small(g_stream, s, &e->b, r)) {
    ramd_launch_assemble(g_stream, s, &e->b, r, 1, RAMD_IN_KEEP, NULL, 0, 0, 0, 1, 0);
    ramd_launch_forward(g_stream, s, &e->b, r, 1, presynaptic_noise);
  }
  if (presynaptic_noise != 0.0f) {
    mail_out(&net->rng, (char *)e->b.rng + (size_t)r * sizeof(rand_ctx), sizeof(rand_ctx));
  }
  mail_out(net->input_layer, d_slot, sizeof(float) * s->I);
  mail_out(net->hidden_layer, e->b.hidden + (size_t)r * s->H, sizeof(float) * s->H);
  mail_out(net->output_layer, e->b.out + (size_t)r * s->O, sizeof(float) * s->O);
  mail_out_flush();
  return net->output_layer;
}

static const int *push_ranges(RamdEngine *e, RecurErrorRange *ranges) {
  if (!ranges) {
    return NULL;
  }
  int n = 0;
  while (ranges[n].start >= 0) {
    n++;
  }
  if (n > 64) {
    fprintf(stderr, "librecur_amd: more than 64 error ranges\n");
    abort();
  }
  mail_in(e->d_ranges, ranges, (n + 1) * sizeof(RecurErrorRange));
  return e->d_ranges;
}

static void calc_deltas_one(RecurNN *net, int accumulate, RecurErrorRange *ranges, unsigned fused) {
  RamdEngine *e = ramd_engine_of(net);
  RamdPriv *p = ramd_priv(net);
  if (p->stream < 0) {
    fprintf(stderr, "librecur_amd: rnn_bptt_calc_deltas on a net without bptt\n");
    abort();
  }
  engine_ensure_device(e);
  top_done_clear(e); /* (a per-net call between a set's one-call loss and its delta call: that set's top backprop is redone) */
  const RamdShape *s = &e->sh;
  int j = p->stream;
  if (fused) {
    /* rnn_bptt_calculate never writes ho_delta: after rnn_bptt_clear_deltas the reference has
     * zeros there (recur-nn.c:681-693), so the pending clear is carried out, not dropped */
    deltas_materialize(e);
  }
  if (e->deltas_zero_pending) { /* the sum into zeros is the sum */
    accumulate = 0;
    e->deltas_zero_pending = 0;
  }
  engine_need_dev(e, RNN_AMD_WEIGHTS | (accumulate ? RNN_AMD_DELTAS : 0));
  e->kept_live = 0; /* (a set call's kept sums: added up just now if this call accumulates, otherwise overwritten) */
  if (fused) {
    /* (the fused path rewrites ih_delta only: the rest of the delta arrays has to be the device's
     * own before they are declared written -- after a regrow the device copy is blank) */
    engine_need_dev(e, RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  }
  float *keep = malloc(sizeof(float) * s->O);
  memcpy(keep, net->bptt->o_error, sizeof(float) * s->O);
  stream_need_host(e, net);
  stream_need_dev(e, net);
  memcpy(net->bptt->o_error, keep, sizeof(float) * s->O);
  free(keep);
  RecurNNBPTT *bp = net->bptt;
  float mef_before = bp->min_error_factor;
  mail_in(e->b.o_error + (size_t)j * s->O, bp->o_error, sizeof(float) * s->O);
  mail_in(e->b.mef + j, &bp->min_error_factor, sizeof(float));
  if (ranges) {
    /* the sparse top path reads last time's h_error */
    err_flush(e);
    mail_in(e->b.err_a + (size_t)j * s->I, bp->h_error, sizeof(float) * s->I);
  }
  push_learn_rates(e, j, 1);
  const int *d_ranges = push_ranges(e, ranges);
  if (e->err_pending) {
    err_flush(e);
  }
  set_uniform_idx(e, j, 1);
  mail_in_flush();
  ramd_launch_calc_deltas(g_stream, s, &e->b, j, 1, accumulate, d_ranges, 0, NULL,
                          net->flags | (fused ? 0x80000000u : 0) | (fused == 2 ? 0x20000000u : 0), NULL);
  if (s->bI && !fused) { /* the fused path passes no bottom error (recur-nn.c:972, 986) */
    if (accumulate) {
      engine_need_dev(e, RNN_AMD_DELTAS);
    }
    ramd_launch_bottom_deltas(g_stream, s, &e->b, j, 1, accumulate, NULL);
    mail_out(net->bottom_layer->o_error, e->b.bcarry + (size_t)e->b.bcarry_cur * s->bO,
             sizeof(float) * s->bO);
  }
  engine_dev_wrote(e, RNN_AMD_DELTAS);
  if (ramd_calc_wrote_images()) { /* the one-workgroup BPTT of a small net leaves them itself */
    e->err_pending = 0;
  } else {
    e->err_pending = 1;
    e->err_row0 = j;
    e->err_nrows = 1;
    err_flush(e);
  }
  mail_out(bp->h_error, e->b.err_a + (size_t)j * s->I, sizeof(float) * s->I);
  mail_out(bp->i_error, e->b.err_b + (size_t)j * s->I, sizeof(float) * s->I);
  mail_out(&bp->min_error_factor, e->b.mef + j, sizeof(float));
  mail_out(&bp->ih_scale, e->b.ih_scale + j, sizeof(float));
  mail_out_flush();
  net->generation++;
  log_bptt(e, net, mef_before);
}

/* recur-nn.h:316 / recur-nn.c:707-772 */
void rnn_bptt_calc_deltas(RecurNN *net, int accumulate_delta, RecurErrorRange *top_error_ranges) {
  calc_deltas_one(net, accumulate_delta, top_error_ranges, 0);
  rnn_log_int(net, "generation", net->generation);
}

/* recur-nn.h:309 / recur-nn.c:681-693 */
void rnn_bptt_clear_deltas(RecurNN *net) {
  RamdEngine *e = ramd_engine_of(net);
  engine_ensure_device(e);
  e->kept_live = 0; /* sums nobody asked for */
  if (e->sh.bI) { /* the bottom layer's error accumulator is cleared with them: at once */
    e->deltas_zero_pending = 0;
    ramd_launch_clear_deltas(g_stream, &e->sh, &e->b);
  } else {
    e->deltas_zero_pending = 1; /* see deltas_materialize */
  }
  engine_dev_wrote(e, RNN_AMD_DELTAS);
}

/* recur-nn.h:313 / recur-nn.c:595-599 */
float rnn_calculate_momentum_soft_start(float generation, float max_momentum, float x) {
  return RAMD_MIN(max_momentum, 1.0f - x / (1.0f + generation + 2.0f * x));
}

static void check_method_arrays(RamdEngine *e, int method) {
  if ((method == RNN_ADADELTA || method == RNN_RPROP) && !e->has_aux) {
    fprintf(stderr, "librecur_amd: learning method %d needs RNN_NET_FLAG_AUX_ARRAYS\n", method);
    abort();
  }
}

/* rnn_apply_learning's arrays in one launch: top layer, recurrent layer and, when there
 * is one, the bottom layer with its own rate scale (recur-nn.c:606-676; the arrays are
 * disjoint, so the reference's order between them does not matter) */
static void apply_all(RamdEngine *e, int method, float lr, float lr_top, float momentum,
                      float mw, const RamdPendingDelta *pend) {
  RamdBuffers *b = &e->b;
  check_method_arrays(e, method);
  float *w[3] = {b->ho_w, b->ih_w, b->bw};
  const float *d[3] = {b->ho_delta, b->ih_delta, b->bdelta};
  float *m[3] = {b->ho_m, b->ih_m, b->bm};
  float *aux[3] = {b->ho_aux, b->ih_aux, b->baux};
  size_t n[3] = {e->ho_size, e->ih_size, (size_t)e->sh.bI * e->sh.bO};
  float rate[3] = {lr_top, lr,
                   e->sh.bI ? lr * e->owner->bottom_layer->learn_rate_scale : 0.0f};
  ramd_launch_apply_multi(g_stream, method, e->sh.bI ? 3 : 2, w, d, m, aux, n, rate, momentum, mw,
                          NULL, pend);
}

/* recur-nn.h:312 / recur-nn.c:601-678 */
static void apply_learning(RecurNN *net, int learning_method, float momentum,
                           const RamdPendingDelta *pend);

void rnn_apply_learning(RecurNN *net, int learning_method, float momentum) {
  apply_learning(net, learning_method, momentum, NULL);
}

static void apply_learning(RecurNN *net, int learning_method, float momentum,
                           const RamdPendingDelta *pend) {
  RamdEngine *e = ramd_engine_of(net);
  engine_ensure_device(e);
  if (!pend && e->kept_live && e->dev_ready) {
    /* the last set call's sums, still planes: this launch adds them up (and stores them) on its way */
    pend = &e->kept;
    e->kept_live = 0;
    engine_need_dev(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS);
  } else {
    engine_need_dev(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  }
  RecurNNBPTT *bptt = net->bptt;
  float mw;
  if (learning_method == RNN_MOMENTUM_SIMPLIFIED_NESTEROV) {
    mw = momentum / (1.0 + momentum);
  } else if (learning_method == RNN_MOMENTUM_CLASSICAL) {
    mw = 1.0f;
  } else {
    mw = bptt->momentum_weight;
  }
  int kernel_method = learning_method;
  if (learning_method == RNN_MOMENTUM_SIMPLIFIED_NESTEROV ||
      learning_method == RNN_MOMENTUM_CLASSICAL || learning_method >= RNN_LAST_LEARNING_METHOD ||
      learning_method < 0) {
    kernel_method = RNN_MOMENTUM_WEIGHTED;
  }
  apply_all(e, kernel_method, bptt->learn_rate, bptt->learn_rate * bptt->ho_scale, momentum, mw,
            pend);
  engine_dev_wrote(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS);
}

/* recur-nn.h:319 / recur-nn.c:782-855 */
void rnn_condition_net(RecurNN *net) {
  u32 mask = net->flags >> RNN_COND_USE_OFFSET;
  u32 m = net->generation % RNN_CONDITIONING_INTERVAL;
  if (((1u << m) & mask) == 0) {
    return;
  }
  RamdEngine *e = ramd_engine_of(net);
  engine_ensure_device(e);
  RamdBuffers *b = &e->b;
  switch (m) {
  case RNN_COND_BIT_SCALE:
    engine_need_dev(e, RNN_AMD_WEIGHTS);
    ramd_launch_scale(g_stream, b->ih_w, e->ih_size, WEIGHT_SCALE);
    ramd_launch_scale(g_stream, b->ho_w, e->ho_size, WEIGHT_SCALE);
    engine_dev_wrote(e, RNN_AMD_WEIGHTS);
    break;
  case RNN_COND_BIT_ZERO:
    engine_need_dev(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS);
    ramd_launch_zero_small(g_stream, b->ih_w, e->ih_size);
    ramd_launch_zero_small(g_stream, b->ho_w, e->ho_size);
    if (net->bptt) {
      ramd_launch_zero_small(g_stream, b->ih_m, e->ih_size);
      ramd_launch_zero_small(g_stream, b->ho_m, e->ho_size);
    }
    engine_dev_wrote(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS);
    break;
  case RNN_COND_BIT_RAND: {
    stream_need_host(e, net); /* the generator may have advanced on the device (noise) */
    rand_ctx tmp, *rng = ramd_shared_rng(net, &tmp); /* every replica takes the same damage */
    int t = ramd_rand_small_int(rng, net->ih_size + net->ho_size);
    float damage = (ramd_cheap_gaussian_noise(rng) * RANDOM_DAMAGE_FACTOR * net->h_size *
                    net->bptt->learn_rate);
    ramd_priv(net)->dev_valid = 0; /* and now it advanced on the host */
    engine_need_dev(e, RNN_AMD_WEIGHTS);
    if (t >= net->ih_size) {
      t -= net->ih_size;
      if (t % net->o_size < net->output_size) {
        ramd_launch_add_at(g_stream, b->ho_w, t, damage);
      }
    } else {
      int col = t % net->h_size;
      if (col >= 1 && col < net->hidden_size + 1) {
        ramd_launch_add_at(g_stream, b->ih_w, t, damage);
      }
    }
    engine_dev_wrote(e, RNN_AMD_WEIGHTS);
  } break;
  case RNN_COND_BIT_TALL_POPPY:
    engine_need_dev(e, RNN_AMD_WEIGHTS);
    ramd_launch_tall_poppy(g_stream, b->ih_w, e->ih_size, RNN_TALL_POPPY_THRESHOLD,
                           RNN_TALL_POPPY_SCALE, e->d_scratch);
    engine_dev_wrote(e, RNN_AMD_WEIGHTS);
    break;
  case RNN_COND_BIT_LAWN_MOWER:
    engine_need_dev(e, RNN_AMD_WEIGHTS);
    ramd_launch_clamp(g_stream, b->ih_w, e->ih_size, -RNN_LAWN_MOWER_THRESHOLD,
                      RNN_LAWN_MOWER_THRESHOLD);
    engine_dev_wrote(e, RNN_AMD_WEIGHTS);
    break;
  }
}

/* recur-nn.h:311 / recur-nn.c:919-1019: the single-net path that updates the
 * weights at once.  Top layer: backprop with the old weights, then the
 * rank-1 update with momentum (no ho_scale, recur-nn.c:927); recurrent layer:
 * BPTT deltas (the per-stream ih_scale is already folded into them) applied
 * with the weighted-momentum rule, every step or every batch_size steps. */
void rnn_bptt_calculate(RecurNN *net, uint batch_size) {
  RamdEngine *e = ramd_engine_of(net);
  RamdPriv *p = ramd_priv(net);
  RecurNNBPTT *bptt = net->bptt;
  int batched = batch_size > 1;
  /* without batching the reference leaves the unscaled sum in ih_delta and multiplies the rate by
   * ih_scale (recur-nn.c:966-975); batched, ih_scale goes into the sum (977-994) */
  calc_deltas_one(net, batched, NULL, batched ? 1 : 2); /* also does generation++ */
  /* generation was already incremented; the reference tests the value before
   * its increment (recur-nn.c:991, 1010) */
  const int due = !batched || ((net->generation - 1) % batch_size) == 0;
  /* the top layer's immediate update and, when due, the recurrent layer's: one launch */
  ramd_launch_fused_updates(g_stream, &e->sh, &e->b, p->stream, bptt->learn_rate, bptt->momentum,
                            bptt->momentum_weight, due, batched ? NULL : e->b.ih_scale + p->stream);
  if (due && batched) { /* ih_delta only (recur-nn.c:991): ho_delta is not this path's */
    HIP_OK(hipMemsetAsync(e->b.ih_delta, 0, e->ih_size * sizeof(float), g_stream));
  }
  engine_dev_wrote(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  if (net->log) {
    rnn_log_int(net, "generation", net->generation);
  }
  rnn_condition_net(net);
}

/* recur-nn.h:322 / recur-nn.c:8-16 */
void rnn_forget_history(RecurNN *net, int bptt_too) {
  RamdEngine *e = ramd_engine_of(net);
  stream_need_host(e, net);
  memset(net->hidden_layer, 0, net->h_size * sizeof(float));
  memset(net->input_layer, 0, (net->hidden_size + 1) * sizeof(float));
  if (bptt_too && net->bptt) {
    memset(net->bptt->history, 0, (size_t)net->bptt->depth * net->i_size * sizeof(float));
  }
  ramd_priv(net)->dev_valid = 0;
}

/* recur-nn.h:320 / recur-nn.c:887-904 */
void rnn_log_net(RecurNN *net) {
  if (net->log == NULL) {
    return;
  }
  if (net->bptt) {
    ramd_need_host(net, RNN_AMD_STREAM);
    float top_error = 0, hidden_error = 0;
    for (int i = 0; i < net->o_size; i++) {
      top_error += fabsf(net->bptt->o_error[i]);
    }
    for (int i = 0; i < net->h_size; i++) {
      hidden_error += fabsf(net->bptt->h_error[i]);
    }
    rnn_log_float(net, "output_error", top_error);
    rnn_log_float(net, "hidden_error", hidden_error);
  }
}

/* ------------------------------------------------ text through one net -- */

/* Runs one net over an encoded text without leaving the device: for every i < len - 1
 * a one_hot_opinion of text[i] (charmodel-helpers.h:16-33) and, from i = skip on, the
 * log2 probability the softmax gives text[i + 1].  With alphabet_len == 0 the softmax is
 * over the whole output row and sums[0] gets the sum of the logs (get_cross_entropy's
 * loop, charmodel-predict.c:62-76; with skip >= len - 1 it is rnn_char_prime's loop,
 * 407-416); otherwise the row is output_size / alphabet_len heads and sums[c] gets head
 * c's sum (rnn_char_multi_cross_entropy's loop, charmodel-multi-predict.c:388-403).
 * The net's state rows stay on the device. */
static void run_text(RecurNN *net, const u8 *text, int len, int skip, int alphabet_len,
                     double *sums, int n_sums) {
  RamdEngine *e = ramd_engine_of(net);
  RamdPriv *p = ramd_priv(net);
  engine_ensure_device(e);
  engine_need_dev(e, RNN_AMD_WEIGHTS);
  stream_need_dev(e, net);
  for (int c = 0; c < n_sums; c++) {
    sums[c] = 0.0;
  }
  if (len < 2) {
    return;
  }
  const RamdShape *s = &e->sh;
  int r = state_row(e, p);
  unsigned char *d_text = dev_alloc(len);
  double *d_acc = alphabet_len ? dev_alloc((size_t)n_sums * sizeof(double)) : NULL;
  h2d(d_text, text, len);
  HIP_OK(hipMemsetAsync(e->b.xent + r, 0, sizeof(double), g_stream));
  if (p->stream >= 0) {
    h2d(e->b.idx + p->stream, &net->bptt->index, sizeof(int));
  }
  dsync();
  unsigned char *old_text = e->b.text;
  int old_len = e->b.text_len;
  e->b.text = d_text;
  e->b.text_len = len;
  set_uniform_idx(e, p->stream >= 0 ? p->stream : e->n_streams, p->stream >= 0 ? 1 : 0);
  for (int i = 0; i < len - 1; i++) {
    if (s->bI) {
      ramd_launch_bottom_forward(g_stream, s, &e->b, r, 1, RAMD_IN_TEXT, NULL, 0, i, 0, 1, 0.0f);
      ramd_launch_assemble(g_stream, s, &e->b, r, 1, RAMD_IN_KEEP, NULL, 0, 0, 0, 1, 0);
    } else {
      ramd_launch_assemble(g_stream, s, &e->b, r, 1, RAMD_IN_TEXT, NULL, 0, i, 0, 1, 0);
    }
    ramd_launch_forward(g_stream, s, &e->b, r, 1, 0.0f);
    if (alphabet_len) {
      if (i >= skip) {
        ramd_launch_multi_xent_accumulate(g_stream, s, &e->b, r, alphabet_len, n_sums, d_acc, 1);
      }
    } else {
      ramd_launch_xent_accumulate(g_stream, s, &e->b, r, i >= skip);
    }
  }
  if (alphabet_len) {
    d2h(sums, d_acc, (size_t)n_sums * sizeof(double));
  } else {
    d2h(sums, e->b.xent + r, sizeof(double));
  }
  dsync();
  e->b.text = old_text;
  e->b.text_len = old_len;
  dev_free(d_text);
  dev_free(d_acc);
  p->dev_valid = 1;
  p->host_valid = 0;
}

double rnn_amd_run_text(RecurNN *net, const u8 *text, int len, int skip) {
  double sum = 0.0;
  run_text(net, text, len, skip, 0, &sum, 1);
  return sum;
}

void rnn_amd_run_text_heads(RecurNN *net, const u8 *text, int len, int skip, int alphabet_len,
                            double *sums) {
  int n_classes = alphabet_len > 0 ? net->output_size / alphabet_len : 0;
  if (n_classes < 1) {
    fprintf(stderr, "librecur_amd: rnn_amd_run_text_heads: %d outputs as heads of %d\n",
            net->output_size, alphabet_len);
    abort();
  }
  run_text(net, text, len, skip, alphabet_len, sums, n_classes);
}

/* ================================================================ batched == */

RnnAmdSet *rnn_amd_set_open(RecurNN **nets, int n_nets) {
  if (!nets || n_nets < 1) {
    return NULL;
  }
  RamdEngine *e = ramd_engine_of(nets[0]);
  /* either training streams in rnn_new_training_set order, or forward-only clones
   * (rnn_clone without bptt, e.g. gstrnnca.c's constructors) in creation order */
  int fwd_only = ramd_priv(nets[0])->stream < 0;
  int row0 = fwd_only ? ramd_priv(nets[0])->fwd : ramd_priv(nets[0])->stream;
  for (int j = 0; j < n_nets; j++) {
    RamdPriv *p = ramd_priv(nets[j]);
    int id = fwd_only ? p->fwd : p->stream;
    if (ramd_engine_of(nets[j]) != e || id != row0 + j || (p->stream < 0) != fwd_only) {
      fprintf(stderr, "librecur_amd: rnn_amd_set_open: nets must be one training set in "
                      "rnn_new_training_set order (or forward-only clones in creation order)\n");
      return NULL;
    }
  }
  engine_ensure_device(e);
  RnnAmdSet *set = ramd_zalloc(sizeof(RnnAmdSet));
  set->eng = e;
  /* the set keeps its own copy of the pointer array: the caller's may be a temporary */
  set->nets = ramd_zalloc(n_nets * sizeof(RecurNN *));
  memcpy(set->nets, nets, n_nets * sizeof(RecurNN *));
  set->n = n_nets;
  set->row0 = row0;
  set->fwd_only = fwd_only;
  set->global_first = 0;
  set->global_count = n_nets;
  set->counts_shard = 0;
  /* A set of ALL the engine's training streams opened inside a group is this rank's shard of the distributed
   * training set: rank r holds global streams [r n, (r + 1) n).  A set of some of them -- the host layers' one-net
   * passes, a validation or side net that one rank alone touches -- stays local: its host draws must not become
   * collectives that the other ranks never enter. */
  if (ramd_dist_active() && !fwd_only && n_nets == e->n_streams) {
    set->global_first = rnn_amd_dist_rank() * n_nets;
    set->global_count = rnn_amd_dist_world() * n_nets;
    set->counts_shard = 1;
    e->sharded_sets++;
    e->sharded = 1;
  }
  return set;
}

/* the engine stays a shard while ANY open set shards it (two sets may be open on one engine and close in any order:
 * ADVICE.md round 4) or it was sharded explicitly (rnn_amd_set_shard, the shard constructor: the nets remain a shard
 * whatever set drives them) */
static void set_unshard(RnnAmdSet *set) {
  RamdEngine *e = set->eng;
  if (set->counts_shard && e->sharded_sets > 0) {
    e->sharded_sets--;
  }
  set->counts_shard = 0;
  e->sharded = e->sharded_sticky || e->sharded_sets > 0;
}

void rnn_amd_set_close(RnnAmdSet *set) {
  if (!set) {
    return;
  }
  ramd_need_host(set->nets[0], RNN_AMD_EVERYTHING);
  set_unshard(set);
  free(set->nets);
  free(set);
}

/* the same without fetching anything: the streams' state stays current on the device only */
void rnn_amd_set_drop(RnnAmdSet *set) {
  if (!set) {
    return;
  }
  set_unshard(set);
  free(set->nets);
  free(set);
}

int rnn_amd_set_size(const RnnAmdSet *set) { return set->n; }

void rnn_amd_set_shard(RnnAmdSet *set, int global_first, int global_count) {
  if (global_first < 0 || global_first + set->n > global_count) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_shard(%d, %d) for a set of %d streams\n", global_first,
            global_count, set->n);
    abort();
  }
  set->global_first = global_first;
  set->global_count = global_count;
  if (global_count != set->n) {
    set->eng->sharded_sticky = 1;
  } else if (global_first == 0) {
    set->eng->sharded_sticky = 0; /* the whole set again: host draws come from the nets' own generators as before */
  }
  set->eng->sharded = set->eng->sharded_sticky || set->eng->sharded_sets > 0;
}

void rnn_amd_set_dist_all_reduce_deltas(RnnAmdSet *set) {
  RamdEngine *e = set->eng;
  if (!ramd_dist_active()) {
    return;
  }
  engine_need_dev(e, RNN_AMD_DELTAS);
  /* ih_delta||ho_delta are one allocation (library-owned or the caller's external buffer) */
  const int tev = ramd_timing_begin(g_stream, RAMD_T_XCHG);
  rnn_amd_dist_all_reduce(e->b.ih_delta, e->ih_size + e->ho_size);
  ramd_timing_end(g_stream, tev);
  engine_dev_wrote(e, RNN_AMD_DELTAS);
}

/* first state row of the set (forward-only rows sit after the Scap training rows) */
static int set_state_row0(const RnnAmdSet *set) {
  return set->fwd_only ? set->eng->sh.Scap + set->row0 : set->row0;
}

static void set_need_training(const RnnAmdSet *set, const char *what) {
  if (set->fwd_only) {
    fprintf(stderr, "librecur_amd: %s needs a training set (nets with bptt)\n", what);
    abort();
  }
}

static void set_streams_to_dev(RnnAmdSet *set) {
  RamdEngine *e = set->eng;
  int any = 0;
  for (int j = 0; j < set->n; j++) {
    RamdPriv *p = ramd_priv(set->nets[j]);
    if (!p->dev_valid) {
      stream_copy(e, set->nets[j], 1);
      p->dev_valid = 1;
      any = 1;
    }
  }
  if (any) {
    dsync();
    top_done_clear(e);
  }
  if (!set->fwd_only) {
    push_indices(e, set->row0, set->n);
    mail_in_flush();
  }
}

static void set_streams_dev_wrote(RnnAmdSet *set) {
  for (int j = 0; j < set->n; j++) {
    RamdPriv *p = ramd_priv(set->nets[j]);
    p->dev_valid = 1;
    p->host_valid = 0;
  }
}

void rnn_amd_set_advance(RnnAmdSet *set) {
  RamdEngine *e = set->eng;
  set_need_training(set, "rnn_amd_set_advance");
  (void)e;
  for (int j = 0; j < set->n; j++) {
    host_advance(set->nets[j]);
  }
  /* the device's copy of the indices follows with the next call's uploads (push_indices): no
   * launch of its own for rnn_bptt_advance's three integer operations per stream */
}

/* hidden_only: stop after the hidden layer's GEMM and leave its K slabs for
 * ramd_launch_text_top; returns their number (0 otherwise) */
static void noise_speculate_from(RnnAmdSet *set, int loss_classes);
static int set_forward(RnnAmdSet *set, int mode, const float *d_dense, int ld, int text_i,
                       float *outputs, int advance, int hidden_only) {
  RamdEngine *e = set->eng;
  top_done_clear(e);
  engine_need_dev(e, RNN_AMD_WEIGHTS);
  set_streams_to_dev(set);
  if (advance) {
    set_need_training(set, "advancing");
    for (int j = 0; j < set->n; j++) {
      host_advance(set->nets[j]); /* the device steps its copy inside the assemble kernel */
      e->idx_pushed[set->row0 + j] = set->nets[j]->bptt->index;
    }
  }
  const int r0 = set_state_row0(set);
  if (set->fwd_only) {
    e->b.uniform_idx = -1;
  } else {
    set_uniform_idx(e, set->row0, set->n);
  }
  if (mode == RAMD_IN_DENSE) {
    e->b.dense_inputs = 1;
  } else if (mode == RAMD_IN_ONE_HOT || mode == RAMD_IN_TEXT) {
    e->b.dense_inputs = 0;
  }
  float noise = set->nets[0]->presynaptic_noise;
  if (noise != 0.0f) {
    /* values generated ahead for exactly this pass, and nothing has touched the generators since? */
    const int k = e->sp_head;
    e->b.noise_spec_use = e->sp[k].pending && e->sp[k].version == e->rng_version && e->sp[k].row0 == r0 &&
                          e->sp[k].n == set->n && e->sp[k].dev == noise && !e->sh.bI;
    e->sp_adopted = -1;
    if (e->b.noise_spec_use) {
      HIP_OK(hipStreamWaitEvent(g_stream, (hipEvent_t)e->sp[k].done, 0));
      e->b.noise_spec = e->sp[k].noise;
      e->b.rng_spec = e->sp[k].states;
      e->sp[k].pending = 0;
      e->sp_head = k ^ 1; /* (the other buffer: the pass after this one, if it has been started) */
      e->sp_adopted = k;
    } else {
      e->sp[0].pending = e->sp[1].pending = 0; /* nothing that was started fits: the second one hangs on the first */
    }
    e->rng_version++; /* this pass moves the generators, either way */
  } else {
    e->sp_adopted = -1;
  }
  for (int j = 1; j < set->n; j++) {
    if (set->nets[j]->presynaptic_noise != noise) {
      fprintf(stderr, "librecur_amd: the nets of a set must share presynaptic_noise\n");
      abort();
    }
  }
  if (e->sh.bI) {
    /* the caller's inputs feed the bottom layer, whose rectified outputs become the real
     * inputs of the current slot: the ring has to step before it runs */
    if (advance) {
      ramd_launch_advance(g_stream, &e->sh, &e->b, set->row0, set->n);
    }
    ramd_launch_bottom_forward(g_stream, &e->sh, &e->b, r0, set->n, mode, d_dense, ld, text_i,
                               set->global_first, set->global_count, noise);
    ramd_launch_assemble(g_stream, &e->sh, &e->b, r0, set->n, RAMD_IN_KEEP, NULL, 0, 0,
                         set->global_first, set->global_count, 0);
  } else {
    if (hidden_only && (advance || mode == RAMD_IN_DENSE) && noise == 0.0f && !set->fwd_only) {
      /* the text step: building the input rows and the hidden layer's GEMM in one launch
       * (the ring index the kernel stores is the host's, which has just stepped); dense inputs on their way to
       * ramd_launch_dense_top likewise */
      int fused = ramd_launch_forward_fused(g_stream, &e->sh, &e->b, r0, set->n, mode, text_i,
                                            set->global_first, set->global_count, 1, d_dense, ld);
      if (fused) {
        set_streams_dev_wrote(set);
        return fused;
      }
    } else if (getenv("RECUR_AMD_TRACE_FWD") && !(!hidden_only && (advance || mode == RAMD_IN_DENSE) && !set->fwd_only &&
                                                 (noise == 0.0f || e->b.noise_spec_use))) {
      fprintf(stderr, "librecur_amd: forward not fused: hidden_only %d advance %d mode %d fwd_only %d noise %g\n", hidden_only,
              advance, mode, set->fwd_only, noise);
    } else if (!hidden_only && (advance || mode == RAMD_IN_DENSE) && !set->fwd_only && (noise == 0.0f || e->b.noise_spec_use)) {
      /* the same launch for the one-hot and text passes that go on to a generic output layer (the multi-head step),
       * and for dense inputs (gstclassify's features, rnnca's neighbourhoods; their callers advance on their own, and
       * the index the kernel stores is the one that is there); presynaptic noise only as the values generated
       * ahead, which the finishing kernel adds */
      int fused = ramd_launch_forward_fused(g_stream, &e->sh, &e->b, r0, set->n, mode, text_i,
                                            set->global_first, set->global_count, 0, d_dense, ld);
      if (fused) {
        ramd_launch_forward_finish(g_stream, &e->sh, &e->b, r0, set->n, fused, 1);
        if (set->early_spec_classes > 0 && e->sp_adopted >= 0) {
          /* the multi-head step: the next pass's noise from here on, beside the output layer and the loss */
          noise_speculate_from(set, set->early_spec_classes);
          set->early_spec_classes = -1; /* done */
        }
        ramd_launch_forward_finish(g_stream, &e->sh, &e->b, r0, set->n, fused, 2);
        e->b.noise_spec_use = 0;
        set_streams_dev_wrote(set);
        if (outputs) {
          d2h(outputs, e->b.out + (size_t)r0 * e->sh.O, (size_t)set->n * e->sh.O * sizeof(float));
          dsync();
        }
        return 0;
      }
    }
    ramd_launch_assemble(g_stream, &e->sh, &e->b, r0, set->n, mode, d_dense, ld, text_i,
                         set->global_first, set->global_count, advance);
  }
  int fwd_ks = 0;
  if (hidden_only) {
    fwd_ks = ramd_launch_forward_hidden(g_stream, &e->sh, &e->b, r0, set->n, noise, 1);
  } else {
    ramd_launch_forward(g_stream, &e->sh, &e->b, r0, set->n, noise);
  }
  e->b.noise_spec_use = 0;
  set_streams_dev_wrote(set);
  if (outputs) {
    d2h(outputs, e->b.out + (size_t)r0 * e->sh.O, (size_t)set->n * e->sh.O * sizeof(float));
    dsync();
  }
  return fwd_ks;
}

void rnn_amd_set_opinion(RnnAmdSet *set, const float *inputs, int ld_inputs, float *outputs) {
  RamdEngine *e = set->eng;
  if (inputs) {
    int w = e->sh.bI ? e->sh.b_in : e->sh.input_size;
    /* (queued: leaves with the ring indices in set_forward's flush) */
    upload_rows_q(e->d_dense, inputs, ld_inputs * sizeof(float), w * sizeof(float), set->n, set->fwd_only);
    set_forward(set, RAMD_IN_DENSE, e->d_dense, w, 0, outputs, 0, 0);
    if (e->sh.bI) {
      /* the layer's one input buffer, shared by every clone, as the per-net loop would leave it:
       * the last stream's inputs (recur-nn.c:88-94) */
      RecurExtraLayer *bl = set->nets[0]->bottom_layer;
      bl->inputs[0] = 1.0f;
      memcpy(bl->inputs + 1, inputs + (size_t)(set->n - 1) * ld_inputs, sizeof(float) * bl->input_size);
    }
  } else {
    set_forward(set, RAMD_IN_KEEP, NULL, 0, 0, outputs, 0, 0);
  }
}

void rnn_amd_set_one_hot_opinion(RnnAmdSet *set, const int *hot, float *outputs) {
  RamdEngine *e = set->eng;
  upload(e->b.hot + set_state_row0(set), hot, set->n * sizeof(int));
  set_forward(set, RAMD_IN_ONE_HOT, NULL, 0, 0, outputs, 0, 0);
}

void rnn_amd_set_put_o_error(RnnAmdSet *set, const float *o_error, int ld) {
  RamdEngine *e = set->eng;
  top_done_clear(e);
  set_need_training(set, "rnn_amd_set_put_o_error");
  set_streams_to_dev(set);
  upload_rows(e->b.o_error + (size_t)set->row0 * e->sh.O, o_error, ld * sizeof(float), e->sh.O * sizeof(float),
              set->n);
  set_streams_dev_wrote(set);
}

void rnn_amd_set_softmax_error(RnnAmdSet *set, const int *target) {
  RamdEngine *e = set->eng;
  top_done_clear(e);
  set_need_training(set, "rnn_amd_set_softmax_error");
  set_streams_to_dev(set);
  if (target) {
    upload(e->b.target + set->row0, target, set->n * sizeof(int));
  }
  ramd_launch_softmax_error(g_stream, &e->sh, &e->b, set->row0, set->n);
  set_streams_dev_wrote(set);
}

static void set_calc_deltas(RnnAmdSet *set, int accumulate, RecurErrorRange *ranges,
                            const u8 *active, unsigned extra_flags, const int *dev_ranges,
                            int range_stride, RamdPendingDelta *defer);

void rnn_amd_set_calc_deltas(RnnAmdSet *set, int accumulate, RecurErrorRange *ranges,
                             const u8 *active) {
  set_calc_deltas(set, accumulate, ranges, active, 0, NULL, 0, NULL);
}

/* the set calls' active mask on the device (b.active, written by nothing else): uploaded unless these very flags are
 * there already -- the class-group loss sends the flags it returns along with its own uploads, because its caller passes
 * them straight on to the delta call (one launch less per generation).  queued: leaves with the caller's next flush */
static void active_mask_to_dev(RamdEngine *e, const u8 *active, int n, int queued) {
  if (e->active_host && e->active_host_n == n && memcmp(e->active_host, active, (size_t)n) == 0) {
    return;
  }
  if (!e->active_host) {
    e->active_host = ramd_zalloc((size_t)e->sh.Scap + 4);
  }
  memcpy(e->active_host, active, (size_t)n);
  e->active_host_n = n;
  if (queued) {
    upload_q(e->b.active, active, (size_t)n);
  } else {
    upload(e->b.active, active, (size_t)n); /* (through the mailbox when the set is a whole number of words) */
  }
}

/* may this set call leave its delta sums as planes (RamdEngine.kept)?  Not with a bottom layer (its deltas follow in
 * a launch of their own that accumulates), an exchange between ranks or a caller's delta buffer (both want the sums
 * as such, at once); the workspace is the engine's own, made on first use: eight planes and the rest rows' planes */
static int kept_deltas_ok(RamdEngine *e) {
  static int on = -1;
  if (on < 0) {
    const char *v = getenv("RECUR_AMD_KEEP_DELTAS");
    on = !(v && *v == '0');
  }
  if (!on || e->sh.bI || e->xchg_world || e->delta_external || rnn_amd_dist_world() > 1) {
    return 0;
  }
  if (!e->d_kept_slab) {
    size_t fl = 8 * e->ih_size + (size_t)64 * 128 * e->sh.H;
    if (fl * sizeof(float) > ((size_t)2 << 30)) {
      return 0;
    }
    e->d_kept_slab = dev_alloc(fl * sizeof(float));
    e->kept_floats = fl;
  }
  return 1;
}

/* dev_ranges: one range list per stream already on the device (range_stride ints apart),
 * instead of the shared host list `ranges` */
static void set_calc_deltas(RnnAmdSet *set, int accumulate, RecurErrorRange *ranges,
                            const u8 *active, unsigned extra_flags, const int *dev_ranges,
                            int range_stride, RamdPendingDelta *defer) {
  RamdEngine *e = set->eng;
  set_need_training(set, "rnn_amd_set_calc_deltas");
  if (e->deltas_zero_pending) { /* the sum into zeros is the sum */
    accumulate = 0;
    e->deltas_zero_pending = 0;
  }
  engine_need_dev(e, RNN_AMD_WEIGHTS | (accumulate ? RNN_AMD_DELTAS : 0));
  e->kept_live = 0; /* (summed just now if this call accumulates, otherwise overwritten) */
  if (!defer && !accumulate && kept_deltas_ok(e)) {
    /* leave the sums as planes for the rnn_apply_learning that normally follows (see RamdEngine.kept) */
    memset(&e->kept, 0, sizeof(e->kept));
    e->kept.own_slab = e->d_kept_slab;
    e->kept.own_slab_floats = e->kept_floats;
    defer = &e->kept;
  }
  set_streams_to_dev(set);
  push_learn_rates(e, set->row0, set->n);
  const int *d_ranges = dev_ranges ? dev_ranges : push_ranges(e, ranges);
  const unsigned char *d_active = NULL;
  if (active) {
    active_mask_to_dev(e, active, set->n, 0);
    d_active = e->b.active;
  }
  if (e->err_pending && d_ranges && e->err_row0 == set->row0 && e->err_nrows == set->n) {
    /* the ranged top backprops keep stale entries of these rows' error images, which have not been rebuilt yet: the
     * launcher sees to it (its per-head form reads them from the planes: no k_err_writeback launch) */
    extra_flags |= RAMD_IMAGES_PENDING;
    e->err_pending = 0;
  } else if (e->err_pending && (d_ranges || e->err_row0 != set->row0 || e->err_nrows != set->n)) {
    err_flush(e);
  }
  if (e->top_done && e->top_done_row0 == set->row0 && e->top_done_n == set->n && !d_ranges &&
      (e->top_done_masked ? (active && e->top_done_mask && memcmp(e->top_done_mask, active, (size_t)set->n) == 0)
                          : !active)) {
    extra_flags |= RAMD_TOP_DONE; /* with the loss (rnn_amd_set_opinion_sigmoid_mse / _grouped_softmax) */
  }
  top_done_clear(e);
  set_uniform_idx(e, set->row0, set->n);
  ramd_launch_calc_deltas(g_stream, &e->sh, &e->b, set->row0, set->n, accumulate, d_ranges,
                          dev_ranges ? range_stride : 0, d_active,
                          set->nets[0]->flags | extra_flags, defer);
  if (e->sh.bI) {
    if (set->global_count != set->n || e->delta_external) {
      fprintf(stderr, "librecur_amd: a bottom layer cannot be trained on a sharded set: its "
                      "error accumulator runs through the streams in order "
                      "(recur-nn.c:377-382)\n");
      abort();
    }
    ramd_launch_bottom_deltas(g_stream, &e->sh, &e->b, set->row0, set->n, accumulate, d_active);
  }
  e->err_pending = 1;
  e->err_row0 = set->row0;
  e->err_nrows = set->n;
  if (defer == &e->kept) {
    e->kept_live = e->kept.slab || e->kept.ho_slab;
  }
  engine_dev_wrote(e, RNN_AMD_DELTAS);
  set_streams_dev_wrote(set);
  for (int j = 0; j < set->n; j++) {
    if (!active || active[j]) {
      set->nets[j]->generation++; /* recur-nn.c:765 */
    }
  }
  if (set->nets[0]->log) {
    log_bptt(e, set->nets[0], set->nets[0]->bptt->min_error_factor);
    pull_scalars(e, set->row0, 1);
    rnn_log_int(set->nets[0], "generation", set->nets[0]->generation);
  }
}

/* gstclassify's loss for the whole set (gstclassify.c:2070-2119), see recur_amd.h */
void rnn_amd_set_grouped_softmax_error(RnnAmdSet *set, int n_groups, const int *group_offset,
                                       const int *group_size, const int *targets,
                                       const float *error_weight, u8 *trained) {
  RamdEngine *e = set->eng;
  set_need_training(set, "rnn_amd_set_grouped_softmax_error");
  top_done_clear(e);
  const RamdShape *s = &e->sh;
  int largest = 1;
  for (int i = 0; i < n_groups; i++) {
    if (group_offset[i] < 0 || group_size[i] < 1 || group_offset[i] + group_size[i] > s->output_size) {
      fprintf(stderr, "librecur_amd: class group %d (%d + %d) outside the %d outputs\n", i,
              group_offset[i], group_size[i], s->output_size);
      abort();
    }
    largest = RAMD_MAX(largest, group_size[i]);
  }
  set_streams_to_dev(set);
  /* one staging block: offsets, sizes, targets, then the weights */
  size_t ints = (size_t)2 * n_groups + (size_t)set->n * n_groups;
  size_t bytes = ints * sizeof(int) + (error_weight ? (size_t)s->output_size * sizeof(float) : 0);
  if (bytes > e->d_group_bytes) {
    dsync();
    dev_free(e->d_group);
    e->d_group = dev_alloc(bytes);
    e->d_group_bytes = bytes;
  }
  int *d = (int *)e->d_group;
  upload_q(d, group_offset, n_groups * sizeof(int));
  upload_q(d + n_groups, group_size, n_groups * sizeof(int));
  upload_q(d + 2 * n_groups, targets, (size_t)set->n * n_groups * sizeof(int));
  float *dw = NULL;
  if (error_weight) {
    dw = (float *)(d + ints);
    upload_q(dw, error_weight, (size_t)s->output_size * sizeof(float));
  }
  if (trained) { /* a stream is trained if any of its groups has a usable target */
    for (int j = 0; j < set->n; j++) {
      trained[j] = 0;
      for (int i = 0; i < n_groups; i++) {
        int t = targets[(size_t)j * n_groups + i];
        trained[j] |= (t >= 0 && t < group_size[i]);
      }
    }
    /* the caller hands these flags to rnn_amd_set_calc_deltas next (gstclassify.c:2120-2127): they travel with this
     * call's uploads, and that call finds them in place (active_mask_to_dev) */
    if (set->n % 4 == 0 && !set->fwd_only) {
      active_mask_to_dev(e, trained, set->n, 1);
    }
  }
  mail_in_flush();
  ramd_launch_grouped_softmax_error(g_stream, s, &e->b, set->row0, set->n, n_groups, largest, d,
                                    d + n_groups, d + 2 * n_groups, dw);
  set_streams_dev_wrote(set);
}

/* The multi-head text model for the whole set, per generation: what
 * charmodel-multi-predict.c:244-256 does per net (rnn_bptt_advance, multi_softmax_error
 * with its one_hot_opinion, rnn_bptt_calc_deltas with the error ranges), stream j
 * accumulating on top of stream j - 1.  The loss half leaves o_error and one range list
 * per stream on the device; the deltas half consumes them.  In between the caller may
 * apply the previous batch's deltas, as text_train does (244-252). */
static const int MULTI_RANGE_STRIDE = RAMD_MULTI_RANGE_STRIDE; /* 2 x (64 + 1) ints and the head bits */

static int multi_heads(RamdEngine *e, int alphabet_len) {
  const RamdShape *s = &e->sh;
  int n_classes = alphabet_len > 0 ? s->output_size / alphabet_len : 0;
  if (alphabet_len < 1 || n_classes < 1 || n_classes > 64) {
    fprintf(stderr, "librecur_amd: %d outputs as heads of %d: 1 to 64 heads are supported\n",
            s->output_size, alphabet_len);
    abort();
  }
  if (!e->d_mranges) {
    e->d_mranges = dev_alloc((size_t)s->Scap * MULTI_RANGE_STRIDE * sizeof(int));
    e->d_mclass = dev_alloc((size_t)s->Scap * sizeof(int));
  }
  /* the sparse top backprop's partial products, one row of h_size per (stream, head): 53 MB at 256 streams of 50
   * heads and 1024 hidden units; sets too large for that keep the one-GEMM form (k_top_backprop_heads) */
  size_t part = (size_t)s->Scap * n_classes * s->H;
  if (alphabet_len >= 24 && alphabet_len <= 128 && part > e->b.mheads_part_floats && part * sizeof(float) <= ((size_t)2 << 30)) {
    dev_free(e->b.mheads_part);
    e->b.mheads_part = dev_alloc(part * sizeof(float));
    e->b.mheads_part_floats = part;
  }
  return n_classes;
}

/* the loss after the opinion: b.target holds each stream's next symbol */
/* The presynaptic noise of the NEXT forward pass of this set, generated now on a second stream.
 * A stream's generator is a sequential recurrence -- 3 x (h_size - 1) dependent steps per pass,
 * one lane per stream, 0.2 ms for 1028 values whatever the number of streams -- so the only
 * way to take it off the generation's critical path is to run it beside something else: it is
 * launched as soon as the last draw before the next pass has been made (the multi-head loss's
 * leak decisions, or the pass itself in the text step) and runs beside the whole backward
 * pass.  It reads the generators and leaves them alone; the forward pass adds the values and
 * adopts the generator states that go with them (k_noise_apply) provided nothing has moved the
 * generators in between (rng_version: uploads of a generator, other passes, other losses),
 * otherwise it generates as before and the speculated values are dropped. */
static void noise_speculate_from(RnnAmdSet *set, int loss_classes);
static void noise_speculate(RnnAmdSet *set) { noise_speculate_from(set, 0); }
/* loss_classes > 0: the early form for the multi-head step, called between the forward pass and the loss.  When that
 * pass adopted the speculated states they still sit in b.rng_spec, which the loss does not touch: the generator is
 * started from there, `loss_classes` heads' worth of leak decisions on (one draw per head other than the stream's
 * own, whatever the outcome), beside the output layer and the loss instead of after them -- at 1024 / 256 / 73 x 50
 * the generator (0.31 ms beside the matrix kernels) had become the generation's critical path. */
static void noise_speculate_from(RnnAmdSet *set, int loss_classes) {
  RamdEngine *e = set->eng;
  const float noise = set->nets[0]->presynaptic_noise;
  static int enabled = -1, two_ahead = -1;
  if (enabled < 0) {
    const char *env = getenv("RECUR_AMD_NOISE_AHEAD");
    enabled = !(env && atoi(env) == 0);
    two_ahead = !(env && atoi(env) == 1); /* (1: one pass ahead only) */
  }
  if (noise == 0.0f || e->sh.bI || set->fwd_only || !enabled) {
    return;
  }
  if (!g_side) {
    HIP_OK(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
    ramd_note_side_stream();
  }
  if (!e->spec_go) {
    HIP_OK(hipEventCreateWithFlags((hipEvent_t *)&e->spec_go, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags((hipEvent_t *)&e->sp[0].done, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags((hipEvent_t *)&e->sp[1].done, hipEventDisableTiming));
  }
  for (int k = 0; k < 2; k++) {
    if (!e->sp[k].noise) {
      e->sp[k].noise = dev_alloc((size_t)e->sh.Scap * e->sh.H * sizeof(float));
      e->sp[k].states = dev_alloc((size_t)e->sh.Scap * sizeof(rand_ctx));
    }
  }
  const int r0 = set->row0;
  int launched = 0;
  /* the next pass (buffer sp_head), unless an earlier call has started it already */
  int k = e->sp_head;
  if (!e->sp[k].pending) {
    const int early = loss_classes > 0 && e->sp_adopted >= 0;
    if (loss_classes > 0 && !early) {
      return; /* (between the pass and the loss only the adopted states are safe to start from) */
    }
    HIP_OK(hipEventRecord((hipEvent_t)e->spec_go, g_stream)); /* the draws made so far, the buffers' last readers */
    HIP_OK(hipStreamWaitEvent(g_side, (hipEvent_t)e->spec_go, 0));
    ramd_launch_noise_speculate(g_side, &e->sh, &e->b, r0, set->n, noise, e->sp[k].noise, e->sp[k].states,
                                early ? e->sp[e->sp_adopted].states : NULL, early ? e->d_mclass + r0 : NULL,
                                early ? loss_classes : 0, 0);
    HIP_OK(hipEventRecord((hipEvent_t)e->sp[k].done, g_side));
    e->sp[k].pending = 1;
    e->sp[k].version = e->rng_version + (early ? 1 : 0); /* (the loss that follows counts as one move) */
    e->sp[k].row0 = r0;
    e->sp[k].n = set->n;
    e->sp[k].dev = noise;
    e->sp[k].assumed_classes = 0;
    launched = 1;
  }
  /* ... and, in the multi-head step, the pass after it: from the states the first one leaves, past the draws of a
   * loss whose classes nobody has yet -- one per head but the stream's own, so on the assumption that every stream's own
   * class is one of the heads (multi_loss checks it when it comes and drops what was built on it otherwise).  The
   * generator then runs a whole generation ahead and is off the critical path even where it is as long as the
   * generation (32 streams per GPU). */
  const int k2 = k ^ 1;
  if (two_ahead && loss_classes > 0 && e->sp[k].pending && !e->sp[k2].pending && e->sp[k].row0 == r0 &&
      e->sp[k].n == set->n && e->sp[k].dev == noise) {
    if (!launched) {
      HIP_OK(hipEventRecord((hipEvent_t)e->spec_go, g_stream)); /* buffer k2's last reader: the pass just made */
      HIP_OK(hipStreamWaitEvent(g_side, (hipEvent_t)e->spec_go, 0));
    }
    ramd_launch_noise_speculate(g_side, &e->sh, &e->b, r0, set->n, noise, e->sp[k2].noise, e->sp[k2].states,
                                e->sp[k].states, NULL, 0, loss_classes - 1);
    HIP_OK(hipEventRecord((hipEvent_t)e->sp[k2].done, g_side));
    e->sp[k2].pending = 1;
    e->sp[k2].version = e->sp[k].version + 2; /* one more pass and one more loss on */
    e->sp[k2].row0 = r0;
    e->sp[k2].n = set->n;
    e->sp[k2].dev = noise;
    e->sp[k2].assumed_classes = loss_classes;
  }
}

static void mclass_note(RamdEngine *e, const int *target_class, int n, int n_classes) {
  int ok = 1;
  for (int j = 0; j < n; j++) {
    ok &= target_class[j] >= 0 && target_class[j] < n_classes;
  }
  e->mclass_in_range = ok;
}

static void multi_loss(RnnAmdSet *set, const int *target_class, int alphabet_len, int n_classes,
                       float leakage, int speculated) {
  RamdEngine *e = set->eng;
  e->mheads_alen = alphabet_len;
  e->b.mheads_alen = alphabet_len;
  e->rng_version++; /* the leak decisions are draws from the streams' generators */
  if (target_class) {
    upload(e->d_mclass + set->row0, target_class, set->n * sizeof(int));
    mclass_note(e, target_class, set->n, n_classes);
  }
  /* noise that was generated past THIS loss before its classes were known */
  for (int k = 0; k < 2; k++) {
    if (e->sp[k].pending && e->sp[k].assumed_classes && e->sp[k].version == e->rng_version &&
        (e->sp[k].assumed_classes != n_classes || !e->mclass_in_range)) {
      e->sp[0].pending = e->sp[1].pending = 0; /* (the other one hangs on it, or is the pass in between: redone) */
    }
  }
  /* u64 threshold = leakage * UINT64_MAX (charmodel-multi-predict.c:27): float arithmetic */
  float tf = leakage * (float)UINT64_MAX;
  unsigned long long threshold = tf >= 18446744073709551615.0f ? UINT64_MAX : (unsigned long long)tf;
  ramd_launch_multi_softmax_error(g_stream, &e->sh, &e->b, set->row0, set->n, alphabet_len, n_classes,
                                  threshold, e->d_mclass + set->row0,
                                  e->d_mranges + (size_t)set->row0 * MULTI_RANGE_STRIDE,
                                  MULTI_RANGE_STRIDE);
  set_streams_dev_wrote(set);
  if (!speculated) {
    noise_speculate(set); /* the next draws are the next forward pass's noise */
  }
}

static void multi_calc_deltas(RnnAmdSet *set, int accumulate, RamdPendingDelta *defer) {
  RamdEngine *e = set->eng;
  set_need_training(set, "rnn_amd_set_multi_calc_deltas");
  if (!e->d_mranges) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_multi_calc_deltas before any multi-head loss\n");
    abort();
  }
  /* (heads of at least 24 columns: a stream's ranges then lie at least 16 columns apart, see k_top_backprop_heads) */
  set_calc_deltas(set, accumulate, NULL, NULL, e->mheads_alen >= 24 ? RAMD_RANGES_ARE_HEADS : 0,
                  e->d_mranges + (size_t)set->row0 * MULTI_RANGE_STRIDE, MULTI_RANGE_STRIDE, defer);
}
void rnn_amd_set_multi_calc_deltas(RnnAmdSet *set, int accumulate) { multi_calc_deltas(set, accumulate, NULL); }

static void multi_step_deltas(RnnAmdSet *set, const int *hot, const int *next, const int *target_class, int alphabet_len,
                              float leakage, int accumulate, RamdPendingDelta *defer);
void rnn_amd_set_multi_step_deltas(RnnAmdSet *set, const int *hot, const int *next,
                                   const int *target_class, int alphabet_len, float leakage,
                                   int accumulate) {
  multi_step_deltas(set, hot, next, target_class, alphabet_len, leakage, accumulate, NULL);
}
static void multi_step_deltas(RnnAmdSet *set, const int *hot, const int *next, const int *target_class, int alphabet_len,
                              float leakage, int accumulate, RamdPendingDelta *defer) {
  RamdEngine *e = set->eng;
  set_need_training(set, "rnn_amd_set_multi_step_deltas");
  int n_classes = multi_heads(e, alphabet_len);
  /* the three small uploads leave with one launch */
  upload_q(e->b.hot + set->row0, hot, set->n * sizeof(int));
  if (target_class) {
    upload_q(e->d_mclass + set->row0, target_class, set->n * sizeof(int));
    mclass_note(e, target_class, set->n, n_classes);
  }
  upload(e->b.target + set->row0, next, set->n * sizeof(int));
  set->early_spec_classes = set->fwd_only ? 0 : n_classes; /* (set_forward's fused branch takes it up) */
  set_forward(set, RAMD_IN_ONE_HOT, NULL, 0, 0, NULL, 1, 0);
  int early = set->early_spec_classes < 0;
  set->early_spec_classes = 0;
  if (!early && e->sp_adopted >= 0 && !set->fwd_only) {
    noise_speculate_from(set, n_classes);
    early = 1;
  }
  multi_loss(set, NULL, alphabet_len, n_classes, leakage, early);
  multi_calc_deltas(set, accumulate, defer);
}

static void check_text_pos(RamdEngine *e, int i, int last_ok, const char *what) {
  if (!e->b.text) {
    fprintf(stderr, "librecur_amd: %s without rnn_amd_set_load_text\n", what);
    abort();
  }
  /* the reference's loops run i over [start, len - 1) (charmodel-predict.c:288,
   * charmodel-multi-predict.c:244); the kernels wrap the per-stream offset once, so i
   * itself has to be in range */
  if (i < 0 || i >= e->b.text_len - 1 + last_ok) {
    fprintf(stderr, "librecur_amd: %s: text position %d outside [0, %d)\n", what, i,
            e->b.text_len - 1 + last_ok);
    abort();
  }
}

/* the same from the text on the device: stream j reads text[o] and is scored against
 * text[o + 1], o as in rnn_amd_set_char_step */
void rnn_amd_set_multi_text_loss(RnnAmdSet *set, int i, const int *target_class, int alphabet_len,
                                 float leakage) {
  RamdEngine *e = set->eng;
  set_need_training(set, "rnn_amd_set_multi_text_loss");
  check_text_pos(e, i, 0, "rnn_amd_set_multi_text_loss");
  int n_classes = multi_heads(e, alphabet_len);
  set_forward(set, RAMD_IN_TEXT, NULL, 0, i, NULL, 1, 0); /* advance + one-hot opinion + b.target */
  multi_loss(set, target_class, alphabet_len, n_classes, leakage, 0);
}

/* rnn_bptt_advance (optional) + one_hot_opinion of the stream's text symbol, nothing else:
 * rnn_char_multitext_spin's step (charmodel-multi-predict.c:293-297); i may be len - 1 */
void rnn_amd_set_text_opinion(RnnAmdSet *set, int i, int advance) {
  RamdEngine *e = set->eng;
  check_text_pos(e, i, 1, "rnn_amd_set_text_opinion");
  set_forward(set, RAMD_IN_TEXT, NULL, 0, i, NULL, advance, 0);
}

/* rnnca's loss on the device (gstrnnca.c:701-714) after rnn_amd_set_opinion: sigmoid of the
 * first n outputs in place, o_error = slope * (target - answer); targets: host [n_nets][ld] */
void rnn_amd_set_sigmoid_mse_error(RnnAmdSet *set, const float *targets, int ld, int n) {
  RamdEngine *e = set->eng;
  top_done_clear(e);
  set_need_training(set, "rnn_amd_set_sigmoid_mse_error");
  if (n < 1 || n > e->sh.output_size || ld < n) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_sigmoid_mse_error over %d of %d outputs (ld %d)\n", n,
            e->sh.output_size, ld);
    abort();
  }
  set_streams_to_dev(set);
  size_t bytes = (size_t)set->n * n * sizeof(float);
  if (bytes > e->d_group_bytes) {
    dsync();
    dev_free(e->d_group);
    e->d_group = dev_alloc(bytes);
    e->d_group_bytes = bytes;
  }
  upload_rows(e->d_group, targets, ld * sizeof(float), n * sizeof(float), set->n);
  ramd_launch_sigmoid_mse_error(g_stream, &e->sh, &e->b, set->row0, set->n, n, (const float *)e->d_group,
                                n);
  set_streams_dev_wrote(set);
}

/* dense inputs -> hidden layer's sums only (they stay in the workspace for ramd_launch_dense_top); 0 where that launch
 * cannot follow */
static int dense_top_ok(RnnAmdSet *set, const float *inputs) {
  RamdEngine *e = set->eng;
  return inputs && !set->fwd_only && !e->sh.bI && ramd_dense_top_ok(&e->sh) &&
         set->nets[0]->presynaptic_noise == 0.0f && set->nets[0]->bptt;
}

void rnn_amd_set_opinion_sigmoid_mse(RnnAmdSet *set, const float *inputs, int ld_inputs, const float *targets, int ld,
                                     int n) {
  RamdEngine *e = set->eng;
  if (!dense_top_ok(set, inputs) || n < 1 || n > e->sh.output_size || ld < n) {
    rnn_amd_set_opinion(set, inputs, ld_inputs, NULL);
    rnn_amd_set_sigmoid_mse_error(set, targets, ld, n);
    return;
  }
  set_need_training(set, "rnn_amd_set_opinion_sigmoid_mse");
  size_t bytes = (size_t)set->n * n * sizeof(float);
  if (bytes > e->d_group_bytes) {
    dsync();
    dev_free(e->d_group);
    e->d_group = dev_alloc(bytes);
    e->d_group_bytes = bytes;
  }
  /* inputs and targets leave together with the ring indices (set_forward's flush) */
  upload_rows_q(e->d_dense, inputs, ld_inputs * sizeof(float), e->sh.input_size * sizeof(float), set->n, 0);
  upload_rows_q(e->d_group, targets, ld * sizeof(float), n * sizeof(float), set->n, 0);
  int fwd_ks = set_forward(set, RAMD_IN_DENSE, e->d_dense, e->sh.input_size, 0, NULL, 0, 1);
  if (!ramd_launch_dense_top(g_stream, &e->sh, &e->b, set->row0, set->n, fwd_ks, (const float *)e->d_group, n, n, 0,
                             NULL, NULL, NULL, NULL)) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_opinion_sigmoid_mse: the top launch declined a shape it had accepted\n");
    abort();
  }
  set_streams_dev_wrote(set);
  e->top_done = 1;
  e->top_done_row0 = set->row0;
  e->top_done_n = set->n;
  e->top_done_masked = 0;
}

void rnn_amd_set_opinion_grouped_softmax(RnnAmdSet *set, const float *inputs, int ld_inputs, int n_groups,
                                         const int *group_offset, const int *group_size, const int *targets,
                                         const float *error_weight, u8 *trained) {
  RamdEngine *e = set->eng;
  const RamdShape *s = &e->sh;
  if (!dense_top_ok(set, inputs) || !trained || set->n % 4 != 0) {
    rnn_amd_set_opinion(set, inputs, ld_inputs, NULL);
    rnn_amd_set_grouped_softmax_error(set, n_groups, group_offset, group_size, targets, error_weight, trained);
    return;
  }
  set_need_training(set, "rnn_amd_set_opinion_grouped_softmax");
  for (int i = 0; i < n_groups; i++) {
    if (group_offset[i] < 0 || group_size[i] < 1 || group_offset[i] + group_size[i] > s->output_size) {
      fprintf(stderr, "librecur_amd: class group %d (%d + %d) outside the %d outputs\n", i, group_offset[i],
              group_size[i], s->output_size);
      abort();
    }
  }
  size_t ints = (size_t)2 * n_groups + (size_t)set->n * n_groups;
  size_t bytes = ints * sizeof(int) + (error_weight ? (size_t)s->output_size * sizeof(float) : 0);
  if (bytes > e->d_group_bytes) {
    dsync();
    dev_free(e->d_group);
    e->d_group = dev_alloc(bytes);
    e->d_group_bytes = bytes;
  }
  int *d = (int *)e->d_group;
  upload_rows_q(e->d_dense, inputs, ld_inputs * sizeof(float), s->input_size * sizeof(float), set->n, 0);
  upload_q(d, group_offset, n_groups * sizeof(int));
  upload_q(d + n_groups, group_size, n_groups * sizeof(int));
  upload_q(d + 2 * n_groups, targets, (size_t)set->n * n_groups * sizeof(int));
  float *dw = NULL;
  if (error_weight) {
    dw = (float *)(d + ints);
    upload_q(dw, error_weight, (size_t)s->output_size * sizeof(float));
  }
  for (int j = 0; j < set->n; j++) { /* a stream is trained if any of its groups has a usable target */
    trained[j] = 0;
    for (int i = 0; i < n_groups; i++) {
      int t = targets[(size_t)j * n_groups + i];
      trained[j] |= (t >= 0 && t < group_size[i]);
    }
  }
  active_mask_to_dev(e, trained, set->n, 1); /* (for the delta call that follows: travels with the rest) */
  int fwd_ks = set_forward(set, RAMD_IN_DENSE, e->d_dense, s->input_size, 0, NULL, 0, 1);
  if (!ramd_launch_dense_top(g_stream, s, &e->b, set->row0, set->n, fwd_ks, NULL, 0, 0, n_groups, d, d + n_groups,
                             d + 2 * n_groups, dw)) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_opinion_grouped_softmax: the top launch declined a shape it had accepted\n");
    abort();
  }
  set_streams_dev_wrote(set);
  e->top_done = 1;
  e->top_done_row0 = set->row0;
  e->top_done_n = set->n;
  e->top_done_masked = 1;
  if (!e->top_done_mask) {
    e->top_done_mask = ramd_zalloc((size_t)e->sh.Scap + 4);
  }
  memcpy(e->top_done_mask, trained, (size_t)set->n); /* (a copy of its own: active_host follows whatever mask is sent next) */
}

/* fill_frame's fast_sigmoid_array(answer, answer, n) (gstrnnca.c:813-814) for every net of the
 * set (training or forward-only), then optionally the n_nets x o_size answers to the host */
void rnn_amd_set_sigmoid_outputs(RnnAmdSet *set, int n, float *outputs) {
  RamdEngine *e = set->eng;
  if (n < 1 || n > e->sh.output_size) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_sigmoid_outputs over %d of %d outputs\n", n,
            e->sh.output_size);
    abort();
  }
  set_streams_to_dev(set);
  const int r0 = set_state_row0(set);
  ramd_launch_sigmoid_outputs(g_stream, &e->sh, &e->b, r0, set->n, n);
  set_streams_dev_wrote(set);
  if (outputs) {
    d2h(outputs, e->b.out + (size_t)r0 * e->sh.O, (size_t)set->n * e->sh.O * sizeof(float));
    dsync();
  }
}

void rnn_amd_set_load_text(RnnAmdSet *set, const u8 *text, int len) {
  RamdEngine *e = set->eng;
  if (!text || len < 2) { /* a step reads text[o] and text[o + 1] */
    fprintf(stderr, "librecur_amd: rnn_amd_set_load_text needs at least 2 symbols (got %d)\n", len);
    abort();
  }
  dsync();
  dev_free(e->b.text);
  e->b.text = dev_alloc((size_t)len + 1); /* zeroed: the "next symbol" of the last one */
  h2d(e->b.text, text, len);
  e->b.text_len = len;
  dsync();
}

static void char_step_deltas(RnnAmdSet *set, int i, RamdPendingDelta *defer) {
  RamdEngine *e = set->eng;
  check_text_pos(e, i, 0, "rnn_amd_set_char_step");
  if (ramd_text_top_ok(&e->sh)) {
    /* advance + hidden layer, then output layer, loss and top backprop in one launch */
    int fwd_ks = set_forward(set, RAMD_IN_TEXT, NULL, 0, i, NULL, 1, 1);
    noise_speculate(set); /* no draws between this pass's noise and the next one's */
    ramd_launch_text_top(g_stream, &e->sh, &e->b, set->row0, set->n, fwd_ks);
    set_calc_deltas(set, 0, NULL, NULL, RAMD_TOP_DONE, NULL, 0, defer);
  } else {
    set_forward(set, RAMD_IN_TEXT, NULL, 0, i, NULL, 1, 0); /* advance + one-hot opinion */
    noise_speculate(set);
    ramd_launch_softmax_error(g_stream, &e->sh, &e->b, set->row0, set->n);
    set_calc_deltas(set, 0, NULL, NULL, 0, NULL, 0, defer);
  }
}

void rnn_amd_set_char_step_deltas(RnnAmdSet *set, int i) { char_step_deltas(set, i, NULL); }

static void apply_learning(RecurNN *net, int learning_method, float momentum,
                           const RamdPendingDelta *pend);

/* The exchange step overlapped with the weight-delta GEMM: the GEMM runs as two
 * row halves (kernels_bptt.hip: g_delta_half_hook); as soon as a half's deltas are complete its sum over the
 * ranks starts on a stream of its own, so the first half's all-reduce (2.2 of the 4.6 MB at the north
 * star) travels over xGMI while the second half is still being multiplied.  The update waits for both.
 * Every rank reduces the same two ranges in the same order, so the replicas stay bit-identical. */
static hipStream_t g_comm_stream = NULL;
static hipEvent_t g_half_ready[2], g_half_summed[2];
static int g_halves_seen = 0;

static void delta_half_ready(void *ctx, int half, size_t first_float, size_t n_floats) {
  RamdEngine *e = ctx;
  if (!g_comm_stream) {
    HIP_OK(hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking));
    ramd_note_side_stream();
    for (int h = 0; h < 2; h++) {
      HIP_OK(hipEventCreateWithFlags(&g_half_ready[h], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&g_half_summed[h], hipEventDisableTiming));
    }
  }
  HIP_OK(hipEventRecord(g_half_ready[half], g_stream));
  HIP_OK(hipStreamWaitEvent(g_comm_stream, g_half_ready[half], 0));
  ramd_dist_all_reduce_on(e->b.ih_delta + first_float, n_floats, g_comm_stream);
  HIP_OK(hipEventRecord(g_half_summed[half], g_comm_stream));
  g_halves_seen |= 1 << half;
}

/* ---- the exchange step as kernel-issued peer traffic (include/recur_amd.h; kernels_apply.hip: k_apply_xchg) ---- */
typedef struct XchgBlob {
  uint64_t pid;
  uint64_t nonce;     /* of this export: the ranks' blobs together name the session (xchg_session_token) */
  uint64_t raw[3];    /* delta, ih_w, ho_w as this process sees them                        */
  uint64_t offset[3]; /* of each inside its allocation (IPC handles name whole allocations) */
  hipIpcMemHandle_t handle[3];
} XchgBlob;

void rnn_amd_set_exchange_export(RnnAmdSet *set, void *blob) {
  RamdEngine *e = set->eng;
  _Static_assert(sizeof(XchgBlob) <= RNN_AMD_EXCHANGE_BLOB_BYTES, "the blob outgrew its public size");
  engine_need_dev(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  deltas_materialize(e);
  XchgBlob b;
  memset(&b, 0, sizeof(b));
  b.pid = (uint64_t)getpid();
  {
    static uint64_t exports = 0;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    b.nonce = ((uint64_t)ts.tv_sec << 30) ^ (uint64_t)ts.tv_nsec ^ (++exports << 48);
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    b.nonce = (b.nonce & ~(uint64_t)0xff) | (uint64_t)(dev & 0xff); /* (low byte: the exporting rank's device, for the join's peer-access check) */
  }
  void *arrays[3] = {e->b.ih_delta, e->b.ih_w, e->b.ho_w};
  for (int k = 0; k < 3; k++) {
    hipDeviceptr_t base = NULL;
    size_t size = 0;
    HIP_OK(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)arrays[k]));
    b.raw[k] = (uint64_t)(uintptr_t)arrays[k];
    b.offset[k] = (uint64_t)((char *)arrays[k] - (char *)base);
    if (hipIpcGetMemHandle(&b.handle[k], (void *)base) != hipSuccess) {
      (void)hipGetLastError(); /* (peers of the same process do not need it) */
      memset(&b.handle[k], 0, sizeof(b.handle[k]));
    }
  }
  memset(blob, 0, RNN_AMD_EXCHANGE_BLOB_BYTES);
  memcpy(blob, &b, sizeof(b));
}

void rnn_amd_set_exchange_leave(RnnAmdSet *set) {
  RamdEngine *e = set->eng;
  if (!e->xchg_world) {
    return;
  }
  dsync();
  for (int p = 0; p < e->xchg_world; p++) {
    for (int k = 0; k < 3; k++) {
      if (e->xchg_opened[p][k]) {
        (void)hipIpcCloseMemHandle(e->xchg_opened[p][k]);
        e->xchg_opened[p][k] = NULL;
      }
    }
  }
  if (e->xchg_flags_host) {
    (void)hipHostUnregister(e->xchg_flags_host);
  }
  e->xchg_flags_host = NULL;
  e->xchg_flags_dev = NULL;
  e->xchg_world = 0;
}

/* The session's name: every rank holds the same `world` blobs in the same order, every export has its own nonce -- a
 * 32-bit hash of them all is the same on every rank and new for every session.  Even and not 0: T = arrived, T | 1 = has
 * read the counters. */
static unsigned xchg_session_token(const void *blobs, int world) {
  unsigned h = 2166136261u; /* FNV-1a */
  const unsigned char *p = blobs;
  for (size_t i = 0; i < (size_t)world * RNN_AMD_EXCHANGE_BLOB_BYTES; i++) {
    h = (h ^ p[i]) * 16777619u;
  }
  h &= ~1u;
  return h ? h : 2u;
}

/* The join is COLLECTIVE between processes: a rendezvous on the host in words 8 .. 15 of the shared counters (the
 * barrier kernels count in words 0 .. 7).  Phase 1: everybody has arrived in THIS session (stale words of an earlier
 * one do not match its token); then every rank reads where the barrier counters stand; phase 2: everybody has read them
 * -- only then may anyone step and move them.  Bounded: RECUR_AMD_XCHG_JOIN_TIMEOUT seconds (default 120). */
static int xchg_rendezvous(unsigned *c, int rank, int world, unsigned token, unsigned *top_out) {
  const char *te = getenv("RECUR_AMD_XCHG_JOIN_TIMEOUT");
  double limit = te && atof(te) > 0 ? atof(te) : 120.0;
  struct timespec t0, t;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int phase = 0; phase < 2; phase++) {
    if (phase == 1) {
      unsigned top = __atomic_load_n(&c[0], __ATOMIC_ACQUIRE);
      for (int p = 1; p < world; p++) {
        const unsigned v = __atomic_load_n(&c[p], __ATOMIC_ACQUIRE);
        if ((int)(v - top) > 0) {
          top = v;
        }
      }
      *top_out = top;
    }
    __atomic_store_n(&c[8 + rank], token | (unsigned)phase, __ATOMIC_RELEASE);
    for (int p = 0; p < world; p++) {
      for (;;) {
        const unsigned v = __atomic_load_n(&c[8 + p], __ATOMIC_ACQUIRE);
        if (v == (token | 1u) || (phase == 0 && v == token)) {
          break;
        }
        clock_gettime(CLOCK_MONOTONIC, &t);
        if ((t.tv_sec - t0.tv_sec) + 1e-9 * (t.tv_nsec - t0.tv_nsec) > limit) {
          fprintf(stderr, "librecur_amd: rnn_amd_set_exchange_join: rank %d has waited %.0f s for rank %d to join (every "
                          "rank calls the join, with the same blobs and counters that all of them map)\n", rank, limit, p);
          return -1;
        }
        usleep(50);
      }
    }
  }
  return 0;
}

int rnn_amd_set_exchange_join(RnnAmdSet *set, int rank, int world, const void *blobs, void *counters, int lockstep) {
  RamdEngine *e = set->eng;
  set_need_training(set, "rnn_amd_set_exchange_join");
  if (world < 1 || world > 8 || rank < 0 || rank >= world || !blobs || (!lockstep && !counters) || e->sh.bI ||
      e->xchg_world || e->delta_external) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_exchange_join(rank %d, world %d): 1..8 ranks, every rank's blob, shared "
                    "counters unless in lock step, no bottom layer, no external delta buffer, not joined already\n",
            rank, world);
    return -1;
  }
  engine_need_dev(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  deltas_materialize(e);
  { /* can this rank's kernels reach every peer's device at all?  Asked BEFORE anything is opened: a kernel that stores
     * through a pointer it may not use faults, and a launcher that tries this exchange beside RCCL (bench.py --exchange
     * auto) must get a -1 to fall back on, not a dead rank */
    int mydev = 0;
    HIP_OK(hipGetDevice(&mydev));
    for (int p = 0; p < world; p++) {
      XchgBlob b;
      memcpy(&b, (const char *)blobs + (size_t)p * RNN_AMD_EXCHANGE_BLOB_BYTES, sizeof(b));
      const int peerdev = (int)(b.nonce & 0xff);
      int can = 1;
      if (p != rank && peerdev != mydev && hipDeviceCanAccessPeer(&can, mydev, peerdev) != hipSuccess) {
        (void)hipGetLastError();
        can = 0;
      }
      if (!can) {
        fprintf(stderr, "librecur_amd: rank %d (device %d) has no peer access to rank %d's device %d: the kernel-issued "
                        "exchange needs it (use the RCCL all-reduce)\n", rank, mydev, p, peerdev);
        return -1;
      }
    }
  }
  float **dst[3] = {e->xchg_delta, e->xchg_ihw, e->xchg_how};
  void *own[3] = {e->b.ih_delta, e->b.ih_w, e->b.ho_w};
  memset(e->xchg_opened, 0, sizeof(e->xchg_opened));
  for (int p = 0; p < world; p++) {
    XchgBlob b;
    memcpy(&b, (const char *)blobs + (size_t)p * RNN_AMD_EXCHANGE_BLOB_BYTES, sizeof(b));
    for (int k = 0; k < 3; k++) {
      if (p == rank) {
        dst[k][p] = own[k];
      } else if (b.pid == (uint64_t)getpid()) {
        dst[k][p] = (float *)(uintptr_t)b.raw[k]; /* another set of this process */
      } else {
        void *base = NULL;
        if (hipIpcOpenMemHandle(&base, b.handle[k], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
          fprintf(stderr, "librecur_amd: rank %d cannot open rank %d's arrays (%s): is there peer access between the "
                          "two GPUs, HSA_ENABLE_IPC_MODE_LEGACY=0 set?\n", rank, p, hipGetErrorString(hipGetLastError()));
          e->xchg_world = p + 1;
          rnn_amd_set_exchange_leave(set);
          return -1;
        }
        e->xchg_opened[p][k] = base;
        dst[k][p] = (float *)((char *)base + b.offset[k]);
      }
    }
  }
  e->xchg_flags_dev = NULL;
  e->xchg_flags_host = NULL;
  if (!lockstep) {
    if (hipHostRegister(counters, 64, hipHostRegisterMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&e->xchg_flags_dev, counters, 0) != hipSuccess) {
      fprintf(stderr, "librecur_amd: the shared counters cannot be mapped (%s)\n", hipGetErrorString(hipGetLastError()));
      e->xchg_world = world;
      rnn_amd_set_exchange_leave(set);
      return -1;
    }
    e->xchg_flags_host = counters;
    ramd_note_side_stream(); /* the peers' work runs beside ours where the ranks share a GPU */
  }
  e->xchg_world = world;
  e->xchg_rank = rank;
  e->xchg_lockstep = lockstep;
  /* The barriers count on from where the shared counters stand (round 6; they counted from 0 and trusted the launcher to
   * have zeroed them: on counters left from an earlier session the first barriers would have passed at once).  Every
   * rank reads the same `world` words -- between the two phases of the rendezvous, when nobody can be stepping -- and
   * takes the same start: the furthest of them, compared as the barrier compares (wrap-safe). */
  e->xchg_seq = 0;
  if (!lockstep) {
    unsigned top = 0;
    if (xchg_rendezvous(counters, rank, world, xchg_session_token(blobs, world), &top) != 0) {
      rnn_amd_set_exchange_leave(set);
      return -1;
    }
    e->xchg_seq = top;
  }
  return 0;
}

/* A 64-bit checksum of this replica -- ih_weights || ho_weights (|| ih_momentum || ho_momentum) -- for launchers that
 * want to KNOW that the ranks' replicas stayed identical (include/recur_amd.h).  through_kernel: summed by a kernel on
 * the library's stream, through the caches the path's kernels read through; otherwise over a device-to-host copy (the
 * copy engine reads memory).  The two agree unless something stored into the arrays behind the caches' back. */
uint64_t rnn_amd_set_replica_checksum(RnnAmdSet *set, int with_momentum, int through_kernel) {
  RamdEngine *e = set->eng;
  engine_need_dev(e, RNN_AMD_WEIGHTS | (with_momentum ? RNN_AMD_MOMENTUMS : 0));
  const float *arrays[4] = {e->b.ih_w, e->b.ho_w, e->b.ih_m, e->b.ho_m};
  const size_t n[4] = {e->ih_size, e->ho_size, e->ih_size, e->ho_size};
  const int n_arrays = with_momentum ? 4 : 2;
  uint64_t sum = 0;
  if (through_kernel) {
    unsigned long long *d = dev_alloc(sizeof(*d));
    ramd_launch_replica_checksum(g_stream, n_arrays, arrays, n, d);
    d2h(&sum, d, sizeof(sum));
    dsync();
    dev_free(d);
    return sum;
  }
  uint64_t first = 0;
  for (int k = 0; k < n_arrays; k++) {
    uint32_t *h = malloc(n[k] * sizeof(uint32_t));
    if (!h) {
      fprintf(stderr, "librecur_amd: rnn_amd_set_replica_checksum: out of memory\n");
      abort();
    }
    d2h(h, arrays[k], n[k] * sizeof(uint32_t));
    dsync();
    for (size_t i = 0; i < n[k]; i++) {
      sum += (uint64_t)h[i] * (2 * (first + i) + 1);
    }
    first += n[k];
    free(h);
  }
  return sum;
}

void rnn_amd_set_exchange_range(const RnnAmdSet *set, int which, size_t *first, size_t *count) {
  const RamdEngine *e = set->eng;
  const size_t n4 = (which ? e->ho_size : e->ih_size) / 4;
  const int world = e->xchg_world ? e->xchg_world : 1, rank = e->xchg_world ? e->xchg_rank : 0;
  const size_t lo = n4 * (size_t)rank / world, hi = n4 * (size_t)(rank + 1) / world;
  *first = 4 * lo;
  *count = 4 * (hi - lo);
}

static void xchg_barrier(RamdEngine *e) {
  if (!e->xchg_lockstep) {
    ramd_launch_xchg_barrier(g_stream, e->xchg_flags_dev, e->xchg_rank, e->xchg_world, ++e->xchg_seq, ramd_abort_word_dev());
  }
}

void rnn_amd_set_apply_exchange(RnnAmdSet *set, int learning_style, float momentum) {
  RamdEngine *e = set->eng;
  RecurNN *net = set->nets[0];
  RecurNNBPTT *bptt = net->bptt;
  if (!e->xchg_world) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_apply_exchange without rnn_amd_set_exchange_join\n");
    abort();
  }
  engine_need_dev(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  deltas_materialize(e);
  float mw; /* (as apply_learning) */
  if (learning_style == RNN_MOMENTUM_SIMPLIFIED_NESTEROV) {
    mw = momentum / (1.0 + momentum);
  } else if (learning_style == RNN_MOMENTUM_CLASSICAL) {
    mw = 1.0f;
  } else {
    mw = bptt->momentum_weight;
  }
  int method = learning_style;
  if (learning_style == RNN_MOMENTUM_SIMPLIFIED_NESTEROV || learning_style == RNN_MOMENTUM_CLASSICAL ||
      learning_style >= RNN_LAST_LEARNING_METHOD || learning_style < 0) {
    method = RNN_MOMENTUM_WEIGHTED;
  }
  check_method_arrays(e, method);
  const int W = e->xchg_world;
  float *w[16];
  const float *d[16];
  for (int p = 0; p < W; p++) { /* segment 0: the top layer, 1: the recurrent layer */
    w[p] = e->xchg_how[p];
    w[W + p] = e->xchg_ihw[p];
    d[p] = e->xchg_delta[p] + e->ih_size;
    d[W + p] = e->xchg_delta[p];
  }
  float *m[2] = {e->b.ho_m, e->b.ih_m}, *aux[2] = {e->b.ho_aux, e->b.ih_aux}, *dout[2] = {e->b.ho_delta, e->b.ih_delta};
  size_t n[2] = {e->ho_size, e->ih_size};
  float rate[2] = {bptt->learn_rate * bptt->ho_scale, bptt->learn_rate};
  const int tev = ramd_timing_begin(g_stream, RAMD_T_XCHG); /* (the two arrivals and the sharded update: what a rank waits and works for) */
  xchg_barrier(e); /* every rank's local sums are complete (and nobody still multiplies with the old weights) */
  ramd_launch_apply_xchg(g_stream, method, e->xchg_rank, W, w, d, m, aux, dout, n, rate, momentum, mw);
  xchg_barrier(e); /* every range of the weights has arrived here */
  ramd_timing_end(g_stream, tev);
  engine_dev_wrote(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
}

/* With ONE rank the sum over the ranks is the rank's own sum and the sharded update is the whole update: the exchange
 * step -- the all-reduce, or the barriers and the peer-pointer update kernel -- is skipped and the generation is the
 * plain one (one-rank cost of either exchange path: 0; round 4: +4.8 us through RCCL, +11 us kernel-issued).
 * RECUR_AMD_DIST_ONE_RANK_EXCHANGE=1 keeps the exchange step in (tests and `bench.py --dist` exercise it on one GPU). */
static int one_rank_exchange_forced(void) {
  static int on = -1;
  if (on < 0) {
    const char *v = getenv("RECUR_AMD_DIST_ONE_RANK_EXCHANGE");
    on = v && *v == '1';
  }
  return on;
}

/* what a one-call generation does before its update: the text step (symbol i of the resident text), or a dense-input
 * generation with rnnca's loss (rnn_amd_set_dense_step_sigmoid_mse) */
typedef struct {
  int i;                /* text position, or -1: dense, or -2: the multi-head step */
  const float *inputs;  /* dense: host [n][ld_inputs] */
  int ld_inputs;
  const float *targets; /* host [n][ld], the first n_targets outputs */
  int ld, n_targets;
  /* the multi-head step (rnn_amd_set_multi_step_deltas's arguments) */
  const int *hot, *next, *target_class;
  int alphabet_len;
  float leakage;
} StepSpec;

static void multi_step_deltas(RnnAmdSet *set, const int *hot, const int *next, const int *target_class, int alphabet_len,
                              float leakage, int accumulate, RamdPendingDelta *defer);
static void step_deltas(RnnAmdSet *set, const StepSpec *sp, RamdPendingDelta *defer) {
  if (sp->i >= 0) {
    char_step_deltas(set, sp->i, defer);
  } else if (sp->i == -2) {
    multi_step_deltas(set, sp->hot, sp->next, sp->target_class, sp->alphabet_len, sp->leakage, 0, defer);
  } else {
    rnn_amd_set_opinion_sigmoid_mse(set, sp->inputs, sp->ld_inputs, sp->targets, sp->ld, sp->n_targets);
    set_calc_deltas(set, 0, NULL, NULL, 0, NULL, 0, defer);
  }
}

static void set_step(RnnAmdSet *set, const StepSpec *sp, int learning_style, float momentum) {
  if (set->eng->xchg_world > 1 || (set->eng->xchg_world == 1 && one_rank_exchange_forced())) {
    /* deltas -> (barrier) -> sharded update with the sum over the ranks in it -> (barrier) */
    step_deltas(set, sp, NULL);
    rnn_amd_set_apply_exchange(set, learning_style, momentum);
    return;
  }
  /* the deltas go straight from the GEMM's K slabs into the update (and into ih_delta) when
   * nothing can look at them in between: library-owned storage, no log on the prototype */
  RamdPendingDelta pend = {0};
  const int dist = ramd_dist_active() && (rnn_amd_dist_world() > 1 || one_rank_exchange_forced());
  int fuse = !set->eng->delta_external && !set->nets[0]->log && !dist;
  if (dist) {
    g_halves_seen = 0;
    ramd_set_delta_half_hook(delta_half_ready, set->eng);
  }
  if (fuse && !set->eng->sh.bI && learning_style == RNN_ADAGRAD) {
    /* ADAGRAD (recur-nn.c:518-524, the multi-head trainer's rule): the same epilogue with the other arithmetic (round 6) */
    const RecurNNBPTT *bptt = set->nets[0]->bptt;
    engine_need_dev(set->eng, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS);
    pend.fuse_want = 1;
    pend.fuse_method = 4;
    pend.fuse_rate = bptt->learn_rate;
    pend.fuse_ho_rate = bptt->learn_rate * bptt->ho_scale;
  } else if (fuse && !set->eng->sh.bI &&
      (learning_style == RNN_MOMENTUM_WEIGHTED || learning_style == RNN_MOMENTUM_SIMPLIFIED_NESTEROV ||
       learning_style == RNN_MOMENTUM_CLASSICAL || learning_style >= RNN_LAST_LEARNING_METHOD || learning_style < 0)) {
    /* the momentum rule (recur-nn.c:482-487, 653-676): the weight-delta GEMM may carry the update out in its own
     * epilogue (kernels_bptt.hip: k_delta_direct), with the rates and weights apply_learning would use */
    const RecurNNBPTT *bptt = set->nets[0]->bptt;
    engine_need_dev(set->eng, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS);
    pend.fuse_want = 1;
    pend.fuse_rate = bptt->learn_rate;
    pend.fuse_ho_rate = bptt->learn_rate * bptt->ho_scale;
    pend.fuse_momentum = momentum;
    pend.fuse_mw = learning_style == RNN_MOMENTUM_SIMPLIFIED_NESTEROV ? (float)(momentum / (1.0 + momentum))
                   : learning_style == RNN_MOMENTUM_CLASSICAL         ? 1.0f
                                                                       : bptt->momentum_weight;
  }
  step_deltas(set, sp, fuse ? &pend : NULL);
  if (pend.fuse_done) { /* weights and momentum are updated, the delta arrays hold the sums */
    engine_dev_wrote(set->eng, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
    return;
  }
  if (dist) {
    /* the one exchange step of the path: this rank's deltas become the sum over all ranks'
     * streams (recur-nn.c:724-739 distributed), then the identical update everywhere */
    ramd_set_delta_half_hook(NULL, NULL);
    if (g_halves_seen == 3) { /* both halves are on their way: the update waits for them */
      HIP_OK(hipStreamWaitEvent(g_stream, g_half_summed[0], 0));
      HIP_OK(hipStreamWaitEvent(g_stream, g_half_summed[1], 0));
    } else { /* a shape the two-halves form does not take: one all-reduce behind the deltas */
      const int tev = ramd_timing_begin(g_stream, RAMD_T_XCHG);
      rnn_amd_dist_all_reduce(set->eng->b.ih_delta, set->eng->ih_size + set->eng->ho_size);
      ramd_timing_end(g_stream, tev);
    }
  }
  apply_learning(set->nets[0], learning_style, momentum, (pend.slab || pend.ho_slab) ? &pend : NULL);
}

/* the multi-head generation as ONE call: rnn_amd_set_multi_step_deltas (accumulate 0) + rnn_apply_learning -- knowing the
 * update that follows, the weight-delta GEMM carries it out in its own epilogue where the rule is ADAGRAD (the trainer's)
 * or the momentum rule: no optimiser launch (17 us of configs[3]'s 340) */
void rnn_amd_set_multi_step(RnnAmdSet *set, const int *hot, const int *next, const int *target_class, int alphabet_len,
                            float leakage, int learning_style, float momentum) {
  StepSpec sp = {-2, NULL, 0, NULL, 0, 0, hot, next, target_class, alphabet_len, leakage};
  set_step(set, &sp, learning_style, momentum);
}

void rnn_amd_set_char_step(RnnAmdSet *set, int i, int learning_style, float momentum) {
  StepSpec sp = {i, NULL, 0, NULL, 0, 0, NULL, NULL, NULL, 0, 0.0f};
  set_step(set, &sp, learning_style, momentum);
}

/* rnnca's maybe_learn for every trainer (gstrnnca.c:693-740) as one call: see recur_amd.h */
void rnn_amd_set_dense_step_sigmoid_mse(RnnAmdSet *set, const float *inputs, int ld_inputs, const float *targets, int ld,
                                        int n, int learning_style, float momentum) {
  if (!inputs) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_dense_step_sigmoid_mse needs the inputs\n");
    abort();
  }
  rnn_bptt_clear_deltas(set->nets[0]); /* (lazy: the delta call below simply does not accumulate) */
  StepSpec sp = {-1, inputs, ld_inputs, targets, ld, n, NULL, NULL, NULL, 0, 0.0f};
  set_step(set, &sp, learning_style, momentum);
}

/* The single-net text step of rnn_char_epoch (charmodel-predict.c:312-321) without a host round trip per
 * symbol: rnn_bptt_advance, one_hot_opinion of text[i], net_error_bptt's loss against text[i + 1] (on the
 * device, into the set's statistics) and rnn_bptt_calculate(net, batch_size) -- the top layer updated at once
 * with bptt->momentum, the recurrent layer every batch_size generations (recur-nn.c:919-1019).  For a set of
 * ONE net; the caller sets bptt->momentum as the reference's loop does. */
void rnn_amd_set_char_step_fused(RnnAmdSet *set, int i, unsigned batch_size) {
  RamdEngine *e = set->eng;
  set_need_training(set, "rnn_amd_set_char_step_fused");
  if (set->n != 1) {
    fprintf(stderr, "librecur_amd: rnn_amd_set_char_step_fused is the single-net path (set of %d)\n", set->n);
    abort();
  }
  check_text_pos(e, i, 0, "rnn_amd_set_char_step_fused");
  RecurNN *net = set->nets[0];
  RecurNNBPTT *bptt = net->bptt;
  const RamdShape *s = &e->sh;
  const int j = set->row0;
  const int batched = batch_size > 1;
  unsigned top_done = 0;
  if (ramd_text_top_ok(s)) { /* advance + hidden layer, then output layer, loss and top backprop in one launch */
    int fwd_ks = set_forward(set, RAMD_IN_TEXT, NULL, 0, i, NULL, 1, 1);
    ramd_launch_text_top(g_stream, s, &e->b, j, 1, fwd_ks);
    top_done = RAMD_TOP_DONE;
  } else {
    set_forward(set, RAMD_IN_TEXT, NULL, 0, i, NULL, 1, 0); /* advance + one-hot opinion */
    ramd_launch_softmax_error(g_stream, s, &e->b, j, 1);
  }
  int accumulate = batched;
  deltas_materialize(e); /* a pending rnn_bptt_clear_deltas: this path never writes ho_delta (see calc_deltas_one) */
  engine_need_dev(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  push_learn_rates(e, j, 1);
  if (e->err_pending && (e->err_row0 != j || e->err_nrows != 1)) {
    err_flush(e);
  }
  set_uniform_idx(e, j, 1);
  ramd_launch_calc_deltas(g_stream, s, &e->b, j, 1, accumulate, NULL, 0, NULL,
                          net->flags | 0x80000000u | (batched ? 0 : 0x20000000u) | top_done, NULL);
  if (ramd_calc_wrote_images()) {
    e->err_pending = 0;
  } else {
    e->err_pending = 1;
    e->err_row0 = j;
    e->err_nrows = 1;
  }
  net->generation++;
  const int due = !batched || ((net->generation - 1) % batch_size) == 0;
  ramd_launch_fused_updates(g_stream, s, &e->b, j, bptt->learn_rate, bptt->momentum, bptt->momentum_weight, due,
                            batched ? NULL : e->b.ih_scale + j);
  if (due && batched) {
    HIP_OK(hipMemsetAsync(e->b.ih_delta, 0, e->ih_size * sizeof(float), g_stream));
  }
  engine_dev_wrote(e, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  set_streams_dev_wrote(set);
  rnn_condition_net(net);
}

void rnn_amd_set_read_stats(RnnAmdSet *set, RnnAmdStats *stats, int clear) {
  RamdEngine *e = set->eng;
  set_need_training(set, "rnn_amd_set_read_stats");
  int n = set->n, r0 = set->row0;
  double *d = malloc(4 * n * sizeof(double));
  long long *l = malloc(2 * n * sizeof(long long));
  d2h(d, e->b.stat_err + r0, n * sizeof(double));
  d2h(d + n, e->b.stat_ent + r0, n * sizeof(double));
  d2h(d + 2 * n, e->b.stat_zero + r0, n * sizeof(double));
  d2h(d + 3 * n, e->b.stat_depth + r0, n * sizeof(double));
  d2h(l, e->b.stat_correct + r0, n * sizeof(long long));
  d2h(l + n, e->b.stat_count + r0, n * sizeof(long long));
  dsync();
  memset(stats, 0, sizeof(*stats));
  for (int j = 0; j < n; j++) {
    stats->error += d[j];
    stats->entropy += d[n + j];
    stats->hidden_zeros += d[2 * n + j];
    stats->bptt_depth_sum += d[3 * n + j];
    stats->correct += l[j];
    stats->count += l[n + j];
  }
  free(d);
  free(l);
  if (clear) {
    HIP_OK(hipMemsetAsync(e->b.stat_err + r0, 0, n * sizeof(double), g_stream));
    HIP_OK(hipMemsetAsync(e->b.stat_ent + r0, 0, n * sizeof(double), g_stream));
    HIP_OK(hipMemsetAsync(e->b.stat_zero + r0, 0, n * sizeof(double), g_stream));
    HIP_OK(hipMemsetAsync(e->b.stat_depth + r0, 0, n * sizeof(double), g_stream));
    HIP_OK(hipMemsetAsync(e->b.stat_correct + r0, 0, n * sizeof(long long), g_stream));
    HIP_OK(hipMemsetAsync(e->b.stat_count + r0, 0, n * sizeof(long long), g_stream));
  }
}

void rnn_amd_set_external_delta(RnnAmdSet *set, void *device_buffer) {
  RamdEngine *e = set->eng;
  float *dst = device_buffer ? (float *)device_buffer : e->delta_own;
  deltas_materialize(e);
  if (dst != e->b.ih_delta) {
    HIP_OK(hipMemcpyAsync(dst, e->b.ih_delta, (e->ih_size + e->ho_size) * sizeof(float),
                          hipMemcpyDeviceToDevice, g_stream));
    dsync();
    e->b.ih_delta = dst;
    e->b.ho_delta = dst + e->ih_size;
  }
  e->delta_external = device_buffer != NULL;
}
