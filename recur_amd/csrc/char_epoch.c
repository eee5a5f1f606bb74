/* char_epoch.c -- one pass of a character model over its training text (gnu11 C): the caller
 * side of the text hot path, so that the reference's text tools keep working when they link
 * librecur_amd instead of recur-nn.o + charmodel-predict.o (charmodel.h:185-204, 234-239).
 *
 * The reference's epoch (charmodel-predict.c:260-405) is one loop with the per-stream work, the
 * statistics and the periodic report written inline.  Here it is three pieces:
 *   a ROUTE that trains one generation -- the whole training set per launch sequence on the device
 *     (rnn_amd_set_char_step), one net on the fused single-net path (rnn_amd_set_char_step_fused),
 *     or, for nets that are not one training set or that log per call, the per-net calls with the
 *     loss taken on the host;
 *   a TALLY of error / entropy / hits, which the device routes keep in HBM until a report asks;
 *   a REPORT every report_interval generations: validation entropy, a console line with a sample of
 *     the net's own prose, the seven log values, net file, images, schedule, weight noise.
 * What a route computes per generation, what a report logs (names and arithmetic) and when the
 * schedule cuts the learn rate are the reference's, pinned by tests/test_char_predict_gpu.py (the
 * erewhon curve through this function and the net's log) and tests/test_charmodel.py.
 */
#include "char_host.h"

/* ------------------------------------------------------------ cross entropy -- */

/* mean bits per symbol of text[skip + 1 ..] given what came before (charmodel-predict.c:62-80):
 * rnn_amd_run_text sums log2 p on the device, one forward pass per symbol, no host visit */
static double bits_per_symbol(RecurNN *net, const u8 *text, int len, int skip) {
  return rnn_amd_run_text(net, text, len, skip) / -(double)(len - skip - 1);
}

int rnn_char_prime(RecurNN *net, RnnCharAlphabet *alphabet, const u8 *text, const int len) {
  (void)alphabet;
  if (!text || !len) {
    return 0;
  }
  rnn_amd_run_text(net, text, len, len); /* every symbol fed, none scored */
  return text[len - 1];
}

double rnn_char_cross_entropy(RecurNN *net, RnnCharAlphabet *alphabet, const u8 *text,
                              const int len, const int skip, const u8 *prefix_text,
                              const int prefix_len) {
  if (prefix_text) {
    rnn_char_prime(net, alphabet, prefix_text, prefix_len);
  }
  return bits_per_symbol(net, text, len, skip);
}

/* ------------------------------------------------------- validation entropy -- */

void rnn_char_init_ventropy(RnnCharVentropy *v, RecurNN *net, const u8 *text, const int len,
                            const int lap) {
  *v = (RnnCharVentropy){.net = net, .text = text, .len = len, .lap = lap, .lapsize = len / lap,
                         .history = calloc(lap, sizeof(float))};
}

void rnn_char_delete_ventropy(RnnCharVentropy *v) { free(v->history); }

/* The validation text is either scored whole, or (lap > 1 and the caller says so) one lap-th of it
 * per call, round robin, the figure being the mean over the laps scored so far
 * (charmodel-predict.c:227-258).  The first symbols of a passage are not scored: a tenth of it, five
 * at most. */
float rnn_char_calc_ventropy(RnnCharModel *model, RnnCharVentropy *v, int lap) {
  (void)model;
  if (v->len <= 0) {
    return v->entropy;
  }
  if (!(lap && v->lap > 1)) {
    v->entropy = bits_per_symbol(v->net, v->text, v->len, RAMD_MIN(v->len / 10, 5));
    v->history[0] = v->entropy;
    return v->entropy;
  }
  v->counter = (v->counter + 1) % v->lap;
  v->history[v->counter] = bits_per_symbol(v->net, v->text + (size_t)v->lapsize * v->counter, v->lapsize,
                                           RAMD_MIN(v->lapsize / 10, 5));
  float total = 0.0f;
  int scored = 0;
  for (int j = 0; j < v->lap; j++) { /* a slot still at zero has not had its turn */
    total += v->history[j];
    scored += v->history[j] != 0;
  }
  v->entropy = scored ? total / scored : 0;
  return v->entropy;
}

/* ------------------------------------------------------------------ schedule -- */

/* The learn-rate schedule keeps a pool of recent validation scores.  Each new score overwrites a
 * RANDOM slot (a draw from the net's generator), and is then compared with the third of the pool that
 * follows that slot: a score that beats none of them ends a plateau -- the learn rate (and, on request,
 * the noise levels) shrink by learn_rate_mul, and no further cut is considered for recent_len reports
 * (charmodel-predict.c:82-118). */
static int beats_none_after(const RnnCharSchedule *s, int slot, float score) {
  const int window = s->recent_len / 3;
  for (int j = 1; j <= window; j++) {
    if (score < s->recent[(slot + j) % s->recent_len]) {
      return 0;
    }
  }
  return 1;
}

static void plateau_schedule(RnnCharModel *model, float score, int verbose) {
  RnnCharSchedule *s = &model->schedule;
  RecurNN *net = model->net;
  if (!s->recent_len || net->bptt->learn_rate <= s->learn_rate_min) {
    return;
  }
  ramd_rng_to_host(net); /* the draw continues the net's stream wherever the device left it */
  const int slot = ramd_rand_small_int(&net->rng, s->recent_len);
  ramd_rng_from_host(net);
  s->recent[slot] = score;
  if (s->timeout) {
    s->timeout--;
    return;
  }
  if (!beats_none_after(s, slot, score)) {
    return;
  }
  /* a plateau: shrink the rate (not below its floor), and the noise levels with it if asked to */
  const float shrink = s->learn_rate_mul;
  float *rate = &net->bptt->learn_rate;
  *rate = RAMD_MAX(*rate * shrink, s->learn_rate_min);
  s->timeout = s->recent_len;
  if (s->adjust_noise) {
    model->periodic_weight_noise *= shrink;
    net->presynaptic_noise *= shrink;
  }
  if (verbose) {
    fprintf(stderr, "generation %d: validation entropy %.4g is no better than %d recent scores: learn rate now %.3g "
                    "(momentum %.3g)\n",
            (int)net->generation, score, s->recent_len / 3, *rate, net->bptt->momentum);
  }
}

void rnn_char_init_schedule(RnnCharSchedule *s, int recent_len, float learn_rate_min,
                            float learn_rate_mul, int adjust_noise) {
  s->eval = plateau_schedule;
  s->adjust_noise = adjust_noise;
  s->timeout = s->recent_len = recent_len;
  if (recent_len) {
    s->learn_rate_mul = learn_rate_mul, s->learn_rate_min = learn_rate_min;
    s->recent = ramd_zalloc(sizeof(float) * recent_len);
    for (int i = 0; i < recent_len; i++) {
      s->recent[i] = 1e10; /* nothing beats an empty slot */
    }
  }
}

/* --------------------------------------------------------------------- epoch -- */

typedef struct {
  float error, entropy; /* sums over stream-steps: target-class error, capped log2(1 - error) */
  int hits;             /* stream-steps whose best guess was the target */
} Tally;

enum Route {
  ROUTE_SET,         /* the training set, one batched generation per call            */
  ROUTE_SET_FUSED,   /* one net, rnn_bptt_calculate's path with the loss on the device */
  ROUTE_NETS,        /* per-net calls over the training nets, loss on the host         */
  ROUTE_NET_FUSED    /* per-net calls on the one net (it has a log file)               */
};

/* softmax error of one stream-step on the host (charmodel-predict.c:18-27 over badmaths.h:113-141):
 * o_error = -softmax(output), + 1 at the target; the first of equal maxima is the guess */
static void host_loss(RecurNN *net, int sym, int target, Tally *t) {
  const float *out = ramd_feed_symbol(net, sym, net->presynaptic_noise);
  float *err = net->bptt->o_error;
  const int n = net->output_size;
  ramd_text_distribution(err, out, n, 0);
  int guess = 0;
  for (int i = 1; i < n; i++) {
    if (err[i] > err[guess]) {
      guess = i;
    }
  }
  for (int i = 0; i < n; i++) {
    err[i] = -err[i];
  }
  err[target] += 1.0f;
  t->hits += guess == target;
  t->error += err[target];
  t->entropy += ramd_capped_log2f(1.0f - err[target]);
}

typedef struct {
  RnnCharModel *model;
  enum Route route;
  RnnAmdSet *set; /* the device routes */
  const u8 *text;
  int len;
  Tally tally; /* the host routes */
} Epoch;

static void open_set(Epoch *ep, RecurNN **nets, int n) {
  ep->set = rnn_amd_set_open(nets, n);
  if (ep->set) { /* the text travels to the device once per epoch; its counters start at zero */
    RnnAmdStats unused;
    rnn_amd_set_load_text(ep->set, ep->text, ep->len);
    rnn_amd_set_read_stats(ep->set, &unused, 1);
  }
}

static void choose_route(Epoch *ep) {
  RnnCharModel *m = ep->model;
  RecurNN *net = m->net;
  /* charmodel-predict.c:291: several nets, another optimiser or the explicit switch take the
   * advance / opinion / calc_deltas ... apply_learning form, everything else rnn_bptt_calculate */
  const int multi_tap = m->n_training_nets > 1 || m->learning_style != RNN_MOMENTUM_WEIGHTED || m->use_multi_tap_path;
  ep->set = NULL;
  if (multi_tap) {
    ep->route = ROUTE_NETS;
    if (m->training_nets && m->training_nets[0] == net) { /* one rnn_new_training_set: the batched calls apply */
      open_set(ep, m->training_nets, m->n_training_nets);
      if (ep->set) {
        ep->route = ROUTE_SET;
      }
    }
  } else {
    ep->route = ROUTE_NET_FUSED;
    if (!net->log && net->bptt) { /* (a net with a log file wants the core's per-call log lines) */
      RecurNN *one[1] = {net};
      open_set(ep, one, 1);
      if (ep->set) {
        ep->route = ROUTE_SET_FUSED;
      }
    }
  }
}

/* generation i of the epoch: stream j reads the text at i + j * spacing, wrapped (charmodel-predict.c:295-298) */
static void train_generation(Epoch *ep, int i, float momentum) {
  RnnCharModel *m = ep->model;
  RecurNN *net = m->net;
  switch (ep->route) {
  case ROUTE_SET:
    rnn_amd_set_char_step(ep->set, i, m->learning_style, momentum);
    break;
  case ROUTE_SET_FUSED:
    net->bptt->momentum = momentum;
    rnn_amd_set_char_step_fused(ep->set, i, m->batch_size);
    break;
  case ROUTE_NETS: {
    const int last = ep->len - 1, spacing = last / m->n_training_nets;
    for (int j = 0; j < m->n_training_nets; j++) {
      RecurNN *stream = m->training_nets[j];
      const int at = (i + j * spacing) % last;
      rnn_bptt_advance(stream);
      host_loss(stream, ep->text[at], ep->text[at + 1], &ep->tally);
      rnn_bptt_calc_deltas(stream, j > 0, NULL); /* the first stream starts the sum, the others add */
    }
    rnn_apply_learning(net, m->learning_style, momentum);
    break;
  }
  case ROUTE_NET_FUSED:
    net->bptt->momentum = momentum;
    rnn_bptt_advance(net);
    host_loss(net, ep->text[i], ep->text[i + 1], &ep->tally);
    rnn_bptt_calculate(net, m->batch_size);
    break;
  }
}

/* what has been tallied since the last report (the device routes keep it in HBM); clears it */
static Tally collect_tally(Epoch *ep) {
  Tally t = ep->tally;
  memset(&ep->tally, 0, sizeof(ep->tally));
  if (ep->set) {
    RnnAmdStats st;
    rnn_amd_set_read_stats(ep->set, &st, 1); /* waits for the device */
    t.error = (float)st.error;
    t.entropy = (float)st.entropy;
    t.hits = (int)st.correct;
  } else {
    rnn_amd_synchronize();
  }
  return t;
}

#define TINT_OFF "\033[00m"
#define TINT_DIM "\033[00;37m"
#define TINT_VAL "\033[00;36m"
#define TINT_WARN "\033[01;33m"

/* the console line of a report: the figures, then a sample of the net's prose -- either one line of it
 * up to the line-end symbol, or a fixed number of symbols */
static void console_line(RnnCharModel *m, RecurNN *confab_net, int *confab_sym, float error, float entropy,
                         float ventropy, float accuracy, double rate, float bias, int size, int line_end) {
  RnnCharAlphabet *a = m->alphabet;
  const int kilo = m->net->generation >> 10;
  /* (the figures are the reference's, the layout is this library's own) */
  fprintf(stderr, TINT_DIM "[%uk]" TINT_OFF " train %.3f bits " TINT_VAL "valid %.3f" TINT_OFF " | err %.0f%% hit %.0f%% | %.0f/s "
                  TINT_DIM ">" TINT_OFF " ",
          (unsigned)kilo, entropy, ventropy, error * 100.0, accuracy * 100.0, rate);
  if (line_end >= 0) {
    const int utf8 = (a->flags & RNN_CHAR_FLAG_UTF8) != 0;
    float *work = malloc(sizeof(float) * confab_net->output_size);
    int shown = 0, sym = *confab_sym;
    for (; shown < size; shown++) {
      sym = ramd_next_symbol(confab_net, sym, bias, 0, confab_net->output_size, 1, work);
      if (sym == line_end) {
        break;
      }
      char bytes[5];
      bytes[ramd_put_codepoint(a->points[sym], bytes, utf8)] = 0;
      fputs(bytes, stderr);
    }
    fputs(shown == size ? TINT_WARN " ..." TINT_OFF "\n" : "\n", stderr); /* (the line did not end by itself) */
    *confab_sym = sym;
    free(work);
  } else {
    char *prose = malloc(4 * (size_t)size + 1);
    rnn_char_confabulate(confab_net, prose, size, 4 * size, a, bias, confab_sym, -1, -1);
    fprintf(stderr, "%s\n", prose);
    free(prose);
  }
}

/* net file and weight pictures of a report, where the model asks for them */
static void periodic_outputs(const RnnCharModel *m) {
  const RnnCharImageSettings *im = &m->images;
  if (m->filename && m->save_net) {
    rnn_save_net(m->net, m->filename, 1);
  }
  if (im->periodic_pgm_dump_string) {
    rnn_multi_pgm_dump(m->net, im->periodic_pgm_dump_string, im->basename);
  }
}

int rnn_char_epoch(RnnCharModel *model, RecurNN *confab_net, RnnCharVentropy *v, const u8 *text,
                   const int len, const int start, const int stop, float confab_bias,
                   int confab_size, int confab_line_end, int quietness,
                   uint diagonal_only_section, uint diagonal_only_friends) {
  RecurNN *net = model->net;
  Epoch ep = {.model = model, .text = text, .len = len};
  choose_route(&ep);
  /* an epoch may begin inside a report interval: the first report averages over what is left of it */
  uint since_report = net->generation % model->report_interval;
  float per_step = 1.0f / ((model->report_interval - since_report) * model->n_training_nets);
  int confab_sym = 0, stopped = 0;
  struct timespec clock;
  clock_gettime(CLOCK_MONOTONIC, &clock);
  const uint diag = diagonal_only_section, friends = diagonal_only_friends;
  if (diag) {
    rnn_clear_diagonal_only_section(net, diag, friends);
  }
  for (int i = start; i < len - 1 && !stopped; i++) {
    train_generation(&ep, i, rnn_calculate_momentum_soft_start(net->generation, model->momentum,
                                                               model->momentum_soft_start));
    if (diag) { /* the section is cleared again after every update */
      rnn_clear_diagonal_only_section(net, diag, friends);
    }
    ramd_image_rows(net, model->images.input_ppm, model->images.error_ppm);
    if (++since_report >= model->report_interval) {
      since_report = 0;
      const Tally t = collect_tally(&ep);
      const double seconds = ramd_seconds_since(&clock);
      const float ventropy = rnn_char_calc_ventropy(model, v, 1);
      const float entropy = -per_step * t.entropy, error = per_step * t.error, accuracy = per_step * t.hits;
      const double rate = 1.0 / per_step / seconds; /* stream-steps per second */
      if (confab_net && confab_size && quietness < 1) {
        console_line(model, confab_net, &confab_sym, error, entropy, ventropy, accuracy, rate, confab_bias, confab_size,
                     confab_line_end);
      }
      /* the net's log (recur-nn.h:337-349; the reference's `plot` script reads these names) */
      const struct {
        char *name; /* (rnn_log_float's prototype, recur-nn.h:337, takes a plain char *) */
        float value;
      } logged[] = {{"t_error", error},   {"t_entropy", entropy},
                    {"v_entropy", ventropy}, {"momentum", net->bptt->momentum},
                    {"accuracy", accuracy}, {"learn-rate", net->bptt->learn_rate},
                    {"per_second", (float)rate}};
      for (size_t k = 0; k < sizeof(logged) / sizeof(logged[0]); k++) {
        rnn_log_float(net, logged[k].name, logged[k].value);
      }
      per_step = 1.0f / (model->report_interval * model->n_training_nets);
      periodic_outputs(model);
      model->schedule.eval(model, ventropy, quietness < 2);
      if (model->periodic_weight_noise) {
        rnn_weight_noise(net, model->periodic_weight_noise);
      }
    }
    stopped = stop && (int)net->generation >= stop;
  }
  /* what was tallied since the last report goes with the epoch, as the reference's locals do */
  rnn_amd_set_close(ep.set);
  return stopped;
}
