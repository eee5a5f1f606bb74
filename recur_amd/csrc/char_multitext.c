/* char_multitext.c -- the multi-head text trainer's host side (gnu11 C; charmodel.h:132-152,
 * 242-265): ONE net whose output row is n_classes heads of alphabet_len symbols each, trained on one
 * class's text at a time with temporal batching.  It is the backend of py-recur-text.c's Net.train /
 * Net.test (py-recur-text.c:759-871) and the caller BASELINE.json configs[3] is quoted on.
 *
 * The reference walks the text with per-symbol host work (charmodel-multi-predict.c:234-301).  Here a
 * pass opens a one-net set with the text resident in HBM and issues, per symbol, the device calls
 * that stand for "advance, opinion, multi-head loss with its leakage draws" and "deltas over the
 * loss's error ranges"; the statistics stay on the device until the pass ends.  The order of
 * operations per symbol -- loss, then (when a batch is full) the update BEFORE that symbol's deltas
 * start the next sum -- is the reference's, as is its habit of applying the NET's momentum rather
 * than the caller's (line 247).  Parity: tests/test_gpu_callers.py, tools/gpu_stress_multitext.py.
 */
#include "char_host.h"

/* a pass's view of one net as a set of one, text loaded; NULL for a net without its own bptt */
static RnnAmdSet *open_text_pass(RecurNN *net, const u8 *text, int len, const char *who) {
  RecurNN *one[1] = {net};
  RnnAmdSet *set = rnn_amd_set_open(one, 1);
  if (!set) {
    fprintf(stderr, "librecur_amd: %s needs a net with its own bptt\n", who);
    abort();
  }
  rnn_amd_set_load_text(set, text, len);
  return set;
}

/* image rows and the periodic weight pictures a caller may have asked for: every `period` symbols,
 * counted across calls by the net's generation */
typedef struct {
  TemporalPPM *input_ppm, *error_ppm;
  const char *pgm_string;
  int period, countdown;
} Pictures;

static Pictures pictures_for(RecurNN *net, TemporalPPM *in, TemporalPPM *err, const char *pgm_string, int period) {
  Pictures p = {in, err, pgm_string, period, 0};
  if (period) {
    p.countdown = period - net->generation % period;
  }
  return p;
}

static void pictures_step(Pictures *p, RecurNN *net) {
  ramd_image_rows(net, p->input_ppm, p->error_ppm);
  if (p->period && --p->countdown == 0) {
    p->countdown = p->period;
    rnn_multi_pgm_dump(net, p->pgm_string, "multi-text");
  }
}

static int pictures_wanted(const Pictures *p) { return p->input_ppm || p->error_ppm || p->period; }

/* ------------------------------------------------------------ sample prose -- */

RnnCharMultiConfab *rnn_char_new_multi_confab(RecurNN *net, RnnCharAlphabet *alphabet, int n_classes,
                                              int target_len, uint confab_period, int caps_marker) {
  /* the heads share one output line of target_len characters, a separator after each */
  const int each = target_len / n_classes - 1;
  if (each < 1) {
    fprintf(stderr, "no room to confabulate %d sub-models in %d characters\n", n_classes, target_len);
    return NULL;
  }
  RnnCharMultiConfab *mc = calloc(1, sizeof(*mc));
  mc->n_classes = n_classes;
  mc->alphabet = alphabet;
  mc->caps_marker = caps_marker;
  mc->period = confab_period;
  mc->char_len = each;
  mc->byte_len = 6 * each + 1;
  mc->nets = calloc(n_classes, sizeof(*mc->nets));
  mc->strings = calloc(n_classes, sizeof(*mc->strings));
  mc->last_char = calloc(n_classes, sizeof(*mc->last_char));
  /* every head writes with a forward-only clone of its own: the heads' hidden states do not mix */
  const u32 borrow = net->flags & ~(RNN_NET_FLAG_OWN_BPTT | RNN_NET_FLAG_OWN_WEIGHTS);
  for (int c = 0; c < n_classes; c++) {
    mc->nets[c] = rnn_clone(net, borrow, RECUR_RNG_SUBSEED, NULL);
    mc->strings[c] = calloc(mc->byte_len, 1);
  }
  return mc;
}

void rnn_char_free_multi_confab(RnnCharMultiConfab *mc) {
  for (uint c = 0; c < mc->n_classes; c++) {
    rnn_delete_net(mc->nets[c]);
    free(mc->strings[c]);
  }
  free(mc->nets);
  free(mc->strings);
  free(mc->last_char);
  free(mc);
}

/* upper case of a lower-case letter as far as the reference takes it: ascii and Greek, where the
 * word-final sigma (962 -> 930, not a letter) becomes a plain capital sigma */
static int capital_of(int c) {
  if (c >= 'a' && c <= 'z') {
    return c - 'a' + 'A';
  }
  if (c >= 945 && c <= 969) {
    return c == 962 ? 931 : c - 32;
  }
  return c;
}

/* char_len characters from every head into its string (charmodel-multi-predict.c:145-197).  The caps
 * marker is a symbol of the alphabet that stands for "the next letter is a capital"; it is not
 * written and does not count. */
static void sample_heads(RnnCharMultiConfab *mc) {
  const int alen = mc->alphabet->len;
  const int utf8 = (mc->alphabet->flags & RNN_CHAR_FLAG_UTF8) != 0;
  const int widest = utf8 ? 5 : 1;
  if ((int)mc->byte_len <= widest) {
    fprintf(stderr, "insufficient space to confabulate (%d bytes)\n", mc->byte_len);
    return;
  }
  float *work = malloc(sizeof(float) * alen);
  for (uint c = 0; c < mc->n_classes; c++) {
    char *out = mc->strings[c];
    int room = mc->byte_len, sym = mc->last_char[c], capital_next = 0;
    for (uint written = 0; written < mc->char_len && room > widest;) {
      sym = ramd_next_symbol(mc->nets[c], sym, mc->bias, (int)c, alen, 0, work);
      int point = mc->alphabet->points[sym];
      if (point == mc->caps_marker) {
        capital_next = 1;
        continue;
      }
      if (capital_next) {
        point = capital_of(point);
        capital_next = 0;
      }
      const int w = ramd_put_codepoint(point, out, utf8);
      out += w;
      room -= w;
      written++;
    }
    *out = 0;
    mc->last_char[c] = sym;
  }
  free(work);
}

/* the heads' strings side by side, `sep` after each (the last one too: charmodel-multi-predict.c:200-230
 * closes the line that way), at most len - 1 bytes */
static void join_heads(const RnnCharMultiConfab *mc, char *dest, int len, const char *sep) {
  int used = 0;
  for (uint c = 0; c < mc->n_classes && used < len - 1; c++) {
    const char *parts[2] = {mc->strings[c], sep};
    for (int k = 0; k < 2; k++) {
      for (const char *s = parts[k]; *s && used < len - 1; s++) {
        dest[used++] = *s;
      }
    }
  }
  dest[used] = 0;
}

#define TINT_OFF "\033[00m"
#define TINT_VAL "\033[00;36m"

static void print_heads(RnnCharMultiConfab *mc, const RecurNN *net) {
  char *line = malloc(mc->byte_len + 1);
  sample_heads(mc);
  join_heads(mc, line, mc->byte_len, TINT_VAL "|" TINT_OFF);
  printf("%8u" TINT_VAL "|" TINT_OFF "%s\n", net->generation, line);
  free(line);
}

/* ------------------------------------------------------------------ training -- */

void rnn_char_multitext_train(RecurNN *net, u8 *text, int len, int alphabet_len, int target_class,
                              float leakage, RnnCharProgressReport *report,
                              RnnCharMultiConfab *confab, int learning_style, float momentum,
                              int batch_size, TemporalPPM *input_ppm, TemporalPPM *error_ppm,
                              const char *periodic_pgm_string, int periodic_pgm_period) {
  (void)momentum; /* the reference's loop applies bptt->momentum (charmodel-multi-predict.c:247) */
  Pictures pics = pictures_for(net, input_ppm, error_ppm, periodic_pgm_string, periodic_pgm_period);
  struct timespec clock;
  clock_gettime(CLOCK_MONOTONIC, &clock);
  const int batch = RAMD_MAX(batch_size, 1);
  int steps = 0;
  if (len >= 2) {
    RnnAmdSet *pass = open_text_pass(net, text, len, "rnn_char_multitext_train");
    RnnAmdStats st;
    rnn_amd_set_read_stats(pass, &st, 1); /* the pass's own counters */
    /* batches are counted by the net's generation, so they run on across the texts of a session */
    int until_update = batch - net->generation % batch;
    const int draw = pictures_wanted(&pics);
    for (; steps < len - 1; steps++, until_update--) {
      /* advance + opinion + the loss of head target_class (always) and of the other heads (each with
       * probability `leakage`, decided by the net's generator on the device) + the merged error ranges */
      rnn_amd_set_multi_text_loss(pass, steps, steps == 0 ? &target_class : NULL, alphabet_len, leakage);
      const int full = until_update == 0;
      if (full) {
        rnn_apply_learning(net, learning_style, net->bptt->momentum);
        until_update = batch;
      }
      rnn_amd_set_multi_calc_deltas(pass, !full); /* a fresh sum after an update, otherwise add */
      if (draw) {
        pictures_step(&pics, net);
      }
    }
    if (report) {
      rnn_amd_set_read_stats(pass, &st, 1);
      report->training_entropy = (float)(-st.entropy) / (len - 1);
      report->training_error = (float)st.error / (len - 1);
    }
    rnn_amd_synchronize();
    rnn_amd_set_drop(pass); /* the net's state stays where it is, on the device */
  }
  if (confab && confab->period && steps % confab->period == 0) {
    print_heads(confab, net);
  }
  if (report) {
    report->per_second = (len - 1) / ramd_seconds_since(&clock);
  }
}

/* charmodel.h:251-254; charmodel-multi-predict.c:283-301: the text through the net without training
 * (advance + opinion with the net's own noise), e.g. to settle the hidden state before a test */
void rnn_char_multitext_spin(RecurNN *net, u8 *text, int len, TemporalPPM *input_ppm,
                             TemporalPPM *error_ppm, const char *periodic_pgm_string,
                             int periodic_pgm_period) {
  if (len < 1) {
    return;
  }
  Pictures pics = pictures_for(net, input_ppm, error_ppm, periodic_pgm_string, periodic_pgm_period);
  if (error_ppm) { /* nothing is learnt here: the error rows of the picture are blank */
    rnn_amd_sync_host(net, RNN_AMD_STREAM);
    memset(net->bptt->o_error, 0, sizeof(float) * net->output_size);
    rnn_amd_host_written(net, RNN_AMD_STREAM);
  }
  /* Every one of the len symbols is an input here, the last one too.  The device's text steps take
   * positions modulo len - 1 (a TRAINING step's input always has a successor), so the resident copy
   * gets one more symbol and position len - 1 is an input, not a wrap to position 0. */
  u8 *resident = malloc((size_t)len + 1);
  memcpy(resident, text, len);
  resident[len] = text[len - 1];
  RnnAmdSet *pass = open_text_pass(net, resident, len + 1, "rnn_char_multitext_spin");
  free(resident);
  const int draw = pictures_wanted(&pics);
  for (int i = 0; i < len; i++) {
    rnn_amd_set_text_opinion(pass, i, 1);
    if (draw) {
      pictures_step(&pics, net);
    }
  }
  rnn_amd_synchronize();
  rnn_amd_set_drop(pass);
}

/* charmodel.h:255-257; charmodel-multi-predict.c:383-408: per head, the mean bits per symbol of the
 * text.  The caller's array is subtracted into and then divided, as the reference treats it. */
void rnn_char_multi_cross_entropy(RecurNN *net, const u8 *text, int len, int alphabet_len,
                                  double *entropy, int ignore_start) {
  const int heads = net->output_size / alphabet_len;
  double *log2_sums = malloc(sizeof(double) * RAMD_MAX(heads, 1));
  rnn_amd_run_text_heads(net, text, len, ignore_start, alphabet_len, log2_sums);
  for (int h = 0; h < heads; h++) {
    entropy[h] = (entropy[h] - log2_sums[h]) / (len - ignore_start - 1);
  }
  free(log2_sums);
}
