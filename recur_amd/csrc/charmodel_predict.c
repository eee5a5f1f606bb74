/* charmodel_predict.c -- the caller side of the text hot path: the epoch loop,
 * validation entropy, learn-rate schedule and confabulation of the reference's
 * charmodel-predict.c, written against this library's own public API
 * (include/recur_amd.h) so that the reference's text tools keep working when they
 * link librecur_amd instead of recur-nn.o + charmodel-predict.o.
 *
 * What differs from the reference is where the loops run.  The multi-tap branch of
 * rnn_char_epoch (charmodel-predict.c:288-311) becomes one rnn_amd_set_char_step per
 * generation with the loss statistics kept on the device, and get_cross_entropy
 * (62-80) becomes rnn_amd_run_text; both fall back to the per-net calls (still on the
 * device) when the nets are not one training set.  Host C, gnu11.
 */
#include "rnn_host.h"
#include "recur_amd_char.h"
#include <time.h>

#define C_NORMAL "\033[00m"
#define C_GREY "\033[00;37m"
#define C_CYAN "\033[00;36m"
#define C_YELLOW "\033[01;33m"

/* pgm_dump.h:227-237.  Only touched when the caller has put such an object into
 * model->images (text-predict.c:575-600 does so with pgm_dump.h's constructor). */
struct _TemporalPPM {
  float *im;
  int width;
  int height;
  int y;
  int id;
  char *basename;
  int counter;
  int mode;
  float **source;
};
enum { TEMPORAL_GREY = 0, TEMPORAL_COLOUR = 1 }; /* pgm_dump.h:222-225 */

/* rnn_dump.c */
void ramd_write_signed_ppm(const float *a, int width, int height, const char *name);
void ramd_write_abs_pgm(const float *a, int width, int height, const char *name);

/* temporal_ppm_add_row + temporal_ppm_write (pgm_dump.h:262-288) */
static void temporal_add_row(TemporalPPM *ppm, const float *row) {
  memcpy(ppm->im + (size_t)ppm->y * ppm->width, row, ppm->width * sizeof(float));
  ppm->y++;
  if (ppm->y == ppm->height) {
    char name[200];
    snprintf(name, sizeof(name), "images/%s-%d-%08d-%dx%d.ppm", ppm->basename, ppm->id,
             ppm->counter, ppm->width, ppm->height);
    if (ppm->mode == TEMPORAL_GREY) {
      ramd_write_abs_pgm(ppm->im, ppm->width, ppm->height, name);
    } else {
      ramd_write_signed_ppm(ppm->im, ppm->width, ppm->height, name);
    }
    ppm->y = 0;
    ppm->counter += ppm->height;
  }
}

/* ---------------------------------------------------------------- small maths -- */

static inline float capped_log2f(float x) { /* charmodel-helpers.h:11-14 */
  return (x < 1e-30f) ? -100.0f : log2f(x);
}

/* badmaths.h:71-111 */
static void softmax_host(float *dest, const float *src, int len) {
  float lo = src[0], hi = src[0];
  for (int i = 1; i < len; i++) {
    hi = RAMD_MAX(hi, src[i]);
    lo = RAMD_MIN(lo, src[i]);
  }
  float adj = 0.0f;
  if (hi > 50.0f) {
    adj = 50.0f - hi;
  } else if (lo < -60.0f) {
    adj = RAMD_MIN(-60.0f - lo, 50.0f - hi);
  }
  float sum = 0.0f;
  for (int i = 0; i < len; i++) {
    float x = ramd_fast_expf(src[i] + adj);
    sum += x;
    dest[i] = x;
  }
  for (int i = 0; i < len; i++) {
    dest[i] /= sum;
  }
}

/* badmaths.h:113-141: writes -softmax into `error`, returns the arg max */
static int softmax_best_guess_host(float *error, const float *src, int len) {
  softmax_host(error, src, len);
  int best = 0;
  float best_e = error[0];
  error[0] = -best_e;
  for (int i = 1; i < len; i++) {
    float e = error[i];
    if (e > best_e) {
      best_e = e;
      best = i;
    }
    error[i] = -e;
  }
  return best;
}

/* badmaths.h:143-156 */
static void biased_softmax_host(float *dest, const float *src, int len, float bias) {
  if (bias == 0) {
    softmax_host(dest, src, len);
    return;
  }
  float *tmp = malloc(len * sizeof(float));
  softmax_host(tmp, src, len);
  for (int i = 0; i < len; i++) {
    tmp[i] = tmp[i] * bias + src[i];
  }
  softmax_host(dest, tmp, len);
  free(tmp);
}

/* charmodel-helpers.h:16-33, including its bottom-layer branch that indexes the
 * layer's inputs from the bias slot */
static float *one_hot_opinion(RecurNN *net, int hot, float presynaptic_noise) {
  float *inputs;
  int len;
  if (net->bottom_layer) {
    inputs = net->bottom_layer->inputs;
    len = net->bottom_layer->input_size;
  } else {
    inputs = net->real_inputs;
    len = net->input_size;
  }
  memset(inputs, 0, len * sizeof(float));
  inputs[hot] = 1.0f;
  return rnn_opinion(net, NULL, presynaptic_noise);
}

/* charmodel-predict.c:18-27 */
static float net_error_bptt(RecurNN *net, float *error, int c, int next, int *correct) {
  float *answer = one_hot_opinion(net, c, net->presynaptic_noise);
  int winner = softmax_best_guess_host(error, answer, net->output_size);
  *correct = (winner == next);
  error[next] += 1.0f;
  return error[next];
}

/* charmodel-predict.c:29-60: sample (or, for a huge bias, pick) the next symbol */
static int guess_next_character(RecurNN *net, int hot, float bias) {
  float *answer = one_hot_opinion(net, hot, 0);
  int len = net->output_size;
  if (bias >= 100) {
    int best = 0;
    float best_score = answer[0];
    for (int i = 1; i < len; i++) {
      if (answer[i] >= best_score) {
        best_score = answer[i];
        best = i;
      }
    }
    return best;
  }
  float *p = malloc(len * sizeof(float));
  biased_softmax_host(p, answer, len, bias);
  /* the draw comes from the net's own generator (rnn_opinion has just refreshed the host
   * struct); the device copy follows */
  int result = -1;
  while (result < 0) {
    float r = ramd_rand_double(&net->rng);
    float accum = 0.0f;
    for (int i = 0; i < len; i++) {
      accum += p[i];
      if (r < accum) {
        result = i;
        break;
      }
    }
  }
  ramd_rng_from_host(net);
  free(p);
  return result;
}

/* ------------------------------------------------------------- cross entropy -- */

/* get_cross_entropy (charmodel-predict.c:62-80) */
static double get_cross_entropy(RecurNN *net, const u8 *text, int len, int skip) {
  double entropy = rnn_amd_run_text(net, text, len, skip);
  entropy /= -(len - skip - 1);
  return entropy;
}

int rnn_char_prime(RecurNN *net, RnnCharAlphabet *alphabet, const u8 *text, const int len) {
  (void)alphabet;
  if (!text || !len) {
    return 0;
  }
  rnn_amd_run_text(net, text, len, len);
  return text[len - 1];
}

double rnn_char_cross_entropy(RecurNN *net, RnnCharAlphabet *alphabet, const u8 *text,
                              const int len, const int skip, const u8 *prefix_text,
                              const int prefix_len) {
  if (prefix_text) {
    rnn_char_prime(net, alphabet, prefix_text, prefix_len);
  }
  return get_cross_entropy(net, text, len, skip);
}

/* ------------------------------------------------------------------ schedule -- */

/* charmodel-predict.c:82-118: when the validation score is no better than any of a
 * third of the remembered scores, cut the learn rate */
static void eval_simple(RnnCharModel *model, float score, int verbose) {
  RecurNN *net = model->net;
  RnnCharSchedule *s = &model->schedule;
  RecurNNBPTT *bptt = net->bptt;
  if (s->recent_len == 0 || bptt->learn_rate <= s->learn_rate_min) {
    return;
  }
  int sample_size = s->recent_len / 3;
  ramd_rng_to_host(net);
  int i = ramd_rand_small_int(&net->rng, s->recent_len);
  ramd_rng_from_host(net);
  s->recent[i] = score;
  if (s->timeout) {
    s->timeout--;
    return;
  }
  i++;
  for (int j = 0; j < sample_size; j++, i++) {
    if (i >= s->recent_len) {
      i = 0;
    }
    if (score < s->recent[i]) {
      return;
    }
  }
  s->timeout = s->recent_len;
  bptt->learn_rate = RAMD_MAX(s->learn_rate_min, bptt->learn_rate * s->learn_rate_mul);
  if (s->adjust_noise) {
    net->presynaptic_noise *= s->learn_rate_mul;
    model->periodic_weight_noise *= s->learn_rate_mul;
  }
  if (verbose) {
    fprintf(stderr, "generation %7d: entropy %.4g exceeds %d recent samples."
                    " setting learn_rate to %.3g. momentum %.3g\n",
            net->generation, score, sample_size, bptt->learn_rate, net->bptt->momentum);
  }
}

void rnn_char_init_schedule(RnnCharSchedule *s, int recent_len, float learn_rate_min,
                            float learn_rate_mul, int adjust_noise) {
  s->recent_len = recent_len;
  if (recent_len) {
    s->recent = ramd_zalloc(recent_len * sizeof(float));
    s->learn_rate_min = learn_rate_min;
    s->learn_rate_mul = learn_rate_mul;
    for (int i = 0; i < recent_len; i++) {
      s->recent[i] = 1e10;
    }
  }
  s->timeout = s->recent_len;
  s->eval = eval_simple;
  s->adjust_noise = adjust_noise;
}

/* ------------------------------------------------------- validation entropy -- */

void rnn_char_delete_ventropy(RnnCharVentropy *v) { free(v->history); }

void rnn_char_init_ventropy(RnnCharVentropy *v, RecurNN *net, const u8 *text, const int len,
                            const int lap) {
  v->net = net;
  v->text = text;
  v->len = len;
  v->lap = lap;
  v->lapsize = len / lap;
  v->history = calloc(lap, sizeof(float));
  v->entropy = 0;
  v->counter = 0;
}

/* charmodel-predict.c:227-258: either the whole validation text, or one lap of it per
 * call with the mean of the laps seen so far */
float rnn_char_calc_ventropy(RnnCharModel *model, RnnCharVentropy *v, int lap) {
  (void)model;
  if (v->len > 0) {
    if (v->lap > 1 && lap) {
      v->counter++;
      if (v->counter == v->lap) {
        v->counter = 0;
      }
      int skip = RAMD_MIN(v->lapsize / 10, 5);
      v->history[v->counter] =
          get_cross_entropy(v->net, v->text + v->lapsize * v->counter, v->lapsize, skip);
      float sum = 0.0f;
      float div = v->lap;
      for (int j = 0; j < v->lap; j++) {
        div -= v->history[j] == 0;
        sum += v->history[j];
      }
      v->entropy = div ? sum / div : 0;
    } else {
      int skip = RAMD_MIN(v->len / 10, 5);
      v->entropy = get_cross_entropy(v->net, v->text, v->len, skip);
      v->history[0] = v->entropy;
    }
  }
  return v->entropy;
}

/* ------------------------------------------------------------ confabulation -- */

/* utf8.h:31-65 */
static int write_code_point(int c, char *dest, int utf8) {
  unsigned code = (unsigned)c;
  if (!utf8 || code < 0x80) {
    dest[0] = (char)c;
    return 1;
  }
  if (code < 0x800) {
    dest[0] = (char)(0xC0 | (code >> 6));
    dest[1] = (char)(0x80 | (code & 63));
    return 2;
  }
  if (code < 0x10000) {
    dest[0] = (char)(0xE0 | (code >> 12));
    dest[1] = (char)(0x80 | ((code >> 6) & 63));
    dest[2] = (char)(0x80 | (code & 63));
    return 3;
  }
  if (code < 0x200000) {
    dest[0] = (char)(0xF0 | (code >> 18));
    dest[1] = (char)(0x80 | ((code >> 12) & 63));
    dest[2] = (char)(0x80 | ((code >> 6) & 63));
    dest[3] = (char)(0x80 | (code & 63));
    return 4;
  }
  return 0;
}

int rnn_char_confabulate(RecurNN *net, char *dest, int char_len, int byte_len, RnnCharAlphabet *a,
                         float bias, int *prev_char, int start_point, int stop_point) {
  int n = *prev_char;
  int utf8 = (a->flags & RNN_CHAR_FLAG_UTF8) != 0;
  const int *alphabet = a->points;
  int safe_end = byte_len - (utf8 ? 5 : 1);
  if (safe_end <= 0) {
    fprintf(stderr, "insufficient space to confabulate (%d bytes)\n", byte_len);
    if (byte_len) {
      *dest = 0;
    }
    return 0;
  }
  int i, j = 0;
  if (start_point >= 0 && char_len > 0) {
    /* run until the requested first character turns up */
    for (i = 0; i < 1000000 && n != start_point; i++) {
      n = guess_next_character(net, n, bias);
    }
    j = write_code_point(alphabet[n], dest, utf8);
    dest[j] = 0;
    if (n != start_point) {
      fprintf(stderr, "start char '%s' not found in first %d characters, giving up\n", dest, i);
    } else {
      fprintf(stderr, "start char '%s' found after %d others\n", dest, i);
    }
  }
  for (i = 0; i < char_len && j < safe_end; i++) {
    n = guess_next_character(net, n, bias);
    j += write_code_point(alphabet[n], dest + j, utf8);
    if (n == stop_point) {
      break;
    }
  }
  dest[j] = 0;
  *prev_char = n;
  return j;
}

/* charmodel-predict.c:184-206: one line to a stream, ending at end_code */
static int fconfab_variable(FILE *f, RecurNN *net, const int end_code, int *prev_char, int max_len,
                            RnnCharAlphabet *a, float bias) {
  int n = *prev_char;
  int utf8 = (a->flags & RNN_CHAR_FLAG_UTF8) != 0;
  int i;
  for (i = 0; i < max_len; i++) {
    n = guess_next_character(net, n, bias);
    if (n == end_code) {
      break;
    }
    char s[5];
    int w = write_code_point(a->points[n], s, utf8);
    s[w] = 0;
    fputs(s, f);
  }
  if (i == max_len) {
    fputs(C_YELLOW "\\" C_NORMAL, f);
  }
  fputc('\n', f);
  *prev_char = n;
  return i;
}

/* --------------------------------------------------------------------- epoch -- */

int rnn_char_epoch(RnnCharModel *model, RecurNN *confab_net, RnnCharVentropy *v, const u8 *text,
                   const int len, const int start, const int stop, float confab_bias,
                   int confab_size, int confab_line_end, int quietness,
                   uint diagonal_only_section, uint diagonal_only_friends) {
  float error = 0.0f, entropy = 0.0f;
  int correct = 0;
  int n_nets = model->n_training_nets;
  int spacing = (len - 1) / n_nets;
  RecurNN *net = model->net;
  RecurNN **nets = model->training_nets;
  uint report_counter = net->generation % model->report_interval;
  float report_scale = 1.0f / ((model->report_interval - report_counter) * n_nets);
  int confab_char = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  if (diagonal_only_section) {
    rnn_clear_diagonal_only_section(net, diagonal_only_section, diagonal_only_friends);
  }
  const int multi_tap = n_nets > 1 || model->learning_style != RNN_MOMENTUM_WEIGHTED ||
                        model->use_multi_tap_path;
  /* the whole set per call when the nets are one training set; the text goes to the
   * device once per epoch */
  RnnAmdSet *set = NULL;
  /* ... and the single-net branch as a set of one: the loss and rnn_bptt_calculate's work stay on the device
   * (a net with a log file keeps the per-call route, which writes the core's log lines) */
  RecurNN *one_net[1] = {net};
  const int fused_set = !multi_tap && !net->log && net->bptt;
  if (fused_set) {
    set = rnn_amd_set_open(one_net, 1);
    if (set) {
      RnnAmdStats drop;
      rnn_amd_set_load_text(set, text, len);
      rnn_amd_set_read_stats(set, &drop, 1);
    }
  }
  if (multi_tap && nets && nets[0] == net) {
    set = rnn_amd_set_open(nets, n_nets);
    if (set) {
      RnnAmdStats drop;
      rnn_amd_set_load_text(set, text, len);
      rnn_amd_set_read_stats(set, &drop, 1);
    }
  }
  const int want_rows = model->images.input_ppm || model->images.error_ppm;
  int result = 0;
  for (int i = start; i < len - 1; i++) {
    float momentum = rnn_calculate_momentum_soft_start(net->generation, model->momentum,
                                                       model->momentum_soft_start);
    if (set && fused_set) {
      net->bptt->momentum = momentum;
      rnn_amd_set_char_step_fused(set, i, model->batch_size);
    } else if (set) {
      rnn_amd_set_char_step(set, i, model->learning_style, momentum);
    } else if (multi_tap) {
      for (int j = 0; j < n_nets; j++) {
        RecurNN *n = nets[j];
        int offset = i + j * spacing;
        if (offset >= len - 1) {
          offset -= len - 1;
        }
        int c;
        rnn_bptt_advance(n);
        float e = net_error_bptt(n, n->bptt->o_error, text[offset], text[offset + 1], &c);
        correct += c;
        error += e;
        entropy += capped_log2f(1.0f - e);
        rnn_bptt_calc_deltas(n, j ? 1 : 0, NULL);
      }
      rnn_apply_learning(net, model->learning_style, momentum);
    } else {
      int c;
      RecurNNBPTT *bptt = net->bptt;
      bptt->momentum = momentum;
      rnn_bptt_advance(net);
      float e = net_error_bptt(net, bptt->o_error, text[i], text[i + 1], &c);
      rnn_bptt_calculate(net, model->batch_size);
      correct += c;
      error += e;
      entropy += capped_log2f(1.0f - e);
    }
    if (diagonal_only_section) { /* a no-op for section 0 in the reference too */
      rnn_clear_diagonal_only_section(net, diagonal_only_section, diagonal_only_friends);
    }
    if (want_rows) {
      if (set) {
        rnn_amd_sync_host(net, RNN_AMD_STREAM);
      }
      if (model->images.input_ppm) {
        temporal_add_row(model->images.input_ppm, net->input_layer);
      }
      if (model->images.error_ppm) {
        temporal_add_row(model->images.error_ppm, net->bptt->o_error);
      }
    }
    report_counter++;
    if (report_counter >= model->report_interval) {
      report_counter = 0;
      if (set) {
        RnnAmdStats st;
        rnn_amd_set_read_stats(set, &st, 1); /* waits for the device */
        error = (float)st.error;
        entropy = (float)st.entropy;
        correct = (int)st.correct;
      } else {
        rnn_amd_synchronize();
      }
      clock_gettime(CLOCK_MONOTONIC, &t1);
      double elapsed = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
      t0 = t1;
      float ventropy = rnn_char_calc_ventropy(model, v, 1);
      {
        int k = net->generation >> 10;
        entropy *= -report_scale;
        error *= report_scale;
        float accuracy = correct * report_scale;
        double per_sec = 1.0 / report_scale / elapsed;
        if (confab_net && confab_size && quietness < 1) {
          if (confab_line_end >= 0) {
            fprintf(stderr, C_GREY "%5dk t%.2f " C_CYAN "v%.2f" C_GREY " %.0f/s |" C_NORMAL, k,
                    entropy, ventropy, per_sec + 0.5);
            fconfab_variable(stderr, confab_net, confab_line_end, &confab_char, confab_size,
                             model->alphabet, confab_bias);
          } else {
            fprintf(stderr, C_GREY "%5dk e.%02d t%.2f v" C_CYAN "%.2f" C_GREY " a.%02d %.0f/s |" C_NORMAL,
                    k, (int)(error * 100 + 0.5), entropy, ventropy, (int)(accuracy * 100 + 0.5),
                    per_sec + 0.5);
            int alloc_size = confab_size * 4;
            char *confab = malloc(alloc_size + 1);
            rnn_char_confabulate(confab_net, confab, confab_size, alloc_size, model->alphabet,
                                 confab_bias, &confab_char, -1, -1);
            fprintf(stderr, "%s" C_GREY "|\n", confab);
            free(confab);
          }
        }
        rnn_log_float(net, "t_error", error);
        rnn_log_float(net, "t_entropy", entropy);
        rnn_log_float(net, "v_entropy", ventropy);
        rnn_log_float(net, "momentum", net->bptt->momentum);
        rnn_log_float(net, "accuracy", accuracy);
        rnn_log_float(net, "learn-rate", net->bptt->learn_rate);
        rnn_log_float(net, "per_second", per_sec);
        correct = 0;
        error = 0.0f;
        entropy = 0.0f;
        report_scale = 1.0f / (model->report_interval * n_nets);
      }
      if (model->save_net && model->filename) {
        rnn_save_net(net, model->filename, 1);
      }
      if (model->images.periodic_pgm_dump_string) {
        rnn_multi_pgm_dump(net, model->images.periodic_pgm_dump_string, model->images.basename);
      }
      model->schedule.eval(model, ventropy, quietness < 2);
      if (model->periodic_weight_noise) {
        rnn_weight_noise(net, model->periodic_weight_noise);
      }
    }
    if (stop && (int)net->generation >= stop) {
      result = 1;
      break;
    }
  }
  if (set) {
    /* statistics gathered since the last report are dropped at the end of an epoch, as
     * the reference's locals are */
    rnn_amd_set_close(set);
  }
  return result;
}

/* ============================================== the multi-head text trainer == */
/* charmodel-multi-predict.c, the caller py-recur-text.c's Net.train / Net.test sit on
 * (py-recur-text.c:759-871): ONE net whose output row is n_classes heads of alphabet_len
 * symbols, trained on one class's text at a time with temporal batching.  The loops below
 * keep the reference's order of operations; what runs per symbol is device work through
 * the set calls (a set of this one net, the text resident in HBM), so no error vector,
 * range list or statistic visits the host inside the loop. */

/* charmodel-multi-predict.c:60-72: image rows and periodic dumps want host copies */
static void multi_inner_cycle_dump(RecurNN *net, TemporalPPM *input_ppm, TemporalPPM *error_ppm,
                                   const char *periodic_pgm_string, int periodic_pgm_period,
                                   int *periodic_pgm_countdown) {
  if (input_ppm || error_ppm) {
    rnn_amd_sync_host(net, RNN_AMD_STREAM);
  }
  if (input_ppm) {
    temporal_add_row(input_ppm, net->input_layer);
  }
  if (error_ppm) {
    temporal_add_row(error_ppm, net->bptt->o_error);
  }
  if (periodic_pgm_period) {
    if (!--*periodic_pgm_countdown) {
      *periodic_pgm_countdown = periodic_pgm_period;
      rnn_multi_pgm_dump(net, periodic_pgm_string, "multi-text");
    }
  }
}

/* charmodel-multi-predict.c:74-119 */
RnnCharMultiConfab *rnn_char_new_multi_confab(RecurNN *net, RnnCharAlphabet *alphabet, int n_classes,
                                              int target_len, uint confab_period, int caps_marker) {
  int len = target_len / n_classes - 1;
  if (len < 1) {
    fprintf(stderr, "no room to confabulate %d sub-models in %d characters\n", n_classes, target_len);
    return NULL;
  }
  RnnCharMultiConfab *mc = calloc(1, sizeof(*mc));
  mc->char_len = len;
  mc->byte_len = len * 6 + 1;
  mc->alphabet = alphabet;
  mc->period = confab_period;
  mc->n_classes = n_classes;
  mc->caps_marker = caps_marker;
  mc->last_char = calloc(n_classes, sizeof(int));
  mc->strings = calloc(n_classes, sizeof(char *));
  mc->nets = calloc(n_classes, sizeof(RecurNN *));
  for (int i = 0; i < n_classes; i++) {
    mc->strings[i] = calloc(mc->byte_len, 1);
    mc->nets[i] = rnn_clone(net, net->flags & ~(RNN_NET_FLAG_OWN_BPTT | RNN_NET_FLAG_OWN_WEIGHTS),
                            RECUR_RNG_SUBSEED, NULL);
  }
  return mc;
}

void rnn_char_free_multi_confab(RnnCharMultiConfab *mc) {
  for (uint i = 0; i < mc->n_classes; i++) {
    free(mc->strings[i]);
    rnn_delete_net(mc->nets[i]);
  }
  free(mc->strings);
  free(mc->nets);
  free(mc->last_char);
  free(mc);
}

/* charmodel-multi-predict.c:122-142: sample the next symbol from head `offset` */
static int offset_guess_next_character(RecurNN *net, float *error, int hot, float bias, uint offset,
                                       int alphabet_len) {
  float *answer = one_hot_opinion(net, hot, 0);
  float *group = answer + alphabet_len * offset;
  biased_softmax_host(error, group, alphabet_len, bias);
  int result = -1;
  while (result < 0) {
    float r = ramd_rand_double(&net->rng);
    float accum = 0.0f;
    for (int i = 0; i < alphabet_len; i++) {
      accum += error[i];
      if (r < accum) {
        result = i;
        break;
      }
    }
  }
  ramd_rng_from_host(net);
  return result;
}

/* charmodel-multi-predict.c:145-197 */
static int multi_confab(RnnCharMultiConfab *mc) {
  const int *alphabet = mc->alphabet->points;
  int total = 0;
  int utf8 = (mc->alphabet->flags & RNN_CHAR_FLAG_UTF8) != 0;
  float *error = malloc(sizeof(float) * mc->alphabet->len);
  int char_width = utf8 ? 5 : 1;
  if ((int)mc->byte_len <= char_width) {
    fprintf(stderr, "insufficient space to confabulate (%d bytes)\n", mc->byte_len);
    free(error);
    return 0;
  }
  for (uint m = 0; m < mc->n_classes; m++) {
    RecurNN *net = mc->nets[m];
    char *d = mc->strings[m];
    int n = mc->last_char[m];
    int bytes_left = mc->byte_len;
    int pending_caps = 0;
    for (int i = 0; i < (int)mc->char_len && bytes_left > char_width;) {
      n = offset_guess_next_character(net, error, n, mc->bias, m, mc->alphabet->len);
      int c = alphabet[n];
      if (c == mc->caps_marker) {
        pending_caps = 1;
      } else {
        if (pending_caps) { /* capitalisation is limited to ascii and greek */
          if (c >= 'a' && c <= 'z') {
            c -= ('a' - 'A');
          } else if (c >= 945 && c <= 969) {
            c -= 32;
            if (c == 930) {
              c++; /* word-final sigma capitalises to plain sigma */
            }
          }
        }
        int w = write_code_point(c, d, utf8);
        d += w;
        bytes_left -= w;
        pending_caps = 0;
        i++;
      }
    }
    *d = '\0';
    total += mc->byte_len - bytes_left;
    mc->last_char[m] = n;
  }
  free(error);
  return total;
}

/* charmodel-multi-predict.c:200-230: the heads' strings joined by `sep` */
static int multi_confab_format_line(RnnCharMultiConfab *mc, char *dest, int len, char *sep) {
  uint i = 0;
  int in_sep = 0;
  int total;
  char *src = mc->strings[0];
  for (total = 0; total < len - 1; total++, src++) {
    char c = *src;
    while (c == '\0') {
      in_sep = !in_sep;
      if (in_sep) {
        src = sep;
      } else {
        i++;
        if (i >= mc->n_classes) {
          break;
        }
        src = mc->strings[i];
      }
      c = *src;
    }
    if (i >= mc->n_classes) {
      break;
    }
    dest[total] = c;
  }
  dest[total] = '\0';
  return total;
}

/* charmodel-multi-predict.c:234-281 (text_train) */
static void multi_text_train(RecurNN *net, u8 *text, int len, int learning_style, int target_class,
                             int batch_size, float leakage, int alphabet_len,
                             RnnCharProgressReport *report, RnnCharMultiConfab *mc,
                             TemporalPPM *input_ppm, TemporalPPM *error_ppm,
                             const char *periodic_pgm_string, int periodic_pgm_period,
                             int periodic_pgm_countdown) {
  RecurNNBPTT *bptt = net->bptt;
  int i = 0;
  if (len >= 2) {
    RecurNN *one[1] = {net};
    RnnAmdSet *set = rnn_amd_set_open(one, 1);
    if (!set) {
      fprintf(stderr, "librecur_amd: rnn_char_multitext_train needs a net with its own bptt\n");
      abort();
    }
    RnnAmdStats st;
    rnn_amd_set_load_text(set, text, len);
    rnn_amd_set_read_stats(set, &st, 1);
    int countdown = batch_size - net->generation % batch_size;
    const int dump = input_ppm || error_ppm || periodic_pgm_period;
    for (i = 0; i < len - 1; i++, countdown--) {
      /* rnn_bptt_advance + multi_softmax_error (its opinion, its leakage draws from the
       * net's generator, the error ranges) */
      rnn_amd_set_multi_text_loss(set, i, i == 0 ? &target_class : NULL, alphabet_len, leakage);
      if (countdown == 0) {
        /* line 247: the net's own momentum, not the caller's */
        rnn_apply_learning(net, learning_style, bptt->momentum);
        countdown = batch_size;
        rnn_amd_set_multi_calc_deltas(set, 0);
      } else {
        rnn_amd_set_multi_calc_deltas(set, 1);
      }
      if (dump) {
        multi_inner_cycle_dump(net, input_ppm, error_ppm, periodic_pgm_string, periodic_pgm_period,
                               &periodic_pgm_countdown);
      }
    }
    if (report) {
      rnn_amd_set_read_stats(set, &st, 1);
      float report_scale = 1.0f / (len - 1);
      report->training_entropy = (float)(-st.entropy) * report_scale;
      report->training_error = (float)st.error * report_scale;
    }
    rnn_amd_synchronize();
    free(set->nets); /* (the set's own copy of the pointer array) */
    free(set);       /* the net's state stays on the device; nothing to copy back here */
  }
  if (mc && mc->period && i % mc->period == 0) {
    char *confab_line = malloc(mc->byte_len + 1);
    multi_confab(mc);
    multi_confab_format_line(mc, confab_line, mc->byte_len, C_CYAN "|" C_NORMAL);
    printf("%8u" C_CYAN "|" C_NORMAL "%s\n", net->generation, confab_line);
    free(confab_line);
  }
}

/* charmodel.h:251-254; charmodel-multi-predict.c:283-301: run a text through the net without
 * training (advance + opinion with the net's noise) */
void rnn_char_multitext_spin(RecurNN *net, u8 *text, int len, TemporalPPM *input_ppm,
                             TemporalPPM *error_ppm, const char *periodic_pgm_string,
                             int periodic_pgm_period) {
  int periodic_pgm_countdown = 0;
  if (periodic_pgm_period) {
    periodic_pgm_countdown = periodic_pgm_period - net->generation % periodic_pgm_period;
  }
  if (len < 1) {
    return;
  }
  if (error_ppm) {
    rnn_amd_sync_host(net, RNN_AMD_STREAM);
    memset(net->bptt->o_error, 0, net->output_size * sizeof(float));
    rnn_amd_host_written(net, RNN_AMD_STREAM);
  }
  RecurNN *one[1] = {net};
  RnnAmdSet *set = rnn_amd_set_open(one, 1);
  if (!set) {
    fprintf(stderr, "librecur_amd: rnn_char_multitext_spin needs a net with its own bptt\n");
    abort();
  }
  {
    /* every one of the len symbols is an input here (charmodel-multi-predict.c:298-302), the last one
     * too; the device text steps read positions modulo len - 1 (a training step's input always has a
     * successor): give the text one more symbol, so that position len - 1 is an input and not a wrap */
    u8 *padded = malloc((size_t)len + 1);
    memcpy(padded, text, len);
    padded[len] = text[len - 1];
    rnn_amd_set_load_text(set, padded, len + 1);
    free(padded);
  }
  const int dump = input_ppm || error_ppm || periodic_pgm_period;
  for (int i = 0; i < len; i++) {
    rnn_amd_set_text_opinion(set, i, 1);
    if (dump) {
      multi_inner_cycle_dump(net, input_ppm, error_ppm, periodic_pgm_string, periodic_pgm_period,
                             &periodic_pgm_countdown);
    }
  }
  rnn_amd_synchronize();
  free(set->nets);
  free(set);
}

/* charmodel.h:242-248; charmodel-multi-predict.c:307-347 */
void rnn_char_multitext_train(RecurNN *net, u8 *text, int len, int alphabet_len, int target_class,
                              float leakage, RnnCharProgressReport *report,
                              RnnCharMultiConfab *confab, int learning_style, float momentum,
                              int batch_size, TemporalPPM *input_ppm, TemporalPPM *error_ppm,
                              const char *periodic_pgm_string, int periodic_pgm_period) {
  struct timespec time_start, time_end;
  int periodic_pgm_countdown = 0;
  (void)momentum; /* unused by the reference too: text_train applies bptt->momentum */
  if (periodic_pgm_period) {
    periodic_pgm_countdown = periodic_pgm_period - net->generation % periodic_pgm_period;
  }
  batch_size = RAMD_MAX(batch_size, 1);
  if (report) {
    clock_gettime(CLOCK_MONOTONIC, &time_start);
  }
  multi_text_train(net, text, len, learning_style, target_class, batch_size, leakage, alphabet_len,
                   report, confab, input_ppm, error_ppm, periodic_pgm_string, periodic_pgm_period,
                   periodic_pgm_countdown);
  if (report) {
    clock_gettime(CLOCK_MONOTONIC, &time_end);
    double elapsed = (time_end.tv_sec - time_start.tv_sec) + 1e-9 * (time_end.tv_nsec - time_start.tv_nsec);
    report->per_second = (len - 1) / elapsed;
  }
}

/* charmodel.h:255-257; charmodel-multi-predict.c:383-408: entropy[j] is subtracted into and
 * then divided, exactly as the reference treats the caller's array */
void rnn_char_multi_cross_entropy(RecurNN *net, const u8 *text, int len, int alphabet_len,
                                  double *entropy, int ignore_start) {
  int n_classes = net->output_size / alphabet_len;
  double *sums = malloc(sizeof(double) * RAMD_MAX(n_classes, 1));
  rnn_amd_run_text_heads(net, text, len, ignore_start, alphabet_len, sums);
  for (int j = 0; j < n_classes; j++) {
    entropy[j] -= sums[j];
    entropy[j] /= (len - ignore_start - 1);
  }
  free(sums);
}
