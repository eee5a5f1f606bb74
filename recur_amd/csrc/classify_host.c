/* classify_host.c -- what a host program needs around a GPU-resident audio classifier: the
 * metadata string, net file name and class-group string of gstclassify (contract: saved nets carry
 * them), and one generation of its training loop -- balanced-training draws included -- as batched
 * device calls (gnu11 C; include/recur_amd_classify.h says which reference lines each piece follows).
 * The plugin itself (GStreamer, feature extraction) stays on the host and out of this library.
 */
#include "char_host.h"
#include "recur_amd_classify.h"
#include <inttypes.h>

/* ------------------------------------------------------------------ metadata -- */

static const char *or_null_text(const char *s) { return s ? s : "(null)"; } /* what glibc's %s prints */

char *rnn_amd_classify_construct_metadata(const RnnAmdClassifyMetadata *m) {
  char *text = NULL;
  const int n = asprintf(&text,
                         "classes %s\nmin-frequency %f\nmax-frequency %f\nknee-frequency %f\nmfccs %d\n"
                         "window-size %d\nbasename %s\ndelta-features %d\nfocus-frequency %f\nlag %f\n"
                         "intensity-feature %d\nconfirmation-lag %f\nfeatures-offset %s\nfeatures-scale %s\n",
                         or_null_text(m->classes), m->min_freq, m->max_freq, m->knee_freq, m->mfccs, m->window_size,
                         or_null_text(m->basename), m->delta_features, m->focus_freq, m->lag, m->intensity_feature,
                         m->confirmation_lag, or_null_text(m->features_offset), or_null_text(m->features_scale));
  if (n < 0) {
    fprintf(stderr, "librecur_amd: out of memory for a classifier's metadata\n");
    abort();
  }
  return text;
}

/* one "name value" item at *at: returns the value's length (the text up to white space, which *value then
 * points at, inside the string) and moves *at past it when the name matched; 0 otherwise.  No length limit: the
 * reference reads text items with %ms (gstclassify.c:841-929) */
static size_t take_item(const char **at, const char *name, const char **value) {
  const char *p = *at;
  while (*p == ' ' || *p == '\n' || *p == '\t') {
    p++;
  }
  const size_t nl = strlen(name);
  if (strncmp(p, name, nl) != 0 || p[nl] != ' ') {
    return 0;
  }
  p += nl + 1;
  while (*p == ' ') {
    p++;
  }
  const size_t k = strcspn(p, " \n\t");
  *value = p;
  *at = p + k;
  return k;
}

int rnn_amd_classify_load_metadata(const char *metadata, RnnAmdClassifyMetadata *m) {
  if (!metadata) {
    fprintf(stderr, "librecur_amd: a classifier net without metadata\n");
    return -1;
  }
  /* eleven items, in this order; reading stops at the first one that is not there (sscanf semantics) */
  enum { TEXT, REAL, WHOLE };
  const struct {
    const char *name;
    int kind;
    void *field;
  } items[] = {{"classes", TEXT, &m->classes},          {"min-frequency", REAL, &m->min_freq},
               {"max-frequency", REAL, &m->max_freq},   {"knee-frequency", REAL, &m->knee_freq},
               {"mfccs", WHOLE, &m->mfccs},             {"window-size", WHOLE, &m->window_size},
               {"basename", TEXT, &m->basename},        {"delta-features", WHOLE, &m->delta_features},
               {"focus-frequency", REAL, &m->focus_freq}, {"lag", REAL, &m->lag},
               {"intensity-feature", WHOLE, &m->intensity_feature}};
  const int wanted = (int)(sizeof(items) / sizeof(items[0]));
  const char *at = metadata, *value;
  int found = 0;
  for (; found < wanted; found++) {
    const size_t len = take_item(&at, items[found].name, &value);
    if (!len) {
      break;
    }
    if (items[found].kind == TEXT) {
      *(const char **)items[found].field = strndup(value, len);
    } else if (items[found].kind == REAL) { /* (the number ends at the white space behind it) */
      *(float *)items[found].field = strtof(value, NULL);
    } else {
      *(int *)items[found].field = (int)strtol(value, NULL, 10);
    }
  }
  if (found != wanted) {
    fprintf(stderr, "librecur_amd: found only %d/%d metadata items\n", found, wanted);
  }
  return wanted - found;
}

void rnn_amd_classify_free_metadata_items(RnnAmdClassifyMetadata *m) {
  free((char *)m->classes);
  free((char *)m->basename);
  m->classes = m->basename = NULL;
}

char *rnn_amd_classify_net_filename(const char *basename, const char *metadata, int n_features,
                                    int bottom_layer, int hidden_size, int top_layer_size,
                                    int sample_rate, int window_size) {
  char layers[64], *name = NULL;
  if (bottom_layer > 0) {
    snprintf(layers, sizeof(layers), "i%d-b%d-h%d-o%d", n_features, bottom_layer, hidden_size, top_layer_size);
  } else {
    snprintf(layers, sizeof(layers), "i%d-h%d-o%d", n_features, hidden_size, top_layer_size);
  }
  if (asprintf(&name, "%s-%0" PRIx32 "-%s-%dHz-w%d.net", basename, ramd_hash32(metadata), layers, sample_rate,
               window_size) < 0) {
    abort();
  }
  return name;
}

int rnn_amd_classify_parse_classes(const char *classes, int *offsets, int *sizes, int max_groups,
                                   int *n_outputs, int *string_len) {
  int groups = 0, letters = 0, run = 0, start = 0, i = 0;
  for (;; i++) {
    const char c = classes[i];
    if (c == ',' || c == 0) { /* a group ends (an empty one still counts: "a,,b" has three) */
      if (groups < max_groups) {
        offsets[groups] = start;
        sizes[groups] = run;
      }
      groups++;
      start = i + 1;
      run = 0;
      if (c == 0) {
        break;
      }
    } else {
      run++;
      letters++;
    }
  }
  if (n_outputs) {
    *n_outputs = letters;
  }
  if (string_len) {
    *string_len = i;
  }
  return groups;
}

int rnn_amd_classify_check_net(const RecurNN *net, const char *metadata, int hidden_size,
                               int bottom_layer, int top_layer_size, int force_load) {
  const int bottom_out = net->bottom_layer ? net->bottom_layer->output_size : 0;
  const int sizes_ok = net->output_size == top_layer_size && net->hidden_size == hidden_size &&
                       !(net->bottom_layer && !bottom_layer);
  const int meta_ok = !net->metadata || force_load || strcmp(net->metadata, metadata) == 0;
  if (sizes_ok && meta_ok) {
    return 0;
  }
  fprintf(stderr, "librecur_amd: this net is not the classifier that was asked for.\n"
                  "  outputs: expected %d, loaded %d\n  hidden:  expected %d, loaded %d\n  bottom:  expected %d, loaded %d\n"
                  "  metadata expected:\n%s\n  metadata loaded:\n%s\n",
          top_layer_size, net->output_size, hidden_size, net->hidden_size, bottom_layer, bottom_out, metadata,
          net->metadata ? net->metadata : "(none)");
  return -1;
}

/* --------------------------------------------------------- balanced training -- */

RnnAmdBalancedTraining *rnn_amd_balanced_new(int n_outputs, float bias) {
  RnnAmdBalancedTraining *b = ramd_zalloc(sizeof(*b));
  b->n_outputs = n_outputs;
  b->bias = bias;
  b->seen = ramd_zalloc(sizeof(u32) * n_outputs);
  b->used = ramd_zalloc(sizeof(u32) * n_outputs);
  b->train_p = ramd_zalloc(sizeof(float) * n_outputs);
  return b;
}

void rnn_amd_balanced_free(RnnAmdBalancedTraining *b) {
  if (b) {
    free(b->seen);
    free(b->used);
    free(b->train_p);
    free(b);
  }
}

void rnn_amd_balanced_begin(RnnAmdBalancedTraining *b) {
  u32 met = 0;
  for (int c = 0; c < b->n_outputs; c++) {
    met += b->seen[c];
  }
  const float share = 1.0f / (met + 1.0f);
  for (int c = 0; c < b->n_outputs; c++) {
    b->train_p[c] = powf(1.0f - b->seen[c] * share, b->bias);
  }
}

/* recur-rng.h:80-85: a float in [0, 1] from 64 random bits */
static float unit_draw(rand_ctx *rng) { return (float)ramd_rand64(rng) * (1.0f / ((float)0xfffffffffffffffeULL)); }

/* ------------------------------------------------------------ one generation -- */

int rnn_amd_classify_generation(RnnAmdSet *set, const float *features, int ld_features, int n_groups,
                                const int *group_offset, const int *group_size, const int *targets,
                                const float *error_weight, RnnAmdBalancedTraining *balance,
                                int learning_style, float momentum_soft_start, int exact_gate) {
  RecurNN *net = set->nets[0]; /* the prototype: its generator makes the balanced-training draws */
  const int channels = set->n;
  rnn_bptt_clear_deltas(net);
  if (balance) {
    rnn_amd_balanced_begin(balance);
  }
  /* every channel's forward pass (its own noise draws first, as in train_channel); without presynaptic noise it draws
   * nothing and can share a call -- and a launch -- with the loss below (rnn_amd_set_opinion_grouped_softmax) */
  const int together = net->presynaptic_noise == 0.0f;
  if (!together) {
    rnn_amd_set_opinion(set, features, ld_features, NULL);
  }
  /* which labelled (channel, group) pairs train: all of them, or the balanced sample -- decided here on
   * the host, in the reference's order, and handed to the device loss as "no target" where not */
  int *use = malloc(sizeof(int) * (size_t)channels * n_groups);
  u8 *trains = malloc(channels);
  int trained_groups = 0;
  if (balance) {
    ramd_rng_to_host(net);
  }
  for (int ch = 0; ch < channels; ch++) {
    trains[ch] = 0;
    for (int g = 0; g < n_groups; g++) {
      const int t = targets[ch * n_groups + g];
      int take = t >= 0 && t < group_size[g];
      if (take && balance) {
        const int out = group_offset[g] + t;
        balance->seen[out]++;
        take = balance->train_p[out] > unit_draw(&net->rng);
        balance->used[out] += take;
      }
      use[ch * n_groups + g] = take ? t : -1;
      trains[ch] |= take;
      trained_groups += take;
    }
  }
  if (balance) {
    ramd_rng_from_host(net);
  }
  RnnAmdStats before, after;
  if (exact_gate) {
    rnn_amd_set_read_stats(set, &before, 0);
  }
  if (together) {
    u8 *trained = malloc(channels); /* (== trains: a channel trains if one of its groups has a target left) */
    rnn_amd_set_opinion_grouped_softmax(set, features, ld_features, n_groups, group_offset, group_size, use, error_weight,
                                        trained);
    free(trained);
  } else {
    rnn_amd_set_grouped_softmax_error(set, n_groups, group_offset, group_size, use, error_weight, NULL);
  }
  if (trained_groups) {
    rnn_amd_set_calc_deltas(set, 1, NULL, trains); /* adds to the cleared deltas */
  }
  rnn_amd_set_advance(set);
  int update = trained_groups > 0;
  if (update && exact_gate) { /* the reference's `if (err_sum)`: the generation's summed error, read back */
    rnn_amd_set_read_stats(set, &after, 0);
    update = after.error - before.error != 0.0;
  }
  if (update) {
    rnn_apply_learning(net, learning_style,
                       rnn_calculate_momentum_soft_start(net->generation, net->bptt->momentum, momentum_soft_start));
  }
  rnn_condition_net(net);
  free(use);
  free(trains);
  return trained_groups;
}
