/* classify_train_amd.c -- a synthetic stand-in for gstclassify's training pipeline on the MI355X core
 * (BASELINE.json configs[2]: 2-class audio, 512 hidden, 128 parallel channels).
 *
 * The plugin's host side -- GStreamer, FFT / MFCC features -- is out of scope; what it does with the
 * features is not: one training net per audio channel sharing weights (gstclassify.c:1313), per chunk
 * maybe_learn (2196-2257): clear the deltas, per channel opinion -> class-group softmax loss (through
 * the balanced-training draw) -> deltas -> advance, then one Nesterov update and rnn_condition_net.
 * Here the features are synthetic: every channel hears class 0 or 1 for random stretches of 50-200
 * windows, the 32 "log-power bins" are noise plus a class-dependent spectral tilt.  The loop is
 * rnn_amd_classify_generation, all channels per launch sequence; the net is saved under the
 * reference's file name with the reference's metadata, loaded back and checked.
 *
 *   classify_train_amd [-H hidden] [-c channels] [-d depth] [-g generations] [-l learn_rate]
 *                      [-b balanced_bias (0 = off)] [-r report_every] [-x (exact `if (err_sum)` gate)]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "recur-nn.h"
#include "recur_amd_classify.h"

#define N_FEATURES 32 /* gstclassify.h:20 */

static float uniform(rand_ctx *r) { /* host-side feature noise: a generator of this program's own */
  uint64_t e = r->a - ((r->b << 7) | (r->b >> 57));
  r->a = r->b ^ ((r->c << 13) | (r->c >> 51));
  r->b = r->c + ((r->d << 37) | (r->d >> 27));
  r->c = r->d + e;
  r->d = e + r->a;
  return (float)(r->d >> 40) / (float)(1 << 24);
}

int main(int argc, char **argv) {
  int hidden = 512, channels = 128, depth = 30, generations = 600, report = 100, exact = 0, opt;
  float lr = 3e-4f, bias = 1.0f;
  while ((opt = getopt(argc, argv, "H:c:d:g:l:b:r:x")) != -1) {
    switch (opt) {
    case 'H': hidden = atoi(optarg); break;
    case 'c': channels = atoi(optarg); break;
    case 'd': depth = atoi(optarg); break;
    case 'g': generations = atoi(optarg); break;
    case 'l': lr = atof(optarg); break;
    case 'b': bias = atof(optarg); break;
    case 'r': report = atoi(optarg); break;
    case 'x': exact = 1; break;
    default: fprintf(stderr, "see the head of tools/classify_train_amd.c\n"); return 2;
    }
  }
  RnnAmdClassifyMetadata md = {"01", 100.f, 1600.f, 700.f, 0, 256, "synthetic", 0, 600.f, 0.f, 0, 0.f, NULL, NULL};
  char *metadata = rnn_amd_classify_construct_metadata(&md);
  int g_off[4], g_size[4], n_outputs = 0;
  const int n_groups = rnn_amd_classify_parse_classes(md.classes, g_off, g_size, 4, &n_outputs, NULL);
  char *filename = rnn_amd_classify_net_filename(md.basename, metadata, N_FEATURES, 0, hidden, n_outputs, 8000,
                                                 md.window_size);
  /* gstclassify's create_net (1060-1126): standard flags, momentum 0.95, Nesterov */
  RecurNN *net = rnn_new(N_FEATURES, hidden, n_outputs, RNN_NET_FLAG_STANDARD, 11, NULL, depth, lr, 0.95f, 0.0f, RNN_RELU);
  rnn_randomise_weights_auto(net);
  net->metadata = strdup(metadata);
  RecurNN **nets = rnn_new_training_set(net, channels);
  RnnAmdSet *set = rnn_amd_set_open(nets, channels);
  RnnAmdBalancedTraining *balance = bias > 0 ? rnn_amd_balanced_new(n_outputs, bias) : NULL;

  float *features = malloc(sizeof(float) * channels * N_FEATURES);
  int *targets = malloc(sizeof(int) * channels * n_groups), *left = calloc(channels, sizeof(int));
  int *cls = calloc(channels, sizeof(int));
  rand_ctx noise = {0xf1ea5eed, 7, 7, 7};
  for (int i = 0; i < 20; i++) uniform(&noise);
  RnnAmdStats st;
  rnn_amd_set_read_stats(set, &st, 1);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  long trained = 0;
  for (int g = 1; g <= generations; g++) {
    for (int ch = 0; ch < channels; ch++) {
      if (left[ch]-- <= 0) { /* the channel's sound changes: class 0 four times as likely as class 1 */
        cls[ch] = uniform(&noise) < 0.2f;
        left[ch] = 50 + (int)(150 * uniform(&noise));
      }
      targets[ch] = uniform(&noise) < 0.9f ? cls[ch] : -1; /* a tenth of the windows carry no label */
      for (int k = 0; k < N_FEATURES; k++) {
        const float tilt = cls[ch] ? (k - 16) * 0.04f : (16 - k) * 0.04f;
        features[ch * N_FEATURES + k] = tilt + 0.6f * (uniform(&noise) + uniform(&noise) - 1.0f);
      }
    }
    trained += rnn_amd_classify_generation(set, features, N_FEATURES, n_groups, g_off, g_size, targets, NULL, balance,
                                           RNN_MOMENTUM_NESTEROV, 2000.0f, exact);
    if (g % report == 0) {
      rnn_amd_set_read_stats(set, &st, 1);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
      t0 = t1;
      printf("generation %6d error %.4f correct %.4f trained %ld", g, st.count ? st.error / st.count : 0.0,
             st.count ? (double)st.correct / st.count : 0.0, trained);
      if (balance) printf(" seen %u/%u used %u/%u", balance->seen[0], balance->seen[1], balance->used[0], balance->used[1]);
      printf(" %.0f channel-windows/s\n", (double)report * channels / s);
      trained = 0;
    }
  }
  /* save under the reference's name, load back, check it is the net this configuration expects */
  rnn_amd_set_close(set);
  int rc = rnn_save_net(net, filename, 0);
  RecurNN *back = rc == 0 ? rnn_load_net(filename) : NULL;
  const int ok = back && rnn_amd_classify_check_net(back, metadata, hidden, 0, n_outputs, 0) == 0;
  printf("net file %s: %s\n", filename, ok ? "saved, loaded, metadata and sizes agree" : "FAILED");
  if (back) {
    RnnAmdClassifyMetadata seen = {0};
    const int missing = rnn_amd_classify_load_metadata(back->metadata, &seen);
    printf("metadata items missing: %d; classes %s window-size %d\n", missing, seen.classes, seen.window_size);
    rnn_amd_classify_free_metadata_items(&seen);
    rnn_delete_net(back);
  }
  unlink(filename);
  rnn_amd_balanced_free(balance);
  rnn_delete_training_set(nets, channels, 0);
  free(features); free(targets); free(left); free(cls); free(metadata); free(filename);
  return ok ? 0 : 1;
}
