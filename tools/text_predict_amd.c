/* text_predict_amd.c -- a small C driver for the text model on the MI355X core.
 *
 * It does what the reference's text-predict does with --multi-tap N
 * (text-predict.c:472-672 -> rnn_char_epoch, charmodel-predict.c:260-405):
 * default alphabet, flat semicircle initialisation, N weight-sharing streams
 * spaced through the text, softmax loss, BPTT, one weighted-momentum update per
 * generation, periodic report of training entropy and of the validation
 * cross-entropy -- but through the batched entry points of recur_amd.h, so the
 * whole generation stays on the GPU.  It is also the worked example of
 * plain C against include/ and librecur_amd.so.
 *
 *   text_predict_amd [-f text] [-H hidden] [-t streams] [-d depth] [-l learn_rate]
 *                    [-m momentum] [-s stop_generation] [-r report_interval]
 *                    [-V validate_chars] [-S seed] [-n net_file]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "recur-nn.h"
#include "recur_amd_char.h"

#define DEFAULT_CHARSET "8 etaonihsrdlucmwfygpb,v.k-;x\"qj'?:z)(_!*&" /* text-predict.c:44 */

#define DEFAULT_COLLAPSE_CHARS "10872}{659/34][@"                    /* text-predict.c:45 */

/* the validation pass of text-predict (text-predict.c:538-541, charmodel-predict.c:62-80):
 * a forward-only clone that borrows the weights, run over the text on the device */
static double cross_entropy(RecurNN *net, RnnCharAlphabet *alphabet, const u8 *text, int len,
                            int skip) {
  RecurNN *v = rnn_clone(net, net->flags & ~(RNN_NET_FLAG_OWN_BPTT | RNN_NET_FLAG_OWN_WEIGHTS),
                         RECUR_RNG_SUBSEED, NULL);
  double entropy = rnn_char_cross_entropy(v, alphabet, text, len, skip, NULL, 0);
  rnn_delete_net(v);
  return entropy;
}

int main(int argc, char **argv) {
  const char *file = "tests/golden/erewhon.txt", *save = NULL;
  int hidden = 199, streams = 64, depth = 30, stop = 2000, report = 200, validate = 4000;
  float lr = 1e-4f, momentum = 0.95f;
  unsigned long long seed = 1;
  int opt;
  while ((opt = getopt(argc, argv, "f:H:t:d:l:m:s:r:V:S:n:")) != -1) {
    switch (opt) {
    case 'f': file = optarg; break;
    case 'H': hidden = atoi(optarg); break;
    case 't': streams = atoi(optarg); break;
    case 'd': depth = atoi(optarg); break;
    case 'l': lr = atof(optarg); break;
    case 'm': momentum = atof(optarg); break;
    case 's': stop = atoi(optarg); break;
    case 'r': report = atoi(optarg); break;
    case 'V': validate = atoi(optarg); break;
    case 'S': seed = strtoull(optarg, NULL, 10); break;
    case 'n': save = optarg; break;
    default:
      fprintf(stderr, "see the comment at the top of %s\n", __FILE__);
      return 2;
    }
  }
  /* alphabet and text (text-predict.c:698-719; charmodel-init.c:334-349) */
  RnnCharAlphabet *alphabet = rnn_char_new_alphabet();
  rnn_char_alphabet_set_flags(alphabet, 1, 0, 1);
  alphabet->len = alphabet->collapsed_len = (int)strlen(DEFAULT_CHARSET);
  for (int i = 0; i < alphabet->len; i++) {
    alphabet->points[i] = alphabet->collapsed_points[i] = (unsigned char)DEFAULT_CHARSET[i];
  }
  int len = 0;
  u8 *text = rnn_char_load_new_encoded_text(file, alphabet, &len, 1);
  if (!text || len < validate + 1000) {
    fprintf(stderr, "cannot use '%s'\n", file);
    return 1;
  }
  const u8 *vtext = text + len - validate;
  len -= validate;

  /* net (text-predict.c:414-437, 360-395) and its training set */
  u32 flags = RNN_NET_FLAG_STANDARD | RNN_NET_FLAG_BPTT_ADAPTIVE_MIN_ERROR;
  RecurNN *net = rnn_new(alphabet->len, hidden, alphabet->len, flags, seed, NULL, depth, lr,
                         momentum, 0.0f, RNN_RELU);
  struct RecurInitialisationParameters p;
  rnn_init_default_weight_parameters(net, &p);
  p.method = RNN_INIT_FLAT;
  p.flat_shape = RNN_INIT_DIST_SEMICIRCLE;
  p.flat_perforation = 0;
  rnn_randomise_weights_clever(net, &p);
  RecurNN **nets = rnn_new_training_set(net, streams);
  RnnAmdSet *set = rnn_amd_set_open(nets, streams);
  rnn_amd_set_load_text(set, text, len);

  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < len - 1 && (int)net->generation < stop; i++) {
    float m = rnn_calculate_momentum_soft_start(net->generation, momentum, 0);
    rnn_amd_set_char_step(set, i, RNN_MOMENTUM_WEIGHTED, m);
    if (net->generation % report == 0) {
      RnnAmdStats st;
      rnn_amd_set_read_stats(set, &st, 1);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
      double v = cross_entropy(net, alphabet, vtext, validate, 5);
      printf("generation %6u t_entropy %.4f v_entropy %.4f accuracy %.3f depth %.1f  %.0f/s\n",
             net->generation, -st.entropy / st.count, v, (double)st.correct / st.count,
             st.bptt_depth_sum / st.count, st.count / secs);
      fflush(stdout);
      clock_gettime(CLOCK_MONOTONIC, &t0);
    }
  }
  if (save) {
    /* the alphabet travels in the net's metadata (text-predict.c:486-497), which is what
     * text_cross_entropy_amd and text_confabulate_amd rebuild it from */
    struct RnnCharMetadata m = {DEFAULT_CHARSET, DEFAULT_COLLAPSE_CHARS, 0, 1, 1};
    free(net->metadata);
    net->metadata = rnn_char_construct_metadata(&m);
    if (rnn_save_net(net, save, 1)) {
      return 1;
    }
    printf("saved %s\n", save);
  }
  rnn_amd_set_close(set);
  rnn_delete_training_set(nets, streams, 0);
  rnn_char_free_alphabet(alphabet);
  free(text);
  return 0;
}
