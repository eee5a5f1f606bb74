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
 *                    [-V validate_chars] [-S seed] [-n net_file] [-G gpus]
 *                    [-a activation] [-L learning_style] [-N presynaptic_noise] [-B ballast]
 *
 * -a / -L take the reference's numbers (text-predict's --activation and --learning-style,
 * text-predict.c:249-262; recur-nn.h:109-131): activation 1 ReLU (default), 2 ReSQRT, 5 ReCLIP20;
 * learning style 0 weighted momentum (default), 1 Nesterov, 2 simplified Nesterov, 3 classical
 * momentum, 4 adagrad, 5 adadelta, 6 rprop (4-6 get the auxiliary arrays; -B is adagrad's / adadelta's
 * starting accumulator, text-predict.c:546-557).
 *
 * -G n shards the t streams over n GPUs, one PROCESS per GPU (forked here before anything
 * touches a device): rank 0 makes the RCCL id and hands it over in shared memory, every
 * rank joins with rnn_amd_dist_init, builds its shard of the training set with
 * rnn_amd_new_training_set_shard, and rnn_amd_set_char_step then sums the weight deltas
 * over the ranks inside the library (one all-reduce per generation).  -G 1 takes the same
 * path with one rank.  Reports come from rank 0 (its own streams' statistics).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/wait.h>
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
  int hidden = 199, streams = 64, depth = 30, stop = 2000, report = 200, validate = 4000, gpus = 0;
  int activation = RNN_RELU, style = RNN_MOMENTUM_WEIGHTED;
  float lr = 1e-4f, momentum = 0.95f, noise = 0.0f, ballast = -1.0f;
  unsigned long long seed = 1;
  int opt;
  while ((opt = getopt(argc, argv, "f:H:t:d:l:m:s:r:V:S:n:G:a:L:N:B:")) != -1) {
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
    case 'G': gpus = atoi(optarg); break;
    case 'a': activation = atoi(optarg); break;
    case 'L': style = atoi(optarg); break;
    case 'N': noise = atof(optarg); break;
    case 'B': ballast = atof(optarg); break;
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
  if ((activation != RNN_RELU && activation != RNN_RESQRT && activation != RNN_RECLIP20) || style < 0 ||
      style >= RNN_LAST_LEARNING_METHOD) {
    fprintf(stderr, "-a %d / -L %d: see the comment at the top of %s\n", activation, style, __FILE__);
    return 2;
  }
  u32 flags = RNN_NET_FLAG_STANDARD | RNN_NET_FLAG_BPTT_ADAPTIVE_MIN_ERROR;
  if (style == RNN_ADADELTA || style == RNN_RPROP) {
    flags |= RNN_NET_FLAG_AUX_ARRAYS; /* text-predict.c:418-421 */
  }
  RecurNN *net = rnn_new(alphabet->len, hidden, alphabet->len, flags, seed, NULL, depth, lr,
                         momentum, noise, activation);
  struct RecurInitialisationParameters p;
  rnn_init_default_weight_parameters(net, &p);
  p.method = RNN_INIT_FLAT;
  p.flat_shape = RNN_INIT_DIST_SEMICIRCLE;
  p.flat_perforation = 0;
  rnn_randomise_weights_clever(net, &p);
  if (style == RNN_ADAGRAD || style == RNN_ADADELTA) { /* the accumulators' starting value (text-predict.c:546-557) */
    rnn_set_momentum_values(net, ballast >= 0 ? ballast : style == RNN_ADAGRAD ? 200.0f : 0.0f); /* text-predict.c:106-107 */
  } else if (style == RNN_RPROP) {
    rnn_set_aux_values(net, 1); /* text-predict.c:558-560 */
  }
  /* -G: one process per GPU, started before any device call */
  int rank = 0, world = gpus > 0 ? gpus : 1;
  pid_t kids[64];
  if (gpus > 0) {
    if (gpus > 64 || streams % gpus) {
      fprintf(stderr, "-G %d: 1 to 64 GPUs, and the %d streams must divide evenly\n", gpus, streams);
      return 2;
    }
    struct {
      volatile int ready;
      char id[RNN_AMD_DIST_ID_BYTES];
    } *shared = mmap(NULL, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (shared == MAP_FAILED) {
      perror("mmap");
      return 1;
    }
    shared->ready = 0;
    fflush(NULL);
    for (int r = 1; r < world; r++) {
      kids[r] = fork();
      if (kids[r] == 0) {
        rank = r;
        break;
      }
    }
    rnn_amd_use_device(rank, NULL);
    if (rank == 0) {
      if (rnn_amd_dist_get_id(shared->id)) {
        return 1;
      }
      __sync_synchronize();
      shared->ready = 1;
    } else {
      while (!shared->ready) {
        usleep(1000);
      }
      __sync_synchronize();
    }
    if (rnn_amd_dist_init(rank, world, shared->id)) {
      return 1;
    }
    streams /= world; /* per rank from here on */
  }
  RecurNN **nets = gpus > 0 ? rnn_amd_new_training_set_shard(net, streams, rank * streams, world * streams)
                            : rnn_new_training_set(net, streams);
  RnnAmdSet *set = rnn_amd_set_open(nets, streams); /* with a group joined: this rank's shard */
  rnn_amd_set_load_text(set, text, len);

  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < len - 1 && (int)net->generation < stop; i++) {
    float m = rnn_calculate_momentum_soft_start(net->generation, momentum, 0);
    rnn_amd_set_char_step(set, i, style, m);
    if (net->generation % report == 0 && rank == 0) {
      RnnAmdStats st;
      rnn_amd_set_read_stats(set, &st, 1);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
      double v = cross_entropy(net, alphabet, vtext, validate, 5);
      printf("generation %6u t_entropy %.4f v_entropy %.4f accuracy %.3f depth %.1f  %.0f/s\n",
             net->generation, -st.entropy / st.count, v, (double)st.correct / st.count,
             st.bptt_depth_sum / st.count, world * st.count / secs);
      fflush(stdout);
      clock_gettime(CLOCK_MONOTONIC, &t0);
    }
  }
  if (save && rank == 0) {
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
  int status = 0;
  if (gpus > 0) {
    rnn_amd_dist_barrier();
    rnn_amd_dist_finalize();
    if (rank != 0) {
      fflush(NULL);
      _exit(0);
    }
    for (int r = 1; r < world; r++) {
      int st = 0;
      waitpid(kids[r], &st, 0);
      if (!WIFEXITED(st) || WEXITSTATUS(st)) {
        fprintf(stderr, "rank %d failed (status %d)\n", r, st);
        status = 1;
      }
    }
  }
  rnn_delete_training_set(nets, streams, 0);
  rnn_char_free_alphabet(alphabet);
  free(text);
  return status;
}
