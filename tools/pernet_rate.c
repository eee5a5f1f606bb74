/* pernet_rate.c -- how fast the reference's OWN per-net call sequence runs on librecur_amd.so.
 *
 * The loop below is rnn_char_epoch's (charmodel-predict.c:260-311) written out with the plain
 * recur-nn.h calls, as an unchanged caller would make them: one_hot_opinion + softmax error on
 * the host (charmodel-helpers.h:16-33, charmodel-predict.c:18-27), then for one net
 * rnn_bptt_calculate, for several nets rnn_bptt_calc_deltas(j ? 1 : 0) and one
 * rnn_apply_learning.  Prints stream-timesteps/s.
 *
 *   pernet_rate [-H hidden] [-t streams] [-d depth] [-s steps]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "recur-nn.h"

static float one_step(RecurNN *net, int in, int target) {
  memset(net->real_inputs, 0, net->input_size * sizeof(float));
  net->real_inputs[in] = 1.0f;
  float *out = rnn_opinion(net, NULL, 0);
  float *err = net->bptt->o_error;
  int n = net->output_size;
  float hi = out[0], sum = 0.0f;
  for (int i = 1; i < n; i++) hi = out[i] > hi ? out[i] : hi;
  for (int i = 0; i < n; i++) sum += (err[i] = expf(out[i] - hi));
  for (int i = 0; i < n; i++) err[i] = -err[i] / sum;
  err[target] += 1.0f;
  return err[target];
}

int main(int argc, char **argv) {
  int hidden = 99, streams = 1, depth = 30, steps = 2000, opt;
  while ((opt = getopt(argc, argv, "H:t:d:s:")) != -1) {
    switch (opt) {
    case 'H': hidden = atoi(optarg); break;
    case 't': streams = atoi(optarg); break;
    case 'd': depth = atoi(optarg); break;
    case 's': steps = atoi(optarg); break;
    default: return 2;
    }
  }
  const int A = 42;
  u32 flags = RNN_NET_FLAG_STANDARD | RNN_NET_FLAG_BPTT_ADAPTIVE_MIN_ERROR;
  RecurNN *net = rnn_new(A, hidden, A, flags, 1, NULL, depth, 1e-4f, 0.95f, 0.0f, RNN_RELU);
  struct RecurInitialisationParameters p;
  rnn_init_default_weight_parameters(net, &p);
  p.method = RNN_INIT_FLAT;
  p.flat_shape = RNN_INIT_DIST_SEMICIRCLE;
  p.flat_perforation = 0;
  rnn_randomise_weights_clever(net, &p);
  RecurNN **nets = rnn_new_training_set(net, streams);
  unsigned x = 12345;
  unsigned char *text = malloc(steps + 64 * streams + 2);
  for (int i = 0; i < steps + 64 * streams + 2; i++) {
    x = x * 1664525u + 1013904223u;
    text[i] = (x >> 16) % A;
  }
  struct timespec t0, t1;
  double err = 0;
  for (int i = 0; i < steps + 50; i++) {
    if (i == 50) clock_gettime(CLOCK_MONOTONIC, &t0); /* after the warm-up */
    if (streams == 1) {
      rnn_bptt_advance(net);
      err += one_step(net, text[i], text[i + 1]);
      rnn_bptt_calculate(net, 1);
    } else {
      for (int j = 0; j < streams; j++) {
        rnn_bptt_advance(nets[j]);
        err += one_step(nets[j], text[i + 64 * j], text[i + 64 * j + 1]);
        rnn_bptt_calc_deltas(nets[j], j ? 1 : 0, NULL);
      }
      rnn_apply_learning(net, RNN_MOMENTUM_WEIGHTED, 0.95f);
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  printf("per-net calls: hidden %d, %d stream(s), depth %d: %.0f stream-timesteps/s (%.1f us per stream-step), mean error %.6f\n",
         hidden, streams, depth, steps * (double)streams / secs, 1e6 * secs / (steps * (double)streams),
         err / ((steps + 50) * (double)streams));
  rnn_delete_training_set(nets, streams, 0);
  free(text);
  return 0;
}
