/* text_confabulate_amd.c -- what the reference's text-confabulate does
 * (text-confabulate.c:50-103): load a text net, optionally prime it with a prefix, and
 * sample characters from its predictions.
 *
 *   text_confabulate_amd -f NET [-B bias] [-n chars] [-p prefix] [-u until_char]
 *                        [-w wait_for_char] [-r seed]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "recur-nn.h"
#include "recur_amd_char.h"

/* init_rand64 of recur-rng.h:33-43: callers of the reference seed net->rng themselves
 * (text-confabulate.c:67) */
static void seed_rng(rand_ctx *x, u64 seed) {
  x->a = 0xf1ea5eed;
  x->b = x->c = x->d = seed;
  for (int i = 0; i < 20; i++) {
    u64 e = x->a - ((x->b << 7) | (x->b >> 57));
    x->a = x->b ^ ((x->c << 13) | (x->c >> 51));
    x->b = x->c + ((x->d << 37) | (x->d >> 27));
    x->c = x->d + e;
    x->d = e + x->a;
  }
}

int main(int argc, char **argv) {
  const char *netfile = NULL, *prefix = NULL, *until = NULL, *wait_for = NULL;
  float bias = 0;
  int chars = 72, opt;
  long long seed = 2;
  while ((opt = getopt(argc, argv, "f:B:n:p:u:w:r:")) != -1) {
    switch (opt) {
    case 'f': netfile = optarg; break;
    case 'B': bias = atof(optarg); break;
    case 'n': chars = atoi(optarg); break;
    case 'p': prefix = optarg; break;
    case 'u': until = optarg; break;
    case 'w': wait_for = optarg; break;
    case 'r': seed = atoll(optarg); break;
    default: fprintf(stderr, "usage: %s -f NET [-B bias] [-n chars] [-p prefix]\n", argv[0]); return 2;
    }
  }
  RecurNN *net = netfile ? rnn_load_net(netfile) : NULL;
  if (!net || !net->metadata) {
    fprintf(stderr, "need -f NET (a text net with metadata)\n");
    return 1;
  }
  RnnCharAlphabet *alphabet = rnn_char_new_alphabet_from_net(net);
  seed_rng(&net->rng, (u64)seed);
  rnn_amd_host_written(net, RNN_AMD_STREAM);
  int prev_char = 0;
  if (prefix) {
    int prefix_len;
    u8 *prefix_text = rnn_char_alloc_encoded_text(alphabet, prefix, (int)strlen(prefix), &prefix_len,
                                                  NULL, false);
    prev_char = rnn_char_prime(net, alphabet, prefix_text, prefix_len);
    free(prefix_text);
  }
  int byte_len = chars * 4 + 5;
  char *t = malloc(byte_len);
  int stop_point = until ? rnn_char_get_codepoint(alphabet, until) : -1;
  int start_point = wait_for ? rnn_char_get_codepoint(alphabet, wait_for) : -1;
  rnn_char_confabulate(net, t, chars, byte_len, alphabet, bias, &prev_char, start_point, stop_point);
  fputs(t, stdout);
  fputs("\n", stdout);
  free(t);
  rnn_char_free_alphabet(alphabet);
  rnn_delete_net(net);
  return 0;
}
