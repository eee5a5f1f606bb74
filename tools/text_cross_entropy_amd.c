/* text_cross_entropy_amd.c -- what the reference's text-cross-entropy does
 * (text-cross-entropy.c:125-207, without the colour mode): load a net saved by a text
 * trainer, rebuild its alphabet from the net's metadata, and print the cross-entropy in bits
 * per character of each text file under the net's predictions.  The whole text runs
 * through the net on the device (rnn_char_cross_entropy -> rnn_amd_run_text).
 *
 *   text_cross_entropy_amd -f NET [-i ignore_first] [-m min_length] [-p prefix] TEXT...
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "recur-nn.h"
#include "recur_amd_char.h"

int main(int argc, char **argv) {
  const char *netfile = NULL, *prefix = NULL;
  int ignore_first = 0, min_length = 0, opt;
  while ((opt = getopt(argc, argv, "f:i:m:p:")) != -1) {
    switch (opt) {
    case 'f': netfile = optarg; break;
    case 'i': ignore_first = atoi(optarg); break;
    case 'm': min_length = atoi(optarg); break;
    case 'p': prefix = optarg; break;
    default: fprintf(stderr, "usage: %s -f NET [-i n] [-m n] [-p prefix] TEXT...\n", argv[0]); return 2;
    }
  }
  if (!netfile || optind >= argc) {
    fprintf(stderr, "usage: %s -f NET [-i n] [-m n] [-p prefix] TEXT...\n", argv[0]);
    return 2;
  }
  RecurNN *net = rnn_load_net(netfile);
  if (!net || !net->metadata) {
    fprintf(stderr, "'%s' is not a text net with metadata\n", netfile);
    return 1;
  }
  RnnCharAlphabet *alphabet = rnn_char_new_alphabet_from_net(net);
  int *char_to_net = rnn_char_new_char_lut(alphabet);
  u8 *prefix_text = NULL;
  int prefix_len = 0;
  if (prefix) {
    prefix_text = rnn_char_alloc_encoded_text(alphabet, prefix, (int)strlen(prefix), &prefix_len, NULL,
                                              false);
  }
  int count = 0;
  for (int i = optind; i < argc; i++) {
    char *raw;
    int raw_len;
    if (rnn_char_alloc_file_contents(argv[i], &raw, &raw_len)) {
      continue;
    }
    if (raw_len >= min_length) {
      int len;
      u8 *text = rnn_char_alloc_encoded_text(alphabet, raw, raw_len, &len, char_to_net, false);
      double entropy =
          rnn_char_cross_entropy(net, alphabet, text, len, ignore_first, prefix_text, prefix_len);
      printf("%s %.5f\n", argv[i], entropy);
      count++;
      free(text);
    }
    free(raw);
  }
  free(char_to_net);
  free(prefix_text);
  rnn_char_free_alphabet(alphabet);
  rnn_delete_net(net);
  fprintf(stderr, "processed %d texts\n", count);
  return 0;
}
