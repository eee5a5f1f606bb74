/* char_sampling.c -- drawing the next symbol of a text from a net's output row, on the host
 * (gnu11 C).  Used by the confabulation calls of char_epoch.c and char_multitext.c; the one
 * forward pass per symbol runs on the device through rnn_opinion.
 *
 * Behaviour follows the reference's samplers (charmodel-predict.c:29-60 for a plain net,
 * charmodel-multi-predict.c:122-142 for one head of a multi-head net; badmaths.h:71-156 for the
 * distribution), which are two copies of one procedure there and one procedure here: feed the
 * previous symbol, turn a slice of the output row into a distribution, draw from it with the
 * net's own generator.  The draws are part of parity (the generator's state is compared), so the
 * arithmetic is the reference's: its clamped softmax over its fast_expf, float accumulation of
 * the cumulative sum, one rand_double per attempt.
 */
#include "char_host.h"

int ramd_put_codepoint(unsigned cp, char *out, int utf8) {
  if (!utf8 || cp < 0x80) {
    out[0] = (char)cp;
    return 1;
  }
  /* 2, 3 or 4 bytes: the lead byte announces the length, every other byte carries six bits */
  static const unsigned char lead[5] = {0, 0, 0xC0, 0xE0, 0xF0};
  const int n = cp < 0x800 ? 2 : cp < 0x10000 ? 3 : cp < 0x200000 ? 4 : 0;
  for (int i = n - 1; i > 0; i--, cp >>= 6) {
    out[i] = (char)(0x80 | (cp & 0x3F));
  }
  if (n) {
    out[0] = (char)(lead[n] | cp);
  }
  return n;
}

/* exp(score + shift) normalised to sum 1.  The shift keeps the largest argument at or below 50 and,
 * where that leaves room, the smallest at or above -60 (badmaths.h:71-111).  dst may be src. */
static void clamped_softmax(float *dst, const float *src, int n) {
  float top = src[0], bottom = src[0];
  for (int i = 1; i < n; i++) {
    top = RAMD_MAX(top, src[i]);
    bottom = RAMD_MIN(bottom, src[i]);
  }
  float shift = 0.0f;
  if (top > 50.0f) {
    shift = 50.0f - top;
  } else if (bottom < -60.0f) {
    shift = RAMD_MIN(-60.0f - bottom, 50.0f - top);
  }
  float total = 0.0f;
  for (int i = 0; i < n; i++) {
    const float w = ramd_fast_expf(src[i] + shift);
    dst[i] = w;
    total += w;
  }
  for (int i = 0; i < n; i++) {
    dst[i] /= total;
  }
}

void ramd_text_distribution(float *p, const float *score, int n, float bias) {
  clamped_softmax(p, score, n);
  if (bias != 0) { /* badmaths.h:143-156: the probabilities, scaled, are added to the scores */
    for (int i = 0; i < n; i++) {
      p[i] = p[i] * bias + score[i];
    }
    clamped_softmax(p, p, n);
  }
}

float *ramd_feed_symbol(RecurNN *net, int hot, float presynaptic_noise) {
  /* the one-hot row goes where rnn_opinion(net, NULL, ..) will look for it */
  RecurExtraLayer *bottom = net->bottom_layer;
  float *row = bottom ? bottom->inputs : net->real_inputs;
  const int width = bottom ? bottom->input_size : net->input_size;
  memset(row, 0, sizeof(float) * width);
  row[hot] = 1.0f;
  return rnn_opinion(net, NULL, presynaptic_noise);
}

/* index of the first cumulative sum that exceeds a uniform draw; a draw beyond the (rounded) total
 * is simply repeated */
static int draw_index(rand_ctx *rng, const float *p, int n) {
  for (;;) {
    const float u = ramd_rand_double(rng);
    float reached = 0.0f;
    for (int i = 0; i < n; i++) {
      reached += p[i];
      if (u < reached) {
        return i;
      }
    }
  }
}

int ramd_next_symbol(RecurNN *net, int hot, float bias, int head, int n, int greedy, float *work) {
  const float *score = ramd_feed_symbol(net, hot, 0) + (size_t)head * n;
  if (greedy && bias >= 100) { /* no draw at all: the LAST of equal maxima, as the reference's >= picks */
    int best = 0;
    for (int i = 1; i < n; i++) {
      if (score[i] >= score[best]) {
        best = i;
      }
    }
    return best;
  }
  ramd_text_distribution(work, score, n, bias);
  /* rnn_opinion has just brought the host struct up to date, generator included; after the draw the
   * device's copy follows the host's */
  const int pick = draw_index(&net->rng, work, n);
  ramd_rng_from_host(net);
  return pick;
}

/* rnn_char_confabulate (charmodel.h:190-192; charmodel-predict.c:137-181): up to char_len symbols
 * into dest as text.  With a start_point the net is first run until that symbol turns up (it becomes
 * the first character written); a stop_point ends the passage after it has been written. */
int rnn_char_confabulate(RecurNN *net, char *dest, int char_len, int byte_len, RnnCharAlphabet *a,
                         float bias, int *prev_char, int start_point, int stop_point) {
  const int utf8 = (a->flags & RNN_CHAR_FLAG_UTF8) != 0;
  const int room = byte_len - (utf8 ? 5 : 1); /* a symbol may need four bytes, then the NUL */
  if (room <= 0) {
    fprintf(stderr, "insufficient space to confabulate (%d bytes)\n", byte_len);
    if (byte_len) {
      dest[0] = 0;
    }
    return 0;
  }
  float *work = malloc(sizeof(float) * net->output_size);
  int sym = *prev_char, used = 0;
  if (start_point >= 0 && char_len > 0) {
    int tries = 0;
    while (sym != start_point && tries < 1000000) {
      sym = ramd_next_symbol(net, sym, bias, 0, net->output_size, 1, work);
      tries++;
    }
    used = ramd_put_codepoint(a->points[sym], dest, utf8);
    dest[used] = 0;
    fprintf(stderr, sym == start_point ? "start char '%s' found after %d others\n"
                                       : "start char '%s' not found in first %d characters, giving up\n",
            dest, tries);
  }
  for (int written = 0; written < char_len && used < room; written++) {
    sym = ramd_next_symbol(net, sym, bias, 0, net->output_size, 1, work);
    used += ramd_put_codepoint(a->points[sym], dest + used, utf8);
    if (sym == stop_point) {
      break;
    }
  }
  dest[used] = 0;
  *prev_char = sym;
  free(work);
  return used;
}
