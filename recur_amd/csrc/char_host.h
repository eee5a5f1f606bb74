/* char_host.h -- what the host-side text layer's three files share (char_sampling.c,
 * char_epoch.c, char_multitext.c).  Not installed: the public face is include/recur_amd_char.h. */
#ifndef RAMD_CHAR_HOST_H
#define RAMD_CHAR_HOST_H
#include "rnn_host.h"
#include "recur_amd_char.h"
#include <time.h>

/* ---- char_sampling.c ---- */
/* code point -> bytes (plain byte when !utf8); returns the bytes written, 0 for an unencodable value */
int ramd_put_codepoint(unsigned cp, char *out, int utf8);
/* badmaths.h:71-156 semantics: p = softmax(score) with the reference's range clamps and its
 * fast_expf; with a bias the distribution is sharpened: softmax(bias * p + score).  In place is fine. */
void ramd_text_distribution(float *p, const float *score, int n, float bias);
/* one forward pass on symbol `hot` (charmodel-helpers.h:16-33: a net with a bottom layer takes the
 * one-hot row in the layer's input buffer), returns the output row */
float *ramd_feed_symbol(RecurNN *net, int hot, float presynaptic_noise);
/* feeds `hot`, then picks the next symbol from outputs [head * n, head * n + n): the best one when
 * `greedy` and bias >= 100, otherwise a draw from the biased distribution with the net's own generator
 * (kept in step with the device).  `work` holds n floats. */
int ramd_next_symbol(RecurNN *net, int hot, float bias, int head, int n, int greedy, float *work);

/* ---- charmodel_meta.c ---- */
uint32_t ramd_hash32(const char *s); /* recur-common.h:207-216: the signature in net file names */

/* ---- rnn_dump.c ---- */
void ramd_temporal_row(TemporalPPM *ppm, const float *row);
/* the two image rows a caller may have asked for (model->images / the multi-text arguments) */
void ramd_image_rows(RecurNN *net, TemporalPPM *input_ppm, TemporalPPM *error_ppm);

static inline float ramd_capped_log2f(float x) { /* charmodel-helpers.h:11-14 */
  return (x < 1e-30f) ? -100.0f : log2f(x);
}
static inline double ramd_seconds_since(struct timespec *t0) { /* and restarts the clock */
  struct timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  double s = (double)(t1.tv_sec - t0->tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0->tv_nsec);
  *t0 = t1;
  return s;
}
#endif
