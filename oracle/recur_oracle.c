/* recur_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see recur_oracle.h).
 *
 * CPU restatement of the reference hot path.  Each function names the
 * reference lines it follows (paths relative to /root/reference).  The
 * arithmetic keeps the reference's operation order so that a build without
 * -ffast-math agrees with the reference built without -ffast-math to rounding.
 */
#include "recur_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define ROT64(x, k) (((x) << (k)) | ((x) >> (64 - (k))))
#define ORC_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define ORC_MAX(a, b) (((a) >= (b)) ? (a) : (b))

/* ---------------------------------------------------------------- PRNG -- */

/* recur-rng.h:22-31 */
uint64_t orc_rand64(OrcRng *x) {
  uint64_t e = x->a - ROT64(x->b, 7);
  x->a = x->b ^ ROT64(x->c, 13);
  x->b = x->c + ROT64(x->d, 37);
  x->c = x->d + e;
  x->d = e + x->a;
  return x->d;
}

/* recur-rng.h:34-43 */
void orc_init_rand64(OrcRng *x, uint64_t seed) {
  x->a = 0xf1ea5eed;
  x->b = x->c = x->d = seed;
  for (int i = 0; i < 20; i++) {
    (void)orc_rand64(x);
  }
}

/* recur-rng.h:57-78: 52 random mantissa bits under exponent 0 -> [1,2) - 1 */
double orc_rand_double(OrcRng *x) {
  union {
    uint64_t i;
    double d;
  } u;
  u.i = (orc_rand64(x) & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
  return u.d - 1.0;
}

/* recur-rng.h:96-100 */
int orc_rand_small_int(OrcRng *x, int cap) {
  double d = orc_rand_double(x) * cap;
  return (int)d;
}

/* recur-rng.h:179-200: Irwin-Hall sum of twelve 16-bit fields */
float orc_cheap_gaussian_noise(OrcRng *x) {
  int64_t a = 0;
  for (int word = 0; word < 3; word++) {
    uint64_t bits = orc_rand64(x);
    for (int k = 0; k < 4; k++) {
      a += (int64_t)(bits & 0xffff);
      bits >>= 16;
    }
  }
  return (float)(a - 0xffff * 6) / (0xffff);
}

/* --------------------------------------------------------- small maths -- */

/* badmaths.h:14-29 */
float orc_fast_expf(float x) {
  int count = 0;
  while (fabsf(x) > 0.2) {
    x *= 0.125;
    count++;
  }
  float a = ((x + 3) * (x + 3) + 3) / ((x - 3) * (x - 3) + 3);
  while (count) {
    a *= a;
    a *= a;
    a *= a;
    count--;
  }
  return a;
}

/* badmaths.h:71-111 */
void orc_softmax(float *dest, const float *src, int len) {
  float sum = 0.0f, adj = 0.0f;
  float lo = src[0], hi = src[0];
  const float max_exp = 50.0f, min_exp = -60.0f;
  for (int i = 1; i < len; i++) {
    hi = ORC_MAX(hi, src[i]);
    lo = ORC_MIN(lo, src[i]);
  }
  if (hi > max_exp) {
    adj = max_exp - hi;
  } else if (lo < min_exp) {
    adj = ORC_MIN(min_exp - lo, max_exp - hi);
  }
  for (int i = 0; i < len; i++) {
    float x = orc_fast_expf(src[i] + adj);
    sum += x;
    dest[i] = x;
  }
  for (int i = 0; i < len; i++) {
    dest[i] /= sum;
  }
}

/* badmaths.h:113-141 */
int orc_softmax_best_guess(float *error, const float *src, int len) {
  orc_softmax(error, src, len);
  int best_i = 0;
  float best_e = error[0];
  error[0] = -best_e;
  for (int i = 1; i < len; i++) {
    float e = error[i];
    if (e > best_e) {
      best_e = e;
      best_i = i;
    }
    error[i] = -e;
  }
  return best_i;
}

/* recur-nn-helpers.h:104-113 */
float orc_soft_clip(float sum, float halfmax) {
  if (halfmax == 0) {
    return sum;
  }
  float x = sum / halfmax;
  float fudge = 0.99 + x * x / 100;
  return 2.0f * x / (1 + x * x * fudge);
}

/* recur-nn-helpers.h:115-124 */
static float softclip_scale(float sum, float halfmax, float *array, int len) {
  if (sum > halfmax) {
    float scale = orc_soft_clip(sum, halfmax);
    for (int i = 0; i < len; i++) {
      array[i] *= scale;
    }
    return scale * sum;
  }
  return sum;
}

/* charmodel-helpers.h:11-13 */
float orc_capped_log2f(float x) { return (x < 1e-30f) ? -100.0f : log2f(x); }

/* recur-nn.c:595-599 */
float orc_momentum_soft_start(float generation, float max_momentum, float x) {
  return ORC_MIN(max_momentum, 1.0f - x / (1.0f + generation + 2.0f * x));
}

/* recur-nn-init.c:87-91 with ALIGNED_VECTOR_LEN of recur-nn-helpers.h:20 */
void orc_padded_sizes(int input_size, int hidden_size, int output_size, int *I,
                      int *H, int *O) {
  *I = (hidden_size + input_size + 1 + 3) & ~3;
  *H = (hidden_size + 1 + 3) & ~3;
  *O = (output_size + 3) & ~3;
}

/* ------------------------------------------------------- set life cycle -- */

static float *zeros(size_t n) {
  float *p = calloc(n ? n : 1, sizeof(float));
  if (!p) {
    fprintf(stderr, "oracle: out of memory (%zu floats)\n", n);
    abort();
  }
  return p;
}

/* rnn_new + new_bptt + rnn_new_training_set allocation semantics
 * (recur-nn-init.c:6-143, 221-243, 296-350): one shared copy of weights,
 * momentums and deltas; per-stream history, layers, error vectors, rng,
 * generation, learn_rate copy and min_error_factor.  Clone seeds are drawn
 * later by orc_set_seed_clones so that weight initialisation (which consumes
 * the prototype's rng first in text-predict.c:414-437, 529-533) can go first. */
OrcSet *orc_set_new(int input_size, int hidden_size, int output_size, int S,
                    int D, int activation, uint32_t flags, float learn_rate,
                    uint64_t seed) {
  OrcSet *z = calloc(1, sizeof(OrcSet));
  z->input_size = input_size;
  z->hidden_size = hidden_size;
  z->output_size = output_size;
  orc_padded_sizes(input_size, hidden_size, output_size, &z->I, &z->H, &z->O);
  z->S = S;
  z->D = D;
  z->activation = activation;
  z->flags = flags;
  size_t ih = (size_t)z->I * z->H, ho = (size_t)z->H * z->O;
  z->ih_w = zeros(ih);
  z->ho_w = zeros(ho);
  z->ih_m = zeros(ih);
  z->ho_m = zeros(ho);
  z->ih_aux = zeros(ih);
  z->ho_aux = zeros(ho);
  z->ih_delta = zeros(ih);
  z->ho_delta = zeros(ho);
  z->ih_delta_tmp = zeros(ih);
  z->hist = zeros((size_t)D * S * z->I);
  z->hidden = zeros((size_t)S * z->H);
  z->output = zeros((size_t)S * z->O);
  z->o_error = zeros((size_t)S * z->O);
  z->err_a = zeros((size_t)S * z->I);
  z->err_b = zeros((size_t)S * z->I);
  z->index = calloc(S, sizeof(int));
  z->learn_rate = zeros(S);
  z->min_error_factor = zeros(S);
  z->ih_scale = zeros(S);
  z->top_error_raw = zeros(S);
  z->top_error_scaled = zeros(S);
  z->bptt_error = zeros(S);
  z->bptt_depth = calloc(S, sizeof(int));
  z->generation = calloc(S, sizeof(uint32_t));
  z->rng = calloc(S, sizeof(OrcRng));
  z->ho_scale = 1.0f;
  z->momentum_weight = 0.5f; /* RNN_MOMENTUM_WEIGHT, recur-nn.h:107 */
  for (int s = 0; s < S; s++) {
    z->learn_rate[s] = learn_rate;
    z->ih_scale[s] = 1.0f;
    /* recur-nn-init.c:73: BASE_MIN_ERROR_FACTOR * h_size */
    z->min_error_factor[s] = 1e-12f * z->H;
    /* new_bptt sets index 0, rnn_new then advances once (recur-nn-init.c:70,133) */
    z->index[s] = (D > 1) ? 1 : 0;
  }
  orc_init_rand64(&z->rng[0], seed);
  return z;
}

/* rnn_new_extra_layer (recur-nn-init.c:158-192) */
void orc_set_add_bottom(OrcSet *z, int n_inputs) {
  z->b_in = n_inputs;
  z->bI = (n_inputs + 1 + 3) & ~3;
  z->bO = (z->input_size + 3) & ~3;
  size_t n = (size_t)z->bI * z->bO;
  z->b_w = zeros(n);
  z->b_m = zeros(n);
  z->b_aux = zeros(n);
  z->b_delta = zeros(n);
  z->b_inputs = zeros(z->bI);
  z->b_outputs = zeros(z->bO);
  z->b_o_error = zeros(z->bO);
  z->b_learn_rate_scale = 1.0f;
}

/* rnn_clone's RECUR_RNG_SUBSEED branch (recur-nn-init.c:300-305), in the
 * order rnn_new_training_set clones (recur-nn-init.c:236-241). */
void orc_set_seed_clones(OrcSet *z) {
  for (int s = 1; s < z->S; s++) {
    uint64_t seed;
    do {
      seed = orc_rand64(&z->rng[0]);
    } while (seed == (uint64_t)-1);
    orc_init_rand64(&z->rng[s], seed);
    z->generation[s] = z->generation[0];
    z->learn_rate[s] = z->learn_rate[0];
  }
}

void orc_set_free(OrcSet *z) {
  if (!z)
    return;
  free(z->ih_w);
  free(z->ho_w);
  free(z->ih_m);
  free(z->ho_m);
  free(z->ih_aux);
  free(z->ho_aux);
  free(z->ih_delta);
  free(z->ho_delta);
  free(z->ih_delta_tmp);
  free(z->b_w);
  free(z->b_m);
  free(z->b_aux);
  free(z->b_delta);
  free(z->b_inputs);
  free(z->b_outputs);
  free(z->b_o_error);
  free(z->hist);
  free(z->hidden);
  free(z->output);
  free(z->o_error);
  free(z->err_a);
  free(z->err_b);
  free(z->index);
  free(z->learn_rate);
  free(z->min_error_factor);
  free(z->ih_scale);
  free(z->top_error_raw);
  free(z->top_error_scaled);
  free(z->bptt_error);
  free(z->bptt_depth);
  free(z->generation);
  free(z->rng);
  free(z);
}

/* randomise_array_flat (recur-nn-init.c:495-545) */
static void fill_flat(OrcRng *rng, float *array, int width, int height, int stride,
                      int offset, float variance, int shape, double perforation) {
  float stddev = sqrtf(variance);
  for (int y = 0; y < height; y++) {
    for (int x = offset; x < width + offset; x++) {
      if (perforation == 0 || orc_rand_double(rng) > perforation) {
        switch (shape) {
        case ORC_DIST_UNIFORM: {
          const double range = sqrtf(12.0f * variance);
          array[y * stride + x] = range * orc_rand_double(rng) - range * 0.5;
        } break;
        default:
        case ORC_DIST_GAUSSIAN:
          array[y * stride + x] = stddev * orc_cheap_gaussian_noise(rng);
          break;
        case ORC_DIST_LOG_NORMAL: {
          float a = orc_cheap_gaussian_noise(rng) * 0.33;
          float b = 0.9 * stddev * orc_fast_expf(a);
          array[y * stride + x] = (orc_rand64(rng) & 1) ? b : -b;
        } break;
        case ORC_DIST_SEMICIRCLE: {
          double a, b;
          do {
            a = orc_rand_double(rng) * 2.0 - 1.0;
            b = orc_rand_double(rng);
          } while (a * a + b * b > 1.0);
          array[y * stride + x] = stddev * 2 * a;
        } break;
        }
      }
    }
  }
}

/* randomise_weights_flat (recur-nn-init.c:547-573) */
void orc_set_init_flat(OrcSet *z, float variance, int shape, double perforation) {
  memset(z->ih_w, 0, sizeof(float) * z->I * z->H);
  memset(z->ho_w, 0, sizeof(float) * z->H * z->O);
  if (perforation < 0) {
    perforation = 0;
  } else if (perforation >= 1.0) {
    return;
  }
  fill_flat(&z->rng[0], z->ih_w, z->hidden_size, z->input_size + z->hidden_size + 1,
            z->H, 1, variance, shape, perforation);
  fill_flat(&z->rng[0], z->ho_w, z->output_size, z->hidden_size + 1, z->O, 0,
            variance, shape, perforation);
  if (z->bI) { /* recur-nn-init.c:566-572 */
    memset(z->b_w, 0, sizeof(float) * z->bI * z->bO);
    fill_flat(&z->rng[0], z->b_w, z->input_size, z->b_in, z->bO, 1, variance, shape,
              perforation);
  }
}

/* ---------------------------------------------------------- forward pass -- */

static inline float *slot_of(OrcSet *z, int s) {
  return z->hist + ((size_t)z->index[s] * z->S + s) * z->I;
}

/* recur-nn.c:696-704 */
void orc_advance(OrcSet *z, int s) {
  z->index[s]++;
  if (z->index[s] == z->D)
    z->index[s] -= z->D;
}

/* calculate_interlayer (recur-nn.c:18-48): axpy order with the zero-row skip */
static void interlayer(const float *in, int in_size, float *out, int out_size,
                       const float *w) {
  memset(out, 0, sizeof(float) * out_size);
  for (int y = 0; y < in_size; y++) {
    float v = in[y];
    if (v) {
      const float *row = w + (size_t)out_size * y;
      for (int x = 0; x < out_size; x++) {
        out[x] += v * row[x];
      }
    }
  }
}

/* rnn_opinion (recur-nn.c:83-154) */
float *orc_opinion(OrcSet *z, int s, const float *inputs, float noise) {
  float *slot = slot_of(z, s);
  float *hid = z->hidden + (size_t)s * z->H;
  float *out = z->output + (size_t)s * z->O;
  int off = z->hidden_size + 1;
  if (z->bI) { /* recur-nn.c:88-103 */
    z->b_inputs[0] = 1.0f;
    if (inputs) {
      memcpy(z->b_inputs + 1, inputs, z->b_in * sizeof(float));
    }
    interlayer(z->b_inputs, z->bI, z->b_outputs, z->bO, z->b_w);
    if (noise) {
      for (int i = 1; i < z->input_size; i++) {
        z->b_outputs[i] += orc_cheap_gaussian_noise(&z->rng[s]) * noise;
      }
    }
    for (int i = 0; i < z->input_size; i++) {
      float x = z->b_outputs[i];
      slot[off + i] = (x > 0.0f) ? x : 0.0f;
    }
  } else if (inputs) {
    memcpy(slot + off, inputs, z->input_size * sizeof(float));
  }
  memcpy(slot, hid, off * sizeof(float));
  slot[0] = 1.0f;
  /* maybe_scale_inputs (recur-nn.c:68-81) */
  {
    float softclip = z->I * 16.0f;
    float sum = 0.0f;
    for (int i = 0; i < z->I; i++) {
      sum += slot[i];
    }
    if (sum > softclip) {
      softclip_scale(sum, softclip, slot, z->I);
    }
  }
  interlayer(slot, z->I, hid, z->H, z->ih_w);
  if (noise) { /* recur-nn.c:120-121: also lands on the pad columns */
    for (int i = 1; i < z->H; i++) {
      hid[i] += orc_cheap_gaussian_noise(&z->rng[s]) * noise;
    }
  }
  if (z->activation == ORC_RESQRT) {
    for (int i = 0; i < z->H; i++) {
      float h = hid[i];
      hid[i] = (h > 0.0f) ? sqrtf(h + 1.0f) - 1.0f : 0.0f;
    }
  } else if (z->activation == ORC_RECLIP20) {
    for (int i = 1; i < z->H; i++) {
      float h = hid[i] - 0.0f;
      h = h < 20.0f ? h : 20.0f;
      hid[i] = (h > 0.0f) ? h : 0.0f;
    }
  } else {
    for (int i = 1; i < z->H; i++) {
      float h = hid[i] - 0.0f;
      hid[i] = (h > 0.0f) ? h : 0.0f;
    }
  }
  hid[0] = 1.0f;
  interlayer(hid, z->H, out, z->O, z->ho_w);
  return out;
}

/* one_hot_opinion (charmodel-helpers.h:16-33) */
float *orc_one_hot_opinion(OrcSet *z, int s, int hot, float noise) {
  if (z->bI) {
    /* with a bottom layer the helper indexes from the layer's bias slot
     * (charmodel-helpers.h:20-23, 30-31): symbol k lights input k - 1, symbol 0 only
     * the bias that rnn_opinion sets anyway, and the last input is never cleared */
    memset(z->b_inputs, 0, z->b_in * sizeof(float));
    z->b_inputs[hot] = 1.0f;
    return orc_opinion(z, s, NULL, noise);
  }
  float *slot = slot_of(z, s);
  float *real = slot + z->hidden_size + 1;
  memset(real, 0, z->input_size * sizeof(float));
  real[hot] = 1.0f;
  return orc_opinion(z, s, NULL, noise);
}

/* net_error_bptt (charmodel-predict.c:18-27) */
float orc_net_error_bptt(OrcSet *z, int s, int c, int next, int *correct) {
  float *error = z->o_error + (size_t)s * z->O;
  float *answer = orc_one_hot_opinion(z, s, c, z->presynaptic_noise);
  int winner = orc_softmax_best_guess(error, answer, z->output_size);
  *correct = (winner == next);
  error[next] += 1.0f;
  return error[next];
}

/* the loss of train_channel (gstclassify.c:2070-2119) after the opinion; returns the
 * number of groups trained */
int orc_grouped_softmax_error(OrcSet *z, int s, int n_groups, const int *group_offset,
                              const int *group_size, const int *targets, const float *weight,
                              int *wins, float *wrongness) {
  float *error = z->o_error + (size_t)s * z->O;
  const float *answer = z->output + (size_t)s * z->O;
  int trained = 0;
  for (int i = 0; i < n_groups; i++) {
    int o = group_offset[i], n = group_size[i], target = targets[i];
    if (target < 0 || target >= n) {
      for (int j = 0; j < n; j++) {
        error[o + j] = 0;
      }
      continue;
    }
    int winner = orc_softmax_best_guess(error + o, answer + o, n);
    *wins += winner == target;
    error[o + target] += 1.0f;
    *wrongness += error[o + target];
    trained++;
  }
  if (trained && weight) {
    for (int i = 0; i < z->output_size; i++) {
      error[i] *= weight[i];
    }
  }
  return trained;
}

/* (Pinned since round 3: tests/golden/ref_callers.npz holds what the reference's own object code gives
 * when walked through these callers' control flow -- tests/golden/make_golden_callers.py -- and
 * tests/test_oracle_callers_golden.py holds orc_multi_softmax_error, orc_multitext_train,
 * orc_multi_cross_entropy, orc_grouped_softmax_error and orc_sigmoid_mse_error to it.) */
/* multi_softmax_error (charmodel-multi-predict.c:17-58): opinion, then every class head
 * of alphabet_len outputs is either trained (its own head always, the others when a draw
 * from the stream's generator falls under leakage) or left at zero; ranges_out receives
 * the merged, aligned (start, len) pairs and the terminator.  Returns the own head's
 * error on the next symbol. */
float orc_multi_softmax_error(OrcSet *z, int s, int c, int next, int target_class,
                              int alphabet_len, float leakage, int *ranges_out) {
  float *error = z->o_error + (size_t)s * z->O;
  float *answer = orc_one_hot_opinion(z, s, c, z->presynaptic_noise);
  int n_classes = z->output_size / alphabet_len;
  float err = 0;
  int j = 0;
  uint64_t threshold = leakage * UINT64_MAX;
  memset(error, 0, z->output_size * sizeof(float));
  for (int i = 0; i < n_classes; i++) {
    int offset = i * alphabet_len;
    if (i == target_class || orc_rand64(&z->rng[s]) < threshold) {
      orc_softmax_best_guess(error + offset, answer + offset, alphabet_len);
      error[offset + next] += 1.0f;
      if (i == target_class) {
        err = error[offset + next];
      }
      int range_start = offset & ~3;
      int range_end = (offset + alphabet_len + 3) & ~3;
      if (j) {
        int pstart = ranges_out[2 * (j - 1)], plen = ranges_out[2 * (j - 1) + 1];
        if (pstart + plen >= range_start) {
          ranges_out[2 * (j - 1) + 1] = range_end - pstart;
          continue;
        }
      }
      ranges_out[2 * j] = range_start;
      ranges_out[2 * j + 1] = range_end - range_start;
      j++;
    }
  }
  ranges_out[2 * j] = -1;
  ranges_out[2 * j + 1] = 0;
  return err;
}

/* get_cross_entropy (charmodel-predict.c:62-80) */
double orc_cross_entropy(OrcSet *z, int s, const uint8_t *text, int len, int skip) {
  float *error = malloc(sizeof(float) * z->output_size);
  double entropy = 0.0;
  int i;
  for (i = 0; i < skip; i++) {
    orc_one_hot_opinion(z, s, text[i], 0);
  }
  for (; i < len - 1; i++) {
    float *answer = orc_one_hot_opinion(z, s, text[i], 0);
    orc_softmax(error, answer, z->output_size);
    entropy += orc_capped_log2f(error[text[i + 1]]);
  }
  free(error);
  return entropy / -(len - skip - 1);
}

/* rnn_char_multi_cross_entropy (charmodel-multi-predict.c:383-408): entropy[j] is the
 * caller's running value (the reference subtracts into it and then divides) */
void orc_multi_cross_entropy(OrcSet *z, int s, const uint8_t *text, int len, int alphabet_len,
                             double *entropy, int ignore_start) {
  float *error = malloc(sizeof(float) * alphabet_len);
  int i, j;
  int n_classes = z->output_size / alphabet_len;
  for (i = 0; i < ignore_start; i++) {
    orc_one_hot_opinion(z, s, text[i], 0);
  }
  for (; i < len - 1; i++) {
    float *answer = orc_one_hot_opinion(z, s, text[i], 0);
    for (j = 0; j < n_classes; j++) {
      float *group = answer + alphabet_len * j;
      orc_softmax(error, group, alphabet_len);
      float e = error[text[i + 1]];
      entropy[j] -= orc_capped_log2f(e);
    }
  }
  for (j = 0; j < n_classes; j++) {
    entropy[j] /= (len - ignore_start - 1);
  }
  free(error);
}

/* fast_sigmoid (badmaths.h:31-36) */
float orc_fast_sigmoid(float x) { return 1.0f / (1.0f + orc_fast_expf(-x * 1.0f)); }

/* rnnca's train_net after the opinion (gstrnnca.c:701-714): sigmoid of the first n answers
 * in place (the answer IS output_layer), error = slope * (target - a) */
void orc_sigmoid_mse_error(OrcSet *z, int s, const float *target, int n) {
  float *answer = z->output + (size_t)s * z->O;
  float *error = z->o_error + (size_t)s * z->O;
  for (int i = 0; i < n; i++) {
    answer[i] = orc_fast_sigmoid(answer[i]);
  }
  for (int i = 0; i < n; i++) {
    float a = answer[i];
    float slope = a * (1.0f - a);
    error[i] = slope * (target[i] - a);
  }
}

/* --------------------------------------------------------- backward pass -- */

/* backprop_single_layer (recur-nn.c:199-228) */
static float backprop_top(const float *w, const float *hid, float *h_err, int H,
                          const float *o_err, int O) {
  float error_sum = 0.0f;
  for (int y = 1; y < H; y++) {
    float e = 0.0f;
    if (hid[y]) {
      const float *row = w + (size_t)y * O;
      for (int x = 0; x < O; x++) {
        e += row[x] * o_err[x];
      }
      error_sum += fabsf(e);
    }
    h_err[y] = e;
  }
  return error_sum;
}

/* backprop_single_layer_sparse (recur-nn.c:156-196): inactive rows keep their
 * old h_err value, and |e| is added once per range while e keeps running. */
static float backprop_top_sparse(const float *w, const float *hid, float *h_err,
                                 int H, const float *o_err, int O,
                                 const int *ranges) {
  float error_sum = 0.0f;
  for (int y = 1; y < H; y++) {
    float e = 0.0f;
    if (hid[y]) {
      const float *row = w + (size_t)y * O;
      for (int i = 0; ranges[2 * i] >= 0; i++) {
        int start = ranges[2 * i] & ~3;
        int len = (ranges[2 * i + 1] + 3) & ~3;
        for (int x = 0; x < len; x++) {
          e += row[start + x] * o_err[start + x];
        }
        error_sum += fabsf(e);
      }
      h_err[y] = e;
    }
  }
  return error_sum;
}

/* single_layer_sgd / _sparse (recur-nn.c:256-301) */
static void top_delta(const float *hid, int H, const float *o_err, int O,
                      float *delta, const int *ranges) {
  for (int y = 0; y < H; y++) {
    float v = hid[y];
    if (v) {
      float *drow = delta + (size_t)y * O;
      if (!ranges) {
        for (int x = 0; x < O; x++) {
          drow[x] += o_err[x] * v;
        }
      } else {
        for (int i = 0; ranges[2 * i] >= 0; i++) {
          int start = ranges[2 * i] & ~3;
          int len = (ranges[2 * i + 1] + 3) & ~3;
          for (int x = 0; x < len; x++) {
            drow[start + x] += o_err[start + x] * v;
          }
        }
      }
    }
  }
}

/* bptt_and_accumulate_error (recur-nn.c:303-450), VECTOR flavour of the inner
 * loop (four running partial sums, recur-nn.c:347-358). */
static float bptt_accumulate(OrcSet *z, int s, float *ih_delta, float *cumulative_input_error,
                             float top_error_sum) {
  const int I = z->I, H = z->H, D = z->D;
  float *h_error = z->err_a + (size_t)s * I;
  float *i_error = z->err_b + (size_t)s * I;
  const float *weights = z->ih_w;
  float error_sum = 0;
  float max_error_sum = 2.0f * top_error_sum + 1;
  float error_sum_ceiling = 1.0f * top_error_sum;
  float min_error_gain = 1e-8f * top_error_sum;
  float min_error_sum =
      ORC_MIN(z->min_error_factor[s] / z->learn_rate[s], min_error_gain);
  int t;
  int offset = z->index[s];
  for (t = D; t > 0; t--, offset += offset ? -1 : D - 1) {
    error_sum = 0.0f;
    const float *inputs = z->hist + ((size_t)offset * z->S + s) * I;
    h_error[0] = 0.0;
    for (int i = z->hidden_size + 1; i < H; i++) {
      h_error[i] = 0.0;
    }
    for (int y = 0; y < I; y++) {
      float input = inputs[y];
      if (input != 0.0f && (z->activation != ORC_RECLIP20 || input < 20.0f)) {
        const float *w_row = weights + (size_t)y * H;
        float *delta_row = ih_delta + (size_t)y * H;
        float ve[4] = {0, 0, 0, 0};
        for (int x = 0; x < H; x += 4) {
          for (int k = 0; k < 4; k++) {
            float ex = h_error[x + k];
            delta_row[x + k] += ex * input;
            ve[k] += w_row[x + k] * ex;
          }
        }
        float e = ve[0] + ve[1] + ve[2] + ve[3];
        if (z->activation == ORC_RESQRT) {
          e /= 2 * (input + 1.0f);
        }
        i_error[y] = e;
        error_sum += e * e;
      } else {
        i_error[y] = 0;
      }
    }
    if (cumulative_input_error) { /* recur-nn.c:377-382 */
      const float *input_error = i_error + z->hidden_size + 1;
      for (int y = 0; y < z->input_size; y++) {
        cumulative_input_error[y] += input_error[y];
      }
    }
    float *tmp = h_error;
    h_error = i_error;
    i_error = tmp;
    if (error_sum <= min_error_sum || error_sum > max_error_sum) {
      break;
    }
  }
  if (error_sum > error_sum_ceiling) {
    z->ih_scale[s] = orc_soft_clip(error_sum, max_error_sum);
    if (cumulative_input_error) {
      /* recur-nn.c:391-399: the accumulator -- ALL of it, earlier streams and generations included,
       * it is the bottom layer's one o_error -- shrinks by ih_scale twice */
      for (int y = 0; y < z->input_size; y++) {
        cumulative_input_error[y] *= z->ih_scale[s] * z->ih_scale[s];
      }
    }
  } else {
    z->ih_scale[s] = 1.0f;
    if (z->flags & ORC_FLAG_ADAPTIVE_MIN_ERROR) {
      int depth_error = D / 4 - t;
      if (z->min_error_factor[s] < 1e-2f &&
          (min_error_gain != min_error_sum || depth_error < 0)) {
        z->min_error_factor[s] *= (1.0f + depth_error * 1e-3);
      }
      z->min_error_factor[s] = ORC_MAX(z->min_error_factor[s], 1e-20f);
    }
  }
  z->bptt_depth[s] = D - t;
  return error_sum;
}

/* recur-nn.c:681-693 */
void orc_clear_deltas(OrcSet *z) {
  memset(z->ih_delta, 0, sizeof(float) * z->I * z->H);
  memset(z->ho_delta, 0, sizeof(float) * z->H * z->O);
  if (z->bI) {
    memset(z->b_o_error, 0, sizeof(float) * z->bO);
    memset(z->b_delta, 0, sizeof(float) * z->bI * z->bO);
  }
}

/* rnn_bptt_calc_deltas (recur-nn.c:707-772) */
void orc_calc_deltas(OrcSet *z, int s, int accumulate, const int *ranges) {
  const size_t ih = (size_t)z->I * z->H;
  float *hid = z->hidden + (size_t)s * z->H;
  float *h_err = z->err_a + (size_t)s * z->I;
  float *o_err = z->o_error + (size_t)s * z->O;
  if (!accumulate) {
    memset(z->ho_delta, 0, sizeof(float) * z->H * z->O);
  }
  float top = ranges ? backprop_top_sparse(z->ho_w, hid, h_err, z->H, o_err, z->O, ranges)
                     : backprop_top(z->ho_w, hid, h_err, z->H, o_err, z->O);
  float top_scaled = softclip_scale(top, z->H * 2.0f, h_err, z->H);
  top_delta(hid, z->H, o_err, z->O, z->ho_delta, ranges);
  float bptt_err;
  if (accumulate) {
    memset(z->ih_delta_tmp, 0, sizeof(float) * ih);
    bptt_err = bptt_accumulate(z, s, z->ih_delta_tmp, z->bI ? z->b_o_error : NULL, top_scaled);
    float scale = z->ih_scale[s];
    if (scale == 1.0f) { /* add_aligned_arrays, recur-nn-helpers.h:68-77 */
      for (size_t i = 0; i < ih; i++) {
        z->ih_delta[i] += z->ih_delta_tmp[i];
      }
    } else {
      for (size_t i = 0; i < ih; i++) {
        z->ih_delta[i] += z->ih_delta_tmp[i] * scale;
      }
    }
  } else {
    memset(z->ih_delta, 0, sizeof(float) * ih);
    bptt_err = bptt_accumulate(z, s, z->ih_delta, z->bI ? z->b_o_error : NULL, top_scaled);
    if (z->ih_scale[s] != 1.0f) {
      float scale = z->ih_scale[s];
      for (size_t i = 0; i < ih; i++) {
        z->ih_delta[i] *= scale;
      }
    }
  }
  if (z->bI) { /* recur-nn.c:750-757: nothing in here ever clears b_o_error */
    if (!accumulate) {
      memset(z->b_delta, 0, sizeof(float) * z->bI * z->bO);
    }
    top_delta(z->b_inputs, z->bI, z->b_o_error, z->bO, z->b_delta, NULL);
  }
  z->top_error_raw[s] = top;
  z->top_error_scaled[s] = top_scaled;
  z->bptt_error[s] = bptt_err;
  z->generation[s]++;
}

/* ------------------------------------------------------------- optimiser -- */

/* recur-nn.c:454-489 */
static void learn_momentum(float *w, const float *delta, float *m, size_t n,
                           float rate, float momentum, float momentum_weight) {
  for (size_t i = 0; i < n; i++) {
    float t = delta[i] * rate;
    float mm = m[i];
    w[i] += t + mm * momentum_weight;
    m[i] = (mm + t) * momentum;
  }
}

/* recur-nn.c:494-509 */
static void learn_nesterov(float *w, const float *delta, float *m, size_t n,
                           float rate, float momentum) {
  for (size_t i = 0; i < n; i++) {
    float t = delta[i] * rate;
    w[i] += t;
    m[i] += t;
  }
  for (size_t i = 0; i < n; i++) {
    m[i] *= momentum;
  }
  for (size_t i = 0; i < n; i++) {
    w[i] += m[i];
  }
}

/* recur-nn.c:511-525 */
static void learn_adagrad(float *w, const float *delta, float *acc, size_t n,
                          float rate) {
  for (size_t i = 0; i < n; i++) {
    float d = delta[i];
    float a = acc[i];
    a += d * d;
    w[i] += d * rate / sqrtf(a);
    acc[i] = a;
  }
}

/* recur-nn.c:527-558 (the live branch is the abs-value variant) */
static void learn_adadelta(float *w, const float *delta, float *gacc, float *sacc,
                           size_t n, float rate, float decay) {
  const float renewal = 1.0f - decay;
  for (size_t i = 0; i < n; i++) {
    float d = delta[i];
    float g = gacc[i];
    float s = sacc[i];
    g *= decay;
    s *= decay;
    g += fabsf(d) * renewal + rate;
    float step = s / g * d;
    s += fabsf(step) * renewal + rate;
    gacc[i] = g;
    sacc[i] = s;
    w[i] += step;
  }
}

/* recur-nn.c:560-593 */
static void learn_rprop(float *w, const float *delta, float *prev_g, float *prev_step,
                        size_t n, float rate, float decay) {
  (void)decay;
  const float shrink = 0.5f, grow = 1.2f;
  const float max_step = 1 * rate;
  const float min_step = 1e-6 * rate;
  for (size_t i = 0; i < n; i++) {
    float d = delta[i];
    float p = prev_g[i];
    float step = prev_step[i];
    if (d * p > 0.0f) {
      step = ORC_MIN(step * grow, max_step);
    } else if (d * p < 0.0f) {
      step = ORC_MAX(step * shrink, min_step);
      d = 0;
    }
    if (d > 0.0f) {
      w[i] += step;
    } else {
      w[i] -= step;
    }
    prev_step[i] = step;
    prev_g[i] = d;
  }
}

/* rnn_apply_learning (recur-nn.c:601-678): top layer first, then recurrent;
 * rates come from the prototype (stream 0). */
void orc_apply_learning(OrcSet *z, int method, float momentum) {
  size_t ih = (size_t)z->I * z->H, ho = (size_t)z->H * z->O;
  float lr = z->learn_rate[0];
  float lr_top = lr * z->ho_scale;
  size_t bn = (size_t)z->bI * z->bO;
  float lr_bottom = lr * z->b_learn_rate_scale;
  if (method == ORC_NESTEROV) {
    learn_nesterov(z->ho_w, z->ho_delta, z->ho_m, ho, lr_top, momentum);
    learn_nesterov(z->ih_w, z->ih_delta, z->ih_m, ih, lr, momentum);
    if (bn) learn_nesterov(z->b_w, z->b_delta, z->b_m, bn, lr_bottom, momentum);
  } else if (method == ORC_ADAGRAD) {
    learn_adagrad(z->ho_w, z->ho_delta, z->ho_m, ho, lr_top);
    learn_adagrad(z->ih_w, z->ih_delta, z->ih_m, ih, lr);
    if (bn) learn_adagrad(z->b_w, z->b_delta, z->b_m, bn, lr_bottom);
  } else if (method == ORC_ADADELTA) {
    learn_adadelta(z->ho_w, z->ho_delta, z->ho_m, z->ho_aux, ho, lr_top, momentum);
    learn_adadelta(z->ih_w, z->ih_delta, z->ih_m, z->ih_aux, ih, lr, momentum);
    if (bn) learn_adadelta(z->b_w, z->b_delta, z->b_m, z->b_aux, bn, lr_bottom, momentum);
  } else if (method == ORC_RPROP) {
    learn_rprop(z->ho_w, z->ho_delta, z->ho_m, z->ho_aux, ho, lr_top, momentum);
    learn_rprop(z->ih_w, z->ih_delta, z->ih_m, z->ih_aux, ih, lr, momentum);
    if (bn) learn_rprop(z->b_w, z->b_delta, z->b_m, z->b_aux, bn, lr_bottom, momentum);
  } else {
    float mw;
    if (method == ORC_SIMPLIFIED_NESTEROV) {
      mw = momentum / (1.0 + momentum);
    } else if (method == ORC_CLASSICAL) {
      mw = 1.0f;
    } else {
      mw = z->momentum_weight;
    }
    learn_momentum(z->ho_w, z->ho_delta, z->ho_m, ho, lr_top, momentum, mw);
    learn_momentum(z->ih_w, z->ih_delta, z->ih_m, ih, lr, momentum, mw);
    if (bn) learn_momentum(z->b_w, z->b_delta, z->b_m, bn, lr_bottom, momentum, mw);
  }
}

/* rnn_condition_net (recur-nn.c:782-855); flags carries the USE bits 16-23 */
void orc_condition(OrcSet *z, uint32_t flags) {
  size_t ih = (size_t)z->I * z->H, ho = (size_t)z->H * z->O;
  uint32_t mask = flags >> 16;
  uint32_t m = z->generation[0] % 8;
  if (((1u << m) & mask) == 0) {
    return;
  }
  switch (m) {
  case 0: /* SCALE */
    for (size_t i = 0; i < ih; i++)
      z->ih_w[i] *= (1.0f - 1e-6f);
    for (size_t i = 0; i < ho; i++)
      z->ho_w[i] *= (1.0f - 1e-6f);
    break;
  case 2: /* ZERO: zero_small_numbers, recur-nn-helpers.h:126-133 */
    for (size_t i = 0; i < ih; i++) {
      z->ih_w[i] = (fabsf(z->ih_w[i]) > 1e-34f) ? z->ih_w[i] : 0.0f;
      z->ih_m[i] = (fabsf(z->ih_m[i]) > 1e-34f) ? z->ih_m[i] : 0.0f;
    }
    for (size_t i = 0; i < ho; i++) {
      z->ho_w[i] = (fabsf(z->ho_w[i]) > 1e-34f) ? z->ho_w[i] : 0.0f;
      z->ho_m[i] = (fabsf(z->ho_m[i]) > 1e-34f) ? z->ho_m[i] : 0.0f;
    }
    break;
  case 6: { /* RAND */
    int t = orc_rand_small_int(&z->rng[0], (int)(ih + ho));
    float damage = orc_cheap_gaussian_noise(&z->rng[0]) * 0.5f * z->H * z->learn_rate[0];
    if (t >= (int)ih) {
      t -= (int)ih;
      int col = t % z->O;
      if (col < z->output_size) {
        z->ho_w[t] += damage;
      }
    } else {
      int col = t % z->H;
      if (col >= 1 && col < z->hidden_size + 1) {
        z->ih_w[t] += damage;
      }
    }
  } break;
  case 4: { /* TALL_POPPY */
    size_t big_i = 0;
    float big_v = fabsf(z->ih_w[0]);
    for (size_t i = 1; i < ih; i++) {
      float v = fabsf(z->ih_w[i]);
      if (v > big_v) {
        big_v = v;
        big_i = i;
      }
    }
    if (big_v > 1.0f) {
      z->ih_w[big_i] *= 0.99f;
    }
  } break;
  case 3: /* LAWN_MOWER */
    for (size_t i = 0; i < ih; i++) {
      z->ih_w[i] = ORC_MAX(z->ih_w[i], -10.0f);
      z->ih_w[i] = ORC_MIN(z->ih_w[i], 10.0f);
    }
    break;
  }
}

/* rnn_bptt_calculate and its helpers (recur-nn.c:919-1019): the single-net
 * path that updates the weights immediately; momentum is bptt->momentum. */
void orc_bptt_calculate(OrcSet *z, int s, unsigned batch_size, float momentum) {
  const size_t ih = (size_t)z->I * z->H;
  float *hid = z->hidden + (size_t)s * z->H;
  float *h_err = z->err_a + (size_t)s * z->I;
  float *o_err = z->o_error + (size_t)s * z->O;
  float rate = z->learn_rate[s];
  float mw = z->momentum_weight;
  /* apply_sgd_top_layer: note no ho_scale (recur-nn.c:927) */
  hid[0] = 1.0f;
  float top = backprop_top(z->ho_w, hid, h_err, z->H, o_err, z->O);
  for (int y = 0; y < z->H; y++) {
    float *mrow = z->ho_m + (size_t)y * z->O;
    float *row = z->ho_w + (size_t)y * z->O;
    if (hid[y]) {
      float m = hid[y] * rate;
      for (int x = 0; x < z->O; x++) {
        float d = o_err[x] * m;
        float mm = mrow[x];
        row[x] += d + mm * mw;
        mm += d;
        mrow[x] = mm * momentum;
      }
    } else {
      for (int x = 0; x < z->O; x++) {
        float mm = mrow[x];
        row[x] += mm * mw;
        mrow[x] = mm * momentum;
      }
    }
  }
  float top_scaled = softclip_scale(top, z->H * 2.0f, h_err, z->H);
  float bptt_err;
  if (batch_size > 1) { /* apply_sgd_with_bptt_batch */
    memset(z->ih_delta_tmp, 0, sizeof(float) * ih);
    bptt_err = bptt_accumulate(z, s, z->ih_delta_tmp, NULL /* recur-nn.c:972, 986 */, top_scaled);
    float scale = z->ih_scale[s];
    if (scale == 1.0f) {
      for (size_t i = 0; i < ih; i++)
        z->ih_delta[i] += z->ih_delta_tmp[i];
    } else {
      for (size_t i = 0; i < ih; i++)
        z->ih_delta[i] += z->ih_delta_tmp[i] * scale;
    }
    if ((z->generation[s] % batch_size) == 0) {
      learn_momentum(z->ih_w, z->ih_delta, z->ih_m, ih, rate, momentum, mw);
      memset(z->ih_delta, 0, sizeof(float) * ih);
    }
  } else { /* apply_sgd_with_bptt */
    memset(z->ih_delta, 0, sizeof(float) * ih);
    bptt_err = bptt_accumulate(z, s, z->ih_delta, NULL /* recur-nn.c:972, 986 */, top_scaled);
    learn_momentum(z->ih_w, z->ih_delta, z->ih_m, ih, rate * z->ih_scale[s], momentum, mw);
  }
  z->top_error_raw[s] = top;
  z->top_error_scaled[s] = top_scaled;
  z->bptt_error[s] = bptt_err;
  z->generation[s]++;
  orc_condition(z, z->flags);
}

/* ---------------------------------------------- multi-tap text generation -- */

/* the j loop of rnn_char_epoch (charmodel-predict.c:293-310) */
void orc_set_char_step_deltas(OrcSet *z, const uint8_t *text, int len, int i) {
  int count = z->global_count ? z->global_count : z->S;
  int spacing = (len - 1) / count;
  for (int j = 0; j < z->S; j++) {
    int offset = i + (z->global_first + j) * spacing;
    if (offset >= len - 1) {
      offset -= len - 1;
    }
    int c;
    orc_advance(z, j);
    float e = orc_net_error_bptt(z, j, text[offset], text[offset + 1], &c);
    z->stat_correct += c;
    z->stat_error += e;
    z->stat_entropy += orc_capped_log2f(1.0f - e);
    z->stat_count++;
    orc_calc_deltas(z, j, j ? 1 : 0, NULL);
    z->stat_depth += z->bptt_depth[j];
    const float *hid = z->hidden + (size_t)j * z->H;
    int zeros = 0;
    for (int k = 0; k < z->H; k++) {
      zeros += (hid[k] == 0.0f);
    }
    z->stat_zeros += zeros / (double)z->hidden_size;
  }
}

/* ... followed by rnn_apply_learning (charmodel-predict.c:311) */
void orc_set_char_step(OrcSet *z, const uint8_t *text, int len, int i, int method,
                       float momentum) {
  orc_set_char_step_deltas(z, text, len, i);
  orc_apply_learning(z, method, momentum);
}

/* text_train of the multi-head trainer (charmodel-multi-predict.c:234-281) for stream s of a
 * set (the reference runs it on ONE net): per symbol advance, multi_softmax_error, then --
 * when the batch countdown has run out -- rnn_apply_learning with the net's OWN momentum
 * (line 247 passes bptt->momentum, not the caller's) before the non-accumulating calc_deltas.
 * error_sum / entropy_sum receive what the progress report averages. */
void orc_multitext_train(OrcSet *z, int s, const uint8_t *text, int len, int alphabet_len,
                         int target_class, float leakage, int learning_style, float bptt_momentum,
                         int batch_size, float *error_sum, float *entropy_sum) {
  int n_classes = z->output_size / alphabet_len;
  int *ranges = malloc(sizeof(int) * 2 * (n_classes + 1));
  float error = 0.0f, entropy = 0.0f;
  if (batch_size < 1) {
    batch_size = 1;
  }
  int countdown = batch_size - z->generation[s] % batch_size;
  for (int i = 0; i < len - 1; i++, countdown--) {
    orc_advance(z, s);
    float e = orc_multi_softmax_error(z, s, text[i], text[i + 1], target_class, alphabet_len, leakage,
                                      ranges);
    if (countdown == 0) {
      orc_apply_learning(z, learning_style, bptt_momentum);
      countdown = batch_size;
      orc_calc_deltas(z, s, 0, ranges);
    } else {
      orc_calc_deltas(z, s, 1, ranges);
    }
    error += e;
    entropy += orc_capped_log2f(1.0f - e);
  }
  free(ranges);
  if (error_sum) *error_sum = error;
  if (entropy_sum) *entropy_sum = entropy;
}
