/* rnn_init.c -- PRNG and weight initialisation (host only, gnu11 C).
 *
 * These run once per net on the host copy of the weights; the device image is
 * refreshed from it at the next device call (ramd_host_wrote).  The random
 * stream must be consumed in exactly the reference's order, because nets are
 * reproducible by seed (recur-rng.h; recur-nn-init.c:382-742).
 */
#include "rnn_host.h"
#include <time.h>

/* ---------------------------------------------------------------- PRNG -- */

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* Bob Jenkins' small fast PRNG, 64-bit flavour (recur-rng.h:22-31) */
uint64_t ramd_rand64(rand_ctx *x) {
  uint64_t e = x->a - rotl64(x->b, 7);
  x->a = x->b ^ rotl64(x->c, 13);
  x->b = x->c + rotl64(x->d, 37);
  x->c = x->d + e;
  x->d = e + x->a;
  return x->d;
}

/* recur-rng.h:34-43 */
void ramd_init_rand64(rand_ctx *x, uint64_t seed) {
  x->a = 0xf1ea5eed;
  x->b = x->c = x->d = seed;
  for (int i = 0; i < 20; i++) {
    (void)ramd_rand64(x);
  }
}

/* recur-rng.h:45-55 */
void ramd_init_rand64_maybe_randomly(rand_ctx *x, uint64_t seed) {
  if (seed == RECUR_RNG_RANDOM_SEED) {
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    seed = (uint64_t)(((uint64_t)t.tv_nsec << 20) + t.tv_sec) ^ (uint64_t)((uintptr_t)x);
    fprintf(stderr, "seeding with %llx\n", (unsigned long long)seed);
  }
  ramd_init_rand64(x, seed);
}

/* recur-rng.h:69-78: random mantissa under a fixed exponent gives [1, 2) */
double ramd_rand_double(rand_ctx *x) {
  union {
    uint64_t i;
    double d;
  } u;
  u.i = (ramd_rand64(x) & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
  return u.d - 1.0;
}

/* recur-rng.h:96-100 */
int ramd_rand_small_int(rand_ctx *x, int cap) { return (int)(ramd_rand_double(x) * cap); }

static inline int rand_range(rand_ctx *x, int start, int cap) {
  return start + ramd_rand_small_int(x, cap - start);
}

/* recur-rng.h:179-200: twelve 16-bit fields of three draws, centred and scaled */
float ramd_cheap_gaussian_noise(rand_ctx *x) {
  int64_t a = 0;
  for (int w = 0; w < 3; w++) {
    uint64_t bits = ramd_rand64(x);
    a += (int64_t)(bits & 0xffff) + (int64_t)((bits >> 16) & 0xffff) +
         (int64_t)((bits >> 32) & 0xffff) + (int64_t)(bits >> 48);
  }
  return (float)(a - 0xffff * 6) / (0xffff);
}

/* badmaths.h:14-29 */
float ramd_fast_expf(float x) {
  int count = 0;
  while (fabsf(x) > 0.2) {
    x *= 0.125;
    count++;
  }
  float a = ((x + 3) * (x + 3) + 3) / ((x - 3) * (x - 3) + 3);
  for (; count; count--) {
    a *= a;
    a *= a;
    a *= a;
  }
  return a;
}

/* ------------------------------------------------------------ flat init -- */

/* one cell of randomise_array_flat (recur-nn-init.c:507-541) */
static float flat_sample(rand_ctx *rng, float variance, float stddev,
                         rnn_init_distribution shape) {
  switch (shape) {
  case RNN_INIT_DIST_UNIFORM: {
    const double range = sqrtf(12.0f * variance);
    return range * ramd_rand_double(rng) - range * 0.5;
  }
  case RNN_INIT_DIST_LOG_NORMAL: {
    float a = ramd_cheap_gaussian_noise(rng) * 0.33;
    float b = 0.9 * stddev * ramd_fast_expf(a);
    return (ramd_rand64(rng) & 1) ? b : -b;
  }
  case RNN_INIT_DIST_SEMICIRCLE: {
    double a, b;
    do { /* rejection-sample the unit half disc */
      a = ramd_rand_double(rng) * 2.0 - 1.0;
      b = ramd_rand_double(rng);
    } while (a * a + b * b > 1.0);
    return stddev * 2 * a;
  }
  case RNN_INIT_DIST_GAUSSIAN:
  default:
    return stddev * ramd_cheap_gaussian_noise(rng);
  }
}

/* recur-nn-init.c:495-545 */
static void fill_flat(rand_ctx *rng, float *array, int width, int height, int stride,
                      int offset, float variance, rnn_init_distribution shape,
                      double perforation) {
  float stddev = sqrtf(variance);
  fprintf(stderr, "initialising using method %d, variance %g\n", shape, variance);
  for (int y = 0; y < height; y++) {
    float *row = array + (size_t)y * stride;
    for (int x = offset; x < width + offset; x++) {
      if (perforation == 0 || ramd_rand_double(rng) > perforation) {
        row[x] = flat_sample(rng, variance, stddev, shape);
      }
    }
  }
}

/* recur-nn-init.c:547-573 */
static void init_flat(RecurNN *net, float variance, rnn_init_distribution shape,
                      double perforation) {
  memset(net->ih_weights, 0, (size_t)net->ih_size * sizeof(float));
  memset(net->ho_weights, 0, (size_t)net->ho_size * sizeof(float));
  if (perforation < 0) {
    perforation = 0;
  } else if (perforation >= 1.0) {
    return;
  }
  fill_flat(&net->rng, net->ih_weights, net->hidden_size,
            net->input_size + net->hidden_size + 1, net->h_size, 1, variance, shape,
            perforation);
  fill_flat(&net->rng, net->ho_weights, net->output_size, net->hidden_size + 1, net->o_size, 0,
            variance, shape, perforation);
  if (net->bottom_layer) { /* recur-nn-init.c:566-572: input_size rows, so not the last one */
    RecurExtraLayer *bl = net->bottom_layer;
    memset(bl->weights, 0, (size_t)bl->i_size * bl->o_size * sizeof(float));
    fill_flat(&net->rng, bl->weights, bl->output_size, bl->input_size, bl->o_size, 1, variance,
              shape, perforation);
  }
}

/* ---------------------------------------------------------- fan-in init -- */

/* recur-nn-init.c:575-591: each destination column collects random sources
 * until their absolute sum is about `sum` */
static void fill_fan_in(rand_ctx *rng, float *weights, int width, int height, int stride,
                        float sum, float kurtosis, float margin) {
  for (int x = 0; x < width; x++) {
    float remainder = sum + margin;
    for (int i = 0; i < height * 2 && remainder > margin; i++) {
      int y = ramd_rand_small_int(rng, height);
      float *cell = weights + (size_t)y * stride + x;
      if (*cell == 0) {
        float w = (ramd_rand_double(rng) * 2 - 1) * remainder * kurtosis;
        *cell += w;
        remainder -= fabsf(w);
      }
    }
  }
}

/* recur-nn-init.c:593-621 */
static void init_fan_in(RecurNN *net, float sum, float kurtosis, float margin,
                        float inputs_weight_ratio) {
  memset(net->ih_weights, 0, (size_t)net->ih_size * sizeof(float));
  memset(net->ho_weights, 0, (size_t)net->ho_size * sizeof(float));
  int hsize = 1 + net->hidden_size;
  if (inputs_weight_ratio > 0) {
    fill_fan_in(&net->rng, net->ih_weights + 1, net->hidden_size, hsize, net->h_size, sum,
                kurtosis, margin);
    fill_fan_in(&net->rng, net->ih_weights + (size_t)hsize * net->h_size + 1, net->hidden_size,
                net->input_size, net->h_size, sum * inputs_weight_ratio, kurtosis, margin);
  } else {
    fill_fan_in(&net->rng, net->ih_weights + 1, net->hidden_size, hsize + net->input_size,
                net->h_size, sum, kurtosis, margin);
  }
  fill_fan_in(&net->rng, net->ho_weights, net->output_size, net->hidden_size, net->o_size, sum,
              kurtosis, margin);
  if (net->bottom_layer) { /* recur-nn-init.c:614-620 */
    RecurExtraLayer *bl = net->bottom_layer;
    memset(bl->weights, 0, (size_t)bl->i_size * bl->o_size * sizeof(float));
    fill_fan_in(&net->rng, bl->weights, bl->output_size, bl->input_size + 1, bl->o_size, sum,
                kurtosis, margin);
  }
}

/* -------------------------------------------------------- runs / loops -- */

/* recur-nn-init.c:384-394 */
static float log_normal_random_sign(rand_ctx *rng, float mean, float stddev, float bound) {
  float x;
  do {
    x = ramd_cheap_gaussian_noise(rng);
  } while (fabsf(x) > bound);
  float w = mean * ramd_fast_expf(x * stddev);
  return (ramd_rand64(rng) & 1) ? w : -w;
}

/* recur-nn-init.c:397-402 */
static void link_random_input(RecurNN *net, int dest, float deviation) {
  int input = rand_range(&net->rng, 0, net->input_size);
  net->ih_weights[(size_t)(net->hidden_size + 1 + input) * net->h_size + dest] =
      ramd_cheap_gaussian_noise(&net->rng) * deviation;
}

/* recur-nn-init.c:408-417 */
static void link_hidden(RecurNN *net, int s, int e, float gain, float input_probability,
                        float input_magnitude) {
  float weight = log_normal_random_sign(&net->rng, gain, 0.25, 3.0);
  net->ih_weights[(size_t)s * net->h_size + e] = weight;
  if (ramd_rand_double(&net->rng) < input_probability) {
    link_random_input(net, e, input_magnitude);
  }
}

/* recur-nn-init.c:419-491: chains (optionally closed into loops) of hidden
 * nodes drawn without replacement from a shrinking pool */
static void init_runs(RecurNN *net, int n_loops, int len_mean, int len_stddev, float gain,
                      float input_probability, float input_magnitude, int loop,
                      int crossing_paths, int inputs_miss, int input_at_start) {
  fprintf(stderr,
          "n_loops %d len_mean %d, len_stddev %d, gain %g, input_probability %g, "
          "input_magnitude %g loop %d crossing_paths %d, inputs_miss %d input_at_start %d\n",
          n_loops, len_mean, len_stddev, gain, input_probability, input_magnitude, loop,
          crossing_paths, inputs_miss, input_at_start);
  int bound = net->hidden_size + 1;
  int *pool = malloc(bound * sizeof(int));
  int used = bound; /* forces a refill on the first pass */
  int total = 0;
  double linked_input_p = inputs_miss ? 0 : input_probability;
  double missing_input_p = inputs_miss ? input_probability : 0;
  for (int n = 0; n < n_loops; n++) {
    int len = ramd_cheap_gaussian_noise(&net->rng) * len_stddev + len_mean + 0.5;
    len = RAMD_MIN(RAMD_MAX(2, len), net->hidden_size);
    if (used + len + inputs_miss >= bound || crossing_paths) {
      for (int k = 0; k < bound; k++) {
        pool[k] = k;
      }
      used = 1;
    }
    int j = rand_range(&net->rng, used, bound);
    int first = pool[j], e = first, s;
    if (input_at_start && input_magnitude) {
      link_random_input(net, e, input_magnitude);
    }
    for (int m = 0; m < len; m++, used++) {
      pool[j] = pool[used];
      s = e;
      if (crossing_paths == 2) {
        e = rand_range(&net->rng, 1, bound);
      } else {
        j = rand_range(&net->rng, used, bound);
        e = pool[j];
      }
      link_hidden(net, s, e, gain, linked_input_p, input_magnitude);
    }
    if (loop) {
      link_hidden(net, e, first, gain, linked_input_p, input_magnitude);
    }
    if (ramd_rand_double(&net->rng) < missing_input_p && used < bound) {
      j = rand_range(&net->rng, used, bound);
      e = pool[j];
      pool[j] = pool[used];
      used++;
      link_random_input(net, e, input_magnitude);
    }
    total += len;
  }
  free(pool);
  fprintf(stderr, "mean loop len %3g\n", (double)total / n_loops);
}

/* recur-nn-init.c:625-646 */
static void runs_submethod(RecurNN *net, struct RecurInitialisationParameters *p) {
  if (p->submethod != p->method) {
    p->method = p->submethod;
    rnn_randomise_weights_clever(net, p);
    p->method = RNN_INIT_RUNS;
    fprintf(stderr, "used submethod %d%s%s\n", p->submethod,
            p->bias_uses_submethod ? ", bias too" : "",
            p->inputs_use_submethod ? ", inputs too" : "");
  }
  float *mem = net->ih_weights;
  size_t rows = p->inputs_use_submethod ? net->h_size : net->i_size;
  if (p->bias_uses_submethod) {
    rows--;
    mem += net->h_size;
  }
  memset(mem, 0, rows * net->h_size * sizeof(float));
}

/* ------------------------------------------------------- public entries -- */

/* recur-nn.h:287 / recur-nn-init.c:649-683 */
void rnn_randomise_weights_clever(RecurNN *net, struct RecurInitialisationParameters *p) {
  ramd_need_host(net, RNN_AMD_WEIGHTS);
  ramd_rng_to_host(net); /* the draws below continue the generator wherever it last advanced */
  switch (p->method) {
  case RNN_INIT_ZERO:
    memset(net->ih_weights, 0, (size_t)net->ih_size * sizeof(float));
    memset(net->ho_weights, 0, (size_t)net->ho_size * sizeof(float));
    break;
  case RNN_INIT_FAN_IN:
    init_fan_in(net, p->fan_in_sum, p->fan_in_step, p->fan_in_min, p->fan_in_ratio);
    break;
  case RNN_INIT_FLAT:
    init_flat(net, p->flat_variance, p->flat_shape, p->flat_perforation);
    break;
  case RNN_INIT_RUNS:
    runs_submethod(net, p);
    init_runs(net, p->run_n, p->run_len_mean, p->run_len_stddev, p->run_gain,
              p->run_input_probability, p->run_input_magnitude, p->run_loop,
              p->run_crossing_paths, p->run_inputs_miss, p->run_input_at_start);
    break;
  default:
    break;
  }
  ramd_rng_from_host(net);
  ramd_host_wrote(net, RNN_AMD_WEIGHTS);
}

/* recur-nn.h:291 / recur-nn-init.c:685-719 */
void rnn_init_default_weight_parameters(RecurNN *net, struct RecurInitialisationParameters *q) {
  memset(q, 0, sizeof(*q));
  q->method = RNN_INIT_FLAT;
  q->submethod = RNN_INIT_FLAT;
  q->fan_in_ratio = net->input_size * 1.0f / net->hidden_size;
  q->fan_in_sum = 3.0;
  q->fan_in_step = 0.3;
  q->fan_in_min = 0.1;
  q->flat_variance = RNN_INITIAL_WEIGHT_VARIANCE_FACTOR / net->h_size;
  q->flat_shape = RNN_INIT_DIST_UNIFORM;
  q->flat_perforation = 0.7;
  q->run_input_probability = .17;
  q->run_input_magnitude = 0.2;
  q->run_gain = 0.17;
  q->run_len_mean = net->hidden_size / 1.0;
  q->run_len_stddev = net->hidden_size / 3.0f;
  q->run_n = net->h_size * 0.085;
  q->run_loop = 1;
}

/* recur-nn.h:288 / recur-nn-init.c:729-735 */
void rnn_randomise_weights_simple(RecurNN *net, const rnn_init_method method) {
  struct RecurInitialisationParameters p;
  rnn_init_default_weight_parameters(net, &p);
  p.method = method;
  rnn_randomise_weights_clever(net, &p);
}

/* recur-nn.h:289 / recur-nn-init.c:721-726 */
void rnn_randomise_weights_auto(RecurNN *net) { rnn_randomise_weights_simple(net, RNN_INIT_FLAT); }

/* recur-nn-helpers.h:84-102 */
static void perforate(float *array, int len, float dropout, rand_ctx *rng) {
  if (dropout == 0.5f) {
    for (int i = 0; i < len;) {
      uint64_t bits = ramd_rand64(rng);
      int end = i + RAMD_MIN(64, len - i);
      for (; i < end; i++) {
        array[i] = (bits & 1) ? array[i] : 0; /* the reference never shifts `bits` */
      }
    }
  } else {
    for (int i = 0; i < len; i++) {
      array[i] = (ramd_rand_double(rng) > dropout) ? array[i] : 0.0f;
    }
  }
}

/* recur-nn.h:324 / recur-nn-init.c:739-742 */
void rnn_perforate_weights(RecurNN *net, float p) {
  ramd_need_host(net, RNN_AMD_WEIGHTS);
  ramd_rng_to_host(net);
  rand_ctx tmp, *rng = ramd_shared_rng(net, &tmp);
  perforate(net->ih_weights, net->ih_size, p, rng);
  perforate(net->ho_weights, net->ho_size, p, rng);
  ramd_rng_from_host(net);
  ramd_host_wrote(net, RNN_AMD_WEIGHTS);
}

/* recur-nn.h:328 / recur-nn.c:857-883 */
void rnn_weight_noise(RecurNN *net, float deviation) {
  ramd_need_host(net, RNN_AMD_WEIGHTS);
  /* on the batched path the device holds the stream's generator (presynaptic noise, the
   * multi-head leakage draws): continue from there and hand the advanced state back */
  ramd_rng_to_host(net);
  rand_ctx tmp, *rng = ramd_shared_rng(net, &tmp); /* sharded sets: every replica adds the same noise */
  int rows = net->hidden_size + 1 + net->input_size;
  for (int y = 0; y < rows; y++) {
    float *row = net->ih_weights + 1 + (size_t)y * net->h_size;
    for (int i = 0; i < net->hidden_size; i++) {
      row[i] += ramd_cheap_gaussian_noise(rng) * deviation;
    }
  }
  for (int y = 0; y < net->hidden_size + 1; y++) {
    float *row = net->ho_weights + (size_t)y * net->o_size;
    for (int i = 0; i < net->output_size; i++) {
      row[i] += ramd_cheap_gaussian_noise(rng) * deviation;
    }
  }
  if (net->bottom_layer) {
    /* recur-nn.c:877-882 walks the [i_size][o_size] matrix with a row stride of i_size
     * and output_size rows; kept as it is */
    RecurExtraLayer *bl = net->bottom_layer;
    for (int y = 0; y < bl->output_size; y++) {
      float *row = bl->weights + 1 + (size_t)y * bl->i_size;
      for (int i = 0; i < bl->input_size; i++) {
        row[i] += ramd_cheap_gaussian_noise(rng) * deviation;
      }
    }
  }
  ramd_rng_from_host(net);
  ramd_host_wrote(net, RNN_AMD_WEIGHTS);
}

static void fill(float *a, size_t n, float x) {
  for (size_t i = 0; i < n; i++) {
    a[i] = x;
  }
}

/* recur-nn.h:330 / recur-nn-init.c:359-369 */
void rnn_set_momentum_values(RecurNN *net, float x) {
  ramd_need_host(net, RNN_AMD_MOMENTUMS);
  fill(net->bptt->ho_momentum, net->ho_size, x);
  fill(net->bptt->ih_momentum, net->ih_size, x);
  if (net->bottom_layer) {
    fill(net->bottom_layer->momentums, (size_t)net->bottom_layer->i_size * net->bottom_layer->o_size, x);
  }
  ramd_host_wrote(net, RNN_AMD_MOMENTUMS);
}

/* recur-nn.h:331 / recur-nn-init.c:370-380 */
void rnn_set_aux_values(RecurNN *net, float x) {
  ramd_need_host(net, RNN_AMD_MOMENTUMS);
  fill(net->bptt->ho_aux, net->ho_size, x);
  fill(net->bptt->ih_aux, net->ih_size, x);
  if (net->bottom_layer && net->bottom_layer->aux) {
    fill(net->bottom_layer->aux, (size_t)net->bottom_layer->i_size * net->bottom_layer->o_size, x);
  }
  ramd_host_wrote(net, RNN_AMD_MOMENTUMS);
}

/* recur-nn.h:333 / recur-nn.c:1082-1134: in the hidden->hidden block, keep
 * only the diagonal (and a "friend" sub-diagonal) of columns start..stop */
void rnn_zap_non_diagonals(RecurNN *net, int start, int stop, int friend_n) {
  int h_end = net->hidden_size + 1;
  int friend_start = start - friend_n;
  if (start >= h_end || start < 0) {
    return;
  }
  if (start > stop) {
    fprintf(stderr, "diagonal zap start is %d, stop is %d; doing nothing\n", start, stop);
    return;
  }
  if (stop > h_end) {
    fprintf(stderr, "net->hidden size is %d, diagonal zap stop is %d; truncating\n",
            net->hidden_size, stop);
    stop = h_end;
  }
  if (friend_n > stop - start || friend_start <= 0) {
    fprintf(stderr, "diagonal friend parameter %d is stupid: start is %d stop %d, size %d "
                    "...ignoring it\n", friend_n, start, stop, h_end);
    friend_n = 0;
  }
  ramd_need_host(net, RNN_AMD_WEIGHTS);
  int span = stop - start;
  for (int y = 0; y < h_end; y++) {
    float *seg = net->ih_weights + (size_t)y * net->h_size + start;
    int keep = -1; /* column of this row's surviving weight, if any */
    if (y >= friend_start && y < start) {
      keep = y - friend_start;
    } else if (y >= start && y < stop) {
      keep = y - start;
    }
    for (int x = 0; x < span; x++) {
      if (x != keep) {
        seg[x] = 0.0f;
      }
    }
  }
  ramd_host_wrote(net, RNN_AMD_WEIGHTS);
}

/* recur-nn.h:334 / recur-nn.c:1136-1145 */
void rnn_clear_diagonal_only_section(RecurNN *net, uint len, uint friends) {
  if (len == 0) {
    /* the reference computes start == stop and zeroes nothing */
    return;
  }
  int h_end = net->hidden_size + 1;
  friends = RAMD_MIN(friends, len);
  rnn_zap_non_diagonals(net, h_end - (int)len, h_end, (int)friends);
}

/* recur-nn.h:294 / recur-nn.c:1027-1076: iterative rescaling toward a target
 * gain, driven by rectified gaussian probes through the hidden->hidden block */
void rnn_scale_initial_weights(RecurNN *net, float target_gain) {
  ramd_need_host(net, RNN_AMD_WEIGHTS);
  ramd_rng_to_host(net);
  int h_size = net->h_size;
  float *in = malloc(sizeof(float) * h_size), *out = malloc(sizeof(float) * h_size);
  double net_adjustment = 1.0, tail_in = 0, tail_out = 0;
  const double generations = 10000;
  for (double j = 1; j < generations; j++) {
    float sum_in = 1, sum_out = 0;
    in[0] = 1;
    for (int i = 1; i < net->hidden_size; i++) {
      float n = RAMD_MAX(ramd_cheap_gaussian_noise(&net->rng), 0);
      in[i] = n;
      sum_in += n * n;
    }
    for (int i = net->hidden_size; i < h_size; i++) {
      in[i] = 0;
      out[i] = 0;
    }
    /* calculate_interlayer over the first hidden_size + 1 rows (recur-nn.c:1050) */
    memset(out, 0, sizeof(float) * h_size);
    for (int y = 0; y < net->hidden_size + 1; y++) {
      float v = in[y];
      if (v) {
        const float *row = net->ih_weights + (size_t)y * h_size;
        for (int x = 0; x < h_size; x++) {
          out[x] += v * row[x];
        }
      }
    }
    out[0] = 1.0f;
    for (int i = 0; i < net->hidden_size; i++) {
      float h = out[i] > 0.0f ? out[i] : 0.0f;
      out[i] = h;
      sum_out += h * h;
    }
    double ratio = sum_out / sum_in;
    double adj = (target_gain * 10 + j) / (ratio * 10 + j);
    net_adjustment *= adj;
    float fadj = adj;
    for (int i = 0; i < net->ih_size; i++) {
      net->ih_weights[i] *= fadj;
    }
    if (j > generations * 0.95) {
      tail_in += sum_in;
      tail_out += sum_out;
    }
  }
  free(in);
  free(out);
  fprintf(stderr, "scaled toward target gain %.3f; hit roughly %.3f; adjusted by %.3f\n",
          target_gain, tail_out / tail_in, net_adjustment);
  ramd_rng_from_host(net);
  ramd_host_wrote(net, RNN_AMD_WEIGHTS);
}

/* recur-nn-init.c:825-844 (Welford) */
static void mean_and_variance(const float *array, int width, int height, int stride, int offset,
                              const char *name) {
  float mean = 0, var = 0, n = 0;
  for (int y = 0; y < height; y++) {
    for (int x = offset; x < width + offset; x++) {
      n++;
      float val = array[(size_t)y * stride + x];
      float delta = val - mean;
      mean += delta / n;
      var += delta * (val - mean);
    }
  }
  var /= n;
  fprintf(stderr, "%s: mean %3g variance %3g (std dev %3g) n %d\n", name, mean, var, sqrt(var),
          (int)n);
}

/* recur-nn.h:296 / recur-nn-init.c:846-861 */
void rnn_print_net_stats(RecurNN *net) {
  ramd_need_host(net, RNN_AMD_WEIGHTS);
  mean_and_variance(net->ih_weights, net->hidden_size, net->hidden_size + net->input_size + 1,
                    net->h_size, 1, "ih_weights");
  mean_and_variance(net->ho_weights, net->output_size, net->hidden_size + 1, net->o_size, 0,
                    "ho_weights");
  if (net->bottom_layer) {
    RecurExtraLayer *bl = net->bottom_layer;
    mean_and_variance(bl->weights, bl->output_size, bl->input_size, bl->o_size, 1,
                      "bottom weights");
  }
}
