/* recur_amd.h -- C ABI of librecur_amd.so, the MI355X (gfx950) implementation of
 * Recur's RNN core.
 *
 * Part 1 of this header is the drop-in contract: the public structs, enums and
 * extern functions that the reference exposes in recur-nn.h (the reference has
 * no FFI layer; callers link recur-nn.o / recur-nn-init.o / recur-nn-io.o
 * statically and poke struct fields).  Every
 * declaration cites the reference line it replaces.  Field order, field types
 * and enum values are ABI and therefore identical; everything else in this
 * repository is written from scratch.
 *
 * Part 2 is additive: batched ("training set at once") entry points and the
 * host/device coherence calls a GPU-resident core needs.  A batched call is
 * defined to be equal to looping the per-net call of Part 1 over the set.
 *
 * Plain pointers and sizes only; no C++ or torch types.
 */
#ifndef RECUR_AMD_H
#define RECUR_AMD_H 1

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* Short integer names used throughout the reference (recur-common.h:83-91). */
#ifndef RECUR_AMD_NO_SHORT_TYPES
typedef uint64_t u64;
typedef int64_t s64;
typedef uint32_t u32;
typedef int32_t s32;
typedef uint16_t u16;
typedef int16_t s16;
typedef uint8_t u8;
typedef int8_t s8;
typedef unsigned int uint;
#endif

/* Jenkins small-fast 64 bit PRNG state (recur-rng.h:15-20).  It is saved in
 * net files as 32 raw bytes, so the layout is part of the file format. */
typedef struct _rand_ctx {
  uint64_t a;
  uint64_t b;
  uint64_t c;
  uint64_t d;
} rand_ctx;

#define RECUR_RNG_RANDOM_SEED (-1ULL) /* recur-rng.h:13: seed from the clock   */
#define RECUR_RNG_SUBSEED (-2ULL)     /* recur-nn.h:15: seed from parent's rng */

/* Numeric constants of the algorithm (recur-nn.h:19-57, 107). */
#define RANDOM_DAMAGE_FACTOR 0.5f
#define MAX_TOP_ERROR_FACTOR 2.0f
#define MAX_ERROR_GAIN 2.0f
#define ERROR_GAIN_CEILING 1.0f
#define BASE_MIN_ERROR_FACTOR 1e-12f
#define MAX_MIN_ERROR_FACTOR 1e-2f
#define ABS_MIN_ERROR_FACTOR 1e-20f
#define MIN_ERROR_GAIN 1e-8f
#define RNN_HIDDEN_PENALTY 0.0f
#define HIDDEN_MEAN_SOFT_TOP 16.0f
#define INPUT_MEAN_SOFT_TOP 16.0f
#define RNN_INITIAL_WEIGHT_VARIANCE_FACTOR 2.0f
#define WEIGHT_SCALE (1.0f - 1e-6f)
#define RNN_CONDITIONING_INTERVAL 8
#define RNN_TALL_POPPY_THRESHOLD 1.0f
#define RNN_TALL_POPPY_SCALE 0.99f
#define RNN_LAWN_MOWER_THRESHOLD 10.0f
#define RNN_MOMENTUM_WEIGHT 0.5f
#define RNN_COND_USE_OFFSET 16

/* Position of each conditioning task in the 8-generation cycle
 * (recur-nn.h:70-76). */
enum {
  RNN_COND_BIT_SCALE = 0U,
  RNN_COND_BIT_ZERO = 2U,
  RNN_COND_BIT_LAWN_MOWER = 3U,
  RNN_COND_BIT_TALL_POPPY = 4U,
  RNN_COND_BIT_RAND = 6U
};

/* net->flags bits (recur-nn.h:78-103). */
enum {
  RNN_NET_FLAG_OWN_BPTT = 1,
  RNN_NET_FLAG_OWN_WEIGHTS = 2,
  RNN_NET_FLAG_LOG_APPEND = 8,
  RNN_NET_FLAG_LOG_HIDDEN_SUM = 16,
  RNN_NET_FLAG_LOG_WEIGHT_SUM = 32,
  RNN_NET_FLAG_BPTT_ADAPTIVE_MIN_ERROR = 64,
  RNN_NET_FLAG_NO_MOMENTUMS = 128,
  RNN_NET_FLAG_NO_DELTAS = 256,
  RNN_NET_FLAG_BOTTOM_LAYER = 1024,
  RNN_NET_FLAG_AUX_ARRAYS = 2048,

  RNN_COND_USE_SCALE = (1 << (RNN_COND_BIT_SCALE + RNN_COND_USE_OFFSET)),
  RNN_COND_USE_ZERO = (1 << (RNN_COND_BIT_ZERO + RNN_COND_USE_OFFSET)),
  RNN_COND_USE_LAWN_MOWER = (1 << (RNN_COND_BIT_LAWN_MOWER + RNN_COND_USE_OFFSET)),
  RNN_COND_USE_TALL_POPPY = (1 << (RNN_COND_BIT_TALL_POPPY + RNN_COND_USE_OFFSET)),
  RNN_COND_USE_RAND = (1 << (RNN_COND_BIT_RAND + RNN_COND_USE_OFFSET)),

  RNN_NET_FLAG_STANDARD = (RNN_NET_FLAG_OWN_BPTT | RNN_NET_FLAG_OWN_WEIGHTS |
                           RNN_COND_USE_ZERO | RNN_NET_FLAG_LOG_HIDDEN_SUM)
};

/* recur-nn.h:109-119 */
typedef enum {
  RNN_MOMENTUM_WEIGHTED = 0,
  RNN_MOMENTUM_NESTEROV,
  RNN_MOMENTUM_SIMPLIFIED_NESTEROV,
  RNN_MOMENTUM_CLASSICAL,
  RNN_ADAGRAD,
  RNN_ADADELTA,
  RNN_RPROP,
  RNN_LAST_LEARNING_METHOD
} rnn_learning_method;

/* recur-nn.h:121-128 */
typedef enum {
  RNN_INIT_ZERO = 0,
  RNN_INIT_FLAT,
  RNN_INIT_FAN_IN,
  RNN_INIT_RUNS,
  RNN_INIT_LAST
} rnn_init_method;

/* recur-nn.h:130-140 (values are stored in net files) */
typedef enum {
  RNN_RELU = 1,
  RNN_RESQRT,
  RNN_RESERVED_ACTIVATION_1,
  RNN_RESERVED_ACTIVATION_2,
  RNN_RECLIP20 = 5,
  RNN_ACTIVATION_LAST
} rnn_activation;

/* recur-nn.h:142-151 */
typedef enum {
  RNN_INIT_DIST_UNIFORM = 1,
  RNN_INIT_DIST_GAUSSIAN,
  RNN_INIT_DIST_LOG_NORMAL,
  RNN_INIT_DIST_SEMICIRCLE,
  RNN_INIT_DIST_DEFAULT
} rnn_init_distribution;

typedef struct _RecurNN RecurNN;
typedef struct _RecurNNBPTT RecurNNBPTT;
typedef struct _RecurExtraLayer RecurExtraLayer;
typedef struct _RecurErrorRange RecurErrorRange;

/* One net == one stream of the synchronic mini-batch (recur-nn.h:158-186).
 * All pointers are host pointers; see rnn_amd_sync_host() below for when the
 * large arrays they point at are current. */
struct _RecurNN {
  int i_size; /* padded: 1 bias + hidden feedback + real inputs, rounded up to 4 */
  int h_size; /* padded: 1 bias + hidden_size */
  int o_size; /* padded output_size */
  int input_size;
  int hidden_size;
  int output_size;
  int ih_size; /* i_size * h_size */
  int ho_size; /* h_size * o_size */
  uint32_t flags;
  FILE *log;
  float *mem; /* private */
  float *input_layer;
  float *hidden_layer;
  float *output_layer;
  float *ih_weights; /* row major [i_size][h_size] */
  float *ho_weights; /* row major [h_size][o_size] */
  float *real_inputs;
  rand_ctx rng;
  RecurNNBPTT *bptt;
  RecurExtraLayer *bottom_layer;
  char *metadata;
  uint32_t generation;
  float presynaptic_noise;
  rnn_activation activation;
};

/* Training state of one stream (recur-nn.h:188-209). */
struct _RecurNNBPTT {
  int depth;
  int index;
  float *i_error;
  float *h_error;
  float *o_error;
  float *ih_momentum;
  float *ho_momentum;
  float *history; /* ring of depth slots, i_size floats each */
  float *ih_delta;
  float *ho_delta;
  float *ih_delta_tmp;
  float *ih_aux;
  float *ho_aux;
  float *mem; /* private */
  float learn_rate;
  float ih_scale;
  float ho_scale;
  float momentum;
  float momentum_weight;
  float min_error_factor;
};

/* Optional dense layer below the recurrent one (recur-nn.h:211-227). */
struct _RecurExtraLayer {
  float *mem;
  float *weights;
  float *momentums;
  float *aux;
  float *delta;
  float *inputs;
  float *outputs;
  float *i_error;
  float *o_error;
  float learn_rate_scale;
  int input_size;
  int output_size;
  int i_size;
  int o_size;
  int overlap;
};

/* recur-nn.h:230-258 */
struct RecurInitialisationParameters {
  rnn_init_method method;
  rnn_init_method submethod;
  int bias_uses_submethod;
  int inputs_use_submethod;

  float fan_in_sum;
  float fan_in_step;
  float fan_in_min;
  float fan_in_ratio;

  float flat_variance;
  rnn_init_distribution flat_shape;
  double flat_perforation;

  float run_input_probability;
  float run_input_magnitude;
  float run_gain;
  float run_len_mean;
  float run_len_stddev;
  int run_n;
  int run_loop;
  int run_crossing_paths;
  int run_inputs_miss;
  int run_input_at_start;
};

/* Column range of the output layer that carries error (recur-nn.h:260-265);
 * arrays of these end with an entry whose start is negative. */
struct _RecurErrorRange {
  int start;
  int len;
};

/* ----- construction / destruction (recur-nn.h:269-300) ------------------- */
RecurNN *rnn_new(uint input_size, uint hidden_size, uint output_size, u32 flags,
                 u64 rng_seed, const char *log_file, int depth, float learn_rate,
                 float momentum, float presynaptic_noise, rnn_activation activation);
RecurNN *rnn_clone(RecurNN *parent, u32 flags, u64 rng_seed, const char *log_file);
RecurExtraLayer *rnn_new_extra_layer(int input_size, int output_size, int overlap,
                                     u32 flags);
RecurNN *rnn_new_with_bottom_layer(int n_inputs, int r_input_size, int hidden_size,
                                   int output_size, u32 flags, u64 rng_seed,
                                   const char *log_file, int bptt_depth,
                                   float learn_rate, float momentum,
                                   float presynaptic_noise, rnn_activation activation,
                                   int convolutional_overlap);
void rnn_set_log_file(RecurNN *net, const char *log_file, int append_dont_truncate);
void rnn_delete_net(RecurNN *net);
RecurNN **rnn_new_training_set(RecurNN *prototype, int n_nets);
void rnn_delete_training_set(RecurNN **nets, int n_nets, int leave_prototype);

/* ----- weight initialisation (recur-nn.h:287-296, 324, 333-334) ---------- */
void rnn_randomise_weights_clever(RecurNN *net, struct RecurInitialisationParameters *p);
void rnn_randomise_weights_simple(RecurNN *net, const rnn_init_method method);
void rnn_randomise_weights_auto(RecurNN *net);
void rnn_init_default_weight_parameters(RecurNN *net,
                                        struct RecurInitialisationParameters *q);
void rnn_scale_initial_weights(RecurNN *net, float target_gain);
void rnn_print_net_stats(RecurNN *net);
void rnn_perforate_weights(RecurNN *net, float p);
void rnn_zap_non_diagonals(RecurNN *net, int start, int stop, int n_friends);
void rnn_clear_diagonal_only_section(RecurNN *net, uint len, uint friends);

/* ----- the hot path (recur-nn.h:302, 309-322) ---------------------------- */
float *rnn_opinion(RecurNN *net, const float *inputs, float presynaptic_noise);
void rnn_bptt_clear_deltas(RecurNN *net);
void rnn_bptt_advance(RecurNN *net);
void rnn_bptt_calculate(RecurNN *net, uint batch_size);
void rnn_apply_learning(RecurNN *net, int learning_style, float momentum);
float rnn_calculate_momentum_soft_start(float generation, float momentum,
                                        float momentum_soft_start);
void rnn_bptt_calc_deltas(RecurNN *net, int accumulate_delta,
                          RecurErrorRange *top_error_ranges);
void rnn_condition_net(RecurNN *net);
void rnn_log_net(RecurNN *net);
void rnn_forget_history(RecurNN *net, int bptt_too);

/* ----- cold helpers (recur-nn.h:304, 328-331) ---------------------------- */
void rnn_multi_pgm_dump(RecurNN *net, const char *dumpees, const char *basename);
void rnn_weight_noise(RecurNN *net, float deviation);
void rnn_set_momentum_values(RecurNN *net, float x);
void rnn_set_aux_values(RecurNN *net, float x);

/* ----- CDB net files, format version 10 (recur-nn.h:306-307) ------------- */
RecurNN *rnn_load_net(const char *filename);
int rnn_save_net(RecurNN *net, const char *filename, int backup);

/* recur-nn.h:337-349: one "name value" line per call on net->log. */
static inline void rnn_log_float(RecurNN *net, char *name, float value) {
  if (net->log) {
    fprintf(net->log, "%s %.5g\n", name, value);
  }
}
static inline void rnn_log_int(RecurNN *net, char *name, int value) {
  if (net->log) {
    fprintf(net->log, "%s %d\n", name, value);
  }
}

/* ======================================================================== */
/* Part 2: additive entry points.                                            */

/* Number of usable HIP devices.  With 0 devices the host-only functions
 * (construction, initialisation, file I/O, PRNG) still work; every function
 * that has to compute on the device prints a message and abort()s.  There is
 * no CPU fallback. */
int rnn_amd_device_count(void);
/* Select the HIP device (default: LOCAL_RANK from the environment, else 0)
 * and optionally an externally created hipStream_t for all launches. */
void rnn_amd_use_device(int device, void *hip_stream);
void *rnn_amd_current_stream(void);
const char *rnn_amd_version(void);

/* Which copies to bring up to date (bit set). */
enum {
  RNN_AMD_WEIGHTS = 1,   /* ih_weights, ho_weights, bottom layer weights     */
  RNN_AMD_MOMENTUMS = 2, /* ih/ho_momentum, ih/ho_aux                       */
  RNN_AMD_DELTAS = 4,    /* ih_delta, ho_delta                              */
  RNN_AMD_STREAM = 8,    /* this net's history, layers, error vectors, scalars */
  RNN_AMD_ALL_STREAMS = 16, /* the same for every net sharing its weights   */
  RNN_AMD_EVERYTHING = 31
};
/* The device copy of the large arrays is authoritative once a net has been
 * used on the device.  rnn_amd_sync_host makes the host arrays behind the
 * struct pointers current; rnn_amd_host_written tells the library that the
 * caller modified them through the pointers (e.g. net->ih_weights[k] += x).
 * Library functions do both themselves; the per-net calls of Part 1 also keep
 * the small per-stream vectors (input/hidden/output layer, o_error, h_error,
 * i_error, bptt scalars) current on return. */
void rnn_amd_sync_host(RecurNN *net, int what);
void rnn_amd_host_written(RecurNN *net, int what);

/* A training set opened for batched work: nets[0] is the prototype and the
 * others are its rnn_new_training_set() clones. */
typedef struct RnnAmdSet RnnAmdSet;
RnnAmdSet *rnn_amd_set_open(RecurNN **nets, int n_nets); /* (the array is copied: it need not outlive the call) */
void rnn_amd_set_close(RnnAmdSet *set); /* brings the host structs up to date first */
void rnn_amd_set_drop(RnnAmdSet *set);  /* the same without the copy-back: state stays on the device */
int rnn_amd_set_size(const RnnAmdSet *set);

/* == rnn_bptt_advance() on every net of the set. */
void rnn_amd_set_advance(RnnAmdSet *set);

/* == rnn_opinion(nets[j], inputs + j * ld_inputs, nets[j]->presynaptic_noise)
 * for every j.  inputs (host, n_nets rows) may be NULL when the inputs were
 * set by one of the calls below.  If outputs is not NULL the n_nets x o_size
 * answers are copied there (host), which synchronises. */
void rnn_amd_set_opinion(RnnAmdSet *set, const float *inputs, int ld_inputs,
                         float *outputs);
/* One-hot inputs (charmodel-helpers.h:16-33): hot[j] is the input index of
 * stream j (host array). */
void rnn_amd_set_one_hot_opinion(RnnAmdSet *set, const int *hot, float *outputs);

/* Upload n_nets x o_size error rows (host) into bptt->o_error of the streams. */
void rnn_amd_set_put_o_error(RnnAmdSet *set, const float *o_error, int ld);
/* Device version of net_error_bptt's loss (charmodel-predict.c:18-27):
 * o_error = onehot(target) - softmax(output) with the reference's fast_expf
 * (badmaths.h:14-29, 71-141).  Per stream it adds the error of the target
 * class, log2(1 - that error) (capped, charmodel-helpers.h:11-13) and the
 * "winner == target" count into device accumulators read by
 * rnn_amd_set_read_stats. */
void rnn_amd_set_softmax_error(RnnAmdSet *set, const int *target);

/* == for every j with (active == NULL || active[j]):
 *      rnn_bptt_calc_deltas(nets[j], accumulate || not the first, ranges)
 * i.e. with accumulate == 0 the deltas are zeroed first, exactly as the
 * reference callers do with "j ? 1 : 0" (charmodel-predict.c:309). */
void rnn_amd_set_calc_deltas(RnnAmdSet *set, int accumulate,
                             RecurErrorRange *top_error_ranges, const u8 *active);

/* The loss of gstclassify's train_channel (gstclassify.c:2070-2119) for every stream, on
 * the device, after an rnn_amd_set_opinion: the outputs are n_groups class groups
 * (group i = outputs [group_offset[i], group_offset[i] + group_size[i])); for stream j and
 * group i with 0 <= targets[j * n_groups + i] < group_size[i] the group's error becomes
 * -softmax with +1 on the target, otherwise zeros (the caller passes -1 where the reference
 * skips a group: unknown target, ignored windows, the balanced-sampling draw); if any group
 * of a stream was trained its whole error row is multiplied by error_weight (may be NULL).
 * trained[j] (may be NULL) receives whether stream j has anything to back-propagate: pass
 * it to rnn_amd_set_calc_deltas as the active mask.  Wins, wrongness and trained groups
 * add to the correct / error / count accumulators of rnn_amd_set_read_stats. */
void rnn_amd_set_grouped_softmax_error(RnnAmdSet *set, int n_groups, const int *group_offset,
                                       const int *group_size, const int *targets,
                                       const float *error_weight, u8 *trained);
/* One generation of the multi-head text model (charmodel-multi-predict.c:17-58, 244-256) for
 * every stream of the set: rnn_bptt_advance, a one-hot opinion of hot[j] with the net's
 * presynaptic noise, the multi-head softmax error against next[j] -- the head
 * target_class[j] always, every other head with probability `leakage` drawn from the
 * stream's own generator --, and rnn_bptt_calc_deltas with the error ranges that loss
 * produces.  accumulate applies to the first stream; the others accumulate.  The output
 * layer is output_size / alphabet_len heads (at most 64).  Loss statistics of the own
 * head go to the accumulators rnn_amd_set_read_stats reads (error, entropy, count). */
void rnn_amd_set_multi_step_deltas(RnnAmdSet *set, const int *hot, const int *next,
                                   const int *target_class, int alphabet_len, float leakage,
                                   int accumulate);
/* The same generation with its update, as ONE call: rnn_amd_set_multi_step_deltas(..., accumulate 0) + rnn_apply_learning(
 * nets[0], learning_style, momentum) (charmodel-multi-predict.c:244-262: text_train's step and its rnn_apply_learning).
 * The same results; knowing the update that follows, the weight-delta GEMM carries it out in its own epilogue where the
 * rule is ADAGRAD (the trainer's default) or the momentum rule -- no optimiser launch (round 6). */
void rnn_amd_set_multi_step(RnnAmdSet *set, const int *hot, const int *next, const int *target_class,
                            int alphabet_len, float leakage, int learning_style, float momentum);

/* The two halves of rnn_amd_set_multi_step_deltas with the text on the device (stream j reads
 * text[o], is scored against text[o + 1], o as in rnn_amd_set_char_step): the loss leaves
 * o_error and one error-range list per stream on the device, the deltas call consumes them.
 * In between a caller may apply the previous batch's deltas, as the reference's text_train
 * does (charmodel-multi-predict.c:244-252).  target_class (host, n_nets) may be NULL to keep
 * the classes of the previous call. */
void rnn_amd_set_multi_text_loss(RnnAmdSet *set, int i, const int *target_class, int alphabet_len,
                                 float leakage);
void rnn_amd_set_multi_calc_deltas(RnnAmdSet *set, int accumulate);
/* rnn_bptt_advance (if advance) + one_hot_opinion of each stream's text symbol and nothing
 * else (rnn_char_multitext_spin's step, charmodel-multi-predict.c:293-297); 0 <= i < len */
void rnn_amd_set_text_opinion(RnnAmdSet *set, int i, int advance);
/* rnnca's loss on the device (gstrnnca.c:701-714, train_net) after an rnn_amd_set_opinion:
 * for every stream the first n outputs become their fast_sigmoid IN PLACE (badmaths.h:33-44,
 * as the reference's fast_sigmoid_array(answer, answer, 3) does to output_layer) and
 * o_error[i] = a (1 - a) (targets[j * ld + i] - a); the error never visits the host. */
void rnn_amd_set_sigmoid_mse_error(RnnAmdSet *set, const float *targets, int ld, int n);
/* rnn_amd_set_opinion (outputs not fetched) + rnn_amd_set_sigmoid_mse_error, resp. + rnn_amd_set_grouped_softmax_error,
 * as ONE call: what train_net does per cell between two calc_deltas (gstrnnca.c:693-714) and train_channel per channel
 * (gstclassify.c:2070-2119).  Same results as the two calls; the hidden layer's activation, the output layer, the loss
 * and the top layer's backprop then share one launch (one workgroup per stream: the hidden row, the outputs and the
 * output error never leave the CU), and the rnn_amd_set_calc_deltas that follows -- with `trained` as its active mask
 * in the class-group case, no error ranges -- finds the top layer's backprop done.  Shapes outside that launch's
 * reach (more than 64 outputs, a bottom layer, presynaptic noise) simply make the two calls. */
void rnn_amd_set_opinion_sigmoid_mse(RnnAmdSet *set, const float *inputs, int ld_inputs, const float *targets, int ld,
                                     int n);
void rnn_amd_set_opinion_grouped_softmax(RnnAmdSet *set, const float *inputs, int ld_inputs, int n_groups,
                                         const int *group_offset, const int *group_size, const int *targets,
                                         const float *error_weight, u8 *trained);
/* rnnca's maybe_learn for every trainer at once (gstrnnca.c:693-740) as ONE call, the dense-input counterpart of
 * rnn_amd_set_char_step: rnn_bptt_clear_deltas, the opinion of `inputs`, the sigmoid-slope loss against `targets` (first n
 * outputs), rnn_bptt_calc_deltas for every stream, rnn_apply_learning(net, learning_style, momentum).  The same results
 * as rnn_bptt_clear_deltas + rnn_amd_set_opinion_sigmoid_mse + rnn_amd_set_calc_deltas(set, 1, NULL, NULL) +
 * rnn_apply_learning; knowing the update that follows, the weight-delta GEMM carries it out in its own epilogue where the
 * rule is the momentum rule (no optimiser launch: 14 us of 879 at 2048 / 512 / 10).  The caller advances the ring first
 * (rnn_amd_set_advance) if its trainer has a history, as the synthetic configs[4] driver does. */
void rnn_amd_set_dense_step_sigmoid_mse(RnnAmdSet *set, const float *inputs, int ld_inputs, const float *targets, int ld,
                                        int n, int learning_style, float momentum);
/* fill_frame's step after the opinion (gstrnnca.c:813-814): fast_sigmoid in place on the first
 * n outputs of every net of the set (training or forward-only clones); outputs (host, n_nets x
 * o_size, may be NULL) receives the answer rows, which synchronises. */
void rnn_amd_set_sigmoid_outputs(RnnAmdSet *set, int n, float *outputs);

/* Text on the device, for a host-free epoch loop (charmodel-predict.c:288-311).  The resident text belongs
 * to the set's ENGINE (the weight-owning net and its clones): every set opened on the same nets sees the
 * text loaded last, and a text step at a position outside it aborts with a message. */
void rnn_amd_set_load_text(RnnAmdSet *set, const u8 *text, int len);
/* One generation of rnn_char_epoch's multi-tap branch for text position i:
 * stream j reads text[off] and is scored against text[off + 1], with
 * off = i + j * ((len - 1) / n_nets), wrapped at len - 1;
 * advance, one-hot opinion, softmax error, calc_deltas(j ? 1 : 0), and then
 * rnn_apply_learning(nets[0], learning_style, momentum).  Nothing is copied to
 * the host. */
void rnn_amd_set_char_step(RnnAmdSet *set, int i, int learning_style, float momentum);
/* The same for ONE net on the fused single-net path (charmodel-predict.c:312-321): advance, opinion of
 * text[i], the loss against text[i + 1] on the device, rnn_bptt_calculate(net, batch_size); the caller
 * sets bptt->momentum first, as the reference's loop does.  No host round trip per symbol. */
void rnn_amd_set_char_step_fused(RnnAmdSet *set, int i, unsigned batch_size);

typedef struct RnnAmdStats {
  double error;   /* sum of target-class errors                */
  double entropy; /* sum of capped log2(1 - error)            */
  long correct;   /* count of winner == target                */
  long count;     /* stream-timesteps accumulated              */
  double bptt_depth_sum; /* sum of executed BPTT steps          */
  double hidden_zeros;   /* sum over stream-steps of the zero fraction of hiddens */
} RnnAmdStats;
/* Reads (and optionally clears) the device accumulators; synchronises. */
void rnn_amd_set_read_stats(RnnAmdSet *set, RnnAmdStats *stats, int clear);

/* ---- multi-GPU: one process per GPU, streams sharded, one all-reduce per generation ----
 * The reference sums the deltas of all streams in one process (recur-nn.c:724-739 over the
 * sharing of recur-nn-init.c:232-241); these calls distribute that sum.  rank 0 obtains an
 * id (128 bytes) and hands it to every rank by whatever means the launcher has (a file, a
 * pipe, MPI, torchrun's store); every rank then joins with the device it selected with
 * rnn_amd_use_device.  While a group is joined, rnn_amd_set_char_step (and
 * rnn_amd_set_dist_all_reduce_deltas for callers that drive calc_deltas / apply_learning
 * themselves) sum ih_delta||ho_delta over the ranks with ONE RCCL all-reduce on the
 * library's stream between the deltas and the update; rnn_amd_set_open shards the text
 * offsets as rnn_amd_set_shard(rank * n_nets, world * n_nets) unless told otherwise.
 * COLLECTIVE BEHAVIOUR of host draws: on a net whose training set is sharded over the group
 * (made by rnn_amd_new_training_set_shard with n_local != global_count, opened with rnn_amd_set_open
 * as ALL of its engine's training streams while the group is joined -- until that set is closed or
 * dropped --, or given a shard by rnn_amd_set_shard), rnn_weight_noise,
 * rnn_perforate_weights and the random damage of rnn_condition_net draw from RANK 0's generator
 * (a 32-byte broadcast) so that the replicas stay identical: EVERY rank must make that call on
 * its replica.  Nets that are not part of a sharded set (validation / confabulation clone
 * families of their own, side nets) are untouched by the group and draw locally.
 * All return 0 on success, -1 on failure (RCCL not loadable, bad arguments), with two exceptions in
 * rnn_amd_dist_init, which END THE PROCESS with _exit(86) after a message on stderr: a rank that has waited
 * RECUR_AMD_RCCL_INIT_TIMEOUT seconds (default 180) for the others inside ncclCommInitRank -- the helper thread
 * sitting in RCCL cannot be cancelled, and a process that has touched the GPU must not re-exec or retry --, and a
 * communicator whose rank / size are not the ones asked for.  A launcher sees exit status 86 of that rank. */
#define RNN_AMD_DIST_ID_BYTES 128
int rnn_amd_dist_get_id(void *id);
int rnn_amd_dist_init(int rank, int world, const void *id);
void rnn_amd_dist_finalize(void);
int rnn_amd_dist_rank(void);  /* 0 when no group is joined */
int rnn_amd_dist_world(void); /* 1 when no group is joined */
/* ---- the exchange step WITHOUT a collective library: reduce-scatter -> sharded optimiser -> all-gather as
 * kernel-issued peer traffic (SURVEY.md section 8e's alternative; the sum it distributes is recur-nn.c:734-739,
 * the update it shards recur-nn.c:601-678).  Every rank owns 1 / world of each weight array; ONE kernel per
 * generation adds the ranks' local delta sums for its range in rank order through peer pointers (replicas stay
 * bit-identical), updates its own weights and momentum there -- the optimiser state of a range lives on its owner
 * only -- and stores the new weights into every rank's weight array; two arrival barriers (kernels on the
 * library's stream that count in host memory all ranks have mapped) keep the ranks' generations apart.  No RCCL
 * launch, no separate all-reduce pass.  Up to 8 ranks, no bottom layer, ranks on GPUs with peer access (or on one
 * GPU: the tests' emulation).  OPT-IN: the RCCL all-reduce above stays the default until a multi-GPU node has
 * measured both.
 *   1. each rank: rnn_amd_set_exchange_export(set, blob)  -- 256 bytes naming its delta and weight arrays (IPC
 *      handles; same-process peers are recognised and take the plain pointers);
 *   2. the launcher hands every blob to every rank (a file, shared memory, a pipe) and provides `counters`: 64 bytes
 *      of zeroed host memory that ALL ranks have mapped (MAP_SHARED memory made before fork, or a file in /dev/shm);
 *      they need no zeroing between sessions (round 6): the barriers count on from where words 0 .. 7 stand at the
 *      join, words 8 .. 15 carry the join's own rendezvous, named by a token made of the session's blobs;
 *   3. each rank: rnn_amd_set_exchange_join(set, rank, world, blobs, counters, 0) -- COLLECTIVE: it returns when every
 *      rank of the session has called it (RECUR_AMD_XCHG_JOIN_TIMEOUT seconds at most, default 120: -1 then); from then
 *      on rnn_amd_set_char_step runs deltas -> barrier -> sharded update -> barrier.
 * lockstep != 0: the ranks are sets of ONE process driven by one thread (counters may be NULL): no barriers are
 * launched, the caller gives the order -- rnn_amd_set_char_step_deltas on every rank, then
 * rnn_amd_set_apply_exchange on every rank.  After a step ih_delta || ho_delta hold the sum over the ranks in the
 * rank's OWN range only, and momentum / aux arrays are current in the own range only (rnn_amd_set_exchange_range).
 * With ONE rank rnn_amd_set_char_step is the plain generation (nothing to exchange); the arrays reached through peer
 * pointers are ordinary device allocations whose visibility between ranks rests on kernel boundaries: on GPUs of
 * different devices that has not run yet.
 * Returns 0, or -1 with a message (a peer's arrays cannot be opened, bad arguments). */
#define RNN_AMD_EXCHANGE_BLOB_BYTES 256
/* A 64-bit checksum of this rank's replica: sum over the 32-bit words of ih_weights || ho_weights (with_momentum: ||
 * ih_momentum || ho_momentum) of word_i * (2 i + 1) mod 2^64.  through_kernel != 0: formed by a kernel on the library's
 * stream, i.e. through the caches the path's own kernels read the arrays through; 0: over a device-to-host copy.  The
 * replicas of a sharded run are bit-identical (RCCL: every rank adds the same sum; kernel-issued: every range has one
 * owner), so a launcher compares the ranks' values after a run -- and the two readings of ONE rank, which differ only
 * if a peer's stores did not reach what this rank's kernels see (bench.py does both and exits non-zero).  In the
 * kernel-issued exchange momentum lives in the owner's range only: compare weights there.  Synchronises. */
uint64_t rnn_amd_set_replica_checksum(RnnAmdSet *set, int with_momentum, int through_kernel);
void rnn_amd_set_exchange_export(RnnAmdSet *set, void *blob);
int rnn_amd_set_exchange_join(RnnAmdSet *set, int rank, int world, const void *blobs, void *counters, int lockstep);
void rnn_amd_set_exchange_leave(RnnAmdSet *set);
/* the update of rnn_apply_learning through the exchange (the second half of rnn_amd_set_char_step) */
void rnn_amd_set_apply_exchange(RnnAmdSet *set, int learning_style, float momentum);
/* this rank's range of the recurrent (which = 0) or top-layer (1) arrays, in floats */
void rnn_amd_set_exchange_range(const RnnAmdSet *set, int which, size_t *first, size_t *count);

/* in-place sum over the ranks of a device buffer of floats, on the library's stream */
void rnn_amd_dist_all_reduce(void *device_buffer, size_t n_floats);
/* max over the ranks of a host value / a barrier (the timing bracket of a benchmark) */
double rnn_amd_dist_max(double x);
void rnn_amd_dist_barrier(void);
/* rnn_new_training_set (recur-nn-init.c:221-243) for one shard of a set of global_count
 * streams: the prototype's generator hands out the clone seeds of ALL global streams in
 * order (recur-nn-init.c:300-305), this process keeps streams [global_first, global_first
 * + n_local): nets[0] is the prototype (which, for global_first > 0, takes the generator of
 * global stream global_first), nets[j] the clone of global stream global_first + j.  With
 * global_first 0 and global_count == n_local it is rnn_new_training_set. */
RecurNN **rnn_amd_new_training_set_shard(RecurNN *prototype, int n_local, int global_first,
                                         int global_count);

/* Multi-GPU: when set, ih_delta||ho_delta of the set live in the caller's
 * device buffer (ih_size + ho_size floats, ih first), so that the caller can
 * all-reduce it (RCCL) between calc_deltas and apply_learning.  Pass NULL to
 * go back to library-owned storage. */
void rnn_amd_set_external_delta(RnnAmdSet *set, void *device_buffer);
/* When the streams of one logical training set are sharded over several
 * processes (one per GPU), this set holds global streams
 * [global_first, global_first + n_nets) of global_count: rnn_amd_set_char_step
 * then spaces the text offsets over global_count streams
 * (charmodel-predict.c:273, 295).  Default: global_first 0, global_count n_nets. */
void rnn_amd_set_shard(RnnAmdSet *set, int global_first, int global_count);
/* Split rnn_amd_set_char_step for that use: everything up to and including
 * calc_deltas, then the update. */
void rnn_amd_set_char_step_deltas(RnnAmdSet *set, int i);
/* sums the set's ih_delta||ho_delta over the joined ranks (no-op without a group) */
void rnn_amd_set_dist_all_reduce_deltas(RnnAmdSet *set);
/* One net run over an encoded text without leaving the device: a one-hot opinion
 * (charmodel-helpers.h:16-33) of text[i] for every i < len - 1, and from i = skip on
 * the log2 of the softmax probability of text[i + 1] (capped at -100 like
 * capped_log2f).  Returns the sum of the logs: get_cross_entropy's loop
 * (charmodel-predict.c:62-76) without its final division; with skip >= len - 1 it is
 * rnn_char_prime's loop (407-416).  Works for nets with or without bptt. */
double rnn_amd_run_text(RecurNN *net, const u8 *text, int len, int skip);
/* The same for a net whose output row is output_size / alphabet_len heads: sums[c] (room for
 * that many doubles) receives head c's sum of capped log2 softmax(head c)[text[i + 1]] over
 * i in [skip, len - 1): rnn_char_multi_cross_entropy's loops (charmodel-multi-predict.c:388-403)
 * without the final negation and division. */
void rnn_amd_run_text_heads(RecurNN *net, const u8 *text, int len, int skip, int alphabet_len,
                            double *sums);
/* Block until all queued device work of the library has finished. */
void rnn_amd_synchronize(void);

/* Timing hook for bench.py: HIP-event time (ms) accumulated over the dominant
 * kernel class since the last reset, and the number of launches.
 * which: 0 = BPTT chain GEMM, 1 = delta GEMM, 2 = forward GEMM, 3 = optimiser, 4 = other launches,
 * 5 = the exchange step between ranks (the RCCL all-reduce of a generation, or the kernel-issued exchange's two
 * arrivals and sharded update: what a rank waits and works for between its deltas and its next forward pass). */
void rnn_amd_kernel_time_enable(int enable);
double rnn_amd_kernel_time_ms(int which, long *launches, int reset);

#ifdef __cplusplus
}
#endif
#endif
