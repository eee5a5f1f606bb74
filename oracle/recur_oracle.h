/* recur_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference RNN core (recur-nn.c,
 * recur-nn-init.c, recur-rng.h, badmaths.h of douglasbagnall/recur) in the
 * batched "set of streams" data layout that librecur_amd.so uses on the device.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or call this; the product library never does.
 *
 * Pinned by tests/test_oracle_golden.py against tests/golden/ (vectors produced
 * by the real reference compiled from /root/reference, see
 * tests/golden/make_golden.py) and, when oracle/_ref/librecur_ref.so is
 * present, against that library live (tests/test_oracle_vs_ref.py).
 *
 * Parity unpinned for these caller-side restatements only: orc_multi_softmax_error,
 * orc_multitext_train, orc_multi_cross_entropy (charmodel-multi-predict.c), orc_grouped_softmax_error
 * (gstclassify.c) and orc_sigmoid_mse_error (gstrnnca.c) restate functions of files that cannot
 * be compiled here (generated path.h, GStreamer); everything they call (softmax, fast_sigmoid,
 * generator, calc_deltas with and without ranges, apply_learning) is pinned.
 */
#ifndef RECUR_ORACLE_H
#define RECUR_ORACLE_H 1
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcRng {
  uint64_t a, b, c, d;
} OrcRng;

/* activations / learning methods use the reference's numeric values */
enum { ORC_RELU = 1, ORC_RESQRT = 2, ORC_RECLIP20 = 5 };
enum {
  ORC_WEIGHTED = 0,
  ORC_NESTEROV,
  ORC_SIMPLIFIED_NESTEROV,
  ORC_CLASSICAL,
  ORC_ADAGRAD,
  ORC_ADADELTA,
  ORC_RPROP
};
enum { ORC_DIST_UNIFORM = 1, ORC_DIST_GAUSSIAN, ORC_DIST_LOG_NORMAL, ORC_DIST_SEMICIRCLE };

#define ORC_FLAG_ADAPTIVE_MIN_ERROR 64u

/* A training set: S streams sharing one set of weights.  Array layouts:
 *   ih_w     [I][H]     ho_w   [H][O]          (reference layout)
 *   hist     [D][S][I]  slot-major ring; stream s uses slot index[s]
 *   hidden   [S][H]     output [S][O]   o_error [S][O]
 *   err_a/b  [S][I]     the h_error / i_error ping-pong pair of each stream
 */
typedef struct OrcSet {
  int input_size, hidden_size, output_size;
  int I, H, O; /* padded sizes */
  int S, D;
  int activation;
  uint32_t flags;
  float *ih_w, *ho_w;
  float *ih_m, *ho_m;     /* momentum / adagrad accumulators */
  float *ih_aux, *ho_aux; /* adadelta / rprop second array */
  float *ih_delta, *ho_delta, *ih_delta_tmp;
  float *hist, *hidden, *output, *o_error, *err_a, *err_b;
  int *index;              /* [S] ring position                     */
  float *learn_rate;       /* [S] each clone's copy (quirk 4)       */
  float *min_error_factor; /* [S]                                   */
  float *ih_scale;         /* [S] out                               */
  float *top_error_raw, *top_error_scaled, *bptt_error; /* [S] out   */
  int *bptt_depth;         /* [S] out: executed steps               */
  uint32_t *generation;    /* [S]                                   */
  OrcRng *rng;             /* [S]                                   */
  float ho_scale, momentum_weight;
  float presynaptic_noise;
  /* running loss statistics of orc_set_char_step */
  double stat_error, stat_entropy;
  long stat_correct, stat_count;
  double stat_depth, stat_zeros;
  /* optional bottom layer (recur-nn.h:211-227; orc_set_add_bottom): ONE layer struct is
   * shared by the prototype and every clone (recur-nn-init.c:345-346), so there is one
   * input buffer, one output buffer and one never-cleared error accumulator */
  int b_in, bI, bO; /* real inputs; padded input (+ bias) and output sizes; 0 = none */
  float *b_w, *b_m, *b_aux, *b_delta; /* [bI][bO] */
  float *b_inputs, *b_outputs, *b_o_error;
  float b_learn_rate_scale;
  /* when this set is one shard of a larger one: its streams are global streams
   * [global_first, global_first + S) of global_count (0 = not sharded) */
  int global_first, global_count;
} OrcSet;

/* ---- PRNG (recur-rng.h) ---- */
uint64_t orc_rand64(OrcRng *x);
void orc_init_rand64(OrcRng *x, uint64_t seed);
double orc_rand_double(OrcRng *x);
int orc_rand_small_int(OrcRng *x, int cap);
float orc_cheap_gaussian_noise(OrcRng *x);

/* ---- small maths (badmaths.h, recur-nn-helpers.h, charmodel-helpers.h) ---- */
float orc_fast_expf(float x);
void orc_softmax(float *dest, const float *src, int len);
int orc_softmax_best_guess(float *error, const float *src, int len);
float orc_soft_clip(float sum, float halfmax);
float orc_capped_log2f(float x);
float orc_momentum_soft_start(float generation, float max_momentum, float x);

/* ---- sizes (recur-nn-init.c:87-91) ---- */
void orc_padded_sizes(int input_size, int hidden_size, int output_size, int *I,
                      int *H, int *O);

/* ---- set life cycle ---- */
OrcSet *orc_set_new(int input_size, int hidden_size, int output_size, int S,
                    int D, int activation, uint32_t flags, float learn_rate,
                    uint64_t seed);
void orc_set_free(OrcSet *set);
/* rnn_new_with_bottom_layer (recur-nn-init.c:158-219): call right after orc_set_new; the
 * set's input_size is then the bottom layer's output size and orc_opinion's `inputs`
 * are the n_inputs values that feed the bottom layer */
void orc_set_add_bottom(OrcSet *set, int n_inputs);
/* seeds streams 1..S-1 from stream 0's rng as rnn_new_training_set does */
void orc_set_seed_clones(OrcSet *set);
/* flat weight init (recur-nn-init.c:495-573) using stream 0's rng */
void orc_set_init_flat(OrcSet *set, float variance, int shape, double perforation);

/* ---- per-stream steps, named after the reference functions ---- */
void orc_advance(OrcSet *set, int s);                                 /* recur-nn.c:696-704 */
float *orc_opinion(OrcSet *set, int s, const float *inputs, float noise); /* recur-nn.c:83-154 */
float *orc_one_hot_opinion(OrcSet *set, int s, int hot, float noise); /* charmodel-helpers.h:16-33 */
float orc_net_error_bptt(OrcSet *set, int s, int c, int next, int *correct); /* charmodel-predict.c:18-27 */
/* gstclassify.c:2070-2119 after the opinion; targets[i] < 0: group i is not trained */
int orc_grouped_softmax_error(OrcSet *set, int s, int n_groups, const int *group_offset,
                              const int *group_size, const int *targets, const float *weight,
                              int *wins, float *wrongness);
/* charmodel-multi-predict.c:17-58; ranges_out: room for n_classes + 1 pairs */
float orc_multi_softmax_error(OrcSet *set, int s, int c, int next, int target_class,
                              int alphabet_len, float leakage, int *ranges_out);
/* ranges: pairs (start, len) ending with start < 0, or NULL */
void orc_calc_deltas(OrcSet *set, int s, int accumulate, const int *ranges); /* recur-nn.c:707-772 */
void orc_clear_deltas(OrcSet *set);                                    /* recur-nn.c:681-693 */
void orc_apply_learning(OrcSet *set, int method, float momentum);      /* recur-nn.c:601-678 */
void orc_condition(OrcSet *set, uint32_t flags);                       /* recur-nn.c:782-855 */
void orc_bptt_calculate(OrcSet *set, int s, unsigned batch_size, float momentum); /* recur-nn.c:999-1019 */
double orc_cross_entropy(OrcSet *set, int s, const uint8_t *text, int len, int skip); /* charmodel-predict.c:62-80 */
void orc_multi_cross_entropy(OrcSet *set, int s, const uint8_t *text, int len, int alphabet_len,
                             double *entropy, int ignore_start); /* charmodel-multi-predict.c:383-408 */
float orc_fast_sigmoid(float x);                                        /* badmaths.h:31-36 */
void orc_sigmoid_mse_error(OrcSet *set, int s, const float *target, int n); /* gstrnnca.c:701-714 */
/* charmodel-multi-predict.c:234-281 (text_train) for stream s */
void orc_multitext_train(OrcSet *set, int s, const uint8_t *text, int len, int alphabet_len,
                         int target_class, float leakage, int learning_style, float bptt_momentum,
                         int batch_size, float *error_sum, float *entropy_sum);

/* ---- whole-set generation of rnn_char_epoch's multi-tap branch
 *      (charmodel-predict.c:288-311) ---- */
void orc_set_char_step(OrcSet *set, const uint8_t *text, int len, int i,
                       int method, float momentum);
void orc_set_char_step_deltas(OrcSet *set, const uint8_t *text, int len, int i);

#ifdef __cplusplus
}
#endif
#endif
