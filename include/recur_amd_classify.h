/* recur_amd_classify.h -- the caller-side pieces of gstclassify's training loop that a
 * GPU-resident classifier needs (additive API; gnu11 C).
 *
 * The reference keeps these as static functions inside its GStreamer plugin (gstclassify.c), where
 * they cannot be linked against.  What they define IS contract, though: the metadata string a
 * classifier net is saved with and checked against on load (gstclassify.c:841-929, 1130-1205), the
 * net file name that embeds that string's hash (689-707), the class-group string ("01,abc": groups
 * of one-letter classes, 709-748), and the balanced-training rule that decides which labelled
 * windows are used (2190-2215 with the draw at 2098-2101).  This header gives them names, so that a
 * host program that feeds features to librecur_amd -- the plugin, ported, or the synthetic driver of
 * tools/classify_train_amd.c -- reads and writes nets that the reference's tools accept and trains
 * on the same sample of windows.
 */
#ifndef RECUR_AMD_CLASSIFY_H
#define RECUR_AMD_CLASSIFY_H
#include "recur_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* gstclassify.h:57-72 (field order is the order of the metadata lines; new items only at the end) */
typedef struct RnnAmdClassifyMetadata {
  const char *classes;
  float min_freq;
  float max_freq;
  float knee_freq;
  int mfccs;
  int window_size;
  const char *basename;
  int delta_features;
  float focus_freq;
  float lag;
  int intensity_feature;
  float confirmation_lag;
  const char *features_offset; /* colon-separated floats, or NULL */
  const char *features_scale;
} RnnAmdClassifyMetadata;

/* the fourteen "name value" lines (gstclassify.c:841-884); the caller frees */
char *rnn_amd_classify_construct_metadata(const RnnAmdClassifyMetadata *m);
/* Reads a metadata string back (gstclassify.c:885-921).  Returns the number of the eleven leading items
 * that were NOT found (0: all there; -1 when the string is NULL).  `classes` and `basename` are
 * allocated (rnn_amd_classify_free_metadata_items).  As in the reference, confirmation-lag and the two
 * features strings are never read back: its scanf template cannot match the line after
 * intensity-feature, so those fields keep what the caller put there. */
int rnn_amd_classify_load_metadata(const char *metadata, RnnAmdClassifyMetadata *m);
void rnn_amd_classify_free_metadata_items(RnnAmdClassifyMetadata *m);
/* <basename>-<hash of the metadata>-i<features>[-b<bottom>]-h<hidden>-o<outputs>-<rate>Hz-w<window>.net
 * (gstclassify.c:689-707); the caller frees */
char *rnn_amd_classify_net_filename(const char *basename, const char *metadata, int n_features,
                                    int bottom_layer, int hidden_size, int top_layer_size,
                                    int sample_rate, int window_size);
/* The class-group string: groups of one-letter classes separated by commas, "01" or "01,abc"
 * (gstclassify.c:709-748).  sizes[g] = letters of group g; offsets[g] = where the group starts IN THE
 * STRING, commas counted -- that is what the reference stores as the group's first output index.
 * Returns the number of groups (at most max_groups are written); *n_outputs (may be NULL) = the count of
 * letters = the net's output size (count_class_group_members); *string_len (may be NULL) = what the
 * reference's parse returns and compares with the net's output size on load (equal for one group). */
int rnn_amd_classify_parse_classes(const char *classes, int *offsets, int *sizes, int max_groups,
                                   int *n_outputs, int *string_len);
/* Whether a net (as loaded) is the one this configuration expects (gstclassify.c:1141-1150): layer
 * sizes, and the metadata string unless force_load.  0 = usable, -1 = not (why goes to stderr). */
int rnn_amd_classify_check_net(const RecurNN *net, const char *metadata, int hidden_size,
                               int bottom_layer, int top_layer_size, int force_load);

/* ---- balanced training (gstclassify.c:2190-2215, 2096-2104) ----
 * Classes that have been seen often are trained on less often: per generation the probability of using a
 * labelled window of class c is (1 - seen[c] / (seen_total + 1)) ^ bias, and the decision is a draw from
 * the PROTOTYPE net's generator, one per (channel, group) with a valid target, in channel order. */
typedef struct RnnAmdBalancedTraining {
  int n_outputs;
  float bias;
  u32 *seen, *used;   /* per output: labelled windows met / trained on */
  float *train_p;     /* per output: this generation's probability      */
} RnnAmdBalancedTraining;
RnnAmdBalancedTraining *rnn_amd_balanced_new(int n_outputs, float bias);
void rnn_amd_balanced_free(RnnAmdBalancedTraining *b);
/* the probabilities of the generation that starts now, from the counts so far */
void rnn_amd_balanced_begin(RnnAmdBalancedTraining *b);

/* One generation of gstclassify's maybe_learn (gstclassify.c:2196-2257) for a training set whose streams
 * are the audio channels, features already extracted: rnn_bptt_clear_deltas; for every channel the
 * opinion (with the net's noise), the class-group loss against targets[channel * n_groups + group]
 * (< 0 or >= the group's size: not labelled), through the balanced-training draw when `balance` is
 * given, the deltas of the channels that trained anything, rnn_bptt_advance AFTER them; then
 * rnn_apply_learning with the soft-started momentum -- if the summed error is non-zero --, and
 * rnn_condition_net.  All channels run as batched device calls.  exact_gate != 0 reads the summed
 * error back (one synchronisation per generation) to decide the update exactly as the reference's
 * `if (err_sum)`; 0 updates whenever any group was trained (the two differ only when every trained
 * group's error is exactly zero).  Returns the number of groups trained. */
int rnn_amd_classify_generation(RnnAmdSet *set, const float *features, int ld_features, int n_groups,
                                const int *group_offset, const int *group_size, const int *targets,
                                const float *error_weight, RnnAmdBalancedTraining *balance,
                                int learning_style, float momentum_soft_start, int exact_gate);

#ifdef __cplusplus
}
#endif
#endif
