/* recur_amd_char.h -- the integer ("bit-exact tables") side of Recur's
 * character models: alphabet discovery, the char -> symbol table and text
 * encoding.  Host-only C; same names, struct layout, argument meaning and error
 * behaviour as the reference's charmodel.h (lines cited per declaration), so the
 * reference's callers (text-predict.c:674-736, py-recur-text.c) can use them
 * unchanged.  SURVEY.md section 8(a), row "Alphabet / LUT / encode". */
#ifndef RECUR_AMD_CHAR_H
#define RECUR_AMD_CHAR_H 1
#include <stdbool.h>
#include "recur_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* charmodel.h:9-13 */
enum {
  RNN_CHAR_FLAG_CASE_INSENSITIVE = 1,
  RNN_CHAR_FLAG_UTF8 = 2,
  RNN_CHAR_FLAG_COLLAPSE_SPACE = 4
};

/* charmodel.h:47-53: points[] are the code points of the symbols in symbol
 * order; collapsed_points[] all map to symbol 0. */
typedef struct _RnnCharAlphabet {
  int *points;
  int *collapsed_points;
  int len;
  int collapsed_len;
  u32 flags;
} RnnCharAlphabet;

RnnCharAlphabet *rnn_char_new_alphabet(void);                 /* charmodel.h:219 */
void rnn_char_free_alphabet(RnnCharAlphabet *a);              /* charmodel.h:223 */
void rnn_char_alphabet_set_flags(RnnCharAlphabet *a, bool case_insensitive, bool utf8,
                                 bool collapse_space);        /* charmodel.h:231-232 */
/* 0 on success, -1 on failure (charmodel.h:166-170; charmodel-init.c:30-157) */
int rnn_char_find_alphabet_s(const char *text, int len, RnnCharAlphabet *alphabet,
                             double threshold, double digit_adjust, double alpha_adjust);
int rnn_char_find_alphabet_f(const char *filename, RnnCharAlphabet *alphabet, double threshold,
                             double digit_adjust, double alpha_adjust);
/* 257 (bytes) or 0x200001 (utf-8) entries, caller frees (charmodel.h:229; -init.c:238-265) */
int *rnn_char_new_char_lut(const RnnCharAlphabet *alphabet);
/* new memory, caller frees (charmodel.h:172-176; charmodel-init.c:270-349) */
u8 *rnn_char_alloc_encoded_text(RnnCharAlphabet *alphabet, const char *text, int byte_len,
                                int *encoded_len, int *char_to_net, bool verbose);
u8 *rnn_char_load_new_encoded_text(const char *filename, RnnCharAlphabet *alphabet,
                                   int *encoded_len, int quietness);
int rnn_char_alloc_file_contents(const char *filename, char **contents, int *len); /* charmodel.h:159 */


/* ------------------------------------------------------------------------
 * The prediction layer (charmodel-predict.c): epoch loop, validation
 * entropy, learn-rate schedule, confabulation.  Struct layouts are the
 * reference's (charmodel.h:15-73), because its callers fill them in directly
 * (text-predict.c:545-600).
 * ------------------------------------------------------------------------ */

/* pgm_dump.h:227-237 defines this; callers that use the image fields include that header
 * themselves (C11 lets the typedef repeat) */
typedef struct _TemporalPPM TemporalPPM;

typedef struct _RnnCharSchedule RnnCharSchedule;
typedef struct _RnnCharModel RnnCharModel;

typedef struct _RnnCharImageSettings { /* charmodel.h:18-24 */
  char *basename;
  bool temporal_pgm_dump;
  TemporalPPM *input_ppm;
  TemporalPPM *error_ppm;
  char *periodic_pgm_dump_string;
} RnnCharImageSettings;

struct _RnnCharSchedule { /* charmodel.h:26-34 */
  float *recent;
  int recent_len;
  int timeout;
  float learn_rate_mul;
  float learn_rate_min;
  int adjust_noise;
  void (*eval)(RnnCharModel *m, float score, int verbose);
};

typedef struct _RnnCharVentropy { /* charmodel.h:36-45 */
  RecurNN *net;
  int counter;
  float *history;
  const u8 *text;
  int len;
  int lap;
  int lapsize;
  float entropy;
} RnnCharVentropy;

struct _RnnCharModel { /* charmodel.h:56-73 */
  RecurNN *net;
  RecurNN **training_nets;
  int n_training_nets;
  uint batch_size;
  char *filename;
  float momentum;
  float momentum_soft_start;
  int learning_style;
  float periodic_weight_noise;
  uint report_interval;
  bool save_net;
  bool use_multi_tap_path;
  RnnCharAlphabet *alphabet;
  RnnCharSchedule schedule;
  RnnCharImageSettings images;
};

typedef struct RnnCharMetadata { /* charmodel.h:75-81 */
  char *alphabet; /* utf-8 or byte string */
  char *collapse_chars;
  bool utf8;
  bool case_insensitive;
  bool collapse_space;
} RnnCharMetadata;

/* charmodel.h:185-186; charmodel-predict.c:120-135 */
void rnn_char_init_schedule(RnnCharSchedule *s, int recent_len, float learn_rate_min,
                            float learn_rate_mul, int adjust_noise);
/* charmodel.h:188, 194-197; charmodel-predict.c:208-258 */
float rnn_char_calc_ventropy(RnnCharModel *model, RnnCharVentropy *v, int lap);
void rnn_char_delete_ventropy(RnnCharVentropy *v);
void rnn_char_init_ventropy(RnnCharVentropy *v, RecurNN *net, const u8 *text, const int len,
                            const int lap);
/* charmodel.h:190-192; charmodel-predict.c:137-181.  Returns the bytes written. */
int rnn_char_confabulate(RecurNN *net, char *dest, int char_len, int byte_len, RnnCharAlphabet *a,
                         float bias, int *prev_char, int start_point, int stop_point);
/* charmodel.h:199-204; charmodel-predict.c:261-405.  One pass over the text from
 * `start`; returns 1 when `stop` generations were reached, else 0.  The multi-tap
 * branch runs as batched device generations (rnn_amd_set_char_step) whenever
 * model->training_nets is one rnn_new_training_set. */
int rnn_char_epoch(RnnCharModel *model, RecurNN *confab_net, RnnCharVentropy *v, const u8 *text,
                   const int len, const int start, const int stop, float confab_bias,
                   int confab_size, int confab_line_end, int quietness,
                   uint diagonal_only_section, uint diagonal_only_friends);
/* charmodel.h:234-239; charmodel-predict.c:407-431 */
int rnn_char_prime(RecurNN *net, RnnCharAlphabet *alphabet, const u8 *text, const int len);
double rnn_char_cross_entropy(RecurNN *net, RnnCharAlphabet *alphabet, const u8 *text,
                              const int len, const int ignore_first, const u8 *prefix_text,
                              const int prefix_len);

/* ---- the multi-head text trainer (charmodel.h:132-152, 242-265;
 *      charmodel-multi-predict.c): py-recur-text.c's Net.train / Net.test backend ---- */
typedef struct _RnnCharProgressReport { /* charmodel.h:132-138 */
  float training_entropy;
  float training_error;
  float training_accuracy;
  float per_second;
} RnnCharProgressReport;

typedef struct RnnCharMultiConfab { /* charmodel.h:140-152 */
  RecurNN **nets;
  int *last_char;
  int caps_marker;
  char **strings;
  uint n_classes;
  uint char_len;
  uint byte_len;
  float bias;
  uint period;
  RnnCharAlphabet *alphabet;
} RnnCharMultiConfab;

/* One pass over one class's text on ONE net whose outputs are output_size / alphabet_len
 * heads: per symbol rnn_bptt_advance, the multi-head loss (own head always, the others
 * with probability `leakage`), rnn_bptt_calc_deltas over the resulting error ranges,
 * accumulating for batch_size symbols between rnn_apply_learning calls (which use the
 * net's own bptt->momentum: the `momentum` argument is unused, as in the reference).
 * The loop's work stays on the device. */
void rnn_char_multitext_train(RecurNN *net, u8 *text, int len, int alphabet_len, int target_class,
                              float leakage, RnnCharProgressReport *report,
                              RnnCharMultiConfab *confab, int learning_style, float momentum,
                              int batch_size, TemporalPPM *input_ppm, TemporalPPM *error_ppm,
                              const char *periodic_pgm_string, int periodic_pgm_period);
void rnn_char_multitext_spin(RecurNN *net, u8 *text, int len, TemporalPPM *input_ppm,
                             TemporalPPM *error_ppm, const char *periodic_pgm_string,
                             int periodic_pgm_period);
/* entropy: n_classes doubles, the caller's running values (normally zeros) */
void rnn_char_multi_cross_entropy(RecurNN *net, const u8 *text, int len, int alphabet_len,
                                  double *entropy, int ignore_start);
RnnCharMultiConfab *rnn_char_new_multi_confab(RecurNN *net, RnnCharAlphabet *alphabet, int n_classes,
                                              int target_len, uint confab_period, int caps_marker);
void rnn_char_free_multi_confab(RnnCharMultiConfab *mc);

/* ---- metadata and names (charmodel.h:206-227; charmodel-init.c:440-800) ---- */
char *rnn_char_uncollapse_text(RnnCharAlphabet *alphabet, const u8 *orig, int len, int *dest_len);
void rnn_char_dump_collapsed_text(const u8 *text, int len, const char *name, const char *alphabet);
char *rnn_char_construct_metadata(const struct RnnCharMetadata *m);
int rnn_char_load_metadata(const char *metadata, struct RnnCharMetadata *m);
void rnn_char_free_metadata_items(struct RnnCharMetadata *m);
char *rnn_char_construct_net_filename(struct RnnCharMetadata *m, const char *basename,
                                      int input_size, int bottom_size, int hidden_size,
                                      int output_size);
int rnn_char_check_metadata(RecurNN *net, struct RnnCharMetadata *m, bool trust_file_metadata,
                            bool force_metadata);
void rnn_char_copy_metadata_items(struct RnnCharMetadata *src, struct RnnCharMetadata *dest);
void rnn_char_dump_alphabet(RnnCharAlphabet *alphabet);
int rnn_char_get_codepoint(RnnCharAlphabet *a, const char *s);
RnnCharAlphabet *rnn_char_new_alphabet_from_net(RecurNN *net);

#ifdef __cplusplus
}
#endif
#endif
