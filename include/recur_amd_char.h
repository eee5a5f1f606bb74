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

#ifdef __cplusplus
}
#endif
#endif
