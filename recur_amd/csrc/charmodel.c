/* charmodel.c -- alphabet discovery, char -> symbol table, text encoding
 * (host only, integer work; see include/recur_amd_char.h).  Behaviour follows
 * charmodel-init.c:15-349 of the reference; pinned by the expected alphabets of
 * the reference's own test table (tests/golden/alphabet_cases.json). */
#include "rnn_host.h"
#include "recur_amd_char.h"
#include <ctype.h>

/* decode one UTF-8 sequence; returns the code point (0 at NUL) or -1 on a
 * malformed sequence, and advances *s (utf8.h of the reference offers the same
 * service as read_utf8_char) */
static int next_code_point(const char **s) {
  const unsigned char *p = (const unsigned char *)*s;
  int c = p[0], extra;
  if (c < 0x80) {
    *s += 1;
    return c;
  }
  if ((c & 0xe0) == 0xc0) {
    c &= 0x1f;
    extra = 1;
  } else if ((c & 0xf0) == 0xe0) {
    c &= 0x0f;
    extra = 2;
  } else if ((c & 0xf8) == 0xf0) {
    c &= 0x07;
    extra = 3;
  } else {
    return -1;
  }
  for (int i = 1; i <= extra; i++) {
    if ((p[i] & 0xc0) != 0x80) {
      return -1;
    }
    c = (c << 6) | (p[i] & 0x3f);
  }
  *s += extra + 1;
  return c;
}

RnnCharAlphabet *rnn_char_new_alphabet(void) {
  RnnCharAlphabet *a = calloc(1, sizeof(*a));
  a->points = calloc(257, sizeof(int));
  a->collapsed_points = calloc(257, sizeof(int));
  return a;
}

void rnn_char_free_alphabet(RnnCharAlphabet *a) {
  if (a) {
    free(a->points);
    free(a->collapsed_points);
    free(a);
  }
}

void rnn_char_alphabet_set_flags(RnnCharAlphabet *a, bool case_insensitive, bool utf8,
                                 bool collapse_space) {
  a->flags = (case_insensitive ? RNN_CHAR_FLAG_CASE_INSENSITIVE : 0) |
             (utf8 ? RNN_CHAR_FLAG_UTF8 : 0) | (collapse_space ? RNN_CHAR_FLAG_COLLAPSE_SPACE : 0);
}

/* charmodel-init.c:15-28: digits and letters can be weighted before the
 * threshold test (the count used to pick the collapse target stays raw) */
static int weighted_count(int c, int count, double digit_adjust, double alpha_adjust) {
  if (count && c < 256) {
    if (isdigit(c)) {
      count = count * digit_adjust + 0.5;
    } else if (isalpha(c)) {
      count = count * alpha_adjust + 0.5;
    }
  }
  return count;
}

int rnn_char_find_alphabet_s(const char *text, int len, RnnCharAlphabet *alphabet,
                             double threshold, double digit_adjust, double alpha_adjust) {
  const int ignore_case = alphabet->flags & RNN_CHAR_FLAG_CASE_INSENSITIVE;
  const int collapse_space = alphabet->flags & RNN_CHAR_FLAG_COLLAPSE_SPACE;
  const int utf8 = alphabet->flags & RNN_CHAR_FLAG_UTF8;
  const int n_chars = utf8 ? 0x200000 : 256;
  int *counts = calloc(n_chars + 1, sizeof(int));
  int n = 0, n_alpha = 0, n_collapsed = 0, prev = 0;
  const char *s = text;
  /* histogram (charmodel-init.c:45-96) */
  for (int i = 0; i < len && s < text + len; i++) {
    int c;
    if (utf8) {
      c = next_code_point(&s);
      if (c < 0) {
        fprintf(stderr, "Unicode Error at %d!\n", i);
        break;
      }
      if (c == 0) {
        break;
      }
    } else {
      c = ((const u8 *)text)[i];
    }
    if (c >= n_chars) {
      fprintf(stderr, "got char %d, but there are only %d slots\n", c, n_chars);
      goto fail;
    }
    if (collapse_space && isspace(c)) {
      c = 32;
      if (c == prev) {
        continue; /* runs of white space count once */
      }
    }
    if (ignore_case && c < 0x80 && isupper(c)) {
      c = tolower(c);
    }
    n++;
    counts[c]++;
    prev = c;
  }
  if (n == 0) {
    goto fail;
  }
  /* the most frequent of the too-rare characters represents them all and goes
   * first in the alphabet (charmodel-init.c:101-120) */
  int min_count = RAMD_MAX(ceil(threshold * n), 1);
  int best_count = 0, best_point = 0;
  for (int c = 0; c < n_chars; c++) {
    int count = counts[c];
    if (count && weighted_count(c, count, digit_adjust, alpha_adjust) < min_count &&
        count > best_count) {
      best_count = count;
      best_point = c;
    }
  }
  if (best_count) {
    alphabet->points[0] = best_point;
    counts[best_point] = 0;
    n_alpha = 1;
  }
  /* everything else in code point order (charmodel-init.c:121-140) */
  for (int c = 0; c < n_chars; c++) {
    int count = counts[c];
    if (!count) {
      continue;
    }
    if (weighted_count(c, count, digit_adjust, alpha_adjust) >= min_count) {
      if (n_alpha == 256) {
        goto fail;
      }
      alphabet->points[n_alpha++] = c;
    } else {
      if (n_collapsed == 256) {
        goto fail;
      }
      alphabet->collapsed_points[n_collapsed++] = c;
    }
  }
  if (n_alpha == 0) {
    goto fail;
  }
  free(counts);
  alphabet->len = n_alpha;
  alphabet->collapsed_len = n_collapsed;
  rnn_char_alphabet_set_flags(alphabet, ignore_case, utf8, collapse_space);
  return 0;
fail:
  fprintf(stderr, "threshold of %f over %d chars led to %d in alphabet, %d collapsed characters\n",
          threshold, n, n_alpha, n_collapsed);
  free(counts);
  alphabet->len = 0;
  alphabet->collapsed_len = 0;
  return -1;
}

/* charmodel-init.c:160-205: whole file into new memory; 0 or an error code */
int rnn_char_alloc_file_contents(const char *filename, char **contents, int *len) {
  FILE *f = fopen(filename, "r");
  if (!f) {
    fprintf(stderr, "could not open '%s'\n", filename);
    *contents = NULL;
    *len = 0;
    return -1;
  }
  fseek(f, 0, SEEK_END);
  long size = ftell(f);
  rewind(f);
  char *c = malloc(size + 1);
  size_t got = fread(c, 1, size, f);
  fclose(f);
  c[got] = 0;
  *contents = c;
  *len = (int)got;
  return 0;
}

int rnn_char_find_alphabet_f(const char *filename, RnnCharAlphabet *alphabet, double threshold,
                             double digit_adjust, double alpha_adjust) {
  char *text;
  int len;
  if (rnn_char_alloc_file_contents(filename, &text, &len)) {
    return -1;
  }
  int err = rnn_char_find_alphabet_s(text, len, alphabet, threshold, digit_adjust, alpha_adjust);
  free(text);
  return err;
}

/* charmodel-init.c:224-235 */
static int symbol_of_space(const RnnCharAlphabet *alphabet) {
  for (int i = 0; i < alphabet->len; i++) {
    if (alphabet->points[i] == ' ') {
      return i;
    }
  }
  return 0; /* no space in the alphabet: the collapse target stands in */
}

/* charmodel-init.c:238-265 */
int *rnn_char_new_char_lut(const RnnCharAlphabet *alphabet) {
  const int case_insensitive = alphabet->flags & RNN_CHAR_FLAG_CASE_INSENSITIVE;
  const int space = symbol_of_space(alphabet);
  const int len = (alphabet->flags & RNN_CHAR_FLAG_UTF8) ? 0x200001 : 257;
  int *lut = malloc(len * sizeof(int));
  for (int i = 0; i < len; i++) {
    lut[i] = space; /* unknown characters read as space */
  }
  for (int i = 0; i < alphabet->collapsed_len; i++) {
    lut[alphabet->collapsed_points[i]] = 0;
  }
  for (int i = 0; i < alphabet->len; i++) {
    int c = alphabet->points[i];
    lut[c] = i;
    if (case_insensitive && islower(c)) {
      lut[toupper(c)] = i;
    }
  }
  return lut;
}

/* charmodel-init.c:270-329 */
u8 *rnn_char_alloc_encoded_text(RnnCharAlphabet *alphabet, const char *text, int byte_len,
                                int *encoded_len, int *char_to_net, bool verbose) {
  const int collapse_space = alphabet->flags & RNN_CHAR_FLAG_COLLAPSE_SPACE;
  const int utf8 = alphabet->flags & RNN_CHAR_FLAG_UTF8;
  const int space = symbol_of_space(alphabet);
  int *lut = char_to_net ? char_to_net : rnn_char_new_char_lut(alphabet);
  u8 *out = malloc((size_t)byte_len * 2 + 4);
  u8 prev = space;
  const char *s = text;
  int i, j = 0;
  for (i = 0; i < byte_len; i++) {
    int chr;
    if (utf8) {
      chr = next_code_point(&s);
      if (chr <= 0) {
        break;
      }
    } else {
      chr = text[i]; /* a plain (signed) char, as in the reference */
      if (chr == 0) {
        break;
      }
    }
    u8 c = lut[chr];
    if (!collapse_space || c != space || prev != space) {
      prev = c;
      out[j++] = c;
    }
  }
  out[j] = 0;
  *encoded_len = j;
  if (!char_to_net) {
    free(lut);
  }
  if (verbose) {
    fprintf(stderr, "original text was %d chars (%d bytes), encoded is %d\n", i, byte_len,
            *encoded_len);
  }
  return realloc(out, j + 1);
}

/* charmodel-init.c:334-349 */
u8 *rnn_char_load_new_encoded_text(const char *filename, RnnCharAlphabet *alphabet,
                                   int *encoded_len, int quietness) {
  char *raw;
  int raw_len;
  if (rnn_char_alloc_file_contents(filename, &raw, &raw_len)) {
    *encoded_len = 0;
    return NULL;
  }
  u8 *enc = rnn_char_alloc_encoded_text(alphabet, raw, raw_len, encoded_len, NULL, quietness < 1);
  free(raw);
  return enc;
}
