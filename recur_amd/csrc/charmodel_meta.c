/* charmodel_meta.c -- the strings around a character model: net metadata, file names,
 * symbol <-> text conversions (charmodel-init.c:352-372, 430-800 of the reference).
 * Host-only C; same names, formats and return values, because saved nets carry the
 * metadata string and the file name embeds its hash. */
#include "rnn_host.h"
#include "recur_amd_char.h"
#include <inttypes.h>

#define C_NORMAL "\033[00m"
#define C_DARK_YELLOW "\033[00;33m"

/* ------------------------------------------------------------- code points -- */

/* utf8.h:31-57: 0 for a code point that does not fit four bytes */
static int put_utf8(unsigned code, char *s) {
  if (code < 0x80) {
    s[0] = (char)code;
    return 1;
  }
  if (code < 0x800) {
    s[0] = (char)(0xC0 | (code >> 6));
    s[1] = (char)(0x80 | (code & 63));
    return 2;
  }
  if (code < 0x10000) {
    s[0] = (char)(0xE0 | (code >> 12));
    s[1] = (char)(0x80 | ((code >> 6) & 63));
    s[2] = (char)(0x80 | (code & 63));
    return 3;
  }
  if (code < 0x200000) {
    s[0] = (char)(0xF0 | (code >> 18));
    s[1] = (char)(0x80 | ((code >> 12) & 63));
    s[2] = (char)(0x80 | ((code >> 6) & 63));
    s[3] = (char)(0x80 | (code & 63));
    return 4;
  }
  return 0;
}

/* utf8.h:79-160 (read_utf8_char): -1 for malformed input, -2 for a lead byte beyond
 * four-byte sequences; over-long encodings are malformed */
static int get_utf8(const char **s) {
  int c = (unsigned char)**s, extra;
  (*s)++;
  if (!(c & 0x80)) {
    return c;
  } else if ((c & 0xE0) == 0xC0) {
    c &= 31;
    extra = 1;
  } else if ((c & 0xF0) == 0xE0) {
    c &= 15;
    extra = 2;
  } else if ((c & 0xF8) == 0xF0) {
    c &= 7;
    extra = 3;
  } else if ((c & 0xC0) == 0x80) {
    return -1;
  } else {
    return -2;
  }
  for (int i = 0; i < extra; i++) {
    int x = (unsigned char)**s;
    (*s)++;
    if ((x & 0xC0) != 0x80) {
      return -1;
    }
    c = (c << 6) + (x & 63);
  }
  int min = 1 << (1 + extra * 5 + (extra == 1));
  return c < min ? -1 : c;
}

/* utf8.h:193-232 */
static char *string_of_points(const int *points, int maxlen, int utf8) {
  char *str = malloc((size_t)maxlen * (utf8 ? 4 : 1) + 1);
  char *s = str;
  for (int i = 0; i < maxlen; i++) {
    int code = points[i];
    if (utf8) {
      int wrote = put_utf8((unsigned)code, s);
      if (wrote == 0) {
        fprintf(stderr, "bad unicode code %d\n", code);
        break;
      }
      s += wrote;
    } else {
      if (!code) {
        break;
      }
      *s++ = (char)code;
    }
  }
  *s = 0;
  return str;
}

/* utf8.h:234-270 */
static int points_of_string(int *points, int len, const char *string, int utf8) {
  int i;
  if (utf8) {
    const char *s = string;
    for (i = 0; i < len; i++) {
      int c = get_utf8(&s);
      if (c <= 0) {
        break;
      }
      points[i] = c;
    }
  } else {
    const unsigned char *s = (const unsigned char *)string;
    for (i = 0; i < len; i++) {
      points[i] = s[i];
      if (!points[i]) {
        break;
      }
    }
  }
  return i;
}

/* charmodel-init.c:351-372 */
void rnn_char_dump_alphabet(RnnCharAlphabet *alphabet) {
  int utf8 = alphabet->flags & RNN_CHAR_FLAG_UTF8;
  char *s = string_of_points(alphabet->points, alphabet->len, utf8);
  char *s2 = string_of_points(alphabet->collapsed_points, alphabet->collapsed_len, utf8);
  fprintf(stderr, "alphabet:  " C_DARK_YELLOW "\xc2\xbb\xc2\xbb" C_NORMAL "%s" C_DARK_YELLOW
                  "\xc2\xab\xc2\xab" C_NORMAL "\n", s);
  for (int i = 0; i < alphabet->len; i++) {
    fprintf(stderr, "%d, ", alphabet->points[i]);
  }
  putc('\n', stderr);
  fprintf(stderr, "collapsed: " C_DARK_YELLOW "\xc2\xbb\xc2\xbb" C_NORMAL "%s" C_DARK_YELLOW
                  "\xc2\xab\xc2\xab" C_NORMAL "\n", s2);
  for (int i = 0; i < alphabet->collapsed_len; i++) {
    fprintf(stderr, "%d, ", alphabet->collapsed_points[i]);
  }
  putc('\n', stderr);
  free(s);
  free(s2);
}

/* charmodel-init.c:788-799: symbol number of the first character of s, or -1 */
int rnn_char_get_codepoint(RnnCharAlphabet *a, const char *s) {
  int p = 0;
  points_of_string(&p, 1, s, a->flags & RNN_CHAR_FLAG_UTF8);
  for (int i = 0; i < a->len; i++) {
    if (a->points[i] == p) {
      return i;
    }
  }
  return -1;
}

/* charmodel-init.c:429-440 */
void rnn_char_dump_collapsed_text(const u8 *text, int len, const char *name,
                                  const char *alphabet) {
  FILE *f = fopen(name, "w");
  if (!f) {
    fprintf(stderr, "could not open '%s'\n", name);
    abort(); /* fopen_or_abort, recur-common.h */
  }
  for (int i = 0; i < len; i++) {
    fputc(alphabet[text[i]], f);
  }
  fclose(f);
}

/* charmodel-init.c:443-478: symbols back to text; stops at a zero code point */
char *rnn_char_uncollapse_text(RnnCharAlphabet *alphabet, const u8 *orig, int len,
                               int *dest_len) {
  int utf8 = alphabet->flags & RNN_CHAR_FLAG_UTF8;
  char *mem = malloc((size_t)(len + 2) * (utf8 ? 4 : 1));
  char *s = mem;
  for (int i = 0; i < len; i++) {
    int code = alphabet->points[orig[i]];
    if (code == 0) {
      break;
    }
    if (utf8) {
      int wrote = put_utf8((unsigned)code, s);
      if (wrote == 0) {
        fprintf(stderr, "bad unicode code %d\n", code);
        break;
      }
      s += wrote;
    } else {
      *s++ = (char)code;
    }
  }
  *s = 0;
  *dest_len = (int)(s - mem);
  return realloc(mem, (size_t)(s - mem) + 1);
}

/* ----------------------------------------------------------------- metadata -- */

/* charmodel-init.c:483-505: everything outside 33..126, and '%', as %xx (lower case) */
static char *urlencode_alloc(const char *orig) {
  size_t len = strlen(orig);
  char *s = malloc(len * 3 + 1);
  static const char hex[] = "0123456789abcdef";
  size_t j = 0;
  for (size_t i = 0; i < len; i++) {
    char c = orig[i];
    if (c > 32 && c < 127 && c != '%') {
      s[j++] = c;
    } else {
      unsigned char u = (unsigned char)c;
      s[j++] = '%';
      s[j++] = hex[u >> 4];
      s[j++] = hex[u & 15];
    }
  }
  s[j] = 0;
  return realloc(s, j + 1);
}

/* charmodel-init.c:507-531 (same hex-digit arithmetic; stops at the end of the input) */
static char *urldecode_alloc(const char *orig) {
  size_t len = strlen(orig);
  char *s = malloc(len + 1);
  size_t i = 0, j = 0;
  while (j < len) {
    char c = orig[j];
    if (c == '%' && j + 2 < len + 1) {
      char hi = orig[j + 1], lo = orig[j + 2];
      char d = (char)((((hi & 0x40) ? hi + 9 : hi) & 15) << 4);
      d += ((lo & 0x40) ? lo + 9 : lo) & 15;
      s[i++] = d;
      j += 3;
    } else {
      s[i++] = c;
      j++;
    }
  }
  s[i] = 0;
  return realloc(s, i + 1);
}

/* charmodel-init.c:534-560 */
char *rnn_char_construct_metadata(const struct RnnCharMetadata *m) {
  char *metadata;
  char *enc_alphabet = urlencode_alloc(m->alphabet);
  char *enc_collapse = urlencode_alloc(m->collapse_chars);
  int ret = asprintf(&metadata,
                     "alphabet %s\n"
                     "collapse_chars %s\n"
                     "utf8 %d\n"
                     "collapse_space %d\n"
                     "case_insensitive %d\n",
                     enc_alphabet, enc_collapse, m->utf8, m->collapse_space, m->case_insensitive);
  if (ret == -1) {
    fprintf(stderr, "can't alloc memory for metadata. or something.\n");
    abort();
  }
  free(enc_alphabet);
  free(enc_collapse);
  return metadata;
}

/* charmodel-init.c:562-628: five "key value" lines in a fixed order; 0 or -1 */
int rnn_char_load_metadata(const char *orig, struct RnnCharMetadata *m) {
  char *metadata = strdup(orig);
  char *s = metadata;
  static const char *const keys[5] = {"alphabet", "collapse_chars", "utf8", "collapse_space",
                                      "case_insensitive"};
  char *values[5];
  for (int k = 0; k < 5; k++) {
    char *key = strsep(&s, " ");
    char *value = strsep(&s, "\n");
    if (!key || strcmp(key, keys[k]) || !value) {
      fprintf(stderr, "Error loading metadata. key is %s, value %s\n", key ? key : "(null)",
              value ? value : "(null)");
      /* what was decoded so far stays in *m, as in the reference */
      if (k > 0) m->alphabet = urldecode_alloc(values[0]);
      if (k > 1) m->collapse_chars = urldecode_alloc(values[1]);
      free(metadata);
      return -1;
    }
    values[k] = value;
  }
  m->alphabet = urldecode_alloc(values[0]);
  m->collapse_chars = urldecode_alloc(values[1]);
  m->utf8 = strtol(values[2], NULL, 10);
  m->collapse_space = strtol(values[3], NULL, 10);
  m->case_insensitive = strtol(values[4], NULL, 10);
  if (s && *s) {
    fprintf(stderr, "Found extra metadata: %s\n", s);
  }
  free(metadata);
  return 0;
}

void rnn_char_free_metadata_items(struct RnnCharMetadata *m) {
  free(m->alphabet);
  free(m->collapse_chars);
}

/* charmodel-init.c:636-650 */
void rnn_char_copy_metadata_items(struct RnnCharMetadata *src, struct RnnCharMetadata *dest) {
  free(dest->alphabet);
  free(dest->collapse_chars);
  dest->alphabet = strdup(src->alphabet);
  dest->collapse_chars = strdup(src->collapse_chars);
  dest->utf8 = src->utf8;
  dest->collapse_space = src->collapse_space;
  dest->case_insensitive = src->case_insensitive;
}

/* recur-common.h:207-216 */
static uint32_t hash32(const char *s) {
  uint32_t sig = 0;
  size_t len = strlen(s);
  for (size_t i = 0; i < len; i++) {
    uint8_t t = (uint8_t)s[i];
    uint32_t x = sig - t;
    sig ^= ((x << 13) | (x >> 19)) + t;
  }
  return sig;
}

/* charmodel-init.c:652-670: <basename>-s<hash of the metadata>-i..[-b..]-h..-o...net */
char *rnn_char_construct_net_filename(struct RnnCharMetadata *m, const char *basename,
                                      int input_size, int bottom_size, int hidden_size,
                                      int output_size) {
  char s[260];
  char *metadata = rnn_char_construct_metadata(m);
  uint32_t sig = hash32(metadata);
  free(metadata);
  if (bottom_size) {
    snprintf(s, sizeof(s), "%s-s%0" PRIx32 "-i%d-b%d-h%d-o%d.net", basename, sig, input_size,
             bottom_size, hidden_size, output_size);
  } else {
    snprintf(s, sizeof(s), "%s-s%0" PRIx32 "-i%d-h%d-o%d.net", basename, sig, input_size,
             hidden_size, output_size);
  }
  fprintf(stderr, "filename: %s\n", s);
  return strdup(s);
}

/* charmodel-init.c:672-719: 0 = consistent (possibly after adopting one side), -1 bad
 * arguments, -2 mismatch with neither flag given */
int rnn_char_check_metadata(RecurNN *net, struct RnnCharMetadata *m, bool trust_file_metadata,
                            bool force_metadata) {
  if (net == NULL || m == NULL) {
    fprintf(stderr, "net is %p, metadata is %p, in %s\n", (void *)net, (void *)m, __func__);
    return -1;
  }
  int ret = 0;
  char *metadata = rnn_char_construct_metadata(m);
  if (net->metadata && strcmp(metadata, net->metadata)) {
    fprintf(stderr, "metadata doesn't match. Expected:\n%s\nLoaded from net:\n%s\n\n", metadata,
            net->metadata);
    if (trust_file_metadata) {
      struct RnnCharMetadata m2 = {0};
      if (rnn_char_load_metadata(net->metadata, &m2)) {
        fprintf(stderr, "The net's metadata doesn't load. Using otherwise determined metadata\n");
      } else {
        fprintf(stderr, "Using the net's metadata. Use --force-metadata to override\n");
        rnn_char_copy_metadata_items(&m2, m);
        rnn_char_free_metadata_items(&m2);
      }
    } else if (force_metadata) {
      fprintf(stderr, "Updating the net's metadata to match that requested "
                      "(because --force-metadata)\n");
      free(net->metadata);
      net->metadata = strdup(metadata);
    } else {
      ret = -2;
    }
  }
  free(metadata);
  return ret;
}

/* charmodel-init.c:733-751 */
RnnCharAlphabet *rnn_char_new_alphabet_from_net(RecurNN *net) {
  RnnCharMetadata m = {0};
  if (net->metadata) {
    rnn_char_load_metadata(net->metadata, &m);
  }
  RnnCharAlphabet *a = rnn_char_new_alphabet();
  rnn_char_alphabet_set_flags(a, m.case_insensitive, m.utf8, m.collapse_space);
  /* (the reference dereferences whatever a failed load left behind; an empty alphabet
   * is returned here instead) */
  a->len = m.alphabet ? points_of_string(a->points, 256, m.alphabet, m.utf8) : 0;
  a->collapsed_len =
      m.collapse_chars ? points_of_string(a->collapsed_points, 256, m.collapse_chars, m.utf8) : 0;
  rnn_char_free_metadata_items(&m);
  if (a->len != net->input_size || a->len != net->output_size) {
    fprintf(stderr, "net sizes in %d out %d, alphabet length %d.\n", net->input_size,
            net->output_size, a->len);
  }
  return a;
}
