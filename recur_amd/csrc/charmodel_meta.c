/* charmodel_meta.c -- the strings around a character model: net metadata, file names,
 * symbol <-> text conversions (charmodel-init.c:352-372, 430-800 of the reference).
 * Host-only C; same names, formats and return values, because saved nets carry the
 * metadata string and the file name embeds its hash. */
#include "char_host.h"
#include <inttypes.h>

#define C_NORMAL "\033[00m"
#define C_DARK_YELLOW "\033[00;33m"

/* ------------------------------------------------------------- code points -- */

/* One UTF-8 sequence from *s (advanced past it).  -1: malformed (a stray or missing continuation
 * byte, an over-long form); -2: a lead byte of a sequence longer than four bytes (utf8.h:79-160's
 * verdicts). */
static int take_codepoint(const char **s) {
  static const struct {
    unsigned char mask, lead, extra;
    int smallest; /* below this the sequence is longer than the value needs */
  } form[] = {{0x80, 0x00, 0, 0}, {0xE0, 0xC0, 1, 0x80}, {0xF0, 0xE0, 2, 0x800}, {0xF8, 0xF0, 3, 0x10000}};
  const unsigned char first = (unsigned char)*(*s)++;
  for (size_t f = 0; f < sizeof(form) / sizeof(form[0]); f++) {
    if ((first & form[f].mask) != form[f].lead) {
      continue;
    }
    int value = first & ~form[f].mask;
    for (int k = 0; k < form[f].extra; k++) {
      const unsigned char more = (unsigned char)*(*s)++;
      if ((more & 0xC0) != 0x80) {
        return -1;
      }
      value = (value << 6) | (more & 0x3F);
    }
    return value < form[f].smallest ? -1 : value;
  }
  return (first & 0xC0) == 0x80 ? -1 : -2;
}

/* n code points as text (bytes when !utf8, where a zero ends the list); the caller frees */
static char *text_of_points(const int *points, int n, int utf8) {
  char *text = malloc((size_t)n * 4 + 1), *w = text;
  for (int i = 0; i < n; i++) {
    if (!utf8 && !points[i]) {
      break;
    }
    const int wrote = ramd_put_codepoint((unsigned)points[i], w, utf8);
    if (!wrote) {
      fprintf(stderr, "bad unicode code %d\n", points[i]);
      break;
    }
    w += wrote;
  }
  *w = 0;
  return text;
}

/* the first `room` code points of a string; returns how many there were (a zero, or in utf-8 a bad
 * sequence, ends it) */
static int points_of_string(int *points, int room, const char *string, int utf8) {
  int n = 0;
  for (const char *s = string; n < room; n++) {
    const int c = utf8 ? take_codepoint(&s) : (unsigned char)*s++;
    if (c <= 0) {
      if (!utf8) {
        points[n] = 0; /* (the byte form stores its terminator, as the reference's loop does) */
      }
      break;
    }
    points[n] = c;
  }
  return n;
}

static void show_points(const char *label, const int *points, int n, int utf8) {
  char *text = text_of_points(points, n, utf8);
  fprintf(stderr, "%s" C_DARK_YELLOW "\xc2\xbb\xc2\xbb" C_NORMAL "%s" C_DARK_YELLOW "\xc2\xab\xc2\xab" C_NORMAL "\n", label, text);
  free(text);
  for (int i = 0; i < n; i++) {
    fprintf(stderr, "%d, ", points[i]);
  }
  fputc('\n', stderr);
}

/* charmodel.h:230: the alphabet and the collapse set, as text and as numbers, on stderr */
void rnn_char_dump_alphabet(RnnCharAlphabet *alphabet) {
  const int utf8 = (alphabet->flags & RNN_CHAR_FLAG_UTF8) != 0;
  show_points("alphabet:  ", alphabet->points, alphabet->len, utf8);
  show_points("collapsed: ", alphabet->collapsed_points, alphabet->collapsed_len, utf8);
}

/* charmodel.h:232: which symbol of the alphabet is the first character of s; -1 if none */
int rnn_char_get_codepoint(RnnCharAlphabet *a, const char *s) {
  int wanted = 0;
  points_of_string(&wanted, 1, s, (a->flags & RNN_CHAR_FLAG_UTF8) != 0);
  for (int sym = 0; sym < a->len; sym++) {
    if (wanted == a->points[sym]) {
      return sym;
    }
  }
  return -1;
}

/* charmodel.h:172: symbols as bytes of a (byte) alphabet string into a file, for inspection */
void rnn_char_dump_collapsed_text(const u8 *text, int len, const char *name,
                                  const char *alphabet) {
  FILE *f = fopen(name, "w");
  if (!f) { /* recur-common.h's fopen_or_abort */
    fprintf(stderr, "could not open '%s'\n", name);
    abort();
  }
  for (const u8 *t = text; t < text + len; t++) {
    fputc(alphabet[*t], f);
  }
  fclose(f);
}

/* charmodel.h:170: symbols back to text (a symbol whose code point is zero ends it) */
char *rnn_char_uncollapse_text(RnnCharAlphabet *alphabet, const u8 *orig, int len,
                               int *dest_len) {
  const int utf8 = (alphabet->flags & RNN_CHAR_FLAG_UTF8) != 0;
  char *text = malloc((size_t)(len + 2) * (utf8 ? 4 : 1)), *w = text;
  for (int i = 0; i < len; i++) {
    const int point = alphabet->points[orig[i]];
    const int wrote = point ? ramd_put_codepoint((unsigned)point, w, utf8) : 0;
    if (!wrote) {
      if (point) {
        fprintf(stderr, "bad unicode code %d\n", point);
      }
      break;
    }
    w += wrote;
  }
  *w = 0;
  *dest_len = (int)(w - text);
  return realloc(text, (size_t)*dest_len + 1);
}

/* ----------------------------------------------------------------- metadata -- */

/* The metadata string keeps its two text fields percent-encoded: every byte outside the printable
 * ascii range 33..126, and '%' itself, as %xx in lower-case hex (charmodel-init.c:483-531). */
static int plain_in_metadata(unsigned char c) { return c > 32 && c < 127 && c != '%'; }

static char *urlencode_alloc(const char *orig) {
  const size_t n = strlen(orig);
  char *enc = malloc(3 * n + 1), *w = enc;
  for (const unsigned char *c = (const unsigned char *)orig; *c; c++) {
    w += plain_in_metadata(*c) ? sprintf(w, "%c", *c) : sprintf(w, "%%%02x", *c);
  }
  *w = 0;
  return realloc(enc, (size_t)(w - enc) + 1);
}

/* value of a hex digit the way the reference reads it: letters (bit 6 set) count from 9, only the low
 * four bits matter -- so a malformed escape decodes to SOMETHING rather than failing */
static int lenient_hex(char c) { return ((c & 0x40) ? c + 9 : c) & 15; }

static char *urldecode_alloc(const char *orig) {
  const size_t n = strlen(orig);
  char *dec = malloc(n + 1), *w = dec;
  for (size_t r = 0; r < n;) {
    if (orig[r] == '%' && r + 2 <= n) {
      *w++ = (char)(lenient_hex(orig[r + 1]) << 4 | lenient_hex(orig[r + 2]));
      r += 3;
    } else {
      *w++ = orig[r++];
    }
  }
  *w = 0;
  return realloc(dec, (size_t)(w - dec) + 1);
}

/* charmodel.h:262: the five lines a saved net carries.  Names and order are the file format. */
char *rnn_char_construct_metadata(const struct RnnCharMetadata *m) {
  char *fields[2] = {urlencode_alloc(m->alphabet), urlencode_alloc(m->collapse_chars)};
  char *text = NULL;
  if (asprintf(&text, "alphabet %s\ncollapse_chars %s\nutf8 %d\ncollapse_space %d\ncase_insensitive %d\n", fields[0],
               fields[1], m->utf8, m->collapse_space, m->case_insensitive) < 0) {
    fprintf(stderr, "librecur_amd: out of memory for a metadata string\n");
    abort();
  }
  free(fields[0]);
  free(fields[1]);
  return text;
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

/* charmodel.h:261: dest becomes a deep copy of src (what dest held is released) */
void rnn_char_copy_metadata_items(struct RnnCharMetadata *src, struct RnnCharMetadata *dest) {
  char *alphabet = strdup(src->alphabet), *collapse = strdup(src->collapse_chars);
  rnn_char_free_metadata_items(dest);
  *dest = *src; /* the three flags */
  dest->alphabet = alphabet;
  dest->collapse_chars = collapse;
}

/* recur-common.h:207-216 */
uint32_t ramd_hash32(const char *s) {
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
  uint32_t sig = ramd_hash32(metadata);
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

/* Does the net's stored description of its text encoding agree with the one the caller is about to
 * use (charmodel.h:258-259; behaviour of charmodel-init.c:672-719)?  Returns 0 when the two are
 * consistent on return -- they were equal, the net had none, or one side was made to follow the other:
 * `trust_file_metadata` makes the CALLER adopt the net's (when it parses), `force_metadata` rewrites
 * the NET's string -- -2 when they differ and neither was asked for, -1 for missing arguments. */
enum meta_resolution { META_SAME, META_ADOPT_NETS, META_OVERWRITE_NETS, META_CONFLICT };

static enum meta_resolution resolve_metadata(const char *nets, const char *callers, bool trust, bool force) {
  if (!nets || strcmp(nets, callers) == 0) {
    return META_SAME;
  }
  return trust ? META_ADOPT_NETS : force ? META_OVERWRITE_NETS : META_CONFLICT;
}

int rnn_char_check_metadata(RecurNN *net, struct RnnCharMetadata *m, bool trust_file_metadata,
                            bool force_metadata) {
  if (!net || !m) {
    fprintf(stderr, "net is %p, metadata is %p, in %s\n", (void *)net, (void *)m, __func__);
    return -1;
  }
  char *callers = rnn_char_construct_metadata(m);
  const enum meta_resolution what = resolve_metadata(net->metadata, callers, trust_file_metadata, force_metadata);
  int verdict = 0;
  if (what != META_SAME) {
    fprintf(stderr, "metadata doesn't match. Expected:\n%s\nLoaded from net:\n%s\n\n", callers, net->metadata);
  }
  switch (what) {
  case META_SAME:
    break;
  case META_ADOPT_NETS: {
    struct RnnCharMetadata stored = {0};
    if (rnn_char_load_metadata(net->metadata, &stored) == 0) {
      fprintf(stderr, "Using the net's metadata. Use --force-metadata to override\n");
      rnn_char_copy_metadata_items(&stored, m);
    } else {
      fprintf(stderr, "The net's metadata doesn't load. Using otherwise determined metadata\n");
    }
    rnn_char_free_metadata_items(&stored);
    break;
  }
  case META_OVERWRITE_NETS:
    fprintf(stderr, "Updating the net's metadata to match that requested (because --force-metadata)\n");
    free(net->metadata);
    net->metadata = callers;
    callers = NULL;
    break;
  case META_CONFLICT:
    verdict = -2;
    break;
  }
  free(callers);
  return verdict;
}

/* charmodel.h:236: the alphabet a net was trained with, from the metadata it carries.  (Where the
 * metadata is missing or does not parse the reference reads freed or unset memory; an empty alphabet
 * comes back here.) */
RnnCharAlphabet *rnn_char_new_alphabet_from_net(RecurNN *net) {
  RnnCharMetadata stored = {0};
  if (net->metadata) {
    rnn_char_load_metadata(net->metadata, &stored);
  }
  RnnCharAlphabet *a = rnn_char_new_alphabet();
  rnn_char_alphabet_set_flags(a, stored.case_insensitive, stored.utf8, stored.collapse_space);
  if (stored.alphabet) {
    a->len = points_of_string(a->points, 256, stored.alphabet, stored.utf8);
  }
  if (stored.collapse_chars) {
    a->collapsed_len = points_of_string(a->collapsed_points, 256, stored.collapse_chars, stored.utf8);
  }
  rnn_char_free_metadata_items(&stored);
  if (net->input_size != a->len || net->output_size != a->len) {
    fprintf(stderr, "the net reads %d symbols and writes %d, its alphabet has %d.\n", net->input_size, net->output_size,
            a->len);
  }
  return a;
}
