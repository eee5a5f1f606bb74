/* cdb.c -- a small reader and writer for D. J. Bernstein's "constant database"
 * file format, which the reference uses as its net container through the
 * third-party tinycdb library (recur-nn-io.c:3, 26-65, 124, 168-184; tinycdb is
 * an unvendored, unpinned distro dependency: Makefile:50-51).  tinycdb is not
 * available here, so the published format is implemented directly:
 *
 *   bytes 0..2047   256 x { u32 table_pos, u32 table_slots }        (little endian)
 *   records         { u32 key_len, u32 data_len, key, data } ...
 *   256 tables      table t holds the records whose hash & 255 == t, as
 *                   2 x count slots of { u32 hash, u32 record_pos }, filled by
 *                   open addressing from slot (hash >> 8) % slots; an empty
 *                   slot has record_pos 0.
 *   hash            h = 5381; for each key byte c: h = (h * 33) ^ c
 *
 * Pinned by tests against the reference's committed fixture
 * test/multi-text-6c34c563i73-h99-o3650.net (read side) and by round trips.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "cdb.h"

static uint32_t cdb_hash(const void *key, uint32_t len) {
  const unsigned char *p = key;
  uint32_t h = 5381;
  while (len--) {
    h = ((h << 5) + h) ^ *p++;
  }
  return h;
}

static void put_u32(unsigned char *p, uint32_t v) {
  p[0] = v & 255;
  p[1] = (v >> 8) & 255;
  p[2] = (v >> 16) & 255;
  p[3] = v >> 24;
}

static uint32_t get_u32(const unsigned char *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* ------------------------------------------------------------------ writer -- */

int cdbw_start(CdbWriter *w, FILE *f) {
  memset(w, 0, sizeof(*w));
  w->f = f;
  unsigned char zero[2048] = {0};
  if (fwrite(zero, 1, sizeof(zero), f) != sizeof(zero)) {
    return -1;
  }
  w->pos = 2048;
  return 0;
}

int cdbw_add(CdbWriter *w, const void *key, uint32_t klen, const void *data, uint32_t dlen) {
  unsigned char hdr[8];
  if ((uint64_t)w->pos + 8 + klen + dlen > 0xffffffffu) {
    return -1; /* the format addresses 4 GiB */
  }
  if (w->n == w->cap) {
    w->cap = w->cap ? w->cap * 2 : 64;
    w->recs = realloc(w->recs, w->cap * sizeof(*w->recs));
    if (!w->recs) {
      return -1;
    }
  }
  put_u32(hdr, klen);
  put_u32(hdr + 4, dlen);
  if (fwrite(hdr, 1, 8, w->f) != 8 || fwrite(key, 1, klen, w->f) != klen ||
      fwrite(data, 1, dlen, w->f) != dlen) {
    return -1;
  }
  w->recs[w->n].hash = cdb_hash(key, klen);
  w->recs[w->n].pos = w->pos;
  w->n++;
  w->pos += 8 + klen + dlen;
  return 0;
}

int cdbw_finish(CdbWriter *w) {
  unsigned char header[2048];
  int rc = 0;
  uint32_t count[256] = {0};
  for (uint32_t i = 0; i < w->n; i++) {
    count[w->recs[i].hash & 255]++;
  }
  for (int t = 0; t < 256 && rc == 0; t++) {
    uint32_t slots = count[t] * 2;
    put_u32(header + 8 * t, w->pos);
    put_u32(header + 8 * t + 4, slots);
    if (!slots) {
      continue;
    }
    uint32_t *tab = calloc(slots * 2, sizeof(uint32_t)); /* (hash, pos) pairs */
    for (uint32_t i = 0; i < w->n; i++) { /* insertion order within a table */
      uint32_t h = w->recs[i].hash;
      if ((h & 255) != (uint32_t)t) {
        continue;
      }
      uint32_t where = (h >> 8) % slots;
      while (tab[2 * where + 1]) {
        if (++where == slots) {
          where = 0;
        }
      }
      tab[2 * where] = h;
      tab[2 * where + 1] = w->recs[i].pos;
    }
    for (uint32_t s = 0; s < slots && rc == 0; s++) {
      unsigned char slot[8];
      put_u32(slot, tab[2 * s]);
      put_u32(slot + 4, tab[2 * s + 1]);
      if (fwrite(slot, 1, 8, w->f) != 8) {
        rc = -1;
      }
    }
    free(tab);
    w->pos += slots * 8;
  }
  if (rc == 0 && (fseek(w->f, 0, SEEK_SET) != 0 || fwrite(header, 1, 2048, w->f) != 2048 ||
                  fflush(w->f) != 0)) {
    rc = -1;
  }
  free(w->recs);
  w->recs = NULL;
  return rc;
}

/* ------------------------------------------------------------------ reader -- */

int cdbr_open(CdbReader *r, const char *filename) {
  memset(r, 0, sizeof(*r));
  FILE *f = fopen(filename, "rb");
  if (!f) {
    return -1;
  }
  if (fseek(f, 0, SEEK_END) != 0) {
    fclose(f);
    return -1;
  }
  long size = ftell(f);
  rewind(f);
  if (size < 2048) {
    fclose(f);
    return -1;
  }
  r->data = malloc(size);
  if (!r->data || fread(r->data, 1, size, f) != (size_t)size) {
    free(r->data);
    r->data = NULL;
    fclose(f);
    return -1;
  }
  fclose(f);
  r->size = (size_t)size;
  return 0;
}

void cdbr_close(CdbReader *r) {
  free(r->data);
  r->data = NULL;
}

/* Returns 1 and the value's location if the key exists, 0 if not, -1 if the
 * file is malformed (the tri-state of tinycdb's cdb_seek, recur-nn-io.c:176). */
int cdbr_find(const CdbReader *r, const void *key, uint32_t klen, const unsigned char **val,
              uint32_t *vlen) {
  uint32_t h = cdb_hash(key, klen);
  const unsigned char *hd = r->data + 8 * (h & 255);
  uint32_t tpos = get_u32(hd), slots = get_u32(hd + 4);
  if (!slots) {
    return 0;
  }
  if ((uint64_t)tpos + (uint64_t)slots * 8 > r->size) {
    return -1;
  }
  uint32_t where = (h >> 8) % slots;
  for (uint32_t probes = 0; probes < slots; probes++) {
    const unsigned char *slot = r->data + tpos + 8 * where;
    uint32_t sh = get_u32(slot), pos = get_u32(slot + 4);
    if (!pos) {
      return 0;
    }
    if (sh == h) {
      if ((uint64_t)pos + 8 > r->size) {
        return -1;
      }
      uint32_t kl = get_u32(r->data + pos), dl = get_u32(r->data + pos + 4);
      if ((uint64_t)pos + 8 + kl + dl > r->size) {
        return -1;
      }
      if (kl == klen && memcmp(r->data + pos + 8, key, klen) == 0) {
        *val = r->data + pos + 8 + kl;
        *vlen = dl;
        return 1;
      }
    }
    if (++where == slots) {
      where = 0;
    }
  }
  return 0;
}

/* Sequential walk over the records in file order (scripts/pycdb.py:36-80 of
 * the reference reads nets this way).  *cursor starts at 0. */
int cdbr_next(const CdbReader *r, size_t *cursor, const unsigned char **key, uint32_t *klen,
              const unsigned char **val, uint32_t *vlen) {
  size_t end = get_u32(r->data); /* the first table starts where the records end */
  for (int t = 0; t < 256; t++) {
    uint32_t p = get_u32(r->data + 8 * t);
    if (p < end) {
      end = p;
    }
  }
  size_t pos = *cursor ? *cursor : 2048;
  if (pos + 8 > end) {
    return 0;
  }
  uint32_t kl = get_u32(r->data + pos), dl = get_u32(r->data + pos + 4);
  if (pos + 8 + kl + dl > end) {
    return -1;
  }
  *key = r->data + pos + 8;
  *klen = kl;
  *val = *key + kl;
  *vlen = dl;
  *cursor = pos + 8 + kl + dl;
  return 1;
}
