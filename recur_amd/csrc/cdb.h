/* cdb.h -- constant-database container used for saved nets (see cdb.c). */
#ifndef RAMD_CDB_H
#define RAMD_CDB_H 1
#include <stdio.h>
#include <stdint.h>
#include <stddef.h>

typedef struct CdbWriter {
  FILE *f;
  uint32_t pos;
  struct {
    uint32_t hash, pos;
  } *recs;
  uint32_t n, cap;
} CdbWriter;

typedef struct CdbReader {
  unsigned char *data;
  size_t size;
} CdbReader;

int cdbw_start(CdbWriter *w, FILE *f);
int cdbw_add(CdbWriter *w, const void *key, uint32_t klen, const void *data, uint32_t dlen);
int cdbw_finish(CdbWriter *w);

int cdbr_open(CdbReader *r, const char *filename);
void cdbr_close(CdbReader *r);
int cdbr_find(const CdbReader *r, const void *key, uint32_t klen, const unsigned char **val,
              uint32_t *vlen);
int cdbr_next(const CdbReader *r, size_t *cursor, const unsigned char **key, uint32_t *klen,
              const unsigned char **val, uint32_t *vlen);
#endif
