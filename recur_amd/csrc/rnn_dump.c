/* rnn_dump.c -- rnn_multi_pgm_dump: debug images of the weight-shaped arrays
 * (recur-nn.h:304; recur-nn-init.c:744-823 choose the array, pgm_dump.h:126-170,
 * 215-220 define the picture: a binary PPM named
 * images/<basename>-<token>-<generation as %08d>-<w>x<h>.ppm whose pixels are
 * red for negative, green for positive, blue for exactly zero, scaled so the
 * largest magnitude is 255).  Cold path, host only. */
#include "char_host.h"

void ramd_write_signed_ppm(const float *a, int width, int height, const char *name) {
  size_t n = (size_t)width * height;
  float biggest = 1e-35f;
  for (size_t i = 0; i < n; i++) {
    float f = fabsf(a[i]);
    if (f > biggest) {
      biggest = f;
    }
  }
  float scale = 255.99f / biggest;
  FILE *fh = fopen(name, "w");
  if (fh == NULL) {
    fprintf(stderr, "could not open '%s' for writing\n", name);
    return;
  }
  fprintf(fh, "P6\n%u %u\n255\n", width, height);
  for (size_t i = 0; i < n; i++) {
    float f = a[i] * scale;
    unsigned char px[3] = {0, 0, 0};
    if (f < 0.0) {
      px[0] = (unsigned char)fabsf(f);
    } else if (f > 0.0) {
      px[1] = (unsigned char)fabsf(f);
    } else {
      px[2] = 180;
    }
    fwrite(px, 1, 3, fh);
  }
  fclose(fh);
}

/* pgm_dump.h:64-81: grey image of magnitudes, scaled so the largest is 255 */
void ramd_write_abs_pgm(const float *a, int width, int height, const char *name) {
  size_t n = (size_t)width * height;
  float biggest = 1e-35f;
  for (size_t i = 0; i < n; i++) {
    float f = fabsf(a[i]);
    if (f > biggest) {
      biggest = f;
    }
  }
  float scale = 255.99f / biggest;
  FILE *fh = fopen(name, "w");
  if (fh == NULL) {
    fprintf(stderr, "could not open '%s' for writing\n", name);
    return;
  }
  fprintf(fh, "P5\n%u %u\n255\n", width, height);
  for (size_t i = 0; i < n; i++) {
    fputc((unsigned char)(fabs(a[i]) * scale), fh);
  }
  fclose(fh);
}

void rnn_multi_pgm_dump(RecurNN *net, const char *dumpees, const char *basename) {
  RecurNNBPTT *bptt = net->bptt;
  char *copy = strdup(dumpees), *cursor = copy, *token;
  ramd_need_host(net, RNN_AMD_WEIGHTS | RNN_AMD_MOMENTUMS | RNN_AMD_DELTAS);
  int aux = (net->flags & RNN_NET_FLAG_AUX_ARRAYS) != 0;
  while ((token = strsep(&cursor, " "))) {
    if (strlen(token) != 3) {
      continue;
    }
    char in = token[0], out = token[1], v = token[2];
    const float *array = NULL;
    int x = 0, y = 0;
    if (out == 'h' && (in == 'i' || in == 'h')) {
      x = net->h_size;
      y = (in == 'i') ? net->i_size : net->hidden_size;
      array = v == 'w' ? net->ih_weights
              : !bptt  ? NULL
              : v == 'm' ? bptt->ih_momentum
              : v == 'd' ? bptt->ih_delta
              : v == 't' ? bptt->ih_delta_tmp
              : (v == 'a' && aux) ? bptt->ih_aux
                                  : NULL;
    } else if (in == 'h' && out == 'o') {
      x = net->o_size;
      y = net->h_size;
      array = v == 'w' ? net->ho_weights
              : !bptt  ? NULL
              : v == 'm' ? bptt->ho_momentum
              : v == 'd' ? bptt->ho_delta
              : (v == 'a' && aux) ? bptt->ho_aux
                                  : NULL;
    } else if (in == 'b' && out == 'i' && net->bottom_layer) { /* recur-nn-init.c:792-808 */
      RecurExtraLayer *bl = net->bottom_layer;
      x = bl->o_size;
      y = bl->i_size;
      array = v == 'w' ? bl->weights
              : v == 'm' ? bl->momentums
              : v == 'd' ? bl->delta
              : (v == 'a' && aux) ? bl->aux
                                  : NULL;
    }
    if (array) {
      char name[200];
      snprintf(name, sizeof(name), "images/%s-%s-%08d-%dx%d.ppm",
               (basename && basename[0]) ? basename : "untitled", token, (int)net->generation, x,
               y);
      ramd_write_signed_ppm(array, x, y, name);
    }
  }
  free(copy);
}

/* ---- rows of a "temporal" image: one row per time step, written out when full ----
 * The struct is pgm_dump.h:227-237's; a caller that wants such images builds it with that header's
 * constructor and hands it over in model->images (text-predict.c:575-600). */
struct _TemporalPPM {
  float *im;
  int width, height;
  int y; /* next row */
  int id;
  char *basename;
  int counter; /* rows written to files so far */
  int mode;    /* pgm_dump.h:222-225: 0 grey magnitudes, 1 signed colour */
  float **source;
};

void ramd_temporal_row(TemporalPPM *ppm, const float *row) {
  memcpy(ppm->im + (size_t)ppm->y * ppm->width, row, sizeof(float) * ppm->width);
  if (++ppm->y < ppm->height) {
    return;
  }
  char name[200]; /* pgm_dump.h:262-288: the file name carries the first row's time step */
  snprintf(name, sizeof(name), "images/%s-%d-%08d-%dx%d.ppm", ppm->basename, ppm->id, ppm->counter, ppm->width,
           ppm->height);
  (ppm->mode == 0 ? ramd_write_abs_pgm : ramd_write_signed_ppm)(ppm->im, ppm->width, ppm->height, name);
  ppm->counter += ppm->height;
  ppm->y = 0;
}

void ramd_image_rows(RecurNN *net, TemporalPPM *input_ppm, TemporalPPM *error_ppm) {
  if (!input_ppm && !error_ppm) {
    return;
  }
  rnn_amd_sync_host(net, RNN_AMD_STREAM); /* the rows live on the device */
  if (input_ppm) {
    ramd_temporal_row(input_ppm, net->input_layer);
  }
  if (error_ppm) {
    ramd_temporal_row(error_ppm, net->bptt->o_error);
  }
}
