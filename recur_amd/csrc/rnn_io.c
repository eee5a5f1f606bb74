/* rnn_io.c -- saved nets: CDB container, format version 10 (gnu11 C, host).
 *
 * Record names, order, sizes and the load-time checks follow recur-nn-io.c:
 * writer 12-147, reader 149-357.  Keys are the literal "obj.attr" strings the
 * reference builds with its QUOTE macros (recur-nn-io.c:47-65).  Values are the
 * raw little-endian bytes of the struct fields / arrays with the padded sizes,
 * so files move between the reference tools and this library in both
 * directions.  The container itself is cdb.c.
 */
#include "rnn_host.h"
#include "cdb.h"
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>

#define SAVE_FORMAT_KEY "save_format_version"
#define MAX_METADATA_SIZE (100u * 1000u * 1000u)

static int put(CdbWriter *w, const char *key, const void *data, size_t len) {
  int rc = cdbw_add(w, key, (uint32_t)strlen(key), data, (uint32_t)len);
  if (rc) {
    fprintf(stderr, "error %d saving '%s'\n", rc, key);
  }
  return rc;
}

#define PUT_FIELD(w, key, field) put((w), (key), &(field), sizeof(field))

/* recur-nn.h:307 / recur-nn-io.c:12-147.  Returns 0, or -1 on any failure. */
int rnn_save_net(RecurNN *net, const char *filename, int backup) {
  char tmpfn[] = "tmp_net_XXXXXX";
  int fd = -1;
  FILE *f = NULL;
  CdbWriter w;
  if (net == NULL || filename == NULL) {
    goto fail;
  }
  ramd_need_host(net, RNN_AMD_WEIGHTS | RNN_AMD_STREAM);
  fd = mkostemp(tmpfn, O_RDWR | O_CREAT);
  if (fd == -1 || (f = fdopen(fd, "w+b")) == NULL) {
    perror("can't open temporary file for writing");
    goto fail;
  }
  if (cdbw_start(&w, f)) {
    goto fail;
  }
  const int version = 10;
  int rc = put(&w, SAVE_FORMAT_KEY, &version, sizeof(version));
  rc |= PUT_FIELD(&w, "net.i_size", net->i_size);
  rc |= PUT_FIELD(&w, "net.h_size", net->h_size);
  rc |= PUT_FIELD(&w, "net.o_size", net->o_size);
  rc |= PUT_FIELD(&w, "net.input_size", net->input_size);
  rc |= PUT_FIELD(&w, "net.hidden_size", net->hidden_size);
  rc |= PUT_FIELD(&w, "net.output_size", net->output_size);
  rc |= PUT_FIELD(&w, "net.ih_size", net->ih_size);
  rc |= PUT_FIELD(&w, "net.ho_size", net->ho_size);
  rc |= PUT_FIELD(&w, "net.generation", net->generation);
  rc |= PUT_FIELD(&w, "net.flags", net->flags);
  rc |= PUT_FIELD(&w, "net.presynaptic_noise", net->presynaptic_noise);
  rc |= PUT_FIELD(&w, "net.activation", net->activation);
  rc |= PUT_FIELD(&w, "net.rng", net->rng);
  rc |= put(&w, "net.ih_weights", net->ih_weights, sizeof(float) * (size_t)net->ih_size);
  rc |= put(&w, "net.ho_weights", net->ho_weights, sizeof(float) * (size_t)net->ho_size);
  if (net->metadata) {
    rc |= put(&w, "net.metadata", net->metadata, strlen(net->metadata) + 1);
  }
  if ((net->flags & RNN_NET_FLAG_OWN_BPTT) && net->bptt) {
    RecurNNBPTT *bptt = net->bptt;
    rc |= PUT_FIELD(&w, "bptt.depth", bptt->depth);
    rc |= PUT_FIELD(&w, "bptt.index", bptt->index);
    rc |= PUT_FIELD(&w, "bptt.learn_rate", bptt->learn_rate);
    rc |= PUT_FIELD(&w, "bptt.ho_scale", bptt->ho_scale);
    rc |= PUT_FIELD(&w, "bptt.momentum", bptt->momentum);
    rc |= PUT_FIELD(&w, "bptt.momentum_weight", bptt->momentum_weight);
    rc |= PUT_FIELD(&w, "bptt.min_error_factor", bptt->min_error_factor);
  }
  if (net->bottom_layer) {
    RecurExtraLayer *bl = net->bottom_layer;
    rc |= PUT_FIELD(&w, "bottom_layer.input_size", bl->input_size);
    rc |= PUT_FIELD(&w, "bottom_layer.output_size", bl->output_size);
    rc |= PUT_FIELD(&w, "bottom_layer.i_size", bl->i_size);
    rc |= PUT_FIELD(&w, "bottom_layer.o_size", bl->o_size);
    rc |= PUT_FIELD(&w, "bottom_layer.learn_rate_scale", bl->learn_rate_scale);
    rc |= PUT_FIELD(&w, "bottom_layer.overlap", bl->overlap);
    rc |= put(&w, "bottom_layer.weights", bl->weights,
              sizeof(float) * (size_t)bl->i_size * bl->o_size);
  }
  if (cdbw_finish(&w) || rc) {
    goto fail;
  }
  fclose(f);
  f = NULL;
  if (backup) {
    /* The reference compares the length of "filename~" with strlen(filename + 2)
     * (recur-nn-io.c:130), which never matches, so no backup is ever made.  Kept. */
    char *backup_filename;
    int size = asprintf(&backup_filename, "%s~", filename);
    if (size != -1) {
      if (size == (int)strlen(filename + 2)) {
        rename(filename, backup_filename);
      }
      free(backup_filename);
    }
  }
  rename(tmpfn, filename);
  return 0;
fail:
  if (f) {
    fclose(f);
    unlink(tmpfn);
  } else if (fd != -1) {
    close(fd);
    unlink(tmpfn);
  }
  fprintf(stderr, "failed to save net %p errno %d filename '%s'\n", (void *)net, errno,
          filename ? filename : "(nil, which is the problem)");
  return -1;
}

/* fixed-size value: must exist with exactly this size (recur-nn-io.c:174-185) */
static int get_fixed(const CdbReader *r, int version, const char *obj, const char *attr,
                     void *dest, size_t size) {
  char key[96];
  const unsigned char *v;
  uint32_t vlen = 0;
  if (version >= 4) {
    snprintf(key, sizeof(key), "%s.%s", obj, attr);
  } else {
    snprintf(key, sizeof(key), "%s", attr);
  }
  int rc = cdbr_find(r, key, (uint32_t)strlen(key), &v, &vlen);
  if (rc < 1) {
    fprintf(stderr, "error %d loading '%s'\n", rc, key);
    return -1;
  }
  if (vlen != size) {
    fprintf(stderr, "size mismatch on '%s' want %zu, found %u\n", key, size, vlen);
    return -1;
  }
  memcpy(dest, v, size);
  return 0;
}

#define GET(obj_name, obj, attr)                                                     \
  do {                                                                               \
    if (get_fixed(&r, version, obj_name, #attr, &(obj).attr, sizeof((obj).attr)))    \
      goto pre_alloc_error;                                                          \
  } while (0)

/* recur-nn.h:306 / recur-nn-io.c:149-357.  NULL on any failure. */
RecurNN *rnn_load_net(const char *filename) {
  CdbReader r;
  RecurNN tmpnet;
  RecurNNBPTT tmpbptt;
  RecurExtraLayer tmpbl;
  RecurNN *net = NULL;
  memset(&tmpnet, 0, sizeof(tmpnet));
  memset(&tmpbptt, 0, sizeof(tmpbptt));
  memset(&tmpbl, 0, sizeof(tmpbl));
  if (cdbr_open(&r, filename)) {
    fprintf(stderr, "can't open '%s' (%s)\n", filename, strerror(errno));
    goto open_error;
  }
  int version = 0;
  {
    const unsigned char *v;
    uint32_t vlen;
    if (cdbr_find(&r, SAVE_FORMAT_KEY, (uint32_t)strlen(SAVE_FORMAT_KEY), &v, &vlen) == 1 &&
        vlen == sizeof(version)) {
      memcpy(&version, v, sizeof(version));
    }
  }
  GET("net", tmpnet, i_size);
  GET("net", tmpnet, h_size);
  GET("net", tmpnet, o_size);
  GET("net", tmpnet, input_size);
  GET("net", tmpnet, hidden_size);
  GET("net", tmpnet, output_size);
  GET("net", tmpnet, ih_size);
  GET("net", tmpnet, ho_size);
  GET("net", tmpnet, rng);
  GET("net", tmpnet, generation);
  GET("net", tmpnet, flags);
  if (version >= 9) {
    GET("net", tmpnet, presynaptic_noise);
  } else {
    tmpnet.presynaptic_noise = 0;
  }
  if (version >= 10) {
    GET("net", tmpnet, activation);
  } else {
    tmpnet.activation = RNN_RELU;
  }
  if (tmpnet.flags & RNN_NET_FLAG_OWN_BPTT) {
    GET("bptt", tmpbptt, depth);
    GET("bptt", tmpbptt, learn_rate);
    GET("bptt", tmpbptt, index);
    GET("bptt", tmpbptt, momentum);
    GET("bptt", tmpbptt, momentum_weight);
    if (version >= 2) {
      GET("bptt", tmpbptt, ho_scale);
    } else {
      tmpbptt.ho_scale = ((float)tmpnet.output_size) / tmpnet.hidden_size;
    }
    if (version >= 3) {
      GET("bptt", tmpbptt, min_error_factor);
    } else {
      tmpbptt.min_error_factor = BASE_MIN_ERROR_FACTOR * tmpnet.h_size;
    }
  }
  if ((tmpnet.flags & RNN_NET_FLAG_BOTTOM_LAYER) && version >= 4) {
    GET("bottom_layer", tmpbl, learn_rate_scale);
    GET("bottom_layer", tmpbl, input_size);
    GET("bottom_layer", tmpbl, output_size);
    GET("bottom_layer", tmpbl, i_size);
    GET("bottom_layer", tmpbl, o_size);
    GET("bottom_layer", tmpbl, overlap);
  }
  if (tmpnet.flags & RNN_NET_FLAG_BOTTOM_LAYER) {
    net = rnn_new_with_bottom_layer(tmpbl.input_size, tmpbl.output_size, tmpnet.hidden_size,
                                    tmpnet.output_size, tmpnet.flags, 0, NULL, tmpbptt.depth,
                                    tmpbptt.learn_rate, tmpbptt.momentum,
                                    tmpnet.presynaptic_noise, tmpnet.activation, tmpbl.overlap);
  } else {
    net = rnn_new(tmpnet.input_size, tmpnet.hidden_size, tmpnet.output_size, tmpnet.flags, 0,
                  NULL, tmpbptt.depth, tmpbptt.learn_rate, tmpbptt.momentum,
                  tmpnet.presynaptic_noise, tmpnet.activation);
  }
  net->rng = tmpnet.rng;
  net->generation = tmpnet.generation;
  if (net->bptt) {
    RecurNNBPTT *bptt = net->bptt;
    /* the ring position is restored by stepping, so input_layer/real_inputs
     * point at the right slot (the reference leaves them at slot 1) */
    bptt->index = tmpbptt.index;
    net->input_layer = bptt->history + (size_t)bptt->index * net->i_size;
    net->real_inputs = net->input_layer + net->hidden_size + 1;
    bptt->momentum_weight = tmpbptt.momentum_weight;
    bptt->ho_scale = tmpbptt.ho_scale;
    bptt->min_error_factor = tmpbptt.min_error_factor;
  }
#define CHECK(attr)                                                                  \
  do {                                                                               \
    if (net->attr != tmpnet.attr) {                                                  \
      fprintf(stderr, "attribute '%s' differs %f vs %f\n", #attr, (float)net->attr,  \
              (float)tmpnet.attr);                                                   \
      goto error;                                                                    \
    }                                                                                \
  } while (0)
  CHECK(i_size);
  CHECK(h_size);
  CHECK(o_size);
  CHECK(input_size);
  CHECK(hidden_size);
  CHECK(output_size);
  CHECK(ih_size);
  CHECK(ho_size);
  CHECK(flags);
  CHECK(presynaptic_noise);
  CHECK(activation);
#undef CHECK
  if (net->bptt && net->bptt->depth != tmpbptt.depth) {
    fprintf(stderr, "attribute 'depth' differs\n");
    goto error;
  }
  {
    const unsigned char *v;
    uint32_t vlen;
    size_t want = sizeof(float) * (size_t)net->ih_size;
    if (cdbr_find(&r, "net.ih_weights", 14, &v, &vlen) < 1 || vlen != want) {
      fprintf(stderr, "array size mismatch on 'net.ih_weights'\n");
      goto error;
    }
    memcpy(net->ih_weights, v, want);
    want = sizeof(float) * (size_t)net->ho_size;
    if (cdbr_find(&r, "net.ho_weights", 14, &v, &vlen) < 1 || vlen != want) {
      fprintf(stderr, "array size mismatch on 'net.ho_weights'\n");
      goto error;
    }
    memcpy(net->ho_weights, v, want);
    if (version >= 5) {
      int rc = cdbr_find(&r, "net.metadata", 12, &v, &vlen);
      if (rc == 1) {
        if (vlen > MAX_METADATA_SIZE) {
          fprintf(stderr, "size of 'net.metadata'(%u) exceeds maximum %u\n", vlen,
                  MAX_METADATA_SIZE);
          goto error;
        }
        net->metadata = malloc(vlen + 1);
        memcpy(net->metadata, v, vlen);
        net->metadata[vlen] = 0;
      } else {
        fprintf(stderr, "error %d loading 'net.metadata'\ncontinuing anyway\n", rc);
      }
    }
    if (net->bottom_layer) { /* recur-nn-io.c:341-344: the weights only; learn_rate_scale
                              * keeps the constructor's 1.0 as in the reference */
      RecurExtraLayer *bl = net->bottom_layer;
      want = sizeof(float) * (size_t)bl->i_size * bl->o_size;
      if (cdbr_find(&r, "bottom_layer.weights", 20, &v, &vlen) < 1 || vlen != want) {
        fprintf(stderr, "array size mismatch on 'bottom_layer.weights'\n");
        goto error;
      }
      memcpy(bl->weights, v, want);
    }
  }
  ramd_host_wrote(net, RNN_AMD_WEIGHTS | RNN_AMD_STREAM);
  cdbr_close(&r);
  fprintf(stderr, "successfully loaded net '%s'\n", filename);
  return net;
error:
  rnn_delete_net(net);
pre_alloc_error:
  cdbr_close(&r);
open_error:
  fprintf(stderr, "loading net failed!\n");
  return NULL;
}
