/* ref_shim.c -- TEST INFRASTRUCTURE.  Built only where /root/reference exists.
 *
 * The reference keeps its PRNG, softmax and soft-clip helpers as static inline
 * functions in headers (recur-rng.h, badmaths.h, recur-nn-helpers.h), so they
 * are not exported by recur-nn.o.  This file is compiled together with the
 * reference's own recur-nn.c and recur-nn-init.c (see oracle/Makefile) and
 * gives those helpers external names that tests/golden/make_golden.py and
 * tests/test_oracle_vs_ref.py can call through ctypes.  It contains no
 * algorithm of its own.
 */
#include "recur-nn.h"
#include "recur-nn-helpers.h"
#include "badmaths.h"
#include "utf8.h"

u64 ref_rand64(rand_ctx *x) { return rand64(x); }
void ref_init_rand64(rand_ctx *x, u64 seed) { init_rand64(x, seed); }
double ref_rand_double(rand_ctx *x) { return rand_double(x); }
int ref_rand_small_int(rand_ctx *x, int cap) { return rand_small_int(x, cap); }
float ref_cheap_gaussian_noise(rand_ctx *x) { return cheap_gaussian_noise(x); }
float ref_fast_expf(float x) { return fast_expf(x); }
float ref_fast_sigmoid(float x) { return fast_sigmoid(x); }
void ref_softmax(float *dest, const float *src, int len) { softmax(dest, src, len); }
int ref_softmax_best_guess(float *error, const float *src, int len) {
  return softmax_best_guess(error, src, len);
}
float ref_soft_clip(float sum, float halfmax) { return soft_clip(sum, halfmax); }
int ref_sizeof_net(void) { return (int)sizeof(RecurNN); }
int ref_sizeof_bptt(void) { return (int)sizeof(RecurNNBPTT); }
void ref_biased_softmax(float *dest, const float *src, int len, float bias) {
  biased_softmax(dest, src, len, bias);
}
u32 ref_hash32(const char *s) { return rnn_hash32(s); }
int ref_write_utf8_char(uint code, char *s) { return write_utf8_char(code, s); }
int ref_read_utf8_char(const char *s, int *consumed) {
  const char *p = s;
  int c = read_utf8_char(&p);
  *consumed = (int)(p - s);
  return c;
}
