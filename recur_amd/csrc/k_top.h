// k_top.h -- the text model's loss on one stream as device functions (k_text_top, kernels_loss.hip; factored out for the
// fused forward + top launch that round 4 built, measured and removed: profiles/NOTES_r04.md section 2).
#pragma once
#include "k_common.h"

// badmaths.h:14-29, kept operation for operation
__device__ __forceinline__ float fast_expf_dev(float x) {
#pragma clang fp contract(off)
  int count = 0;
  while (fabsf(x) > 0.2) {
    x *= 0.125;
    count++;
  }
  float a = ((x + 3) * (x + 3) + 3) / ((x - 3) * (x - 3) + 3);
  while (count) {
    a *= a;
    a *= a;
    a *= a;
    count--;
  }
  return a;
}

// The softmax loss of one stream (charmodel-predict.c:18-27, badmaths.h:71-141) by ONE wave: sout[o_size] the
// outputs (LDS), shid[h_size] the hidden row (LDS; its zeros are counted for the statistics), target the
// stream's next symbol, pad_oe this lane's current o_error value (for the pad columns, which stay what they
// were).  Leaves the exponentials in sex, the error row in serr (LDS) and in `err` (global), and in tstat[0..3]
// the error on the target, its log2 likelihood and "best guess == target" (tstat[3], the zero count: text_count_zeros_wave).
/* the hidden row's zeros, for the statistics (tstat[3]): another wave's work beside the softmax (1.2 us of the one
 * wave's 3.4 when it counted them itself, round 5's stamps) */
__device__ __forceinline__ void text_count_zeros_wave(const RamdShape &s, int lane, const float *shid, float *tstat) {
  int zeros = 0;
  for (int i = lane; i < s.H; i += 64) zeros += (shid[i] == 0.0f);
  for (int off = 32; off > 0; off >>= 1) zeros += __shfl_down(zeros, off, 64);
  if (lane == 0) tstat[3] = (float)zeros; /* exact: h_size < 2^24 */
}
__device__ __forceinline__ void text_softmax_wave(const RamdShape &s, int lane, const float *shid, const float *sout,
                                                  float *sex, float *serr, float *err, int target, float pad_oe,
                                                  float *tstat) {
#pragma clang fp contract(off)
  const int len = s.output_size;
  TT_STAMP(8);
  float lo = sout[0], hi = sout[0];
  for (int i = lane; i < len; i += 64) {
    hi = fmaxf(hi, sout[i]);
    lo = fminf(lo, sout[i]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    lo = fminf(lo, __shfl_xor(lo, off, 64));
  }
  float adj = 0.0f;
  if (hi > 50.0f) adj = 50.0f - hi;
  else if (lo < -60.0f) adj = fminf(-60.0f - lo, 50.0f - hi);
  TT_STAMP(9);
  for (int i = lane; i < len; i += 64) sex[i] = fast_expf_dev(sout[i] + adj);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* one wave: its LDS writes are ordered */
  TT_STAMP(10);
  float sum = 0.0f;
  if (s.O <= 64) { /* the exponentials in order, four float4 reads in flight instead of a read per addition */
    for (int i0 = 0; 4 * i0 < len; i0 += 4) {
      float4 q[4];
#pragma unroll
      for (int i = 0; i < 4; i++) q[i] = *reinterpret_cast<const float4 *>(sex + 4 * (4 * (i0 + i) < s.O ? i0 + i : 0));
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (4 * (i0 + i) + 0 < len) sum += q[i].x;
        if (4 * (i0 + i) + 1 < len) sum += q[i].y;
        if (4 * (i0 + i) + 2 < len) sum += q[i].z;
        if (4 * (i0 + i) + 3 < len) sum += q[i].w;
      }
    }
  } else {
    for (int i = 0; i < len; i++) sum += sex[i];
  }
  TT_STAMP(11);
  float best_e = -1.0f;
  int best_i = 0x7fffffff;
  for (int i = lane; i < s.O; i += 64) {
    float oe;
    if (i < len) {
      float e = sex[i] / sum;
      oe = (i == target) ? -e + 1.0f : -e;
      err[i] = oe;
      if (e > best_e) {
        best_e = e;
        best_i = i;
      }
    } else {
      oe = i < 64 ? pad_oe : err[i]; /* the pad of o_error stays what it was (zero) */
    }
    serr[i] = oe;
  }
  for (int off = 32; off > 0; off >>= 1) {
    float oe = __shfl_xor(best_e, off, 64);
    int oi = __shfl_xor(best_i, off, 64);
    if (oe > best_e || (oe == best_e && oi < best_i)) {
      best_e = oe;
      best_i = oi;
    }
  }
  TT_STAMP(12);
  if (lane == 0) {
    float e = -(sex[target] / sum) + 1.0f;
    float l = 1.0f - e;
    tstat[0] = e;
    tstat[1] = (l < 1e-30f) ? -100.0f : log2f(l);
    tstat[2] = (best_i == target) ? 1.0f : 0.0f;
  }
}
