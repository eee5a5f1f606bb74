"""The device's noise value (k_common.h: dev_cheap_gaussian_pair) replaces the reference's division
(recur-rng.h:179-201: (float)(sum - 6 * 0xffff) / 0xffff) by x * RN(1 / 65535) with one fma correction of the
remainder.  That is the same float for every possible sum of twelve 16-bit fields -- checked here for all 786,421
of them with the host's correctly rounded fmaf (the device's v_fma_f32 / v_mul_f32 are IEEE operations too)."""
import os
import subprocess
import tempfile

SRC = r"""
#include <math.h>
#include <stdio.h>
int main(void) {
  const float b = 65535.0f;
  const float rc = 1.0f / b;
  long bad = 0;
  for (int a = 0; a <= 12 * 65535; a++) {
    float x = (float)(a - 0xffff * 6);
    float q0 = x * rc;
    float r = fmaf(-q0, b, x);
    float q1 = fmaf(r, rc, q0);
    if (q1 != x / b) bad++;
  }
  printf("%ld\n", bad);
  return 0;
}
"""


def test_corrected_reciprocal_multiply_equals_the_division_for_every_noise_value():
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "d.c"), os.path.join(d, "d")
        with open(c, "w") as f:
            f.write(SRC)
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, c, "-lm"], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.strip()
    assert out == "0"
