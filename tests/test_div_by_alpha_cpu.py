"""ftrl_step divides by alpha twice per step (ftrl.h:63,70) and every gradient is a sum divided by
the minibatch's row count (lr_worker.cc:117, fm_worker.cc:150-156).  The kernels compute x / d for
such a call-invariant divisor as the double product with the reciprocal, rounded to float, and do
the division itself only where the result is in or next to the subnormal range (xf_device.h:
div_by_const) — three instructions instead of an IEEE fp32 division's eleven in VALU-bound
kernels.  This must be the SAME float for every x: checked here exhaustively, all 2^32 bit
patterns of x, for the reference's default alpha, divisors with exact midpoints among their
subnormal quotients (50000, the bench's row count: the guard's reason) and other row counts, on the host (IEEE double multiply
and double -> float rounding are what the GPU's v_mul_f64 / v_cvt_f32_f64 do; the GPU side is
covered by the bit-exact parity tests).  Without the guard the unguarded product differs for
d = 50000 at 1308 values of x, all with subnormal quotients (exact rounding midpoints exist
there: x = 25000 * 2^-149): checked too."""
import os
import shutil
import subprocess

import pytest

SRC = r"""
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline float div_by_const(float x, float d, double inv_d) {  /* xf_device.h */
  const float q = (float)((double)x * inv_d);
  if (fabsf(q) >= 0x1p-125f || x == 0.0f) return q;
  return x / d;
}
int main(int argc, char **argv) {
  unsigned long long bad_total = 0;
  for (int ai = 1; ai < argc; ++ai) {
    const float d = strtof(argv[ai], NULL);
    const double inv = 1.0 / (double)d;
    unsigned long long bad = 0, bad_plain = 0;
#pragma omp parallel for reduction(+ : bad, bad_plain) schedule(static)
    for (long long i = 0; i < (1LL << 32); ++i) {
      uint32_t u = (uint32_t)i, a, b, c;
      float x;
      memcpy(&x, &u, 4);
      if (!isfinite(x)) continue;
      volatile float q1 = x / d;
      float t = q1, q2 = div_by_const(x, d, inv), q3 = (float)((double)x * inv);
      memcpy(&a, &t, 4);
      memcpy(&b, &q2, 4);
      memcpy(&c, &q3, 4);
      bad += a != b;
      bad_plain += a != c;
    }
    printf("d %.9g: %llu (unguarded %llu)\n", d, bad, bad_plain);
    bad_total += bad;
  }
  return bad_total != 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_double_product_with_the_reciprocal_is_the_fp32_quotient_for_every_float(tmp_path):
    c = tmp_path / "t.c"
    c.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", str(c), "-o", str(exe), "-lm"],
                   check=True)
    divisors = ["0.05", "10", "50000", "16667", "16777215"]
    out = subprocess.run([str(exe)] + divisors, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == len(divisors) and all(": 0 (" in ln for ln in lines), out.stdout
    # the guard is there for a reason: the bare product mis-rounds subnormal quotients of 50000
    assert not lines[2].endswith("(unguarded 0)"), out.stdout
