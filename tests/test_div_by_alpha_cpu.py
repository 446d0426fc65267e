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


def _compile(tmp_path, src, name):
    c = tmp_path / (name + ".c")
    c.write_text(src)
    exe = tmp_path / name
    r = subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", str(c), "-o", str(exe),
                        "-lm"], capture_output=True, text=True)
    if r.returncode != 0:     # (no libgomp on this machine, say: nothing to check with)
        pytest.skip("cannot build the checker: " + r.stderr[-300:])
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_double_product_with_the_reciprocal_is_the_fp32_quotient_for_every_float(tmp_path):
    """2 x 10^10 divisions: minutes on a machine with few cores (the time limit is the test's own)"""
    exe = _compile(tmp_path, SRC, "t")
    divisors = ["0.05", "10", "50000", "16667", "16777215"]
    out = subprocess.run([str(exe)] + divisors, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == len(divisors) and all(": 0 (" in ln for ln in lines), out.stdout
    # the guard is there for a reason: the bare product mis-rounds subnormal quotients of 50000
    assert not lines[2].endswith("(unguarded 0)"), out.stdout
    # ... and the reference's alpha is a divisor that does not need it: what the table's own
    # check on the GPU (xf_table.hip: div_exact_for) finds for it, every x, before it lets the
    # step drop the guard
    assert lines[0].endswith("(unguarded 0)"), out.stdout


RANDOM_SRC = SRC.split("int main")[0] + r"""
static uint64_t s = 0x9e3779b97f4a7c15ull;
static uint32_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
int main(void) {   /* alpha is the user's to set: random divisors, random and subnormal-range x */
  unsigned long long bad = 0;
  for (int di = 0; di < 20000; ++di) {
    uint32_t ud = rnd() & 0x7FFFFFFFu;
    float d;
    memcpy(&d, &ud, 4);
    if (!isfinite(d) || d == 0.0f) continue;
    const double inv = 1.0 / (double)d;
    for (int xi = 0; xi < 2000; ++xi) {
      uint32_t ux = rnd();
      if (xi & 1) ux = (ux & 0x80FFFFFFu) | ((rnd() % 40) << 23);   /* tiny x: subnormal quotients */
      float x;
      memcpy(&x, &ux, 4);
      if (!isfinite(x)) continue;
      volatile float q1 = x / d;
      float t = q1, q2 = div_by_const(x, d, inv);
      uint32_t a, b;
      memcpy(&a, &t, 4);
      memcpy(&b, &q2, 4);
      if (isnan(t) && isnan(q2)) continue;
      bad += a != b;
    }
  }
  printf("%llu\n", bad);
  return bad != 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_guarded_product_is_the_quotient_for_random_divisors(tmp_path):
    """the general claim behind div_by_const (alpha is user-settable): 4 x 10^7 random (x, d)
    pairs, half of them with x in the range whose quotients are subnormal or next to it"""
    exe = _compile(tmp_path, RANDOM_SRC, "r")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout + out.stderr
