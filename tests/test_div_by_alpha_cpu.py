"""ftrl_step divides by alpha twice per step (ftrl.h:63,70).  The kernels compute x / alpha as
(float)((double)x * (1.0 / (double)alpha)) (xf_device.h: div_by_alpha) — three instructions
instead of an IEEE fp32 division's eleven in a VALU-bound step.  This must be the SAME float for
every x: checked here exhaustively, all 2^32 bit patterns of x, for the reference's default alpha
and two others, on the host (IEEE double multiply and double -> float rounding are what the GPU's
v_mul_f64 / v_cvt_f32_f64 do; the GPU side is covered by the bit-exact parity tests)."""
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
int main(int argc, char **argv) {
  unsigned long long bad_total = 0;
  for (int ai = 1; ai < argc; ++ai) {
    const float alpha = strtof(argv[ai], NULL);
    const double inv = 1.0 / (double)alpha;
    unsigned long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long i = 0; i < (1LL << 32); ++i) {
      uint32_t u = (uint32_t)i, a, b;
      float x;
      memcpy(&x, &u, 4);
      if (!isfinite(x)) continue;
      volatile float q1 = x / alpha;
      float t = q1, q2 = (float)((double)x * inv);
      memcpy(&a, &t, 4);
      memcpy(&b, &q2, 4);
      bad += a != b;
    }
    printf("alpha %.9g: %llu\n", alpha, bad);
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
    out = subprocess.run([str(exe), "0.05", "0.3", "1.9999999"], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(": 0\n") == 3, out.stdout
