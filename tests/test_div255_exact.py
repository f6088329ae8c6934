"""The trilinear filter's division by 255 (pt_volume.hpp: div255) is the product with the rounded reciprocal, corrected
once through the exact remainder.  tests/cpp/div255_exact.c compares it with the IEEE division for every float in
[0, 255]; the two must agree bit for bit, or frames drawn with the filter would differ from the oracle's."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reciprocal_with_one_correction_is_the_division_for_every_float_up_to_255():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "div255_exact")
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "cpp", "div255_exact.c"), "-lm", "-lpthread"],
                       check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=600).stdout
    assert out.strip().splitlines()[-1] == "mismatches 0", out
