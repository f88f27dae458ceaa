"""host/seam/demod_mod_hip.c, the parts without GPU work: the reference's own decoders linked against the seam (oracle/_ref/*_seam)
take soft symbols / hard bits on stdin through the seam's f32soft_read() / find_softbinhead() / find_binhead() and must print what the
all-CPU reference binaries print.  (The sample-input side of the seam is tests/test_gpu_seam.py.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def _both(binary, args, stdin):
    seam, ref = os.path.join(REF, binary + "_seam"), os.path.join(REF, binary)
    if not (os.path.exists(seam) and os.path.exists(ref)):
        pytest.skip("oracle/_ref seam binaries not built (needs the reference sources: make -C oracle ref)")
    a = subprocess.run([seam] + args, input=stdin, capture_output=True, timeout=120)
    b = subprocess.run([ref] + args, input=stdin, capture_output=True, timeout=120)
    assert a.returncode == b.returncode == 0 and a.stdout == b.stdout and len(b.stdout) > 1000, (binary, args)


def test_seam_softin_m10_m20():
    _both("m10mod", ["--json", "--ptu", "-vvv", "--softin"], make_golden.m10_field_symbols(make_golden.M10_FIELD_SCENARIOS["m10f_mixed_bad_10"]).tobytes())
    _both("m20mod", ["--json", "--ptu", "-vv", "--softinv"], (-make_golden.m20_field_symbols(make_golden.M20_FIELD_SCENARIOS["m20f_fw6_12"])).tobytes())


def test_seam_softin_and_bin_rs41_dfm():
    soft = make_golden.fields_softbits(make_golden.FIELD_SCENARIOS[sorted(make_golden.FIELD_SCENARIOS)[0]])
    _both("rs41mod", ["--ptu2", "--json", "--jsnsubfrm1", "--softin", "-i"], (-soft).astype("<f4").tobytes())
    _both("rs41mod", ["-r", "--ecc2", "--bin"], (soft > 0).astype(np.uint8).tobytes())
    dsym = make_golden.dfm_field_symbols(dict(kind="09", n=30, sn=18012345))
    _both("dfm09mod", ["-vv", "--ecc", "--json", "--dist", "--auto", "--softin"], dsym.astype("<f4").tobytes())
