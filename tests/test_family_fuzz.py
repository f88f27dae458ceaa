"""A fixed-seed slice of tests/fuzz/fuzz_family.py in the CPU suite: the native bit-rate tiers (RS92 with its position solver among them) against the compiled reference decoders on damaged
soft-bit streams with random option sets (noise, bursts, scaling, zeros, inversion, truncation).  `python tests/fuzz/fuzz_family.py <seed> <n>` runs more."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "lms6Xmod")), reason="compiled reference not present")


@pytest.mark.parametrize("seed", [11, 12])
def test_differential_fuzz_slice(seed):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "radiosonde_auto_rx_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import fuzz_family
    assert fuzz_family.run(seed, 40) == 0
