"""auto_rx's experimental decode chains for the rest of the family — `fsk_demod --cs16 -b lo -u hi -s --stats=5 2 48000 <baud> - - | <decoder> --softin ...`
(decode.py:1200-1209 LMS6, :1240-1250 iMet-54, :1283-1293 MRZ, :1369-1379 Meisei) — both halves from this repo (modem on the GPU, native
bit-rate tier) against both halves of the compiled reference, on the same 48 kHz cs16 captures: same stdout."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
BIN = os.path.join(ROOT, "host", "bin")


def _chain(d, modem_args, dec, dec_args, data):
    env = dict(os.environ, SONDE_JSN_VERSION="oracle")
    m = subprocess.run([os.path.join(d, "fsk_demod")] + modem_args, input=data, capture_output=True, timeout=300, env=env)
    assert m.returncode == 0, m.stderr[-300:]
    r = subprocess.run([os.path.join(d, dec)] + dec_args, input=m.stdout, capture_output=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-300:]
    return m.stdout, r.stdout


CASES = [
    ("lms6Xmod", ["--json", "--softin", "--vit2", "-i"], 4800, 10000, dict(fn="lms6_capture", kw=dict(seconds=6.0, noise_sigma=0.05, seed=81)), b'"type": "LMS"', 4),
    ("imet54mod", ["--ecc", "--json", "--softin", "-i", "--ptu"], 4800, 10000, dict(fn="imet54_capture", kw=dict(seconds=6.5, noise_sigma=0.05, seed=82)), b'"type": "IMET5"', 3),
    ("mp3h1mod", ["--auto", "--json", "--softin", "--ptu"], 2400, 10000, dict(fn="mrz_capture", kw=dict(seconds=18.5, noise_sigma=0.05, seed=83)), b"[OK]", 10),
    ("meisei100mod", ["--softin", "--json", "--ptu", "--ecc"], 2400, 15000, dict(fn="meisei_capture", kw=dict(seconds=8.0, noise_sigma=0.05, seed=84)), b'"type": "MEISEI"', 4),
]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "fsk_demod")), reason="compiled reference not present")
@pytest.mark.parametrize("dec,dec_args,baud,lim,cap,needle,least", CASES, ids=[c[0] for c in CASES])
def test_modem_to_decoder_chain_matches_reference(dec, dec_args, baud, lim, cap, needle, least):
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    x = getattr(synth, cap["fn"])(sr=48_000, **cap["kw"])
    modem = ["--cs16", "-b", str(-lim), "-u", str(lim), "-s", "--stats=5", "2", "48000", str(baud), "-", "-"]
    soft_a, out_a = _chain(BIN, modem, dec, dec_args, x.tobytes())
    soft_b, out_b = _chain(REF, modem, dec, dec_args, x.tobytes())
    assert len(soft_a) == len(soft_b)
    assert out_a == out_b, (out_a[:600], out_b[:600])
    assert out_a.count(needle) >= least, out_a[:800]
