"""Shared access to tests/golden (fixtures generated from the compiled reference by tools/make_golden.py)."""
import functools, json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden  # noqa: E402  (only CASES + capture(); does not touch oracle/_ref on import)

GOLDEN = os.path.join(ROOT, "tests", "golden")
NAMES = list(make_golden.CASES)


@functools.lru_cache(maxsize=None)
def load(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    g["consts"] = json.loads(str(g["consts"]))
    g["lines"] = [str(s) for s in g["lines"]]
    return g


@functools.lru_cache(maxsize=4)
def capture(name):
    x, fq = make_golden.capture(make_golden.CASES[name])
    return x, fq, make_golden.CASES[name]["sr"]


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64))))) if np.size(a) else 0.0


DFM_NAMES = list(make_golden.DFM_CASES)


@functools.lru_cache(maxsize=4)
def dfm_capture(name):
    x, fq, ecc = make_golden.dfm_capture(make_golden.DFM_CASES[name])
    return x, fq, make_golden.DFM_CASES[name]["sr"], ecc


SCAN_NAMES = list(make_golden.SCAN_CASES)


@functools.lru_cache(maxsize=None)
def load_scan(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    g["consts"] = json.loads(str(g["consts"]))
    g["stdout"] = str(g["stdout"])
    g["rc"] = int(g["rc"])
    return g


@functools.lru_cache(maxsize=4)
def scan_capture(name):
    """-> (int16 samples, fq, CLI stdin bytes, case dict)"""
    case = make_golden.SCAN_CASES[name]
    x, fq, stdin = make_golden.scan_capture(case)
    return x, fq, stdin, case


FSK_NAMES = list(make_golden.FSK_CASES)


@functools.lru_cache(maxsize=None)
def load_fsk(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    g["consts"] = json.loads(str(g["consts"]))
    if "rs41_lines" in g:
        g["rs41_lines"] = [str(s) for s in g["rs41_lines"]]
    return g


@functools.lru_cache(maxsize=4)
def fsk_capture(name):
    case = make_golden.FSK_CASES[name]
    return make_golden.fsk_capture(case), case


AUDIO_NAMES = list(make_golden.AUDIO_CASES)


@functools.lru_cache(maxsize=4)
def audio_capture(name):
    case = make_golden.AUDIO_CASES[name]
    pcm, wav = make_golden.audio_capture(case)
    return pcm, wav, case


IQDEC_NAMES = list(make_golden.IQDEC_CASES)


@functools.lru_cache(maxsize=None)
def load_iqdec(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    g["stderr"] = str(g["stderr"])
    return g


def need_ref() -> bool:
    """True when the compiled reference (oracle/_ref) is there.  When it is not: fail, unless SONDE_ALLOW_NO_REF=1 says the run knowingly does
    without (then False: the caller leaves that comparison out).  Replaces silent `if have_ref():` gates inside tests."""
    import os
    import pytest
    from oracle import bind
    if bind.have_ref():
        return True
    if os.environ.get("SONDE_ALLOW_NO_REF") == "1":
        return False
    pytest.fail("oracle/_ref (compiled reference) is missing: build it where /root/reference exists (make -C oracle ref) or set SONDE_ALLOW_NO_REF=1")
