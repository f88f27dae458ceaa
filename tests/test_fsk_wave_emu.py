"""The wave form of the 2-/4-FSK modem (radiosonde_auto_rx_amd/csrc/sonde_fsk_wave.h — the source hipcc compiles into k_fsk_wave) executed on the
CPU under tests/emu/wave_emu.h — every thread a fiber, every cross-lane operation and barrier a rendezvous — against the recordings of the
compiled reference modem (tests/golden/fsk_*.npz, utils/fsk.c driven the way utils/fsk_demod.c drives it).  Checks the whole schedule without a
GPU: the walker / worker hand-over and its snapshot protocol, the estimate made ahead, pieces cut by frame ends, launches that end inside a
frame, the state carried from launch to launch.  Same tolerances as tests/test_gpu_fsk.py (nin, tone estimates, hard decisions, Sf exact;
soft decisions 1e-6 of their RMS); the device's own instruction sequence for the oscillator is covered there."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from golden_cases import load_fsk, fsk_capture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emu", "fsk_wave_emu.cpp")
SO = os.path.join(ROOT, "tests", "emu", "libfsk_wave_emu.so")
CSRC = os.path.join(ROOT, "radiosonde_auto_rx_amd", "csrc")
DEPS = [SRC, os.path.join(ROOT, "tests", "emu", "wave_emu.h")] + [os.path.join(CSRC, n) for n in ("sonde_fsk_wave.h", "sonde_fsk_dev.h", "sonde_fsk_tables.h")]


class Rec(C.Structure):
    _fields_ = [("nin", C.c_int), ("nin_next", C.c_int), ("f_est", C.c_float * 4), ("norm_rx_timing", C.c_float), ("ppm", C.c_float),
                ("EbNodB", C.c_float), ("snr_est", C.c_float)]


def _build(so, defs=()):
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in DEPS):
        tmp = so + ".%d.tmp" % os.getpid()
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas", "-shared", "-fPIC", *defs, "-o", tmp, SRC])
        os.replace(tmp, so)
    L = C.CDLL(so)
    L.emu_fsk_run.argtypes = [C.c_int] * 13 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return L


@pytest.fixture(scope="module")
def emu():
    return _build(SO)


@pytest.fixture(scope="module")
def emu_wrong_guess():
    """the same source with the estimator's guess of the next frame's start made wrong for every nominal-length frame (SPEC_TEST_WRONG): every
    speculative estimate is thrown away, Sf restored and the estimate redone — results must not move"""
    return _build(SO.replace(".so", "_wrong.so"), ("-DSPEC_TEST_WRONG=1",))


def run(emu, name, frames, chunk, split, M=2, fin=None):
    g = load_fsk(name)
    x, case = fsk_capture(name)
    sr = case["cap"]["sr"]
    per = 1 if case["fmt"] == 1 else 2
    need = int(g["nin"][:frames].sum())
    x = np.ascontiguousarray(x[:per * need])
    lower = (-sr // 2 if case["fmt"] != 1 else 0) if case["lower"] is None else case["lower"]
    upper = sr // 2 if case["upper"] is None else case["upper"]
    nsym = case["nsym"]
    if fin is None:
        fin = 1 if 2 * M * (nsym + 1) * case["P"] * 8 <= 16384 else 0      # the launcher's rule (sonde_launch_fsk)
    sd = np.zeros(frames * nsym + 16, np.float32)
    recs = (Rec * (frames + 4))()
    Sf = np.zeros(g["consts"]["Ndft"], np.float32)
    ns = C.c_longlong(0)
    n = emu.emu_fsk_run(sr, case["Rs"], M, case["P"], nsym, case["fmt"], lower, upper, 1 if case["mask"] else 0, case["mask"] or 100, 0, 1 if split else 0, fin,
                        x.ctypes.data, need, chunk, sd.ctypes.data, len(sd), C.addressof(recs), len(recs), Sf.ctypes.data, C.byref(ns))
    return g, n, sd, recs, Sf, int(ns.value)


def check(g, n, sd, recs, frames, nsym):
    assert n == frames
    assert [recs[i].nin for i in range(n)] == g["nin"][:n].tolist()
    assert [recs[i].nin_next for i in range(n)] == g["nin_next"][:n].tolist()
    assert np.array_equal(np.array([list(recs[i].f_est)[:2] for i in range(n)], np.float32), g["f_est"][:n])
    ref = g["sd"][:n].ravel()
    got = sd[:n * nsym]
    rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    d = got.astype(np.float64) - ref
    assert np.sqrt(np.mean(d ** 2)) < 1e-6 * rms and np.abs(d).max() < 1e-5 * rms
    assert np.array_equal(got < 0, ref < 0)
    assert np.abs(np.array([recs[i].norm_rx_timing for i in range(n)]) - g["norm_rx_timing"][:n]).max() < 2e-7
    assert np.abs(np.array([recs[i].ppm for i in range(n)]) - g["ppm"][:n]).max() < 1e-3
    assert np.abs(np.array([recs[i].EbNodB for i in range(n)]) - g["EbNodB"][:n]).max() < 5e-3
    assert np.abs(np.array([recs[i].snr_est for i in range(n)]) - g["snr_est"][:n]).max() < 5e-3


@pytest.mark.parametrize("rot", [2, 3, 4])
def test_wave_modem_with_the_roles_on_other_wavefronts(emu, monkeypatch, rot):
    """role_rot: the launcher turns the order of worker / walker / estimator / finisher by the channel number (SIMD balance); results must not depend on it"""
    monkeypatch.setenv("EMU_ROLE_ROT", str(rot))
    for name, frames in (("fsk_dfm_50k", 12), ("fsk_m10_48080", 40), ("fsk_rs41_48k_mask", 3)):
        _, case = fsk_capture(name)
        g, n, sd, recs, Sf, ns = run(emu, name, frames, case["cap"]["sr"], True)
        check(g, n, sd, recs, frames, case["nsym"])


def test_wave_modem_with_every_guessed_frame_start_wrong(emu_wrong_guess):
    for name, frames in (("fsk_dfm_50k", 40), ("fsk_m10_48080", 120), ("fsk_rs41_48k_peak", 40)):
        _, case = fsk_capture(name)
        g, n, sd, recs, Sf, ns = run(emu_wrong_guess, name, frames, case["cap"]["sr"], True)
        check(g, n, sd, recs, frames, case["nsym"])


CASES = [("fsk_rs41_48k_mask", 5), ("fsk_dfm_50k", 12), ("fsk_m10_48080", 40), ("fsk_rs41_48k_cu8", 4), ("fsk_rs41_48k_real", 12), ("fsk_rs41_48k_peak", 12)]


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("name,frames", CASES)
def test_wave_modem_matches_reference(emu, name, frames, split):
    """one launch per second of signal (the bench's call pattern)"""
    g0 = load_fsk(name)
    frames = min(frames, len(g0["nin"]))
    _, case = fsk_capture(name)
    g, n, sd, recs, Sf, ns = run(emu, name, frames, case["cap"]["sr"], split)
    check(g, n, sd, recs, frames, case["nsym"])
    assert ns == int(g["nin"][:frames].sum())
    if frames == len(g["nin"]):
        assert np.array_equal(Sf, g["Sf"])


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("name,frames,chunk", [("fsk_rs41_48k_mask", 5, 7001), ("fsk_m10_48080", 40, 333), ("fsk_dfm_50k", 12, 1010), ("fsk_m10_48080", 30, 97)])
def test_wave_modem_launches_that_end_inside_frames(emu, name, frames, chunk, split):
    """calls shorter than a frame, calls that end inside one: the oscillators, the f_dc tail, Sf and the pending length go from launch to launch"""
    _, case = fsk_capture(name)
    g, n, sd, recs, Sf, ns = run(emu, name, frames, chunk, split)
    check(g, n, sd, recs, frames, case["nsym"])


def test_wave_modem_spectrum_is_the_references_bit_for_bit(emu):
    """all frames of a short capture: the smoothed spectrum behind the last one equals the reference's (kiss_fft's own butterfly order, four elements per lane)"""
    name = "fsk_dfm_50k"
    g0 = load_fsk(name)
    frames = len(g0["nin"])
    _, case = fsk_capture(name)
    g, n, sd, recs, Sf, ns = run(emu, name, frames, case["cap"]["sr"], True)
    check(g, n, sd, recs, frames, case["nsym"])
    assert np.array_equal(Sf, g["Sf"])


@pytest.mark.parametrize("name,frames,fin", [("fsk_dfm_50k", 12, 0), ("fsk_m10_48080", 40, 0), ("fsk_rs41_48k_real", 12, 0), ("fsk_rs41_48k_mask", 3, 1)])
def test_wave_modem_with_the_finisher_forced_on_or_off(emu, name, frames, fin):
    """the launcher turns the finisher wave on where f_int fits into LDS twice (short frames); either way the worker / finisher split must not show in the results"""
    _, case = fsk_capture(name)
    g, n, sd, recs, Sf, ns = run(emu, name, frames, case["cap"]["sr"], True, fin=fin)
    check(g, n, sd, recs, frames, case["nsym"])


@pytest.mark.parametrize("split", [False, True])
def test_wave_modem_4fsk_matches_the_compiled_reference(emu, split):
    """M = 4 (fsk.c: four tone estimates, four down-converters / integrators, two soft decisions per symbol :793-802) through the same source, against
    `oracle/_ref/fsk_demod --cs16 -p 5 -s 4 48000 2400` run here on the same synthetic four-tone signal"""
    from golden_cases import need_ref
    need_ref()
    import sys
    sys.path.insert(0, ROOT)
    from tools import synth
    rng = np.random.default_rng(4)
    bits = rng.integers(0, 2, 2 * 50 * 60)
    x = synth.mfsk_capture(bits, 48000, 2400, 4, f_low=-3600.0, shift=2400.0, noise_sigma=0.12, seed=9)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "fsk_demod"), "--cs16", "-p", "5", "-s", "4", "48000", "2400", "-", "-"], input=x.tobytes(), capture_output=True, timeout=120)
    ref = np.frombuffer(r.stdout, np.float32)
    nsym, M = 50, 4
    frames = len(ref) // (nsym * 2)
    assert frames >= 20
    n = len(x) // 2
    sd = np.zeros(frames * nsym * 2 + 64, np.float32)
    recs = (Rec * (frames + 8))()
    Sf = np.zeros(1024, np.float32)
    ns = C.c_longlong(0)
    fin = 1 if 2 * M * (nsym + 1) * 5 * 8 <= 16384 else 0
    got_frames = emu.emu_fsk_run(48000, 2400, M, 5, nsym, 2, -24000, 24000, 0, 100, 0, 1 if split else 0, fin,
                                 np.ascontiguousarray(x).ctypes.data, n, 48000, sd.ctypes.data, len(sd), C.addressof(recs), len(recs), Sf.ctypes.data, C.byref(ns))
    assert got_frames == frames
    got = sd[:frames * nsym * 2]
    rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    d = got.astype(np.float64) - ref[:len(got)]
    assert np.sqrt(np.mean(d ** 2)) < 1e-6 * rms and np.array_equal(got < 0, ref[:len(got)] < 0)
