"""ctypes bindings for the CPU oracle (liboracle.so) and the compiled reference (oracle/_ref).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(HERE, "_ref")

RS41_HDR = b"0000100001101101010100111000100001000100011010010100100000011111"
CONST_NAMES = ("N", "M", "L", "K", "delay", "dectaps", "decM", "lut_len", "lpiq_taps", "lpfm_taps", "if_sr")


def build(ref: bool | None = None) -> None:
    """Compile liboracle.so (always) and oracle/_ref (only where /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    if ref is None:
        ref = os.path.isdir("/root/reference/demod/mod")
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        _lib = C.CDLL(path)
        _lib.ora_rs41_decode.restype = C.c_int
        _lib.ora_streams.restype = C.c_int
    return _lib


def _buf(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def ora_rs41_decode(iq: np.ndarray, sr: int, *, bps: int = 16, iq_mode: int = 5, fq: float = 0.0,
                    lp_iq: bool = True, lp_fm: bool = False, afc: bool = False, ecc: int = 2,
                    thres: float = 0.7, max_frames: int = 64, want_soft: bool = True):
    """Oracle equivalent of `rs41mod -r --ecc<ecc> --IQ fq [--lpIQ] [--dc] - sr bps`."""
    raw = np.ascontiguousarray(iq)
    frames = np.zeros((max_frames, 518), np.uint8)
    flen = np.zeros(max_frames, np.int32)
    ec = np.zeros(max_frames, np.int32)
    meta = np.zeros((max_frames, 4), np.float64)
    soft = np.zeros((max_frames, 4080), np.float32) if want_soft else None
    pre = np.zeros((max_frames, 518), np.float32)
    n = lib().ora_rs41_decode(_buf(raw), C.c_size_t(raw.nbytes), sr, bps, iq_mode, C.c_double(fq),
                              (1 if lp_iq else 0) | (2 if lp_fm else 0), int(afc), ecc, C.c_float(thres),
                              max_frames, _buf(frames), _buf(flen), _buf(ec), _buf(meta),
                              _buf(soft) if want_soft else None, _buf(pre))
    if n < 0:
        raise RuntimeError("ora_rs41_decode failed")
    lines = []
    out = C.create_string_buffer(2 * 518 + 32)
    for i in range(n):
        k = lib().ora_rs41_rawline(_buf(frames[i]), int(flen[i]), int(ec[i]), out)
        lines.append(out.raw[:k].decode())
    return dict(n=n, frames=frames[:n], flen=flen[:n], ecc=ec[:n], mv=meta[:n, 0], mv_pos=meta[:n, 1].astype(np.int64),
                s_in_after=meta[:n, 2].astype(np.int64), soft=None if soft is None else soft[:n],
                pre_ecc=pre[:n].astype(np.uint8), lines=lines)


def _streams(fn, cfgargs, iq, max_if, want_iq):
    raw = np.ascontiguousarray(iq)
    iqo = np.zeros((max_if, 2), np.float32) if want_iq else None
    fm = np.zeros(max_if, np.float32)
    bufs = np.zeros(max_if, np.float32)
    consts = np.zeros(11, np.int32)
    n = fn(raw, cfgargs, max_if, iqo, fm, bufs, consts)
    if n < 0:
        raise RuntimeError("streams failed: %d" % n)
    return dict(n=n, iq=None if iqo is None else iqo[:n], fm=fm[:n], bufs=bufs[:n],
                consts=dict(zip(CONST_NAMES, (int(v) for v in consts))))


def ora_streams(iq, sr, *, bps=16, iq_mode=5, fq=0.0, lp_iq=True, lp_fm=False, afc=False, baud=4800.0,
                bt=0.5, h=0.6, lpiq_bw=7400, lpfm_bw=6000, hdr=RS41_HDR, symlen=1, symhd=1,
                max_if=None, want_iq=True):
    if max_if is None:
        max_if = len(iq) // 2 + 16
    mask = (1 if lp_iq else 0) | (2 if lp_fm else 0)

    def call(raw, _, max_if, iqo, fm, bufs, consts):
        return lib().ora_streams(_buf(raw), C.c_size_t(raw.nbytes), sr, bps, iq_mode, C.c_double(-fq), mask,
                                 int(afc), C.c_float(baud), C.c_float(bt), C.c_float(h), lpiq_bw, lpfm_bw,
                                 C.c_char_p(hdr), symlen, symhd, max_if,
                                 None if iqo is None else _buf(iqo), _buf(fm), _buf(bufs), _buf(consts))
    return _streams(call, None, iq, max_if, want_iq)


# ----------------------------------------------------------------------------- compiled reference
class RefCfg(C.Structure):
    _fields_ = [("sr", C.c_int), ("bps", C.c_int), ("opt_iq", C.c_int), ("opt_lp", C.c_int),
                ("opt_dc", C.c_int), ("opt_iqdc", C.c_int), ("opt_min", C.c_int), ("opt_nolut", C.c_int),
                ("xlt_fq", C.c_double), ("baud", C.c_float), ("symlen", C.c_int), ("symhd", C.c_int),
                ("BT", C.c_float), ("h", C.c_float), ("lpIQ_bw", C.c_int), ("lpFM_bw", C.c_int),
                ("hdr", C.c_char_p)]


def have_ref() -> bool:
    return os.path.exists(os.path.join(REFDIR, "libref_demod.so")) and os.path.exists(os.path.join(REFDIR, "rs41mod"))


_reflibs: dict = {}


def reflib(name: str = "libref_demod.so") -> C.CDLL:
    if name not in _reflibs:
        _reflibs[name] = C.CDLL(os.path.join(REFDIR, name))
    return _reflibs[name]


def _refcfg(sr, bps, iq_mode, fq, lp_iq, lp_fm, afc, baud, bt, h, lpiq_bw, lpfm_bw, hdr, symlen, symhd, iqdc=False):
    return RefCfg(sr, bps, iq_mode, (1 if lp_iq else 0) | (2 if lp_fm else 0), int(afc), int(iqdc), 0, 0, -fq, baud,
                  symlen, symhd, bt, h, lpiq_bw, lpfm_bw, hdr)


def ref_streams(iq, sr, *, bps=16, iq_mode=5, fq=0.0, lp_iq=True, lp_fm=False, afc=False, baud=4800.0,
                bt=0.5, h=0.6, lpiq_bw=7400, lpfm_bw=6000, hdr=RS41_HDR, symlen=1, symhd=1,
                max_if=None, want_iq=True, libname="libref_demod.so"):
    if max_if is None:
        max_if = len(iq) // 2 + 16
    cfg = _refcfg(sr, bps, iq_mode, fq, lp_iq, lp_fm, afc, baud, bt, h, lpiq_bw, lpfm_bw, hdr, symlen, symhd)
    L = reflib(libname)
    L.ref_streams.restype = C.c_int

    def call(raw, _, max_if, iqo, fm, bufs, consts):
        return L.ref_streams(C.byref(cfg), _buf(raw), C.c_size_t(raw.nbytes), max_if,
                             None if iqo is None else _buf(iqo), _buf(fm), _buf(bufs), _buf(consts))
    return _streams(call, None, iq, max_if, want_iq)


def ref_softframes(iq, sr, *, bps=16, iq_mode=5, fq=0.0, lp_iq=True, lp_fm=False, afc=False, baud=4800.0,
                   bt=0.5, h=0.6, lpiq_bw=7400, lpfm_bw=6000, hdr=RS41_HDR, symlen=1, symhd=1,
                   thres=0.7, hdmax=4, bitofs=2, l=2.0, nbits=4080, max_hits=64, libname="libref_demod.so", iqdc=False):
    raw = np.ascontiguousarray(iq)
    cfg = _refcfg(sr, bps, iq_mode, fq, lp_iq, lp_fm or (afc and iq_mode == 5), afc, baud, bt, h, lpiq_bw,
                  lpfm_bw, hdr, symlen, symhd, iqdc)
    hits = np.zeros((max_hits, 4), np.float64)
    sb = np.zeros((max_hits, nbits), np.float32)
    sb1 = np.zeros((max_hits, nbits), np.float32)
    consts = np.zeros(11, np.int32)
    L = reflib(libname)
    L.ref_softframes.restype = C.c_int
    n = L.ref_softframes(C.byref(cfg), _buf(raw), C.c_size_t(raw.nbytes), C.c_float(thres), hdmax, bitofs,
                         C.c_float(l), nbits, max_hits, _buf(hits), _buf(sb), _buf(sb1), _buf(consts))
    if n < 0:
        raise RuntimeError("ref_softframes failed")
    return dict(n=n, mv=hits[:n, 0], mv_pos=hits[:n, 1].astype(np.int64), nbits=hits[:n, 2].astype(np.int64),
                s_in_after=hits[:n, 3].astype(np.int64), soft=sb[:n], soft1=sb1[:n],
                consts=dict(zip(CONST_NAMES, (int(v) for v in consts))))


def ora_softframes(iq, sr, *, bps=16, iq_mode=5, fq=0.0, lp_iq=True, lp_fm=False, afc=False, baud=4800.0,
                   bt=0.5, h=0.6, lpiq_bw=7400, lpfm_bw=6000, hdr=RS41_HDR, symlen=1, symhd=1,
                   thres=0.7, hdmax=4, bitofs=2, l=2.0, nbits=4080, max_hits=64, iqdc=False):
    """the CPU restatement with the arguments (and the result layout) of ref_softframes(): any sonde of the family"""
    raw = np.ascontiguousarray(iq)
    cfg = _refcfg(sr, bps, iq_mode, fq, lp_iq, lp_fm or (afc and iq_mode == 5), afc, baud, bt, h, lpiq_bw, lpfm_bw, hdr, symlen, symhd, iqdc)
    hits = np.zeros((max_hits, 4), np.float64)
    sb = np.zeros((max_hits, nbits), np.float32)
    sb1 = np.zeros((max_hits, nbits), np.float32)
    L = lib()
    L.ora_softframes.restype = C.c_int
    n = L.ora_softframes(C.byref(cfg), _buf(raw), C.c_size_t(raw.nbytes), C.c_float(thres), hdmax, bitofs, C.c_float(l), nbits, max_hits, _buf(hits), _buf(sb), _buf(sb1))
    if n < 0:
        raise RuntimeError("ora_softframes failed")
    return dict(n=n, mv=hits[:n, 0], mv_pos=hits[:n, 1].astype(np.int64), nbits=hits[:n, 2].astype(np.int64), s_in_after=hits[:n, 3].astype(np.int64), soft=sb[:n], soft1=sb1[:n])


def ref_run(binary: str, args: list[str], data: bytes | np.ndarray, timeout: float = 120.0):
    """Run a compiled reference binary with `data` on stdin -> (stdout, stderr, returncode)."""
    if isinstance(data, np.ndarray):
        data = data.tobytes()
    r = subprocess.run([os.path.join(REFDIR, binary)] + list(args), input=data, capture_output=True, timeout=timeout)
    return r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace"), r.returncode


DFM_RAWHDR = b"10011010100110010101101001010101"


def ora_dfm_decode(iq, sr, *, bps=16, iq_mode=5, fq=0.0, lp_iq=True, ecc=1, thres=0.65, max_lines=256, max_hits=64,
                   want_soft=True):
    """Oracle equivalent of `dfm09mod -r --ecc[2] --IQ fq [--lpIQ] - sr bps`: text lines + per-hit meta/soft bits."""
    raw = np.ascontiguousarray(iq)
    lines = C.create_string_buffer(max_lines * 96)
    meta = np.zeros((max_hits, 4), np.float64)
    soft = np.zeros((max_hits, 2224), np.float32) if want_soft else None
    nh = C.c_int(0)
    L = lib()
    L.ora_dfm_decode.restype = C.c_int
    n = L.ora_dfm_decode(_buf(raw), C.c_size_t(raw.nbytes), sr, bps, iq_mode, C.c_double(fq), 1 if lp_iq else 0, ecc,
                         C.c_float(thres), max_lines, lines, max_hits, _buf(meta), _buf(soft) if want_soft else None, C.byref(nh))
    if n < 0:
        raise RuntimeError("ora_dfm_decode failed")
    out = [lines.raw[i * 96:(i + 1) * 96].split(b"\0")[0].decode() for i in range(n)]
    h = nh.value
    return dict(n=n, lines=out, nhits=h, mv=meta[:h, 0], mv_pos=meta[:h, 1].astype(np.int64), nbits=meta[:h, 2].astype(np.int64),
                s_in_after=meta[:h, 3].astype(np.int64), soft=None if soft is None else soft[:h])


# ----------------------------------------------------------------------------- scanner (scan/dft_detect.c)
class RefScanCfg(C.Structure):
    _fields_ = [("sr", C.c_int), ("bps", C.c_int), ("opt_iq", C.c_int), ("opt_dc", C.c_int), ("opt_min", C.c_int),
                ("fq", C.c_double), ("bw_khz", C.c_double), ("nch", C.c_int)]


SCAN_TYPES = ("DFM9", "RS41", "RS92", "LMS6", "IMET5", "MK2LMS", "M10", "MEISEI", "RD94RD41", "MRZ", "MTS01",
              "C34C50", "WXR301", "WXRPN9", "IMET1AB", "IMETafsk")


def ref_scan_windows(raw, sr, *, bps=16, iq_mode=5, fq=0.0, dc=False, opt_min=False, bw_khz=0.0, max_win=256,
                     want_fm=0):
    """Per-window score/position/dc of every template from the reference's own getCorrDFT (dft_detect.c:357).

    dft_detect.c keeps its state in file statics, so every call loads a private copy of the harness."""
    import shutil
    import tempfile
    raw = np.ascontiguousarray(raw)
    src = os.path.join(REFDIR, "libref_scan.so")
    with tempfile.NamedTemporaryFile(suffix=".so", delete=False) as t:
        tmp = t.name
    shutil.copyfile(src, tmp)
    try:
        L = C.CDLL(tmp)
        L.ref_scan_windows.restype = C.c_int
        cfg = RefScanCfg(sr, bps, iq_mode, int(dc), int(opt_min), fq, bw_khz, 1)
        mv = np.zeros((max_win, 16), np.float32)
        mpos = np.zeros((max_win, 16), np.uint32)
        mp = np.zeros((max_win, 16), np.int32)
        dcs = np.zeros((max_win, 16), np.float32)
        herrs = np.zeros((max_win, 16), np.int32)
        m10 = np.zeros((max_win, 16), np.uint32)
        pos = np.zeros(max_win, np.uint32)
        consts = np.zeros(32, np.int32)
        fm = np.zeros((4, want_fm), np.float32) if want_fm else None
        n = L.ref_scan_windows(C.byref(cfg), _buf(raw), C.c_size_t(raw.nbytes), max_win, _buf(mv), _buf(mpos), _buf(mp),
                               _buf(dcs), _buf(herrs), _buf(m10), _buf(pos), _buf(consts),
                               _buf(fm) if want_fm else None, want_fm)
    finally:
        os.unlink(tmp)
    if n < 0:
        raise RuntimeError("ref_scan_windows failed: %d" % n)
    names = ("K", "N", "delay", "M", "sr_if", "decM", "lpfm_taps", "lpiq_taps", "ntpl")
    cd = dict(zip(names, (int(v) for v in consts[:9])))
    cd["L"] = [int(v) for v in consts[9:25]]
    return dict(n=n, mv=mv[:n], mpos=mpos[:n], mp=mp[:n], dc=dcs[:n], herrs=herrs[:n], m10=m10[:n], pos=pos[:n],
                consts=cd, fm=fm)


# ----------------------------------------------------------------------------- 2-FSK modem (utils/fsk.c)
class RefFskCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("Fs", "Rs", "P", "nsym", "format", "lower", "upper", "mask", "tone_spacing")]


class RefFskFrame(C.Structure):
    _fields_ = [("nin", C.c_int), ("nin_next", C.c_int), ("f_est", C.c_float * 2), ("norm_rx_timing", C.c_float),
                ("ppm", C.c_float), ("EbNodB", C.c_float), ("snr_est", C.c_float)]


def ref_fsk_run(raw, Fs, Rs, *, P=8, nsym=50, fmt=2, lower=None, upper=None, mask=0, tone_spacing=100, max_frames=4096):
    """Per-frame internals + soft decisions of the reference modem (fsk_demod_sd) on an in-memory capture."""
    raw = np.ascontiguousarray(raw)
    ns = raw.size // (2 if fmt != 1 else 1)
    if lower is None:
        lower = -Fs // 2 if fmt != 1 else 0
    if upper is None:
        upper = Fs // 2
    L = reflib("libref_fsk.so")
    L.ref_fsk_run.restype = C.c_int
    cfg = RefFskCfg(Fs, Rs, P, nsym, fmt, lower, upper, int(bool(mask)), tone_spacing)
    sd = np.zeros(max_frames * nsym, np.float32)
    fr = (RefFskFrame * max_frames)()
    Sf = np.zeros(1024, np.float32)
    consts = np.zeros(4, np.int32)
    n = L.ref_fsk_run(C.byref(cfg), _buf(raw), C.c_size_t(ns), max_frames, _buf(sd), fr, _buf(Sf), _buf(consts))
    if n < 0:
        raise RuntimeError("ref_fsk_run failed")
    keys = ("nin", "nin_next", "norm_rx_timing", "ppm", "EbNodB", "snr_est")
    out = {k: np.array([getattr(fr[i], k) for i in range(n)]) for k in keys}
    out["f_est"] = np.array([[fr[i].f_est[0], fr[i].f_est[1]] for i in range(n)], np.float32).reshape(n, 2)
    Ts, N, Ndft, Nmem = (int(v) for v in consts)
    out.update(n=n, sd=sd[:n * nsym].reshape(n, nsym), Sf=Sf[:Ndft].copy(), consts=dict(Ts=Ts, N=N, Ndft=Ndft, Nmem=Nmem))
    return out
