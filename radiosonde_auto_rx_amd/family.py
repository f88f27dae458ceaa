"""The rest of the decoder family behind the engine's generic sonde description: descriptors (what the reference's decoders put into dsp_t and pass to
find_header / read_softbit*) and ctypes bindings of their bit-rate tiers (include/sonde_lms6.h, sonde_meisei.h, sonde_imet54.h, sonde_mrz.h,
sonde_mts01.h, sonde_rs92.h).  Host-side; `FamilyDecoder.hit(h)` takes one dict of Engine(sonde="generic").fetch_hits() and returns the characters the
reference decoder prints for that header hit; `.json_objects(text)` picks the JSON lines out of them.

The scanner's type names (dft_detect.c:172-191) are the keys."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from .engine import SondeError, lib

_I = C.c_int32


def _opts(names):
    return [(n, _I) for n in names] + [("version", C.c_char * 32), ("reserved", _I * 4)]


class Lms6Opts(C.Structure):
    _fields_ = [(n, _I) for n in ("raw", "ecc", "vit", "json", "typ", "gpsweek", "jsn_freq_khz")] + [("version", C.c_char * 32), ("reserved", _I * 4)]


class MeiseiOpts(C.Structure):
    _fields_ = _opts(("raw", "verbose", "dbg", "ecc", "json", "ptu", "ims100", "ref_year", "jsn_freq_khz"))


class Imet54Opts(C.Structure):
    _fields_ = _opts(("raw", "verbose", "ecc", "ptu", "silent", "json", "inv", "aut", "jsn_freq_khz"))


class MrzOpts(C.Structure):
    _fields_ = [(n, _I) for n in ("raw", "verbose", "dbg", "ptu", "uniq", "color", "json", "inv", "aut", "bits_ofs", "jsn_freq_khz")] + \
               [("version", C.c_char * 32), ("bits_ofs_given", _I), ("reserved", _I * 3)]


class Mts01Opts(C.Structure):
    _fields_ = _opts(("raw", "verbose", "json", "jsn_freq_khz"))


class Rs92Opts(C.Structure):
    _fields_ = [(n, _I) for n in ("raw", "verbose", "aux", "ecc", "ptu", "inv", "ngp", "dbg", "json", "gps_verbose", "gps_iter", "gps_vel", "exsat", "gpsepoch")] + \
               [("dop_limit", C.c_float), ("d_err", C.c_float), ("jsn_freq_khz", _I), ("version", C.c_char * 32), ("reserved", _I * 4)]


# RS92 sends raw GPS ranges: its decoder needs orbit data (auto_rx downloads a RINEX navigation file per flight, decode.py:423-446)
RS92_ORBITS = {"ephemeris": None, "almanac": None}


def set_rs92_orbits(ephemeris: str | None = None, almanac: str | None = None):
    RS92_ORBITS.update(ephemeris=ephemeris, almanac=almanac)


# generic = sonde_generic_t; thres / keep_soft / auto = engine configuration; pol: what the decoder wants — "raw" (un-flip the engine's bits when the
# header score is negative), "engine" (bits in the polarity in effect, as stored)
FAMILY = {
    "LMS6": dict(generic=dict(header="0101011000001000" "0001110010010111" "0001101010100111" "0011110100111110", baud=4800.0, bt=1.2, h=0.9, symlen=1, symhd=1,
                              hdmax=10, bitofs=0, nbits=261 * 16 - 80, l_win=-1.0, lpiq_bw=16000, lpfm_bw=6000),      # lms6Xmod.c:100,1283-1305,1358
                 thres=0.65, auto=True, pol="raw", prefix="lms6", opts=Lms6Opts, kw=dict(ecc=1, vit=2, json=1), sep_hz=8000.0),
    "MEISEI": dict(generic=dict(header="101010101011010100101011001101001100101011001101", baud=2400.0, bt=1.2, h=2.4, symlen=1, symhd=1,
                                hdmax=1, bitofs=0, nbits=1152, l_win=-1.0, lpiq_bw=16000, lpfm_bw=4000),               # meisei100mod.c:200,626-644,690
                   thres=0.7, auto=True, pol="engine", prefix="meisei", opts=MeiseiOpts, kw=dict(ecc=1, json=1, ptu=1), sep_hz=12000.0),
    "IMET5": dict(generic=dict(header="0000000001" "0101010101" "0001001001" "0001001001", baud=4798.0, bt=1.0, h=0.8, symlen=1, symhd=1,
                               hdmax=4, bitofs=1, nbits=2200, l_win=2.0, lpiq_bw=7400, lpfm_bw=6000),                  # imet54mod.c:91-98,945-962,1013
                  thres=0.7, auto=False, pol="engine", prefix="imet54", opts=Imet54Opts, kw=dict(ecc=1, json=1, ptu=1), sep_hz=8000.0),
    "MRZ": dict(generic=dict(header="100110011001100110011001100110011001" "10101010", baud=2399.0, bt=1.0, h=2.0, symlen=2, symhd=2,
                             hdmax=2, bitofs=2, nbits=386, l_win=2.0, lpiq_bw=9000, lpfm_bw=6000),                     # mp3h1mod.c:117,1112-1131,1181
                thres=0.76, auto=False, pol="engine", prefix="mrz", opts=MrzOpts, kw=dict(json=1, ptu=1, uniq=1), sep_hz=10000.0),
    "RS92": dict(generic=dict(header="10100110011001101001" "1010011001100110100110101010100110101001", baud=4800.0, bt=0.5, h=0.8, symlen=2, symhd=2,
                              hdmax=3, bitofs=2, nbits=2340, l_win=4.0, lpiq_bw=8000, lpfm_bw=6000),                   # rs92mod.c:88-92,1914-1943,1992
                 thres=0.7, auto=False, pol="engine", prefix="rs92", opts=Rs92Opts, kw=dict(verbose=1, aux=1, ecc=2, gps_vel=4, json=1, gpsepoch=-1), sep_hz=8000.0),
    "MTS01": dict(generic=dict(header="10101010" "10101010" "10110100" "00101011", baud=1200.0, bt=1.5, h=0.9, symlen=1, symhd=1,
                               hdmax=2, bitofs=0, nbits=1048, l_win=2.0, lpiq_bw=4000, lpfm_bw=4000),                  # mts01mod.c:47-48,514-533,575
                  thres=0.76, auto=True, pol="raw", prefix="mts01", opts=Mts01Opts, kw=dict(json=1), sep_hz=6000.0),
}


# An LMS6 whose blocks turn out to be LMS-X (lms6Xmod.c:1436-1462): same filters and header, bit clock 4797.8 Bd and 4720 bits per block.  Not a scanner type:
# the receivers move a sonde to an engine of this description when its decoder object reports the change (FamilyDecoder.lms_type), and back.
FAMILY["LMSX"] = dict(FAMILY["LMS6"], generic=dict(FAMILY["LMS6"]["generic"], slice_baud=4797.8, nbits=300 * 16 - 80))
LMS_BASE = {"LMSX": "LMS6"}                      # receiver bookkeeping: the scanner's name of a sonde's type


class FamilyDecoder:
    """one decoder object of type `typ` (a key of FAMILY): state lives across hits like the reference's gpx_t"""

    def __init__(self, typ: str, *, freq_khz: int = 0, version: str = "sonde_hip", **kw):
        self.typ, self.f = typ, FAMILY[typ]
        L = lib()
        p = self.f["prefix"]
        self._create, self._destroy = getattr(L, f"sonde_{p}_dec_create"), getattr(L, f"sonde_{p}_dec_destroy")
        self._create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        self._destroy.argtypes = [C.c_void_p]
        okw = {k: v for k, v in kw.items() if k not in ("ephemeris", "almanac")}
        o = self.f["opts"](**{**self.f["kw"], **okw}, jsn_freq_khz=freq_khz, version=version.encode())
        self._h = C.c_void_p()
        if self._create(C.byref(o), C.byref(self._h)) < 0:
            raise SondeError(f"sonde_{p}_dec_create: unsupported options")
        self._buf = C.create_string_buffer(1 << 16)
        if typ == "LMS6":
            self._fn = L.sonde_lms6_dec_block
            self._fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _I, C.c_float, C.c_float, C.c_double, C.c_char_p, C.c_size_t]
            L.sonde_lms6_dec_block_bits.argtypes = [C.c_void_p]
            self._last_pos = 0
        else:
            self._fn = getattr(L, f"sonde_{p}_dec_frame")
            self._fn.argtypes = [C.c_void_p, C.c_void_p, _I, C.c_char_p, C.c_size_t]
        if typ == "MRZ":
            L.sonde_mrz_dec_frame_bits.argtypes = [C.c_void_p]
        if typ == "RS92":                                              # without orbit data: frames and JSON-less text only, as `rs92mod` without -e / -a
            for kind in ("almanac", "ephemeris"):
                path = kw.get(kind) or RS92_ORBITS[kind]
                fn = getattr(L, f"sonde_rs92_dec_load_{kind}")
                fn.argtypes = [C.c_void_p, C.c_char_p]
                if path and fn(self._h, os.fsencode(path)) < 0:
                    raise SondeError(f"RS92: {kind} file {path} not readable as such")

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def hit(self, h: dict, if_sr: int = 48000) -> str:
        soft = np.ascontiguousarray(h["soft"], np.float32)
        if self.f["pol"] == "raw" and h["mv"] < 0:
            soft = -soft
        n = len(soft)
        if self.typ == "LMS6":
            n = min(n, lib().sonde_lms6_dec_block_bits(self._h))           # a decoder that went over to LMS-X wants more bits than a fixed description slices
            d = (int(h["mv_pos"]) - self._last_pos) & 0xFFFFFFFF
            rate = 4800.0 * if_sr / d if d else float("inf")
            if getattr(self, "moved", False):                              # first block on another engine (receivers, LMS6 <-> LMS-X): no header position to take the
                rate, self.moved = 4800.0, False                          # frame rate from; a rate outside 4000..5000 would send the decoder back (lms6Xmod.c:959)
            self._last_pos = int(h["mv_pos"])
            k = self._fn(self._h, soft.ctypes.data, None, n, h["mv"], rate, (h["mv_pos"] + n * if_sr / 4800.0) / if_sr, self._buf, len(self._buf))
        else:
            if self.typ == "MRZ":
                n = min(n, lib().sonde_mrz_dec_frame_bits(self._h))
            k = self._fn(self._h, soft.ctypes.data, n, self._buf, len(self._buf))
        if k < 0:
            raise SondeError(f"{self.typ}: decoder call failed ({k})")
        return self._buf.raw[:k].decode(errors="replace")

    def lms_type(self):
        """LMS6 only: ("LMS6" | "LMSX" = the description the decoder wants its demodulator on, whether the last block made it change)"""
        ch = _I(0)
        L = lib()
        L.sonde_lms6_dec_type.argtypes = [C.c_void_p, C.POINTER(_I)]
        t = L.sonde_lms6_dec_type(self._h, C.byref(ch))
        return ("LMSX" if (t & 0xFF) == 10 else "LMS6"), bool(ch.value)

    @staticmethod
    def json_objects(text: str):
        return [json.loads(l) for l in text.splitlines() if l.startswith("{")]
