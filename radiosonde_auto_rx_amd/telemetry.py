"""ctypes bindings of the telemetry tier (include/sonde_rs41.h, sonde_dfm.h, sonde_m10.h, sonde_m20.h): frames -> the reference's text / JSON.

Host-side, no GPU involved.  `Rs41Telemetry.decode(frame_dict)` takes what Engine.fetch_frames() returns and gives back the
characters `rs41mod` prints for that frame; `.json(frame_dict)` parses the JSON object out of them (None if the frame did not
qualify for one — block CRCs of ID, time and position must be good, like the reference)."""
from __future__ import annotations

import ctypes as C
import json

from .engine import SondeDfmFrame, SondeFrame, SondeM10Frame, SondeM20Frame, SondeError, lib


class Rs41Opts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("verbose", "ptu", "dewp", "json", "jsn_subfrm", "silent", "jsn_freq_khz")] + \
               [("version", C.c_char * 32), ("sat", C.c_int32), ("aux", C.c_int32), ("reserved", C.c_int32 * 2)]


class Rs41Telemetry:
    def __init__(self, *, ptu: int = 2, verbose: int = 0, jsn_subfrm: int = 0, freq_khz: int = 0, version: str = "sonde_hip", silent: bool = False):
        L = lib()
        L.sonde_rs41_dec_create.argtypes = [C.POINTER(Rs41Opts), C.POINTER(C.c_void_p)]
        L.sonde_rs41_dec_frame.argtypes = [C.c_void_p, C.POINTER(SondeFrame), C.c_char_p, C.c_size_t]
        L.sonde_rs41_dec_destroy.argtypes = [C.c_void_p]
        o = Rs41Opts(verbose=verbose, ptu=ptu, json=1, jsn_subfrm=jsn_subfrm, silent=int(silent), jsn_freq_khz=freq_khz, version=version.encode())
        self._h = C.c_void_p()
        if L.sonde_rs41_dec_create(C.byref(o), C.byref(self._h)) < 0:
            raise SondeError("sonde_rs41_dec_create: unsupported options")
        self._buf = C.create_string_buffer(16384)

    def close(self):
        if getattr(self, "_h", None):
            lib().sonde_rs41_dec_destroy(self._h)
            self._h = None

    __del__ = close

    def decode(self, frame: dict) -> str:
        f = SondeFrame(channel=frame["channel"], len=frame["len"], ecc=frame["ecc"], mv_pos=frame["mv_pos"], mv=frame["mv"], nbytes=frame["nbytes"])
        raw = bytes(frame["frame"])
        C.memmove(f.frame, raw + bytes(518 - len(raw)), 518)
        n = lib().sonde_rs41_dec_frame(self._h, C.byref(f), self._buf, len(self._buf))
        if n < 0:
            raise SondeError("sonde_rs41_dec_frame failed")
        return self._buf.value.decode(errors="replace")

    def json(self, frame: dict):
        for line in self.decode(frame).splitlines():
            if line.startswith("{"):
                return json.loads(line)
        return None


class DfmOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("verbose", "ptu", "ecc", "dist", "json", "sat", "raw", "opt_auto", "jsn_freq_khz")] + \
               [("version", C.c_char * 32), ("dbg", C.c_int32), ("reserved", C.c_int32 * 3)]


class DfmTelemetry:
    """`dfm09mod -vv --ecc --json --dist --auto` behind Engine(sonde="dfm").fetch_dfm(): feed EVERY frame dict in order (a DFM
    spreads one fix over nine packets); decode() returns the text of that frame (mostly empty), json() the parsed object or None."""

    def __init__(self, *, verbose: int = 2, ptu: int = 1, dist: bool = True, freq_khz: int = 0, version: str = "sonde_hip"):
        L = lib()
        L.sonde_dfm_dec_create.argtypes = [C.POINTER(DfmOpts), C.POINTER(C.c_void_p)]
        L.sonde_dfm_dec_frame.argtypes = [C.c_void_p, C.POINTER(SondeDfmFrame), C.c_char_p, C.c_size_t]
        L.sonde_dfm_dec_destroy.argtypes = [C.c_void_p]
        o = DfmOpts(verbose=verbose, ptu=ptu, ecc=1, dist=int(dist), json=1, opt_auto=1, jsn_freq_khz=freq_khz, version=version.encode())
        self._h = C.c_void_p()
        if L.sonde_dfm_dec_create(C.byref(o), C.byref(self._h)) < 0:
            raise SondeError("sonde_dfm_dec_create: unsupported options")
        self._buf = C.create_string_buffer(8192)

    def close(self):
        if getattr(self, "_h", None):
            lib().sonde_dfm_dec_destroy(self._h)
            self._h = None

    __del__ = close

    def decode(self, frame: dict) -> str:
        f = SondeDfmFrame(channel=frame["channel"], frame_in_hit=frame["frame_in_hit"], mv_pos=frame["mv_pos"], mv=frame["mv"],
                          frm_count=frame["frm_count"], inv=frame["inv"])
        for i, v in enumerate(frame["ecc"]):
            f.ecc[i] = v
        C.memmove(f.conf, frame["conf"], 7); C.memmove(f.dat1, frame["dat1"], 13); C.memmove(f.dat2, frame["dat2"], 13)
        if lib().sonde_dfm_dec_frame(self._h, C.byref(f), self._buf, len(self._buf)) < 0:
            raise SondeError("sonde_dfm_dec_frame failed")
        return self._buf.value.decode(errors="replace")

    def json(self, frame: dict):
        for line in self.decode(frame).splitlines():
            if line.startswith("{"):
                return json.loads(line)
        return None


class MxxOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("verbose", "ptu", "json", "silent", "raw", "jsn_freq_khz")] + \
               [("version", C.c_char * 32), ("reserved", C.c_int32 * 4)]


class _MxxTelemetry:
    _name, _frame, _flen = "", None, 0

    def __init__(self, *, verbose: int = 1, ptu: bool = True, freq_khz: int = 0, version: str = "sonde_hip", silent: bool = False):
        L = lib()
        self._create, self._dec, self._destroy = (getattr(L, "sonde_%s_dec_%s" % (self._name, n)) for n in ("create", "frame", "destroy"))
        self._create.argtypes = [C.POINTER(MxxOpts), C.POINTER(C.c_void_p)]
        self._dec.argtypes = [C.c_void_p, C.POINTER(self._frame), C.c_char_p, C.c_size_t]
        self._destroy.argtypes = [C.c_void_p]
        o = MxxOpts(verbose=verbose, ptu=int(ptu), json=1, silent=int(silent), jsn_freq_khz=freq_khz, version=version.encode())
        self._h = C.c_void_p()
        if self._create(C.byref(o), C.byref(self._h)) < 0:
            raise SondeError("sonde_%s_dec_create: unsupported options" % self._name)
        self._buf = C.create_string_buffer(8192)

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def decode(self, frame: dict) -> str:
        f = self._frame(**{k: frame[k] for k, _ in self._frame._fields_ if k != "frame"})
        raw = bytes(frame["frame"])
        C.memmove(f.frame, raw + bytes(self._flen - len(raw)), self._flen)
        if self._dec(self._h, C.byref(f), self._buf, len(self._buf)) < 0:
            raise SondeError("sonde_%s_dec_frame failed" % self._name)
        return self._buf.value.decode(errors="replace")

    def json(self, frame: dict):
        for line in self.decode(frame).splitlines():
            if line.startswith("{"):
                return json.loads(line)
        return None


class M10Telemetry(_MxxTelemetry):
    """`m10mod -v --ptu --json` behind Engine(sonde="m10").fetch_mxx()"""
    _name, _frame, _flen = "m10", SondeM10Frame, 124


class M20Telemetry(_MxxTelemetry):
    """`m20mod -v --ptu --json` behind Engine(sonde="m20").fetch_mxx()"""
    _name, _frame, _flen = "m20", SondeM20Frame, 172
