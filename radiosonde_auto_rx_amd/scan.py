"""ctypes binding of the batched scanner in libsonde_hip.so (include/sonde_scan.h).

Python mirror of the reference's `dft_detect` CLI for many channels at once (auto_rx/autorx/scan.py:541-547 is
the caller it replaces).  No CPU fallback: the constructor raises without the in-tree HIP library / a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import ABI_VERSION, _chk, lib

NTPL = 16
AUDIO, IFIQ, BBIQ = 0, 1, 5
TYPES = ("DFM9", "RS41", "RS92", "LMS6", "IMET5", "MK2LMS", "M10", "MEISEI", "RD94RD41", "MRZ", "MTS01",
         "C34C50", "WXR301", "WXRPN9", "IMET1AB", "IMETafsk")


class ScanCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "device", "n_channels", "sample_rate", "bits", "iq_mode", "opt_dc",
                                         "opt_min", "opt_cont", "opt_d2", "opt_lband", "audio_channels", "audio_select",
                                         "max_chunk")] + \
               [("bw_khz", C.c_float), ("ths", C.c_float), ("time_limit", C.c_float), ("disable_mask", C.c_uint32),
                ("opt_exact", C.c_int32), ("reserved", C.c_int32 * 3)]


class Detection(C.Structure):
    _fields_ = [("channel", C.c_int32), ("tpl", C.c_int32), ("tn", C.c_int32), ("type", C.c_char * 12), ("score", C.c_float),
                ("sample", C.c_uint32), ("df", C.c_float), ("freq_hz", C.c_float), ("m10_bytes", C.c_uint32),
                ("printed", C.c_int32)]


class ScanInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("if_sr", "decM", "dectaps", "lpiq_taps", "lpfm_taps", "K", "N", "delay", "L2")] + \
               [("L", C.c_int32 * NTPL), ("ring_len", C.c_int32), ("reserved", C.c_int32 * 3)]


class ScanWindow(C.Structure):
    _fields_ = [("channel", C.c_int32), ("pos", C.c_uint32), ("mp", C.c_int32 * NTPL), ("mv", C.c_float * NTPL),
                ("mpos", C.c_uint32 * NTPL), ("dc", C.c_float * NTPL), ("herrs", C.c_int32 * NTPL), ("m10", C.c_uint32 * NTPL)]


_proto_done = False


def _lib():
    global _proto_done
    L = lib()
    if not _proto_done:
        L.sonde_scan_create.argtypes = [C.POINTER(ScanCfg), C.POINTER(C.c_double), C.POINTER(C.c_void_p)]
        L.sonde_scan_destroy.argtypes = [C.c_void_p]
        L.sonde_scan_info.argtypes = [C.c_void_p, C.POINTER(ScanInfo)]
        L.sonde_scan_process_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        L.sonde_scan_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        L.sonde_scan_wait_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.sonde_scan_finish.argtypes = [C.c_void_p]
        L.sonde_scan_fetch.argtypes = [C.c_void_p, C.POINTER(Detection), C.c_int32]
        L.sonde_scan_channel_done.argtypes = [C.c_void_p, C.c_int32]
        L.sonde_scan_result.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.sonde_scan_line.argtypes = [C.c_void_p, C.POINTER(Detection), C.c_int, C.c_char_p, C.c_size_t]
        L.sonde_scan_last_windows.argtypes = [C.c_void_p, C.POINTER(ScanWindow), C.c_int32]
        L.sonde_scan_read_fm.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]
        L.sonde_scan_kernel_ms.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        _proto_done = True
    return L


class Scanner:
    """Batched `dft_detect [--IQ fq | --iq] [--dc] [--bw k] [-t s] [-c] - <sr> 16` for n channels on one GPU."""

    def __init__(self, sample_rate: int, *, fq=None, n_channels: int | None = None, iq_mode: int = BBIQ, dc: bool = False,
                 bw_khz: float = 0.0, opt_min: bool = False, cont: bool = False, d2: bool = False, lband: bool = False,
                 ths: float = 0.0, time_limit: float = 0.0, max_chunk: int | None = None, device: int = 0,
                 audio_channels: int = 1, audio_select: int = 0, disable_mask: int = 0, bits: int = 16, exact: bool = False):
        """exact=True: every (window, template) goes through the reference's transform network (per-window parity taps); the default scores
        them on the matrix cores first and runs the network for those within 0.03 of their threshold — same detections, scores, exit codes."""
        if fq is None:
            fq = np.zeros(n_channels or 1)
        fq = np.atleast_1d(np.asarray(fq, dtype=np.float64))
        self.n_channels = len(fq)
        self.sample_rate = sample_rate
        self.iq_mode = iq_mode
        self.audio_channels = audio_channels
        self._dtype = {8: np.uint8, 32: np.float32}.get(bits, np.int16)
        cfg = ScanCfg(ABI_VERSION, device, self.n_channels, sample_rate, bits, iq_mode, int(dc), int(opt_min), int(cont), int(d2),
                      int(lband), audio_channels, audio_select, max_chunk or sample_rate, bw_khz, ths, time_limit, disable_mask, int(exact))
        h = C.c_void_p()
        _chk(_lib().sonde_scan_create(C.byref(cfg), fq.ctypes.data_as(C.POINTER(C.c_double)), C.byref(h)))
        self._h = h
        info = ScanInfo()
        _chk(_lib().sonde_scan_info(h, C.byref(info)))
        self.info = {n: getattr(info, n) for n, _ in ScanInfo._fields_ if n not in ("reserved", "L")}
        self.info["L"] = list(info.L)

    def close(self):
        if getattr(self, "_h", None):
            _lib().sonde_scan_destroy(self._h)
            self._h = None

    __del__ = close

    def process_host(self, x: np.ndarray, shared: bool = False):
        """x: int16 (bits=8: uint8) [n_channels, 2*n] (IQ forms) or [n_channels, n*audio_channels] (FM audio).
        shared=True: x is ONE wideband stream [2*n] that every channel mixes its own fq out of (channel stride 0)."""
        x = np.ascontiguousarray(x, dtype=self._dtype)
        if x.ndim == 1:
            x = x[None, :]
        assert shared or x.shape[0] == self.n_channels
        per = 2 if self.iq_mode != AUDIO else self.audio_channels
        n = x.shape[1] // per
        _chk(_lib().sonde_scan_process_host(self._h, x.ctypes.data_as(C.c_void_p), 0 if shared else n, n))

    def process_device(self, ptr: int, ch_stride: int, n: int):
        _chk(_lib().sonde_scan_process_device(self._h, C.c_void_p(ptr), ch_stride, n))

    def wait_stream(self, stream: int):
        """The scanner's stream waits for what is queued on `stream` (Channelizer.stream(), a torch stream's cuda_stream): no host wait in between."""
        _chk(_lib().sonde_scan_wait_stream(self._h, C.c_void_p(stream)))

    def finish(self):
        """End of input: decide a pending IMET check with the samples that exist."""
        _chk(_lib().sonde_scan_finish(self._h))

    def fetch(self, verbose: bool = False):
        out = []
        buf = (Detection * 64)()
        line = C.create_string_buffer(256)
        while True:
            k = _chk(_lib().sonde_scan_fetch(self._h, buf, 64))
            for i in range(k):
                d = buf[i]
                _lib().sonde_scan_line(self._h, C.byref(d), int(verbose), line, 256)
                out.append(dict(channel=d.channel, tpl=d.tpl, tn=d.tn, type=d.type.decode(), score=d.score, sample=d.sample,
                                df=d.df, freq_hz=d.freq_hz, m10_bytes=d.m10_bytes, printed=bool(d.printed),
                                line=line.value.decode()))
            if k < 64:
                return out

    def done(self, ch: int = 0) -> bool:
        return bool(_chk(_lib().sonde_scan_channel_done(self._h, ch)))

    def result(self, ch: int = 0) -> int:
        code = C.c_int32(0)
        _chk(_lib().sonde_scan_result(self._h, ch, C.byref(code)))
        return code.value

    def last_windows(self):
        n = _chk(_lib().sonde_scan_last_windows(self._h, None, 0))
        buf = (ScanWindow * max(n, 1))()
        _chk(_lib().sonde_scan_last_windows(self._h, buf, n))
        return [dict(channel=w.channel, pos=w.pos, mp=np.array(w.mp), mv=np.array(w.mv), mpos=np.array(w.mpos),
                     dc=np.array(w.dc), herrs=np.array(w.herrs), m10=np.array(w.m10)) for w in buf[:n]]

    def read_fm(self, ch: int, stream: int, first: int, count: int) -> np.ndarray:
        out = np.zeros(count, np.float32)
        _chk(_lib().sonde_scan_read_fm(self._h, ch, stream, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def kernel_ms(self, name: str):
        ms, n = C.c_double(0), C.c_int64(0)
        _chk(_lib().sonde_scan_kernel_ms(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
