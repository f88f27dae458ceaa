"""ctypes binding of the polyphase channelizer in libsonde_hip.so (include/sonde_chan.h): one wideband cs16 stream -> M uniformly
spaced IF-rate channels (float32 IQ) in one pass.  The reference has no counterpart (one mixer + FIR per frequency and process,
demod_mod.c:737-754); what comes out is the `--iq` / `--iq2` / `--iq3` float input of dft_detect and the decoders."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import ABI_VERSION, _chk, lib


class ChanCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "device", "sample_rate", "M", "D", "P", "max_chunk")] + [("reserved", C.c_int32 * 5)]


class ChanInfo(C.Structure):
    _fields_ = [("out_rate_num", C.c_int32), ("out_rate_den", C.c_int32), ("taps", C.c_int32), ("max_frames", C.c_int32),
                ("spacing_hz", C.c_float), ("reserved", C.c_int32 * 3)]


_proto = False


def _lib():
    global _proto
    L = lib()
    if not _proto:
        L.sonde_chan_create.argtypes = [C.POINTER(ChanCfg), C.POINTER(C.c_void_p)]
        L.sonde_chan_destroy.argtypes = [C.c_void_p]
        L.sonde_chan_info.argtypes = [C.c_void_p, C.POINTER(ChanInfo)]
        L.sonde_chan_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.sonde_chan_process_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.sonde_chan_sync.argtypes = [C.c_void_p]
        L.sonde_chan_stream.argtypes = [C.c_void_p]; L.sonde_chan_stream.restype = C.c_void_p
        L.sonde_chan_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        _proto = True
    return L


class Channelizer:
    def __init__(self, sample_rate: int, M: int = 256, D: int = 200, P: int = 16, *, max_chunk: int | None = None, device: int = 0):
        cfg = ChanCfg(ABI_VERSION, device, sample_rate, M, D, P, max_chunk or sample_rate)
        h = C.c_void_p()
        _chk(_lib().sonde_chan_create(C.byref(cfg), C.byref(h)))
        self._h = h
        info = ChanInfo()
        _chk(_lib().sonde_chan_info(h, C.byref(info)))
        self.M, self.D, self.P, self.sample_rate = M, D, P, sample_rate
        self.out_rate = info.out_rate_num / info.out_rate_den
        self.taps, self.max_frames, self.spacing_hz = info.taps, info.max_frames, info.spacing_hz

    def channel_freq(self, k: int) -> float:
        """centre of channel k relative to the stream's centre, Hz"""
        return (k if k < self.M // 2 else k - self.M) * self.sample_rate / self.M

    def nearest_channel(self, f_hz: float) -> int:
        return int(round(f_hz * self.M / self.sample_rate)) % self.M

    def process_device(self, in_ptr: int, n_samples: int, out_ptr: int, out_stride: int) -> int:
        return _chk(_lib().sonde_chan_process_device(self._h, C.c_void_p(in_ptr), n_samples, C.c_void_p(out_ptr), out_stride))

    def process_host(self, x: np.ndarray, out_ptr: int, out_stride: int) -> int:
        x = np.ascontiguousarray(x, dtype=np.int16)
        return _chk(_lib().sonde_chan_process_host(self._h, x.ctypes.data_as(C.c_void_p), len(x) // 2, C.c_void_p(out_ptr), out_stride))

    def sync(self):
        _chk(_lib().sonde_chan_sync(self._h))

    def stream(self) -> int:
        """The hipStream_t the channelizer queues its work on (for Scanner.wait_stream / an engine's wait)."""
        return int(_lib().sonde_chan_stream(self._h) or 0)

    def kernel_ms(self):
        ms, n = C.c_double(), C.c_int64()
        _chk(_lib().sonde_chan_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None):
            _lib().sonde_chan_destroy(self._h)
            self._h = None

    __del__ = close
