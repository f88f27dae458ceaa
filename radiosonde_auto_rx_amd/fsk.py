"""ctypes binding of the batched 2-FSK modem in libsonde_hip.so (include/sonde_fsk.h).

Python mirror of the reference's `fsk_demod` (utils/fsk_demod.c, the codec2 modem auto_rx pipes IQ into) for many
channels at once.  No CPU fallback: the constructor raises without the in-tree HIP library / a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import ABI_VERSION, _chk, lib

S16, CS16, CU8, CF32 = 1, 2, 3, 4


class FskCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "device", "n_channels", "Fs", "Rs", "M", "P", "nsym", "format",
                                         "fsk_lower", "fsk_upper", "mask", "tone_spacing", "max_chunk")] + \
               [("burst_mode", C.c_int32), ("raw_eye", C.c_int32), ("reserved", C.c_int32 * 2)]


class FskInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("Ts", "N", "Ndft", "Nmem", "Nbits")] + [("tc", C.c_float), ("reserved", C.c_int32 * 4)]


class FskFrame(C.Structure):
    _fields_ = [("nin", C.c_int32), ("nin_next", C.c_int32), ("f_est", C.c_float * 4), ("norm_rx_timing", C.c_float),
                ("ppm", C.c_float), ("EbNodB", C.c_float), ("snr_est", C.c_float)]


_proto = False


def _lib():
    global _proto
    L = lib()
    if not _proto:
        L.sonde_fsk_create.argtypes = [C.POINTER(FskCfg), C.POINTER(C.c_void_p)]
        L.sonde_fsk_destroy.argtypes = [C.c_void_p]
        L.sonde_fsk_info.argtypes = [C.c_void_p, C.POINTER(FskInfo)]
        L.sonde_fsk_process_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        L.sonde_fsk_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        L.sonde_fsk_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        L.sonde_fsk_wait.argtypes = [C.c_void_p]
        L.sonde_fsk_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(FskFrame), C.c_int32, C.POINTER(C.c_int32)]
        L.sonde_fsk_stats.argtypes = [C.c_void_p, C.c_int32, C.POINTER(FskFrame), C.c_void_p, C.POINTER(C.c_int64)]
        L.sonde_fsk_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.sonde_fsk_eye.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.sonde_fsk_process_host_var.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
        L.sonde_fsk_reset_channel.argtypes = [C.c_void_p, C.c_int32]
        L.sonde_fsk_fetch_bits.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.sonde_softin_dev_create.argtypes = [C.c_int32] * 6 + [C.POINTER(C.c_void_p)]
        L.sonde_softin_dev_destroy.argtypes = [C.c_void_p]
        L.sonde_softin_dev_push_fsk.argtypes = [C.c_void_p, C.c_void_p]
        L.sonde_softin_dev_submit_fsk.argtypes = [C.c_void_p, C.c_void_p]
        L.sonde_softin_dev_submit_fsk_behind.argtypes = [C.c_void_p, C.c_void_p]
        L.sonde_softin_dev_collect.argtypes = [C.c_void_p]
        L.sonde_softin_dev_push_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        L.sonde_softin_dev_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.sonde_softin_dev_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 5
        L.sonde_softin_dev_fetch_dfm.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.sonde_softin_dev_fetch_m10.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        _proto = True
    return L


class FskModem:
    """Batched `fsk_demod [--cs16|--cu8] -s [-b lo] [-u hi] [--mask S] [--nsym N] [-p P] (2|4) Fs Rs - -` for n channels."""

    def __init__(self, Fs: int, Rs: int, *, n_channels: int = 1, P: int = 8, nsym: int = 50, fmt: int = CS16,
                 lower: int | None = None, upper: int | None = None, mask: int = 0, max_chunk: int | None = None, device: int = 0, M: int = 2):
        self.n_channels, self.Fs, self.Rs, self.nsym, self.fmt, self.M = n_channels, Fs, Rs, nsym, fmt, M
        if lower is None:
            lower = -Fs // 2 if fmt != S16 else 0
        if upper is None:
            upper = Fs // 2
        cfg = FskCfg(ABI_VERSION, device, n_channels, Fs, Rs, M, P, nsym, fmt, lower, upper, int(mask > 0), mask if mask else 100,
                     max_chunk or Fs)
        h = C.c_void_p()
        _chk(_lib().sonde_fsk_create(C.byref(cfg), C.byref(h)))
        self._h = h
        info = FskInfo()
        _chk(_lib().sonde_fsk_info(h, C.byref(info)))
        self.info = {n: getattr(info, n) for n, _ in FskInfo._fields_ if n != "reserved"}

    def close(self):
        if getattr(self, "_h", None):
            _lib().sonde_fsk_destroy(self._h)
            self._h = None

    __del__ = close

    def process_host(self, x: np.ndarray):
        """x: [n_channels, n*k] int16 (k = 2 for cs16) or uint8 pairs (cu8)."""
        x = np.ascontiguousarray(x)
        if x.ndim == 1:
            x = x[None, :]
        per = 1 if self.fmt == S16 else 2
        n = x.shape[1] // per
        _chk(_lib().sonde_fsk_process_host(self._h, x.ctypes.data_as(C.c_void_p), n, n))

    def process_host_var(self, chunks):
        """Channels fed independently (what the broker does): chunks[c] = samples of channel c for this call (None / empty = nothing)."""
        per = 1 if self.fmt == S16 else 2
        ptrs = (C.c_void_p * self.n_channels)()
        ns = (C.c_int32 * self.n_channels)()
        keep = []
        for c in range(self.n_channels):
            x = chunks[c] if c < len(chunks) else None
            if x is None or len(x) == 0:
                ptrs[c] = None; ns[c] = 0
                continue
            x = np.ascontiguousarray(x); keep.append(x)
            ptrs[c] = x.ctypes.data; ns[c] = x.shape[-1] // per
        _chk(_lib().sonde_fsk_process_host_var(self._h, ptrs, ns))

    def reset_channel(self, ch: int):
        """Back to the fsk_create_hbr() state for one channel (a new stream starts on it)."""
        _chk(_lib().sonde_fsk_reset_channel(self._h, ch))

    def process_device(self, ptr: int, ch_stride: int, n: int):
        _chk(_lib().sonde_fsk_process_device(self._h, C.c_void_p(ptr), ch_stride, n))

    def submit_device(self, ptr: int, ch_stride: int, n: int):
        """process_device in two halves: everything enqueued on the engine's stream, no waiting; wait() (or any other call) blocks until it is through"""
        _chk(_lib().sonde_fsk_submit_device(self._h, C.c_void_p(ptr), ch_stride, n))

    def wait(self):
        _chk(_lib().sonde_fsk_wait(self._h))

    def fetch(self, ch: int = 0):
        """-> (soft decisions [frames, Nbits], list of per-frame dicts) of the last process call."""
        cap = 4096
        fr = (FskFrame * cap)()
        nf = C.c_int32(0)
        nbits = self.info["Nbits"]
        sd = np.zeros(cap * nbits, np.float32)
        nb = _chk(_lib().sonde_fsk_fetch(self._h, ch, sd.ctypes.data_as(C.c_void_p), len(sd), fr, cap, C.byref(nf)))
        recs = [dict(nin=fr[i].nin, nin_next=fr[i].nin_next, f_est=tuple(fr[i].f_est[m] for m in range(self.M)), norm_rx_timing=fr[i].norm_rx_timing,
                     ppm=fr[i].ppm, EbNodB=fr[i].EbNodB, snr_est=fr[i].snr_est) for i in range(nf.value)]
        return sd[:nb].reshape(-1, nbits), recs

    def stats(self, ch: int = 0):
        last = FskFrame()
        Sf = np.zeros(self.info["Ndft"], np.float32)
        n = C.c_int64(0)
        _chk(_lib().sonde_fsk_stats(self._h, ch, C.byref(last), Sf.ctypes.data_as(C.c_void_p), C.byref(n)))
        return dict(f_est=tuple(last.f_est[m] for m in range(self.M)), ppm=last.ppm, EbNodB=last.EbNodB, snr_est=last.snr_est,
                    norm_rx_timing=last.norm_rx_timing, nin=last.nin_next, Sf=Sf, samples=n.value)

    def eye(self, ch: int = 0) -> np.ndarray:
        """Eye diagram of the last modem frame: [8 traces, 2P/ceil(2P/160) samples], normalised (MODEM_STATS.rx_eye)."""
        buf = np.zeros(8 * 160, np.float32)
        ntr, nes = C.c_int32(0), C.c_int32(0)
        n = _chk(_lib().sonde_fsk_eye(self._h, ch, buf.ctypes.data_as(C.c_void_p), C.byref(ntr), C.byref(nes)))
        return buf[:n].reshape(ntr.value, nes.value)

    def kernel_ms(self):
        ms, n = C.c_double(0), C.c_int64(0)
        _chk(_lib().sonde_fsk_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class SoftinDev:
    """Batched `rs41mod --softin [-i] [--ecc|--ecc2]` on the device (include/sonde_fsk.h sonde_softin_dev_*): the consumer of a modem
    engine's soft decisions where they lie — auto_rx's pipe `fsk_demod ... | rs41mod --softin -i` (auto_rx/autorx/decode.py:901-909)
    without the soft-decision stream crossing to the host.  No CPU fallback."""

    def __init__(self, n_channels: int, *, ecc: int = 2, softinv: bool = False, inv: bool = True, auto: bool = False, kind: str = "rs41"):
        """kind: "rs41" (rs41mod --softin), "dfm" (dfm09mod --softin: ecc 0 / 1 = --ecc / 2 = --ecc2) or "m10" (m10mod --softin)"""
        from .engine import SONDE_RS41, SONDE_DFM09, SONDE_M10
        h = C.c_void_p()
        self.kind, self.ecc = kind, ecc
        _chk(_lib().sonde_softin_dev_create(n_channels, {"rs41": SONDE_RS41, "dfm": SONDE_DFM09, "m10": SONDE_M10}[kind], ecc, int(softinv), int(inv), int(auto), C.byref(h)))
        self._h, self.n_channels = h, n_channels

    def close(self):
        if getattr(self, "_h", None):
            _lib().sonde_softin_dev_destroy(self._h)
            self._h = None

    __del__ = close

    def push_fsk(self, modem: "FskModem"):
        """consume what the modem's last process call left in device memory"""
        _chk(_lib().sonde_softin_dev_push_fsk(self._h, modem._h))

    def submit_fsk(self, modem: "FskModem"):
        """push_fsk without waiting for the result: waits for the modem's launch, then enqueues the consumer on its own stream; the modem can be given its next second
        (submit_device) before collect() — it keeps the soft decisions of two launches"""
        _chk(_lib().sonde_softin_dev_submit_fsk(self._h, modem._h))

    def submit_fsk_behind(self, modem: "FskModem"):
        """the consumer over the modem's launch BEFORE the one in flight (order: wait, submit_device, collect, submit_fsk_behind): nothing waits"""
        _chk(_lib().sonde_softin_dev_submit_fsk_behind(self._h, modem._h))

    def collect(self):
        _chk(_lib().sonde_softin_dev_collect(self._h))

    def push_device(self, ptr: int, ch_stride: int, n_bits: int):
        _chk(_lib().sonde_softin_dev_push_device(self._h, C.c_void_p(ptr), ch_stride, n_bits))

    def fetch(self, max_frames: int = 4096):
        """-> list of dicts (channel, len, ecc, mv, mv_pos, frame bytes, line = the `rs41mod -r` text) of the frames completed since the last fetch"""
        from .engine import SondeFrame, lib
        buf = (SondeFrame * max_frames)()
        n = _chk(_lib().sonde_softin_dev_fetch(self._h, buf, max_frames))
        out = []
        line = C.create_string_buffer(1200)
        for i in range(n):
            f = buf[i]
            ll = lib().sonde_rs41_rawline(C.byref(f), line, 1200)
            out.append(dict(channel=f.channel, len=f.len, ecc=f.ecc, mv=f.mv, mv_pos=f.mv_pos, nbytes=f.nbytes, frame=bytes(f.frame), line=line.raw[:ll].decode()))
        return out

    def fetch_dfm(self, max_frames: int = 8192):
        """DFM consumers: dicts with ecc (per block), conf / dat1 / dat2 nibbles, frame_in_hit, frm_count, inv, line = the `dfm09mod -r [--ecc]` text"""
        from .engine import SondeDfmFrame, lib
        buf = (SondeDfmFrame * max_frames)()
        n = _chk(_lib().sonde_softin_dev_fetch_dfm(self._h, buf, max_frames))
        out = []
        line = C.create_string_buffer(160)
        for i in range(n):
            f = buf[i]
            ll = lib().sonde_dfm_rawline(C.byref(f), self.ecc, line, 128)
            out.append(dict(channel=f.channel, frame_in_hit=f.frame_in_hit, ecc=tuple(f.ecc), mv=f.mv, mv_pos=f.mv_pos, frm_count=f.frm_count, inv=f.inv,
                            conf=bytes(f.conf), dat1=bytes(f.dat1), dat2=bytes(f.dat2), rawbits=bytes(f.rawbits), line=line.raw[:ll].decode()))
        return out

    def fetch_m10(self, max_frames: int = 4096, verbose: int = 1):
        """M10 consumers: dicts with len, cs_ok, cs_calc, frame bytes, line = the `m10mod -r [-v]` text"""
        from .engine import SondeM10Frame, lib
        buf = (SondeM10Frame * max_frames)()
        n = _chk(_lib().sonde_softin_dev_fetch_m10(self._h, buf, max_frames))
        out = []
        line = C.create_string_buffer(420)
        for i in range(n):
            f = buf[i]
            ll = lib().sonde_m10_rawline(C.byref(f), verbose, line, 420)
            out.append(dict(channel=f.channel, nbits=f.nbits, len=f.len, cs_ok=f.cs_ok, cs_calc=f.cs_calc, mv=f.mv, mv_pos=f.mv_pos, frame=bytes(f.frame), line=line.raw[:ll].decode()))
        return out

    def counts(self):
        v = [C.c_int64(0) for _ in range(5)]
        _chk(_lib().sonde_softin_dev_counts(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("frames", "ecc_ok", "repaired", "symbols", "dropped"), [x.value for x in v]))
