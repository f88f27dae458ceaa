"""ctypes binding of libsonde_hip.so (include/sonde_hip.h) — the Python host side used by tests and bench.py.

There is no CPU fallback: if the in-tree HIP library is missing or no GPU is present the constructor raises.
PyTorch is only plumbing here (device buffers for resident input, torch.distributed in bench.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SONDE_HIP_LIB", os.path.join(_HERE, "libsonde_hip.so"))     # override: A/B experiments with another build

SONDE_RS41 = 41
SONDE_DFM09 = 9
SONDE_M10 = 10
SONDE_M20 = 20
LP_IQ, LP_FM = 1, 2
TAP_DECIM, TAP_IFIQ, TAP_FM, TAP_BUFS, TAP_CORR = range(5)
ABI_VERSION = 3


class SondeCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "device", "n_channels", "sample_rate", "bits", "sonde_type",
                                         "opt_lp", "opt_dc", "opt_min", "lpiq_bw", "ecc_level")] + \
               [("thres", C.c_float), ("max_chunk", C.c_int32), ("max_frames", C.c_int32), ("keep_soft", C.c_int32),
                ("pipeline", C.c_int32), ("input", C.c_int32), ("audio_channels", C.c_int32), ("audio_select", C.c_int32),
                ("if_rate", C.c_int32), ("opt_iqdc", C.c_int32), ("opt_inv", C.c_int32), ("opt_nolut", C.c_int32), ("m10_noskip", C.c_int32), ("opt_auto", C.c_int32), ("if_tune", C.c_int32)]


class SondeFrame(C.Structure):
    _fields_ = [("channel", C.c_int32), ("len", C.c_int32), ("ecc", C.c_int32), ("mv_pos", C.c_uint32),
                ("mv", C.c_float), ("nbytes", C.c_int32), ("frame", C.c_uint8 * 518), ("pad", C.c_uint8 * 2)]


class SondeDfmFrame(C.Structure):
    _fields_ = [("channel", C.c_int32), ("frame_in_hit", C.c_int32), ("ecc", C.c_int32 * 3), ("mv_pos", C.c_uint32),
                ("mv", C.c_float), ("conf", C.c_uint8 * 7), ("dat1", C.c_uint8 * 13), ("dat2", C.c_uint8 * 13), ("pad", C.c_uint8 * 3),
                ("frm_count", C.c_float), ("inv", C.c_int32), ("rawbits", C.c_uint8 * 35), ("pad2", C.c_uint8)]


class SondeM10Frame(C.Structure):
    _fields_ = [("channel", C.c_int32), ("nbits", C.c_int32), ("len", C.c_int32), ("cs_ok", C.c_int32), ("cs_calc", C.c_uint32),
                ("mv_pos", C.c_uint32), ("mv", C.c_float), ("frame", C.c_uint8 * 124)]


class SondeM20Frame(C.Structure):
    _fields_ = [("channel", C.c_int32), ("nbits", C.c_int32), ("len", C.c_int32), ("cs_ok", C.c_int32), ("cs_calc", C.c_uint32),
                ("blk_ok", C.c_int32), ("fw", C.c_int32), ("mv_pos", C.c_uint32), ("mv", C.c_float), ("frame", C.c_uint8 * 172)]


class SondeHit(C.Structure):
    _fields_ = [("channel", C.c_int32), ("nbits", C.c_int32), ("mv_pos", C.c_uint32), ("mv", C.c_float)]


class SondeGeneric(C.Structure):
    _fields_ = [("header", C.c_char * 68), ("baud", C.c_float), ("bt", C.c_float), ("h", C.c_float), ("symlen", C.c_int32), ("symhd", C.c_int32),
                ("hdmax", C.c_int32), ("bitofs", C.c_int32), ("nbits", C.c_int32), ("skip_bits", C.c_int32), ("l_win", C.c_float),
                ("lpiq_bw", C.c_int32), ("lpfm_bw", C.c_int32), ("slice_baud", C.c_float), ("reserved", C.c_int32 * 3)]


class SondeGroup(C.Structure):
    """sonde_group_t: one decoder command line of a mixed engine (include/sonde_hip.h)"""
    _fields_ = [("sonde_type", C.c_int32), ("ecc_level", C.c_int32), ("lpiq_bw", C.c_int32), ("opt_inv", C.c_int32), ("opt_auto", C.c_int32),
                ("m10_noskip", C.c_int32), ("thres", C.c_float), ("reserved", C.c_int32)]


class SondeInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("if_sr", "decM", "dectaps", "lut_len", "lpiq_taps", "lpfm_taps",
                                         "L", "M", "K", "N", "delay")] + \
               [("sps", C.c_float), ("ring_len", C.c_int32), ("reserved", C.c_int32 * 3)]


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the in-tree library (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-j8", "-C", src])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the engine has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.sonde_strerror.restype = C.c_char_p
        L.sonde_engine_stream.restype = C.c_void_p
        L.sonde_engine_create.argtypes = [C.POINTER(SondeCfg), C.POINTER(C.c_double), C.POINTER(C.c_void_p)]
        L.sonde_engine_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        L.sonde_engine_process_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
        L.sonde_engine_fetch_frames.argtypes = [C.c_void_p, C.POINTER(SondeFrame), C.c_int32]
        L.sonde_engine_fetch_frames_lagged.argtypes = [C.c_void_p, C.POINTER(SondeFrame), C.c_int32, C.c_int32]
        L.sonde_engine_finish.argtypes = [C.c_void_p, C.POINTER(SondeFrame), C.c_int32]
        L.sonde_engine_fetch_soft.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.sonde_engine_fetch_dfm.argtypes = [C.c_void_p, C.POINTER(SondeDfmFrame), C.c_int32, C.c_int32]
        L.sonde_dfm_rawline.argtypes = [C.POINTER(SondeDfmFrame), C.c_int, C.c_char_p, C.c_size_t]
        L.sonde_engine_create_generic.argtypes = [C.POINTER(SondeCfg), C.POINTER(C.c_double), C.POINTER(SondeGeneric), C.POINTER(C.c_void_p)]
        L.sonde_engine_fetch_hits.argtypes = [C.c_void_p, C.POINTER(SondeHit), C.c_int32, C.c_int32]
        L.sonde_engine_set_threshold.argtypes = [C.c_void_p, C.c_float]
        L.sonde_engine_fetch_m10.argtypes = [C.c_void_p, C.POINTER(SondeM10Frame), C.c_int32, C.c_int32]
        L.sonde_engine_fetch_m10_lagged.argtypes = [C.c_void_p, C.POINTER(SondeM10Frame), C.c_int32, C.c_int32]
        L.sonde_engine_fetch_dfm_lagged.argtypes = [C.c_void_p, C.POINTER(SondeDfmFrame), C.c_int32, C.c_int32]
        L.sonde_engine_create_mixed.argtypes = [C.POINTER(SondeCfg), C.POINTER(C.c_double), C.POINTER(SondeGroup), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
        L.sonde_engine_group_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(SondeInfo)]
        L.sonde_engine_fetch_m20.argtypes = [C.c_void_p, C.POINTER(SondeM20Frame), C.c_int32, C.c_int32]
        L.sonde_m10_rawline.argtypes = [C.POINTER(SondeM10Frame), C.c_int, C.c_char_p, C.c_size_t]
        L.sonde_m20_rawline.argtypes = [C.POINTER(SondeM20Frame), C.c_int, C.c_char_p, C.c_size_t]
        L.sonde_engine_read_tap.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]
        L.sonde_engine_sync.argtypes = [C.c_void_p]
        L.sonde_engine_samples_to_dc_boundary.argtypes = [C.c_void_p]
        L.sonde_engine_samples_to_dc_boundary.restype = C.c_int64
        L.sonde_engine_destroy.argtypes = [C.c_void_p]
        L.sonde_engine_info.argtypes = [C.c_void_p, C.POINTER(SondeInfo)]
        L.sonde_engine_profile.argtypes = [C.c_void_p, C.c_int]
        L.sonde_engine_kernel_ms.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.sonde_engine_stream.argtypes = [C.c_void_p]
        L.sonde_rs41_rawline.argtypes = [C.POINTER(SondeFrame), C.c_char_p, C.c_size_t]
        L.sonde_engine_host_ecc_frames.argtypes = [C.c_void_p]
        L.sonde_engine_host_ecc_frames.restype = C.c_longlong
        L.sonde_engine_set_device_ecc.argtypes = [C.c_void_p, C.c_int32]
        L.sonde_rs41_ecc_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def rs41_ecc_device(frames: np.ndarray, flen, level: int = 2):
    """rs41_ecc() of a batch of de-whitened 518-byte RS41 frames on the device (sonde_rs41_ecc_device, include/sonde_hip.h).
    -> (repaired frames [n, 518] u8, ecc [n] i32, codes [n, 2] i32 = the two rs_decode values, syndromes [n, 48] u8 of the first pass)"""
    fr = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, 518).copy()
    n = fr.shape[0]
    fl = np.ascontiguousarray(np.broadcast_to(np.asarray(flen, dtype=np.int32), (n,)))
    ecc, codes, synd = np.zeros(n, np.int32), np.zeros((n, 2), np.int32), np.zeros((n, 48), np.uint8)
    _chk(lib().sonde_rs41_ecc_device(fr.ctypes.data, fl.ctypes.data, n, level, ecc.ctypes.data, codes.ctypes.data, synd.ctypes.data))
    return fr, ecc, codes, synd


class SondeError(RuntimeError):
    pass


def snap_fq(fq: float, sr: int) -> float:
    """The carrier the reference's mixer table really mixes with: fq snapped to a multiple of 16 Hz (demod_mod.c:1265-1288,
    valid where 16 divides the sample rate); the engine applies the same rule itself (design_mixer in sonde_design.cpp)."""
    return int(round(fq * sr / 16.0)) * 16 / sr


def _chk(rc: int) -> int:
    if rc < 0:
        raise SondeError(f"libsonde_hip: {lib().sonde_strerror(rc).decode()} ({rc})")
    return rc


class Engine:
    """Batched equivalent of `rs41mod --IQ <fq> [--lpIQ] - <sr> 16` for n channels on one GPU.
    iq_mode: 5 = --IQ fq (baseband, mixed and decimated), 1 / 2 / 3 = --iq0 / --iq2 / --iq3 (IF-rate IQ, fq unused);
    audio=True = FM audio (opt_iq 0)."""

    def __init__(self, fq, sample_rate: int, *, device: int = 0, lp_iq: bool = True, lp_fm: bool = False,
                 ecc: int = 2, thres: float = 0.0, max_chunk: int | None = None, max_frames: int = 0,
                 keep_soft: bool = False, opt_min: bool = False, lpiq_bw: int = 0, opt_dc: bool = False,
                 sonde: str = "rs41", pipeline: bool = False, audio: bool = False, audio_channels: int = 1, audio_select: int = 0,
                 if_rate: int = 0, bits: int = 16, iq_mode: int = 5, iqdc: bool = False, inv: bool = False, auto: bool = False, nolut: bool = False,
                 generic: dict | None = None, if_tune: bool = False):
        fq = np.atleast_1d(np.asarray(fq, dtype=np.float64))
        self.n_channels = len(fq)
        self.sample_rate = sample_rate
        self.sonde = sonde
        self.ecc = ecc
        self._per_sample = audio_channels if audio else 2      # input words (int16, or uint8 for bits=8) per sample
        self._dtype = {8: np.uint8, 32: np.float32}.get(bits, np.int16)
        cfg = SondeCfg(ABI_VERSION, device, self.n_channels, sample_rate, bits, {"rs41": SONDE_RS41, "dfm": SONDE_DFM09, "m10": 10, "m20": 20, "frontend": 0, "generic": 99}[sonde],
                       (LP_IQ if lp_iq else 0) | (LP_FM if lp_fm else 0), int(opt_dc), int(opt_min), lpiq_bw, ecc,
                       thres, max_chunk or sample_rate, max_frames, int(keep_soft), int(pipeline),
                       1 if audio else {5: 0, 1: 2, 2: 3, 3: 4}[iq_mode], audio_channels, audio_select, if_rate, int(iqdc), int(inv), int(nolut), 0, int(auto), int(if_tune))
        h = C.c_void_p()
        if sonde == "generic":      # any other 2-FSK sonde: generic = dict(header=, baud=, bt=, h=, symlen=, symhd=, hdmax=, bitofs=, nbits=, ...) (sonde_generic_t)
            g = SondeGeneric(**{k: (v.encode() if k == "header" else v) for k, v in (generic or {}).items()})
            cfg.keep_soft = 1
            _chk(lib().sonde_engine_create_generic(C.byref(cfg), fq.ctypes.data_as(C.POINTER(C.c_double)), C.byref(g), C.byref(h)))
        else:
            _chk(lib().sonde_engine_create(C.byref(cfg), fq.ctypes.data_as(C.POINTER(C.c_double)), C.byref(h)))
        self._h = h
        info = SondeInfo()
        _chk(lib().sonde_engine_info(h, C.byref(info)))
        self.info = {n: getattr(info, n) for n, _ in SondeInfo._fields_ if n != "reserved"}
        self.nbits = 4080 if sonde == "rs41" else generic["nbits"] if sonde == "generic" else {"m10": 968, "m20": 1320}.get(sonde, 2224)
        self._max_frames = max_frames or 4 * self.n_channels

    def close(self):
        if getattr(self, "_h", None):
            lib().sonde_engine_destroy(self._h)
            self._h = None

    __del__ = close

    # -- input -------------------------------------------------------------------------------
    def process_host(self, iq: np.ndarray, n_samples: int | None = None, shared: bool = False):
        """iq: int16 (bits=8: uint8, 32: float32) array [n_channels, 2*stride] (or [2*stride] for one channel).
        shared=True: iq is ONE wideband stream [2*n] that every channel mixes its own fq out of (channel stride 0)."""
        if shared:
            iq = np.ascontiguousarray(iq, dtype=self._dtype).reshape(-1)
            n = iq.shape[0] // self._per_sample
            _chk(lib().sonde_engine_process_host(self._h, iq.ctypes.data_as(C.c_void_p), 0, n_samples or n))
            return
        iq = np.ascontiguousarray(iq, dtype=self._dtype).reshape(self.n_channels, -1)
        stride = iq.shape[1] // (self._per_sample)
        _chk(lib().sonde_engine_process_host(self._h, iq.ctypes.data_as(C.c_void_p), stride, n_samples or stride))

    def process_device(self, ptr: int, ch_stride: int, n_samples: int):
        _chk(lib().sonde_engine_process_device(self._h, C.c_void_p(ptr), ch_stride, n_samples))

    def finish_channel(self, ch: int):
        """End of one channel's stream: the frame in progress on it is emitted with the bits that exist (fetch afterwards)."""
        _chk(lib().sonde_engine_finish_channel(self._h, ch))

    def restart_channel(self, ch: int):
        """A new stream starts on the channel with the next samples (IF-rate / FM-audio engines only; include/sonde_hip.h)."""
        _chk(lib().sonde_engine_restart_channel(self._h, ch))

    def tune_channel(self, ch: int, fq: float):
        """if_tune engines: fine-tuning offset (cycles per IF sample) of one channel"""
        lib().sonde_engine_tune_channel.argtypes = [C.c_void_p, C.c_int32, C.c_double]
        _chk(lib().sonde_engine_tune_channel(self._h, ch, float(fq)))

    def samples_to_dc_boundary(self) -> int:
        return int(lib().sonde_engine_samples_to_dc_boundary(self._h))

    def sync(self):
        _chk(lib().sonde_engine_sync(self._h))

    # -- output ------------------------------------------------------------------------------
    def set_summary(self, device_ptr: int, channel_base: int = 0):
        """Per-channel detection summaries (sonde_summary_t, 32 B each) are written into this device buffer from now on
        (radiosonde_auto_rx_amd.shard.summary_buffer); 0 switches them off."""
        _chk(lib().sonde_engine_set_summary(self._h, C.c_void_p(device_ptr), channel_base))

    def set_summary_snapshots(self, device_ptr: int) -> int:
        """device memory for 2 x n_channels records: every process call leaves a copy of the summaries in half (call & 1), stable once a
        lagged fetch has waited for that call (include/sonde_hip.h).  Returns the half the next call fills."""
        return _chk(lib().sonde_engine_set_summary_snapshots(self._h, C.c_void_p(device_ptr)))

    def overflowed(self) -> bool:
        """True if the device-side frame queue overflowed since the last call (oldest frames overwritten before a fetch read them);
        the fetch_* methods return what they could read either way"""
        return bool(_chk(lib().sonde_engine_overflowed(self._h)))

    def host_ecc_frames(self) -> int:
        """RS41 frames whose Reed-Solomon decoder ran on the host so far (whole frames are decoded in k_framesync; include/sonde_hip.h)"""
        return int(lib().sonde_engine_host_ecc_frames(self._h))

    def set_device_ecc(self, on: bool):
        _chk(lib().sonde_engine_set_device_ecc(self._h, 1 if on else 0))

    def fetch_frames(self, max_frames: int | None = None, with_soft: bool = False, finish: bool = False):
        """Frames completed so far; finish=True = end of input (also emits the frame in progress, like the reference at EOF)."""
        n = max_frames or self._max_frames
        buf = (SondeFrame * n)()
        fn = lib().sonde_engine_finish if finish else lib().sonde_engine_fetch_frames
        k = _chk(fn(self._h, buf, n))
        frames = []
        line = C.create_string_buffer(1200)
        for i in range(k):
            f = buf[i]
            ll = lib().sonde_rs41_rawline(C.byref(f), line, 1200)
            frames.append(dict(channel=f.channel, len=f.len, ecc=f.ecc, mv=f.mv, mv_pos=f.mv_pos, nbytes=f.nbytes,
                               frame=bytes(f.frame), line=line.raw[:ll].decode()))
        if with_soft:
            soft = np.zeros((max(k, 1), self.nbits), np.float32)
            _chk(lib().sonde_engine_fetch_soft(self._h, soft.ctypes.data_as(C.c_void_p), k))
            for i in range(k):
                frames[i]["soft"] = soft[i].copy()
        return frames

    def fetch_dfm(self, finish: bool = False, with_soft: bool = False):
        """DFM engines: decoded frames (dicts with the `dfm09mod -r` text line); with_soft adds per-hit soft bits."""
        n = 8 * self._max_frames
        buf = (SondeDfmFrame * n)()
        k = _chk(lib().sonde_engine_fetch_dfm(self._h, buf, n, int(finish)))
        line = C.create_string_buffer(128)
        frames = []
        for i in range(k):
            f = buf[i]
            ll = lib().sonde_dfm_rawline(C.byref(f), self.ecc, line, 128)
            frames.append(dict(channel=f.channel, frame_in_hit=f.frame_in_hit, ecc=list(f.ecc), mv=f.mv, mv_pos=f.mv_pos,
                               conf=bytes(f.conf), dat1=bytes(f.dat1), dat2=bytes(f.dat2), frm_count=f.frm_count, inv=f.inv,
                               line=line.raw[:ll].decode()))
        if with_soft:
            soft = np.zeros((self._max_frames, self.nbits), np.float32)
            nh = _chk(lib().sonde_engine_fetch_soft(self._h, soft.ctypes.data_as(C.c_void_p), self._max_frames))
            return frames, soft[:nh]
        return frames

    def fetch_dfm_raw(self, finish: bool = False, lag: int = 0):
        """DFM engines: the decoded frames as a ctypes array of sonde_dfm_frame_t and their count — no per-frame Python work (the batch caller's form).
        lag = 1: only frames of calls before the latest one (sonde_engine_fetch_dfm_lagged)."""
        n = 8 * self._max_frames
        if getattr(self, "_dfmbuf", None) is None:
            self._dfmbuf = (SondeDfmFrame * n)()
        if lag and not finish:
            return self._dfmbuf, _chk(lib().sonde_engine_fetch_dfm_lagged(self._h, self._dfmbuf, n, lag))
        return self._dfmbuf, _chk(lib().sonde_engine_fetch_dfm(self._h, self._dfmbuf, n, int(finish)))

    def fetch_m10_raw(self, finish: bool = False, lag: int = 0):
        """M10 engines: the frames as a ctypes array of sonde_m10_frame_t and their count — no per-frame Python work."""
        n = 4 * self._max_frames
        if getattr(self, "_m10buf", None) is None:
            self._m10buf = (SondeM10Frame * n)()
        if lag and not finish:
            return self._m10buf, _chk(lib().sonde_engine_fetch_m10_lagged(self._h, self._m10buf, n, lag))
        return self._m10buf, _chk(lib().sonde_engine_fetch_m10(self._h, self._m10buf, n, int(finish)))

    def fetch_hits(self, finish: bool = False):
        """Function-level seam: header hits (score, position, polarity) with their soft bits, any sonde type (needs keep_soft)."""
        n = self._max_frames
        buf = (SondeHit * n)()
        k = _chk(lib().sonde_engine_fetch_hits(self._h, buf, n, int(finish)))
        soft = np.zeros((max(k, 1), self.nbits), np.float32)
        if k:
            _chk(lib().sonde_engine_fetch_soft(self._h, soft.ctypes.data_as(C.c_void_p), k))
        return [dict(channel=buf[i].channel, nbits=buf[i].nbits, mv=buf[i].mv, mv_pos=buf[i].mv_pos, soft=soft[i, :buf[i].nbits].copy()) for i in range(k)]

    def fetch_mxx(self, finish: bool = False, verbose: int = 1):
        """M10 / M20 engines: frames as dicts (bytes, checksum verdicts, the `m10mod -r [-v]` / `m20mod -r [-v]` text line)."""
        m20 = self.sonde == "m20"
        n = 4 * self._max_frames
        buf = ((SondeM20Frame if m20 else SondeM10Frame) * n)()
        L = lib()
        k = _chk((L.sonde_engine_fetch_m20 if m20 else L.sonde_engine_fetch_m10)(self._h, buf, n, int(finish)))
        line = C.create_string_buffer(420)
        frames = []
        for i in range(k):
            f = buf[i]
            ll = (L.sonde_m20_rawline if m20 else L.sonde_m10_rawline)(C.byref(f), verbose, line, 420)
            d = dict(channel=f.channel, nbits=f.nbits, len=f.len, cs_ok=f.cs_ok, cs_calc=f.cs_calc, mv=f.mv, mv_pos=f.mv_pos,
                     frame=bytes(f.frame), line=line.raw[:max(ll, 0)].decode())
            if m20:
                d.update(blk_ok=f.blk_ok, fw=f.fw)
            frames.append(d)
        return frames

    FRAME_DTYPE = np.dtype([("channel", "<i4"), ("len", "<i4"), ("ecc", "<i4"), ("mv_pos", "<u4"), ("mv", "<f4"),
                            ("nbytes", "<i4"), ("frame", "u1", (518,)), ("pad", "u1", (2,))])

    def fetch_frames_np(self, max_frames: int | None = None, lag: int = 0) -> np.ndarray:
        """Frames completed so far as one structured array (layout of sonde_frame_t); no per-frame Python work.
        lag = 1: only frames of calls before the latest one (lets the latest call's GPU work keep running)."""
        n = max_frames or self._max_frames
        if getattr(self, "_fbuf", None) is None or len(self._fbuf) < n:
            self._fbuf = np.zeros(n, self.FRAME_DTYPE)
        k = _chk(lib().sonde_engine_fetch_frames_lagged(self._h, self._fbuf.ctypes.data_as(C.POINTER(SondeFrame)), n, lag))
        return self._fbuf[:k].copy()

    def read_tap(self, channel: int, tap: int, first: int, count: int) -> np.ndarray:
        width = 2 if tap in (TAP_DECIM, TAP_IFIQ) else 1
        out = np.zeros((count, width), np.float32)
        _chk(lib().sonde_engine_read_tap(self._h, channel, tap, first, count, out.ctypes.data_as(C.c_void_p)))
        return out if width == 2 else out[:, 0]

    # -- profiling ---------------------------------------------------------------------------
    def profile(self, enable: bool | int = True):
        """True / 2: HIP events around every kernel; 1: around the dominant kernel only; False: none."""
        _chk(lib().sonde_engine_profile(self._h, 2 if enable is True else int(enable)))

    def kernel_ms(self, name: str):
        ms, n = C.c_double(), C.c_int64()
        _chk(lib().sonde_engine_kernel_ms(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @property
    def stream(self) -> int:
        return lib().sonde_engine_stream(self._h)


_KIND = {"rs41": SONDE_RS41, "dfm": SONDE_DFM09, "m10": SONDE_M10, "m20": SONDE_M20}
SONDE_MIXED = 100


class MixedEngine(Engine):
    """Mixed-type engine (sonde_engine_create_mixed): channel c carries a sonde of type kinds[c] ("rs41" / "dfm" / "m10" / "m20"); ONE decimator launch per call
    serves all channels, the IF-rate stages run per type.  ecc = {kind: level} (rs41mod --ecc2 = 2, dfm09mod --ecc = 1).  Frames come back per type:
    fetch_frames* (RS41), fetch_dfm* (DFM), fetch_m10* / fetch_mxx (M10 / M20), channel numbers are the caller's."""

    def __init__(self, fq, kinds, sample_rate: int, *, device: int = 0, lp_iq: bool = True, ecc: dict | None = None, thres: dict | None = None,
                 max_chunk: int | None = None, max_frames: int = 0, opt_min: bool = False, bits: int = 16, pipeline: bool = True):
        fq = np.atleast_1d(np.asarray(fq, dtype=np.float64))
        kinds = list(kinds)
        assert len(kinds) == len(fq)
        ecc = {"rs41": 2, "dfm": 1, "m10": 0, "m20": 0, **(ecc or {})}
        thres = thres or {}
        self.kinds = kinds
        self.group_kinds = list(dict.fromkeys(kinds))              # groups in order of first appearance
        self.n_channels = len(fq)
        self.sample_rate = sample_rate
        self.sonde = "mixed"
        self.ecc = ecc["dfm"]                                      # (Engine.fetch_dfm prints the raw line with it)
        self._ecc = ecc
        self._per_sample = 2
        self._dtype = {8: np.uint8}.get(bits, np.int16)
        cfg = SondeCfg(ABI_VERSION, device, self.n_channels, sample_rate, bits, SONDE_MIXED, LP_IQ if lp_iq else 0, 0, int(opt_min), 0, 0,
                       0.0, max_chunk or sample_rate, max_frames, 0, int(pipeline), 0, 1, 0, 0, 0, 0, 0, 0, 0, 0)
        groups = (SondeGroup * len(self.group_kinds))()
        for i, k in enumerate(self.group_kinds):
            groups[i] = SondeGroup(_KIND[k], ecc[k], 0, 0, 0, 0, float(thres.get(k, 0.0)), 0)
        gof = np.asarray([self.group_kinds.index(k) for k in kinds], dtype=np.int32)
        h = C.c_void_p()
        _chk(lib().sonde_engine_create_mixed(C.byref(cfg), fq.ctypes.data_as(C.POINTER(C.c_double)), groups, len(self.group_kinds),
                                              gof.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(h)))
        self._h = h
        info = SondeInfo()
        _chk(lib().sonde_engine_info(h, C.byref(info)))
        self.info = {n: getattr(info, n) for n, _ in SondeInfo._fields_ if n != "reserved"}
        self.nbits = 4080
        self._max_frames = max_frames or 4 * self.n_channels

    def group_info(self, channel: int):
        t, info = C.c_int32(), SondeInfo()
        _chk(lib().sonde_engine_group_info(self._h, channel, C.byref(t), C.byref(info)))
        return t.value, {n: getattr(info, n) for n, _ in SondeInfo._fields_ if n != "reserved"}

    def fetch_mxx(self, finish: bool = False, verbose: int = 1, m20: bool = False):
        self.sonde = "m20" if m20 else "m10"
        try:
            return Engine.fetch_mxx(self, finish=finish, verbose=verbose)
        finally:
            self.sonde = "mixed"
