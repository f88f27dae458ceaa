"""CPU-side checks of the C ABI: libsonde_hip.so loads and exports every entry point include/sonde_hip.h declares;
host-only helpers (RS codec, CRC, raw line) agree with the oracle.  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from radiosonde_auto_rx_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    return C.CDLL(engine.LIB_PATH)


def test_exports():
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))) if h.endswith(".h"))
    names = set(re.findall(r"\b(sonde_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 35 and "sonde_scan_create" in names and "sonde_fsk_create" in names
    L = _lib()
    for n in sorted(names):
        assert hasattr(L, n), n


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from radiosonde_auto_rx_amd.engine import Engine, SondeError
    with pytest.raises(SondeError):
        Engine([0.1], 2_400_000)


def test_scanner_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from radiosonde_auto_rx_amd.engine import SondeError
    from radiosonde_auto_rx_amd.scan import Scanner
    with pytest.raises(SondeError):
        Scanner(48000, n_channels=1, iq_mode=1)


def test_modem_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from radiosonde_auto_rx_amd.engine import SondeError
    from radiosonde_auto_rx_amd.fsk import FskModem
    with pytest.raises(SondeError):
        FskModem(48000, 4800)


def test_host_rs_codec_matches_oracle(oracle):
    L = _lib()
    O = oracle.lib()
    rng = np.random.default_rng(9)
    for trial in range(200):
        cw = np.zeros(255, np.uint8)
        cw[24:] = rng.integers(0, 256, 231, dtype=np.uint8)
        a = cw.copy(); b = cw.copy()
        L.sonde_rs255_encode(a.ctypes.data_as(C.c_void_p)); O.ora_rs255_encode(b.ctypes.data_as(C.c_void_p))
        assert (a == b).all()
        nerr = int(rng.integers(0, 18))
        pos = rng.choice(255, nerr, replace=False)
        a[pos] ^= rng.integers(1, 256, nerr, dtype=np.uint8)
        b = a.copy()
        ra = L.sonde_rs255_decode(a.ctypes.data_as(C.c_void_p))
        rb = O.ora_rs255_decode(b.ctypes.data_as(C.c_void_p), None, None)
        assert ra == rb and (a == b).all(), (trial, nerr, ra, rb)


def test_host_crc_kat():
    p = (C.c_ubyte * 17)(*([0] * 17))
    assert _lib().sonde_crc16(p, 17) == 0xC7EC


def test_generic_descriptor_is_validated_before_the_device_is_touched():
    """sonde_engine_create_generic: malformed descriptors are SONDE_E_ARG (no GPU needed to find out); a well-formed one gets as far as the
    device check (SONDE_E_NOGPU here) — the generic path has no CPU fallback either."""
    from radiosonde_auto_rx_amd.engine import SondeCfg, SondeGeneric, ABI_VERSION
    import torch
    E_ARG, E_NOGPU = -1, -2                                # SONDE_E_ARG, SONDE_E_NOGPU (include/sonde_hip.h)
    L = _lib()
    L.sonde_engine_create_generic.argtypes = [C.POINTER(SondeCfg), C.POINTER(C.c_double), C.POINTER(SondeGeneric), C.POINTER(C.c_void_p)]
    fq = (C.c_double * 1)(0.0)

    def create(sonde_type=99, gen=True, **kw):
        cfg = SondeCfg(abi_version=ABI_VERSION, n_channels=1, sample_rate=48000, bits=16, sonde_type=sonde_type, opt_lp=1, max_chunk=48000, keep_soft=1)
        d = dict(header=b"10101010101101001010110011010011", baud=2400.0, bt=1.2, h=2.4, symlen=1, symhd=1, hdmax=1, bitofs=0, nbits=1152)
        d.update(kw)
        g = SondeGeneric(**d)
        h = C.c_void_p()
        return L.sonde_engine_create_generic(C.byref(cfg), fq, C.byref(g) if gen else None, C.byref(h))

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert create() == E_NOGPU
    assert create(gen=False) == E_ARG                      # SONDE_GENERIC without a descriptor
    assert create(sonde_type=9) == E_NOGPU                 # a descriptor with a PRESET type: its baud replaces the preset's (--br); everything else is ignored
    assert create(sonde_type=9, baud=0.0) == E_ARG         # ... and must be a rate
    assert create(header=b"1010") == E_ARG                 # header shorter than 8 symbols
    assert create(baud=0.0) == E_ARG
    assert create(symlen=3) == E_ARG
    assert create(symlen=1, symhd=2) == E_ARG              # header symbols per bit cannot exceed the frame's
    assert create(nbits=9000) == E_ARG                     # more than the soft-bit buffers are sized for
    assert create(nbits=0) == E_ARG
