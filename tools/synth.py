"""Deterministic synthetic radiosonde captures (test + bench input generator).

The reference ships no sample captures; parity is pinned by
running the compiled reference next to the engine on captures made here.
Everything is seeded numpy -> identical bytes in this container and on the GPU box.

RS41 frame layout follows the block map in the reference
(demod/mod/rs41mod.c:336-392, header bytes :172, whitening mask :175-182,
RS(255,231) interleave :1729-1732, CRC-16/CCITT-FALSE :284-304).
Modulation: 4800 Bd GFSK, BT 0.5, +-2400 Hz deviation.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# GF(2^8), primitive polynomial 0x11D, alpha = 2  (bch_ecc_mod.h:83-89)
# --------------------------------------------------------------------------
_GF_EXP = np.zeros(512, dtype=np.int32)
_GF_LOG = np.zeros(256, dtype=np.int32)


def _gf_init() -> None:
    x = 1
    for i in range(255):
        _GF_EXP[i] = x
        _GF_LOG[x] = i
        x <<= 1
        if x & 0x100:
            x ^= 0x11D
    _GF_EXP[255:510] = _GF_EXP[0:255]


_gf_init()


def gf_mul(a: int, b: int) -> int:
    if a == 0 or b == 0:
        return 0
    return int(_GF_EXP[_GF_LOG[a] + _GF_LOG[b]])


def _rs_generator(nroots: int = 24) -> list[int]:
    # g(x) = prod_{i=0}^{23} (x - alpha^i), coefficient list low->high (b = 0)
    g = [1]
    for i in range(nroots):
        root = int(_GF_EXP[i])
        ng = [0] * (len(g) + 1)
        for j, c in enumerate(g):
            ng[j] ^= gf_mul(c, root)
            ng[j + 1] ^= c
        g = ng
    return g


_RS_G = _rs_generator()


def rs255_231_parity(msg231: bytes | np.ndarray) -> np.ndarray:
    """Parity p[0..23] so that cw = [p, msg] (cw[i] = coeff of x^i) is a codeword."""
    msg = [int(v) for v in msg231]
    assert len(msg) == 231
    rem = [0] * 24 + msg  # m(x) * 1 with zero low coefficients
    for d in range(254, 23, -1):  # long division by monic g, high -> low
        c = rem[d]
        if c:
            for j in range(25):
                rem[d - 24 + j] ^= gf_mul(_RS_G[j], c)
    return np.array(rem[:24], dtype=np.uint8)


def crc16_ccitt_false(data: bytes | np.ndarray) -> int:
    rem = 0xFFFF
    for b in bytes(data):
        rem ^= b << 8
        for _ in range(8):
            rem = ((rem << 1) ^ 0x1021) & 0xFFFF if rem & 0x8000 else (rem << 1) & 0xFFFF
    return rem


RS41_HEADER_BYTES = bytes([0x86, 0x35, 0xF4, 0x40, 0x93, 0xDF, 0x1A, 0x60])
RS41_MASK = bytes([
    0x96, 0x83, 0x3E, 0x51, 0xB1, 0x49, 0x08, 0x98, 0x32, 0x05, 0x59, 0x0E, 0xF9, 0x44, 0xC6, 0x26,
    0x21, 0x60, 0xC2, 0xEA, 0x79, 0x5D, 0x6D, 0xA1, 0x54, 0x69, 0x47, 0x0C, 0xDC, 0xE8, 0x5C, 0xF1,
    0xF7, 0x76, 0x82, 0x7F, 0x07, 0x99, 0xA2, 0x2C, 0x93, 0x7C, 0x30, 0x63, 0xF5, 0x10, 0x2E, 0x61,
    0xD0, 0xBC, 0xB4, 0xB6, 0x06, 0xAA, 0xF4, 0x23, 0x78, 0x6E, 0x3B, 0xAE, 0xBF, 0x7B, 0x4C, 0xC1])
RS41_FRAME_LEN = 320


def _put_block(frame: bytearray, pos: int, blk_id: int, payload: bytes) -> None:
    frame[pos] = blk_id
    frame[pos + 1] = len(payload)
    frame[pos + 2:pos + 2 + len(payload)] = payload
    crc = crc16_ccitt_false(payload)
    frame[pos + 2 + len(payload)] = crc & 0xFF
    frame[pos + 3 + len(payload)] = crc >> 8


def rs41_cal_table(seed: int = 1, typ: str = "RS41-SGP", rsm: str = "RSM421", freq_khz: int = 403010, fw: int = 0x4EF5,
                   burst_kill: int = 0, kill_timer: int = 0xFFFF, burst_timer: int = 0x7788) -> bytes:
    """A plausible 51 x 16 byte calibration / configuration table (the subframes an RS41 cycles through, one per frame):
    PTU coefficients as little-endian floats at the offsets the decoders read them from, frequency, firmware, timers,
    type strings, CRC-16 of bytes 2..799 in front.  Values are made up but sane (T around -40..+30 C, RH 0..100 %)."""
    import struct
    rng = np.random.default_rng(seed)
    c = bytearray(rng.integers(0, 256, 51 * 16, dtype=np.uint8).tobytes())

    def put(ofs, *vals):
        for k, v in enumerate(vals):
            c[ofs + 4 * k:ofs + 4 * k + 4] = struct.pack("<f", v)
    put(61, 750.0, 1100.0)                                   # reference resistors
    put(69, 0.0, 47.0)                                       # reference capacitors
    put(77, -243.911, 0.187654, 8.2e-06)                     # platinum polynomial, sensor 1
    put(89, 1.00312, -0.0213, 0.00021)                       # calibration T1
    put(117, 44.93, 5.02)                                    # humidity capacitance calibration
    put(125, *[float(v) for v in (rng.standard_normal(42) * np.repeat(10.0 ** -np.arange(7), 6) * 20.0)])
    put(293, -243.911, 0.187654, 8.2e-06)                    # platinum polynomial, humidity-sensor temperature
    put(305, 1.00177, 0.0153, -0.00013)
    put(606, *[float(v) for v in (rng.standard_normal(18) * 0.3)])
    put(630, 1.45)                                           # calP[24]: scale of the pressure ratio
    put(678, 0.021, -0.0042, 0.0011)                         # pressure correction of the humidity capacitance
    put(698, *[float(v) for v in (rng.standard_normal(12) * 0.05)])
    c[2] = (((freq_khz - 400000) % 40) // 10) << 6           # 10 kHz steps in the two top bits, 40 kHz steps in the next byte
    c[3] = (freq_khz - 400000) // 40
    c[0x10 + 5:0x10 + 7] = int(fw).to_bytes(2, "little")
    c[0x20 + 11] = burst_kill
    c[0x20 + 7:0x20 + 9] = int(kill_timer).to_bytes(2, "little")
    t9 = typ.encode("ascii")[:9].ljust(9, b"\0")
    c[0x210 + 8:0x210 + 16] = t9[:8]
    c[0x220] = t9[8]
    c[0x221] = 0
    c[0x222:0x22A] = rsm.encode("ascii")[:8].ljust(8, b"\0")
    c[0x310 + 6:0x310 + 8] = int(burst_timer).to_bytes(2, "little")
    c[0x320:0x322] = (0xFFFF).to_bytes(2, "little")          # countdown (variable subframe, outside the CRC)
    c[0:2] = crc16_ccitt_false(bytes(c[2:800])).to_bytes(2, "little")
    return bytes(c)


def rs41_ptu_counts(rng: np.random.Generator) -> bytes:
    """0x2A bytes of the measurement block: four sensors x (measurement, reference 1, reference 2) 24-bit counts, then the
    pressure-sensor temperature (int16, 1/100 C) and padding — in ranges that give finite physical values."""
    m = []
    for lo, hi in ((131000, 189000), (520000, 570000), (131000, 189000), (300000, 420000)):
        f1, f2 = 130000 + int(rng.integers(0, 500)), 190000 + int(rng.integers(0, 500))
        if lo >= 300000:
            f1, f2 = 290000 + int(rng.integers(0, 500)), 430000 + int(rng.integers(0, 500))
        if lo >= 500000:
            f1, f2 = 480000 + int(rng.integers(0, 500)), 560000 + int(rng.integers(0, 500))
        m += [int(rng.integers(lo, hi)), f1, f2]
    b = bytearray(0x2A)
    for k, v in enumerate(m):
        b[3 * k:3 * k + 3] = int(v).to_bytes(3, "little")
    b[36:38] = bytes(rng.integers(0, 256, 2, dtype=np.uint8))
    b[38:40] = int(rng.integers(-6000, 3000)).to_bytes(2, "little", signed=True)
    b[40:42] = bytes(rng.integers(0, 256, 2, dtype=np.uint8))
    return bytes(b)


def rs41_frame(frame_no: int, sonde_id: str = "S1234567", *, week: int = 2280,
               itow_ms: int | None = None, ecef_cm=(412345600, 61234500, 480123400),
               vel_cms=(123, -45, 510), nsats: int = 9, batt_dV: int = 27,
               rng: np.random.Generator | None = None, cal_table: bytes | None = None, ptu_counts: bool = False,
               xdata: list | None = None, gnss2: bool = False, corrupt_crc: int | None = None) -> bytes:
    """One valid RS41 frame with CRCs and RS parity: standard 320 bytes, or 518 bytes when xdata (list of ASCII strings, one
    0x7E block each) is given.  cal_table: the 51 x 16 table whose subframe (frame_no mod 51) the status block carries
    (default: random bytes); ptu_counts: physical measurement counts instead of random bytes in the PTU block;
    gnss2: the newer block layout (0x8226 position + UTC date/time, 0x8329 satellites) instead of the three u-blox 6 blocks;
    corrupt_crc: block id whose CRC is broken BEFORE the RS parity is computed (ECC passes, the block CRC does not)."""
    rng = rng or np.random.default_rng(frame_no)
    flen = 518 if xdata is not None else RS41_FRAME_LEN
    f = bytearray(flen)
    f[0:8] = RS41_HEADER_BYTES
    f[0x38] = 0xF0 if xdata is not None else 0x0F
    # 0x79 FRAME block, 0x28 bytes
    p = bytearray(0x28)
    p[0:2] = int(frame_no & 0xFFFF).to_bytes(2, "little")
    p[2:10] = sonde_id.encode("ascii")[:8].ljust(8, b"0")
    p[10] = batt_dV
    calidx = frame_no % 51
    p[0x52 - 0x3B] = calidx
    sub = bytes(rng.integers(0, 256, 16, dtype=np.uint8))
    p[0x53 - 0x3B:0x63 - 0x3B] = sub if cal_table is None else cal_table[16 * calidx:16 * calidx + 16]
    _put_block(f, 0x39, 0x79, bytes(p))
    # 0x7A PTU block, 0x2A bytes (opaque measurement counts unless ptu_counts)
    rnd = bytes(rng.integers(0, 256, 0x2A, dtype=np.uint8))
    _put_block(f, 0x65, 0x7A, rs41_ptu_counts(rng) if ptu_counts else rnd)
    if gnss2:
        import datetime
        t = datetime.datetime(2024, 5, 17, 11, 42, 7) + datetime.timedelta(seconds=frame_no)
        p = bytearray(0x26)
        for i in range(3):
            p[4 * i:4 * i + 4] = int(ecef_cm[i]).to_bytes(4, "little", signed=True)
            p[12 + 2 * i:14 + 2 * i] = int(vel_cms[i]).to_bytes(2, "little", signed=True)
        p[18:20] = t.year.to_bytes(2, "little"); p[20] = t.month; p[21] = t.day; p[22] = t.hour; p[23] = t.minute; p[24] = t.second
        p[25] = (7 * frame_no) % 100
        p[26:] = bytes(rng.integers(0, 256, 0x26 - 26, dtype=np.uint8))
        _put_block(f, 0x93, 0x82, bytes(p))
        q = bytearray(rng.integers(0, 256, 0x29, dtype=np.uint8).tobytes())
        q[4 + 21:4 + 21 + 16] = bytes([0x37 if k < 5 else 0x00 for k in range(16)])     # 10 satellites with a status nibble
        _put_block(f, 0xBD, 0x83, bytes(q))
        _put_block(f, 0xEA, 0x76, bytes(320 - 0xEA - 4))
        return _rs41_finish(f, corrupt_crc)
    # 0x7C GPS1, 0x1E bytes: week, iTOW, sats
    p = bytearray(rng.integers(0, 256, 0x1E, dtype=np.uint8).tobytes())
    p[0:2] = int(week).to_bytes(2, "little")
    if itow_ms is None:
        itow_ms = 1000 * (100000 + frame_no)
    p[2:6] = int(itow_ms).to_bytes(4, "little")
    _put_block(f, 0x93, 0x7C, bytes(p))
    # 0x7D GPS2, 0x59 bytes
    _put_block(f, 0xB5, 0x7D, bytes(rng.integers(0, 256, 0x59, dtype=np.uint8)))
    # 0x7B GPS3, 0x15 bytes: ECEF pos (cm), vel (cm/s), nSV
    p = bytearray(0x15)
    for i in range(3):
        p[4 * i:4 * i + 4] = int(ecef_cm[i]).to_bytes(4, "little", signed=True)
        p[12 + 2 * i:14 + 2 * i] = int(vel_cms[i]).to_bytes(2, "little", signed=True)
    p[18] = nsats
    p[19] = 10
    p[20] = 12
    _put_block(f, 0x112, 0x7B, bytes(p))
    if xdata is None:
        # 0x76 zero block, 0x11 bytes
        _put_block(f, 0x12B, 0x76, bytes(0x11))
        assert 0x12B + 2 + 0x11 + 2 == RS41_FRAME_LEN
    else:
        pos = 0x12B
        for k, xs in enumerate(xdata):                        # 0x7E blocks: instrument byte + ASCII payload
            pl = bytes([k]) + xs.encode("ascii")
            _put_block(f, pos, 0x7E, pl)
            pos += 2 + len(pl) + 2
        _put_block(f, pos, 0x76, bytes(518 - pos - 4))        # zero block up to the end of the long frame
    return _rs41_finish(f, corrupt_crc)


def _rs41_finish(f: bytearray, corrupt_crc: int | None = None) -> bytes:
    if corrupt_crc is not None:
        pos = 0x39
        while pos < len(f) - 4:
            ln = f[pos + 1]
            if f[pos] == corrupt_crc:
                f[pos + 2 + ln] ^= 0x5A
                break
            pos += ln + 4
    # two interleaved RS(255,231) codewords, message zero-padded beyond byte 320
    msg1 = np.zeros(231, dtype=np.uint8)
    msg2 = np.zeros(231, dtype=np.uint8)
    body = np.frombuffer(bytes(f[56:]), dtype=np.uint8)
    msg1[:len(body[0::2])] = body[0::2]
    msg2[:len(body[1::2])] = body[1::2]
    f[8:32] = rs255_231_parity(msg1).tobytes()
    f[32:56] = rs255_231_parity(msg2).tobytes()
    return bytes(f)


def rs41_onair_bits(frame: bytes, preamble_bytes: int = 40) -> np.ndarray:
    """Whitened, LSB-first bit stream with 0101.. preamble in front."""
    fr = np.frombuffer(frame, dtype=np.uint8)
    mask = np.frombuffer(RS41_MASK, dtype=np.uint8)
    x = fr ^ mask[np.arange(len(fr)) % 64]
    bits = np.unpackbits(x, bitorder="little")
    pre = np.tile(np.array([0, 1], dtype=np.uint8), preamble_bytes * 4)
    return np.concatenate([pre, bits])


# --------------------------------------------------------------------------
# GFSK modulator
# --------------------------------------------------------------------------
def _gauss_taps(sps: float, bt: float, span: int = 4) -> np.ndarray:
    n = int(round(span * sps)) | 1
    t = (np.arange(n) - (n - 1) / 2) / sps
    sigma = np.sqrt(np.log(2.0)) / (2 * np.pi * bt)
    g = np.exp(-0.5 * (t / sigma) ** 2)
    return g / g.sum()


def gfsk_baseband(bits: np.ndarray, sr: int, baud: float, dev_hz: float, bt: float = 0.5,
                  manchester: bool = False) -> np.ndarray:
    """Complex unit-amplitude GFSK burst at baseband (bit 1 -> +dev)."""
    from scipy.signal import oaconvolve
    if manchester:  # 1 -> 10, 0 -> 01 (caller passes symbols if it wants another map)
        sym = np.empty(2 * len(bits), dtype=np.uint8)
        sym[0::2] = bits
        sym[1::2] = 1 - bits
        bits = sym
    nrz = 2.0 * bits.astype(np.float64) - 1.0
    sps = sr / baud
    nsamp = int(round(len(bits) * sps))
    idx = np.minimum((np.arange(nsamp) / sps).astype(np.int64), len(bits) - 1)
    f = oaconvolve(nrz[idx], _gauss_taps(sps, bt), mode="same") * dev_hz
    phase = 2 * np.pi * np.cumsum(f) / sr
    return np.exp(1j * phase)


def rs41_capture(sr: int = 2_400_000, seconds: float = 2.2, fq: float = 0.1, *, n_frames: int | None = None,
                 first_frame_no: int = 1234, sonde_id: str = "S1234567", t_first: float = 0.15,
                 amp: float = 0.5, noise_sigma: float = 0.01, dc: complex = 0.0, seed: int = 1,
                 bit_errors: int = 0, dev_hz: float = 2400.0, f_offset_hz: float = 0.0,
                 return_frames: bool = False, frame_kw: dict | None = None):
    """Interleaved int16 IQ capture: one RS41 frame per second at carrier fq*sr (+f_offset_hz).

    bit_errors: number of random on-air bit flips per frame (inside the 320 data bytes) to
    exercise the RS decoder.  Returns int16 array of shape [2*N] (and the clean frames).
    """
    rng = np.random.default_rng(seed)
    n = int(round(sr * seconds))
    x = np.zeros(n, dtype=np.complex128)
    frames = []
    k = 0
    while True:
        t0 = t_first + k * 1.0
        if n_frames is not None and k >= n_frames:
            break
        frm = rs41_frame(first_frame_no + k, sonde_id, rng=np.random.default_rng(seed * 1000 + k), **(frame_kw or {}))
        bits = rs41_onair_bits(frm)
        if bit_errors:
            pos = rng.choice(np.arange(40 * 8 + 64, len(bits)), size=bit_errors, replace=False)
            bits = bits.copy()
            bits[pos] ^= 1
        burst = gfsk_baseband(bits, sr, 4800.0, dev_hz)
        s0 = int(round(t0 * sr))
        if s0 + len(burst) > n:
            break
        x[s0:s0 + len(burst)] += amp * burst
        frames.append(frm)
        k += 1
    if fq != 0.0 or f_offset_hz != 0.0:
        x *= np.exp(2j * np.pi * (fq + f_offset_hz / sr) * np.arange(n))
    x += dc
    x += noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, dtype=np.int16)
    out[0::2] = np.clip(np.round(x.real * 32767 * 0.9), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.round(x.imag * 32767 * 0.9), -32768, 32767).astype(np.int16)
    if return_frames:
        return out, frames
    return out


def snap_fq(fq: float, sr: int) -> float:
    """Snap a normalised carrier to a multiple of 16 Hz (the LUT periodicity of
    demod_mod.c:1265-1288) so the reference mixes with exactly this frequency."""
    hz = int(round(fq * sr / 16.0)) * 16
    return hz / sr


# --------------------------------------------------------------------------
# DFM09: 2500 Bd Manchester GFSK, 280-bit frames (16 header + 56 conf + 2 x 104 data), Hamming(8,4),
# bit-interleaved blocks (reference demod/mod/dfm09mod.c:141-142 header, :181-195 G/H, :231 deinterleave)
# --------------------------------------------------------------------------
DFM_HEADER_BITS = "0100010111001111"       # 0x45CF


def _hamming84(nib: int) -> list[int]:
    d = [(nib >> 3) & 1, (nib >> 2) & 1, (nib >> 1) & 1, nib & 1]          # big endian nibble
    return d + [d[1] ^ d[2] ^ d[3], d[0] ^ d[2] ^ d[3], d[0] ^ d[1] ^ d[3], d[0] ^ d[1] ^ d[2]]


def _dfm_block(nibbles) -> list[int]:
    L = len(nibbles)
    cw = [_hamming84(n) for n in nibbles]
    out = [0] * (8 * L)
    for j in range(8):                      # transmitted order: str[L*j + i] = bit j of codeword i
        for i in range(L):
            out[L * j + i] = cw[i][j]
    return out


def dfm_frame_bits(conf_nibbles, dat1_nibbles, dat2_nibbles) -> np.ndarray:
    assert len(conf_nibbles) == 7 and len(dat1_nibbles) == 13 and len(dat2_nibbles) == 13
    bits = [int(c) for c in DFM_HEADER_BITS] + _dfm_block(conf_nibbles) + _dfm_block(dat1_nibbles) + _dfm_block(dat2_nibbles)
    return np.array(bits, dtype=np.uint8)


def dfm_capture(sr: int = 480_000, seconds: float = 3.0, fq: float = 0.0, *, amp: float = 0.5, noise_sigma: float = 0.01,
                seed: int = 1, bit_errors_per_frame: int = 0, dev_hz: float = 2400.0, t_first: float = 0.1,
                return_frames: bool = False):
    """Continuous DFM09 transmission starting at t_first: back-to-back 280-bit frames, Manchester (1 -> 01, 0 -> 10)."""
    rng = np.random.default_rng(seed)
    n = int(round(sr * seconds))
    nfr = int((seconds - t_first) * 2500.0 / 560.0)
    frames, allbits = [], []
    for k in range(nfr):
        conf = [int(v) for v in rng.integers(0, 16, 7)]
        d1 = [int(v) for v in rng.integers(0, 16, 13)]
        d2 = [int(v) for v in rng.integers(0, 16, 13)]
        d1[12] = k % 9                       # data block id nibble
        fb = dfm_frame_bits(conf, d1, d2)
        if bit_errors_per_frame:
            pos = rng.choice(np.arange(16, 280), size=bit_errors_per_frame, replace=False)
            fb = fb.copy(); fb[pos] ^= 1
        frames.append((conf, d1, d2))
        allbits.append(fb)
    bits = np.concatenate(allbits)
    sym = np.empty(2 * len(bits), dtype=np.uint8)
    sym[0::2] = 1 - bits                     # bit 1 -> symbols 0,1 ; bit 0 -> symbols 1,0
    sym[1::2] = bits
    burst = gfsk_baseband(sym, sr, 2500.0, dev_hz, bt=0.5)
    x = np.zeros(n, dtype=np.complex128)
    s0 = int(round(t_first * sr))
    m = min(len(burst), n - s0)
    x[s0:s0 + m] = amp * burst[:m]
    if fq != 0.0:
        x *= np.exp(2j * np.pi * fq * np.arange(n))
    x += noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, dtype=np.int16)
    out[0::2] = np.clip(np.round(x.real * 32767 * 0.9), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.round(x.imag * 32767 * 0.9), -32768, 32767).astype(np.int16)
    if return_frames:
        return out, frames
    return out


# --------------------------------------------------------------------------
# M10 / M20 (scanner tests): raw-symbol header + differential-Manchester type bytes
# --------------------------------------------------------------------------
M10_RAWHEADER = "1001100110010100110010011001" "1010"   # scan/dft_detect.c:72-74, last 4 symbols = first two frame bits


def m10_checksum(data: bytes) -> int:
    """checkM10 of the M10 / M20 frames: 16-bit register, one step per byte"""
    c = 0
    for b in data:
        b = ((b >> 1) | ((b & 1) << 7)) & 0xFF
        b ^= (b >> 2) & 0xFF
        t6 = (c & 1) ^ ((c >> 2) & 1) ^ ((c >> 4) & 1)
        t7 = ((c >> 1) & 1) ^ ((c >> 3) & 1) ^ ((c >> 5) & 1)
        t = (c & 0x3F) | (t6 << 6) | (t7 << 7)
        sreg = (c >> 7) & 0xFF
        sreg ^= (sreg >> 2) & 0xFF
        c = (((c & 0xFF) << 8) | ((b ^ t ^ sreg) & 0xFF)) & 0xFFFF
    return c


def m10_frame(k: int = 0, *, gtop: bool = False, lat=48.1, lon=11.6, alt_m=1234.5, sn=(0x84, 0x12, 0x3A, 0x45, 0x67), rng=None,
              good_checksum: bool = True) -> bytes:
    """One 101-byte M10 (Trimble GPS, type 0x9F) or M10+ (Gtop GPS, 0xAF) frame with plausible sensor words and a valid checksum."""
    rng = rng or np.random.default_rng(k)
    f = bytearray(rng.integers(0, 256, 101, dtype=np.uint8).tobytes())
    f[0] = 0x64; f[1] = 0xAF if gtop else 0x9F
    if gtop:
        f[0x04:0x08] = int(round((lat + 1e-4 * k) * 1e6)).to_bytes(4, "big", signed=True)
        f[0x08:0x0C] = int(round((lon - 2e-4 * k) * 1e6)).to_bytes(4, "big", signed=True)
        f[0x0C:0x0F] = (int(round((alt_m + 5.2 * k) * 100)) & 0xFFFFFF).to_bytes(3, "big")
        for p, v in ((0x0F, 321 + k), (0x11, -1234 + 3 * k), (0x13, 498)):
            f[p:p + 2] = int(v).to_bytes(2, "big", signed=True)
        f[0x15:0x18] = (114207 + k).to_bytes(3, "big")
        f[0x18:0x1B] = (170524).to_bytes(3, "big")
    else:
        f[2] = 0x20
        for p, v in ((0x04, 640 + 5 * k), (0x06, -2468 + 7 * k), (0x08, 996)):
            f[p:p + 2] = int(v).to_bytes(2, "big", signed=True)
        f[0x0A:0x0E] = (4 * 86400_000 + 42_127_250 + 1000 * k).to_bytes(4, "big")
        unit = (1 << 30) / 90.0
        f[0x0E:0x12] = int(round((lat + 1e-4 * k) * unit)).to_bytes(4, "big", signed=True)
        f[0x12:0x16] = int(round((lon - 2e-4 * k) * unit)).to_bytes(4, "big", signed=True)
        f[0x16:0x1A] = int(round((alt_m + 5.2 * k) * 1000)).to_bytes(4, "big", signed=True)
        f[0x1E] = 9; f[0x1F] = 18
        f[0x20:0x22] = (2314 - 2048).to_bytes(2, "big")             # week number after the 10-bit rollovers
    f[0x32:0x35] = (14_100_000 + 1000 * k).to_bytes(3, "little")    # reference capacitance count x 1000
    f[0x35:0x38] = (13_500_000 + 9000 * k).to_bytes(3, "little")    # humidity capacitance count x 1000
    f[0x3E] = k % 3                                                  # thermistor range
    f[0x3F:0x41] = (0xA000 | (900 + 310 * (k % 9))).to_bytes(2, "little")
    f[0x45:0x47] = (655 + k % 5).to_bytes(2, "little")              # battery ADC
    f[0x48:0x4A] = (2790 + 3 * k).to_bytes(2, "little")             # MCU temperature diode
    f[0x59:0x5B] = (1500 + 40 * (k % 7)).to_bytes(2, "little")      # second NTC
    f[0x5D:0x62] = bytes(sn)
    f[0x62] = k & 0xFF
    cs = m10_checksum(bytes(f[:99]))
    if not good_checksum:
        cs ^= 0x0101
    f[99] = cs >> 8; f[100] = cs & 0xFF
    return bytes(f)


def m20_frame(k: int = 0, *, fw: int = 6, lat=52.2, lon=-4.3, alt_m=2345.6, sn24: int = (1 << 23) | (4321 << 10) | (2 << 7) | 101,
              rng=None, good_checksum: bool = True, blk: str = "ok", week: int = 2314, pressure_hpa: float = 0.0) -> bytes:
    """One 70-byte M20 frame (type 0x20): sensor words, the 0x16-byte block with its own check word (blk = "ok" | "zero" | "bad";
    firmware >= 7 reuses the low check byte as pressure LSB) and the frame checksum over bytes 0..0x43."""
    rng = rng or np.random.default_rng(k)
    f = bytearray(rng.integers(0, 256, 70, dtype=np.uint8).tobytes())
    f[0] = 0x45; f[1] = 0x20
    f[0x02:0x04] = (30100 + 850 * (k % 11)).to_bytes(2, "little")    # humidity capacitance word
    f[0x04:0x06] = ((k % 3) * 4096 + 700 + 290 * (k % 10)).to_bytes(2, "little")   # thermistor ADC incl. range bits
    f[0x06:0x08] = (1400 + 55 * (k % 13)).to_bytes(2, "little")      # NTC on the humidity sensor
    f[0x08:0x0B] = (int(round((alt_m + 4.7 * k) * 100)) & 0xFFFFFF).to_bytes(3, "big")
    for p, v in ((0x0B, 512 + 9 * k), (0x0D, -1777 + 5 * k), (0x18, 503 - k)):
        f[p:p + 2] = int(v).to_bytes(2, "big", signed=True)
    f[0x0F:0x12] = (3 * 86400 + 51_723 + k).to_bytes(3, "big")
    f[0x12:0x15] = int(sn24).to_bytes(3, "little")
    f[0x15] = (17 + k) & 0xFF
    f[0x1A:0x1C] = int(week).to_bytes(2, "big")                      # < 1304 gets one 10-bit rollover added by the decoder
    f[0x1C:0x20] = int(round((lat + 1e-4 * k) * 1e6)).to_bytes(4, "big", signed=True)
    f[0x20:0x24] = int(round((lon - 2e-4 * k) * 1e6)).to_bytes(4, "big", signed=True)
    pv = int(round(pressure_hpa * 4096))
    f[0x24:0x26] = ((pv >> 8) & 0xFFFF).to_bytes(2, "little")
    f[0x26] = 198 + k % 20                                           # battery
    f[0x2F:0x31] = (31000 + 3 * k).to_bytes(2, "little")             # humidity calibration word
    f[0x43] = fw
    bc = m10_checksum(bytes([0x16]) + bytes(f[2:2 + 0x14]))
    if blk == "zero":
        bc = 0
    elif blk == "bad":
        bc ^= 0x1010
    f[0x16] = bc >> 8; f[0x17] = bc & 0xFF
    if fw >= 7:
        f[0x16] = pv & 0xFF
    cs = m10_checksum(bytes(f[:0x44]))
    if not good_checksum:
        cs ^= 0x0101
    f[0x44] = cs >> 8; f[0x45] = cs & 0xFF
    return bytes(f)


def m10_symbols(type_bytes=(0x64, 0x9F), n_payload_bytes: int = 99, rng=None, data: bytes | None = None) -> np.ndarray:
    """Raw 2-FSK symbols of one M10-style frame: 1001.. preamble, the 32-symbol header, then the frame bytes as
    Manchester pairs whose first symbol repeats the previous pair's for a 1 and flips for a 0 (what frm_M10 of
    dft_detect.c:932-977 undoes).  The header's last two pairs already carry the first two bits of byte 0."""
    rng = rng or np.random.default_rng(0)
    if data is None:
        data = bytes(type_bytes) + bytes(int(v) for v in rng.integers(0, 256, n_payload_bytes))
    bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))          # MSB first
    sym = [int(c) for c in "1001" * 12] + [int(c) for c in M10_RAWHEADER]
    prev = int(M10_RAWHEADER[30])
    for b in bits[2:]:
        s = prev if b else 1 - prev
        sym += [s, 1 - s]
        prev = s
    return np.array(sym, dtype=np.uint8)


def m10_capture(sr: int = 48_000, seconds: float = 3.0, fq: float = 0.0, *, type_bytes=(0x64, 0x9F), baud: float = 9616.0,
                amp: float = 0.5, noise_sigma: float = 0.01, seed: int = 1, dev_hz: float = 3300.0, t_first: float = 0.35,
                period: float = 1.0, f_offset_hz: float = 0.0, frame_fn=None) -> np.ndarray:
    """Interleaved int16 IQ: continuous carrier, one M10/M20-style frame per `period` seconds, 1001.. idle pattern between.
    frame_fn(k) -> frame bytes of the k-th frame (default: random payload behind type_bytes)."""
    rng = np.random.default_rng(seed)
    n = int(round(sr * seconds))
    nsym = int(seconds * baud) + 8
    sym = np.tile(np.array([1, 0, 0, 1], dtype=np.uint8), nsym // 4 + 1)[:nsym]
    k = 0
    while True:
        s0 = int(round((t_first + k * period) * baud)) // 4 * 4
        fr = m10_symbols(type_bytes, rng=np.random.default_rng(seed * 77 + k), data=frame_fn(k) if frame_fn else None)
        if s0 + len(fr) > nsym:
            break
        sym[s0:s0 + len(fr)] = fr
        k += 1
    x = amp * gfsk_baseband(sym, sr, baud, dev_hz, bt=1.0)[:n]
    if len(x) < n:
        x = np.concatenate([x, np.zeros(n - len(x))])
    if fq != 0.0 or f_offset_hz != 0.0:
        x = x * np.exp(2j * np.pi * (fq + f_offset_hz / sr) * np.arange(n))
    x = x + noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, dtype=np.int16)
    out[0::2] = np.clip(np.round(x.real * 32767 * 0.9), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.round(x.imag * 32767 * 0.9), -32768, 32767).astype(np.int16)
    return out


# Other frame-based 2-FSK sondes of the reference's demod/mod family (seam tests): raw header + random payload.  `header` is the raw-symbol
# string the decoder hands to the demodulator (rs92mod.c:88-92, imet54mod.c:90-95, mp3h1mod.c:118, mts01mod.c:49-50, meisei100mod.c:200-201).
FAMILY = {
    "rs92mod": dict(baud=4800.0, dev_hz=2400.0, bt=0.5, manchester=True, nbits=2340,
                    header="10100110011001101001" "1010011001100110100110101010100110101001"),
    "imet54mod": dict(baud=4798.0, dev_hz=2000.0, bt=1.0, manchester=False, nbits=2200, header="0000000001" "0101010101" "0001001001" "0001001001"),
    "mp3h1mod": dict(baud=2399.0, dev_hz=2400.0, bt=1.0, manchester=True, nbits=386, header="100110011001100110011001100110011001" "10101010"),
    "mts01mod": dict(baud=1200.0, dev_hz=640.0, bt=1.5, manchester=False, nbits=1048, header="10101010" "10101010" "10110100" "00101011"),
    "meisei100mod": dict(baud=2400.0, dev_hz=2900.0, bt=1.2, manchester=False, nbits=1152, header="101010101011010100101011001101001100101011001101"),
}


def family_capture(name: str, sr: int = 48_000, seconds: float = 4.3, fq: float = 0.0, *, amp: float = 0.5, noise_sigma: float = 0.02, seed: int = 1,
                   t_first: float = 0.3, period: float = 1.0, f_offset_hz: float = 0.0, invert: bool = False) -> np.ndarray:
    """Interleaved int16 IQ: carrier with an alternating idle pattern, every `period` seconds the decoder's raw header followed by random
    payload bits (Manchester pairs where the sonde uses them).  The payload is not a valid frame: the decoders' raw output prints it anyway."""
    f = FAMILY[name]
    rng = np.random.default_rng(seed)
    baud = f["baud"]
    n = int(round(sr * seconds))
    nsym = int(seconds * baud) + 8
    sym = np.tile(np.array([1, 0], dtype=np.uint8), nsym // 2 + 1)[:nsym]
    hdr = np.array([int(c) for c in f["header"]], dtype=np.uint8)
    k = 0
    while True:
        s0 = int(round((t_first + k * period) * baud)) // 2 * 2
        bits = rng.integers(0, 2, f["nbits"] + 40, dtype=np.uint8)
        if f["manchester"]:
            pay = np.empty(2 * len(bits), np.uint8); pay[0::2] = bits; pay[1::2] = 1 - bits
        else:
            pay = bits
        fr = np.concatenate([hdr, pay])
        if s0 + len(fr) > nsym:
            break
        sym[s0:s0 + len(fr)] = fr
        k += 1
    if invert:
        sym = 1 - sym
    x = amp * gfsk_baseband(sym, sr, baud, f["dev_hz"], bt=f["bt"])[:n]
    if len(x) < n:
        x = np.concatenate([x, np.zeros(n - len(x))])
    if fq != 0.0 or f_offset_hz != 0.0:
        x = x * np.exp(2j * np.pi * (fq + f_offset_hz / sr) * np.arange(n))
    x = x + noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, dtype=np.int16)
    out[0::2] = np.clip(np.round(x.real * 32767 * 0.9), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.round(x.imag * 32767 * 0.9), -32768, 32767).astype(np.int16)
    return out


def fm_audio(iq: np.ndarray, gain: float = 0.25) -> np.ndarray:
    """FM-discriminator audio (int16 mono) of an interleaved int16 IQ capture at the same rate — the kind of input
    the reference's FM chain feeds to the decoders / dft_detect as WAV."""
    z = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    w = z[1:] * np.conj(z[:-1])
    s = np.concatenate([[0.0], np.angle(w) / np.pi])
    return np.clip(np.round(s * gain * 32767 * 4), -32768, 32767).astype(np.int16)


def to_u8(x: np.ndarray) -> np.ndarray:
    """int16 samples -> 8-bit unsigned (rtl_sdr's cu8 / 8-bit WAV): the top byte around 128."""
    return np.clip((np.asarray(x).astype(np.int32) >> 8) + 128, 0, 255).astype(np.uint8)


def to_f32(x: np.ndarray, scale: float = 0.73) -> np.ndarray:
    """int16 samples -> float32 in [-1, 1) times a factor that makes the values non-representable as int16 / 32768."""
    return (np.asarray(x).astype(np.float32) * np.float32(scale / 32768.0)).astype(np.float32)


def wav_bytes(pcm: np.ndarray, sr: int, nch: int = 1, bits: int = 16) -> bytes:
    """Minimal RIFF/WAVE container (16-bit PCM, or 8-bit unsigned from uint8 samples) around interleaved samples."""
    import struct
    data = np.ascontiguousarray(pcm, dtype={8: np.uint8, 16: "<i2", 32: "<f4"}[bits]).tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, nch, sr, sr * nch * bits // 8, nch * bits // 8, bits)
    return hdr + b"data" + struct.pack("<I", len(data)) + data


def wideband_capture(sr: int, seconds: float, signals, *, noise_sigma: float = 0.01, seed: int = 1) -> np.ndarray:
    """One wideband int16 IQ stream carrying several sondes (BASELINE config 3): signals = list of dicts
    {kind: "rs41"|"dfm"|"m10", fq, t_first, amp, ...}; each burst is generated at baseband and shifted to fq*sr."""
    rng = np.random.default_rng(seed)
    n = int(round(sr * seconds))
    x = np.zeros(n, dtype=np.complex128)
    for k, sg in enumerate(signals):
        kind, fq, t0, amp = sg["kind"], sg["fq"], sg.get("t_first", 0.05), sg.get("amp", 0.1)
        if kind == "rs41":
            bits = rs41_onair_bits(rs41_frame(1000 + k, "W%07d" % k, rng=np.random.default_rng(seed * 31 + k)))
            b = gfsk_baseband(bits, sr, 4800.0, sg.get("dev_hz", 2400.0))
        elif kind == "dfm":
            r2 = np.random.default_rng(seed * 37 + k)
            fb = np.concatenate([dfm_frame_bits([int(v) for v in r2.integers(0, 16, 7)], [int(v) for v in r2.integers(0, 16, 13)],
                                                [int(v) for v in r2.integers(0, 16, 13)]) for _ in range(int(seconds * 2500 / 560) + 1)])
            sym = np.empty(2 * len(fb), dtype=np.uint8); sym[0::2] = 1 - fb; sym[1::2] = fb
            b = gfsk_baseband(sym, sr, 2500.0, 2400.0, bt=0.5)
        else:
            nsym = int((seconds - t0) * 9616.0)
            sym = np.tile(np.array([1, 0, 0, 1], dtype=np.uint8), nsym // 4 + 1)[:nsym]
            fr = m10_symbols(sg.get("type_bytes", (0x64, 0x9F)), rng=np.random.default_rng(seed * 41 + k))
            s0 = int(0.02 * 9616) // 4 * 4
            sym[s0:s0 + len(fr)] = fr[:max(0, nsym - s0)]
            b = gfsk_baseband(sym, sr, 9616.0, 3300.0, bt=1.0)
        s0 = int(round(t0 * sr))
        m = min(len(b), n - s0)
        x[s0:s0 + m] += amp * b[:m] * np.exp(2j * np.pi * fq * np.arange(s0, s0 + m))
    x += noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, dtype=np.int16)
    out[0::2] = np.clip(np.round(x.real * 32767 * 0.9), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.round(x.imag * 32767 * 0.9), -32768, 32767).astype(np.int16)
    return out


def imet_capture(sr: int = 48_000, seconds: float = 3.0, *, f_offset_hz: float = 0.0, amp: float = 0.5, noise_sigma: float = 0.02,
                 seed: int = 1, dev_hz: float = 3000.0, t_first: float = 0.3, space_hz: float = 2200.0) -> np.ndarray:
    """iMet-style AFSK on FM (Bell 202: mark 1200 Hz, space `space_hz`): carrier with a 1200 Hz preamble tone, then 1200 Bd
    data alternating between mark and space tones.  What dft_detect's IMETafsk template + spectrum check look for
    (scan/dft_detect.c:103-107,1533-1607)."""
    rng = np.random.default_rng(seed)
    n = int(round(sr * seconds))
    t = np.arange(n) / sr
    tone = np.full(n, 1200.0)
    s0 = int(round(t_first * sr))
    bits = rng.integers(0, 2, int((seconds - t_first - 0.25) * 1200))
    pre = int(0.25 * sr)                                   # 0.25 s of mark tone (preamble), then data
    spb = sr / 1200.0
    idx = np.minimum(((np.arange(n - s0 - pre)) / spb).astype(int), len(bits) - 1)
    tone[s0 + pre:] = np.where(bits[idx] == 1, 1200.0, space_hz)
    audio = np.sign(np.sin(2 * np.pi * np.cumsum(tone) / sr))          # square-ish AFSK audio as the FM modulator
    audio[:s0] = 0.0
    phase = 2 * np.pi * np.cumsum(dev_hz * audio + f_offset_hz) / sr
    x = amp * np.exp(1j * phase) + noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, dtype=np.int16)
    out[0::2] = np.clip(np.round(x.real * 32767 * 0.9), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.round(x.imag * 32767 * 0.9), -32768, 32767).astype(np.int16)
    return out


def mfsk_capture(bits: np.ndarray, sr: int, baud: int, M: int = 2, *, f_low: float = 1500.0, shift: float = 400.0, amp: float = 0.4,
                 noise_sigma: float = 0.02, seed: int = 1) -> np.ndarray:
    """Continuous-phase M-FSK (M = 2 or 4) as interleaved complex int16: symbol m = tone f_low + m * shift, MSB first for 4-FSK
    (the bit order of the reference's fsk_mod, utils/fsk.c:302-312)."""
    bits = np.asarray(bits, dtype=np.int64)
    if M == 4:
        sym = 2 * bits[0::2][:len(bits) // 2] + bits[1::2][:len(bits) // 2]
    else:
        sym = bits
    sps = sr // baud
    f = np.repeat(f_low + shift * sym, sps).astype(np.float64)
    ph = 2 * np.pi * np.cumsum(f) / sr
    rng = np.random.default_rng(seed)
    z = amp * np.exp(1j * ph) + noise_sigma * (rng.standard_normal(len(ph)) + 1j * rng.standard_normal(len(ph)))
    out = np.empty(2 * len(z), np.int16)
    out[0::2] = np.clip(np.round(z.real * 32767), -32768, 32767)
    out[1::2] = np.clip(np.round(z.imag * 32767), -32768, 32767)
    return out


def fsk_test_frame_bits(n_frames: int) -> np.ndarray:
    """The 100-bit test frame of the reference's fsk_demod --testframes (utils/fsk_demod.c:30,247-251: srand(158324), rand() & 1 from
    the C library), repeated."""
    import ctypes
    import ctypes.util
    libc = ctypes.CDLL(ctypes.util.find_library("c") or "libc.so.6")
    libc.srand(158324)
    frame = np.array([libc.rand() & 1 for _ in range(100)], dtype=np.int64)
    return np.tile(frame, n_frames)


# ---------------------------------------------------------------- LMS6 (CCSDS-style: RS(255,223) blocks behind a K = 7 rate-1/2 convolutional code)
_GF187 = None


def _gf187():
    """GF(2^8) / 0x187 exp / log tables (the CCSDS field of the reference's RS(255,223), bch_ecc_mod.h:99)"""
    global _GF187
    if _GF187 is None:
        exp, log = [0] * 512, [0] * 256
        x = 1
        for i in range(255):
            exp[i] = x; log[x] = i
            x <<= 1
            if x & 0x100:
                x ^= 0x187
        for i in range(255, 512):
            exp[i] = exp[i - 255]
        _GF187 = (exp, log)
    return _GF187


def rs255_223_ccsds_parity(msg223) -> np.ndarray:
    """32 parity bytes of the reference's RS(255,223) (generator roots alpha^(11 (112 + i)), i = 0..31; codeword = parity[0..31] | message[0..222],
    lowest degree first, like rs_encode in bch_ecc_mod.c:832)"""
    exp, log = _gf187()

    def mul(a, b):
        return 0 if a == 0 or b == 0 else exp[log[a] + log[b]]
    g = [1]
    for i in range(32):
        root = exp[(11 * (112 + i)) % 255]
        ng = [0] * (len(g) + 1)
        for k, c in enumerate(g):                # g(x) * (x + root), coefficients lowest degree first
            ng[k] ^= mul(c, root)
            ng[k + 1] ^= c
        g = ng
    rem = [0] * 32                               # remainder of x^32 m(x) mod g(x), highest message coefficient first
    for m in reversed([int(v) for v in msg223]):
        fb = m ^ rem[31]
        rem = [0] + rem[:31]
        if fb:
            for k in range(32):
                rem[k] ^= mul(fb, g[k])
    return np.array(rem, dtype=np.uint8)


def lms6_frame(k: int = 0, *, sn: int = 8123456, lat=47.5, lon=8.7, alt_m=12345.6, tow_ms: int = 3 * 86400_000 + 12 * 3600_000) -> bytes:
    """One 223-byte LMS6 data frame (lms6Xmod.c:446-460 field map: 24 54 00 00 | SN u32 | frame nr u16 | GPS tow ms u32 | .. lat / lon i32 deg 2^31/180..
    | alt i32 mm | vE vN vU i24 | .. | CRC16 (poly 0x1021, init 0) over the first 221 bytes, big endian)"""
    f = bytearray(223)
    f[0:4] = bytes([0x24, 0x54, 0x00, 0x00])
    f[4:8] = int(sn).to_bytes(4, "big")
    f[8:10] = int(100 + k).to_bytes(2, "big")
    f[10:14] = int(tow_ms + 1000 * k).to_bytes(4, "big")
    f[18:22] = int(round(lat * (2 ** 31) / 180.0)).to_bytes(4, "big", signed=True)
    f[22:26] = int(round(lon * (2 ** 31) / 180.0)).to_bytes(4, "big", signed=True)
    f[26:30] = int(round((alt_m + 5.0 * k) * 1000)).to_bytes(4, "big", signed=True)
    for j, v in enumerate((310, -120, 480)):
        f[30 + 3 * j:33 + 3 * j] = int(v).to_bytes(3, "big", signed=True)
    rng = np.random.default_rng(1000 + k)
    f[39:221] = rng.integers(0, 256, 221 - 39, dtype=np.uint8).tobytes()
    crc = 0
    for b in f[:221]:
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    f[221:223] = crc.to_bytes(2, "big")
    return bytes(f)


def lmsx_frame(k: int = 0, **kw) -> bytes:
    """LMS-X variant of lms6_frame(): frame sync 24 46 05 00 (lms6Xmod.c:109), same CRC position"""
    f = bytearray(lms6_frame(k, **kw))
    f[0:4] = bytes([0x24, 0x46, 0x05, 0x00])
    crc = 0
    for b in f[:221]:
        crc ^= b << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    f[221:223] = crc.to_bytes(2, "big")
    return bytes(f)


def lms6_onair_bits(n_blocks: int, lmsx: bool = False) -> np.ndarray:
    """Raw channel bits of n_blocks consecutive LMS6 blocks: [00 58 f3 3f b8 | 223-byte frame | 32 RS parity] bytes, LSB first, through the
    K = 7 rate-1/2 code of lms6Xmod.c:116-117 (c0 from 1001111, c1 from 1101101, oldest bit first), every second channel bit inverted."""
    data = []
    for k in range(n_blocks):
        fr = np.frombuffer(lmsx_frame(k) if lmsx else lms6_frame(k), np.uint8)
        par = rs255_223_ccsds_parity(fr[::-1])          # rs_cw[254 - j] = block byte j: the first frame byte is the highest coefficient
        data += [0x00, 0x58, 0xF3, 0x3F, 0xB8] + list(fr) + list(par[::-1])
        if lmsx:
            data += [0] * 40                            # LMS-X: one block per 300 bytes (RAWBITBLOCK_LEN, lms6Xmod.c:91)
    bits = np.unpackbits(np.array(data, np.uint8)[:, None], axis=1, bitorder="little").ravel().astype(np.int64)
    pa = np.array([1, 0, 0, 1, 1, 1, 1]); pb = np.array([1, 1, 0, 1, 1, 0, 1])
    hist = np.concatenate([np.zeros(6, np.int64), bits])
    win = np.lib.stride_tricks.sliding_window_view(hist, 7)          # win[n] = bits n-6 .. n, oldest first
    c0 = (win @ pa) & 1
    c1 = ((win @ pb) & 1) ^ 1
    return np.stack([c0, c1], axis=1).ravel().astype(np.uint8)


def lms6_capture(sr: int = 48_000, seconds: float = 4.0, fq: float = 0.0, *, amp: float = 0.5, noise_sigma: float = 0.02, seed: int = 1,
                 baud: float = 4800.0, lmsx: bool = False) -> np.ndarray:
    """LMS6-403 GFSK capture (h = 0.9, BT = 1.2 as the decoder assumes, lms6Xmod.c:1274-1275), continuous blocks from t = 0.
    lmsx: 300-byte blocks with the LMS-X frame sync (give baud = 4797.8)."""
    n_blocks = int(seconds * baud / ((300 if lmsx else 260) * 16)) + 2
    bits = lms6_onair_bits(n_blocks, lmsx)
    n = int(seconds * sr)
    z = gfsk_baseband(bits, sr, baud, dev_hz=0.9 * baud / 2, bt=1.2)[:n]
    if len(z) < n:
        z = np.concatenate([z, np.zeros(n - len(z), z.dtype)])
    t = np.arange(n)
    rng = np.random.default_rng(seed)
    z = amp * z * np.exp(2j * np.pi * fq * t) + noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, np.int16)
    out[0::2] = np.clip(np.round(z.real * 32767), -32768, 32767)
    out[1::2] = np.clip(np.round(z.imag * 32767), -32768, 32767)
    return out


# ---------------------------------------------------------------------------------------------------------------- Meisei iMS-100 / RS-11G
def _bch46(msg34) -> list:
    """46 transmitted bits of one block: 34 message bits + 12 check bits of the BCH(63,51) code shortened to (46,34), generator
    x^12+x^10+x^8+x^5+x^4+x^3+1 (meisei100mod.c:96-98); transmitted bit j is the coefficient of x^(45-j) (:738)."""
    g = 0b1010100111001                                  # x^12 .. x^0
    r = 0
    for b in msg34:                                      # long division of m(x) x^12, highest power first
        r = (r << 1) | int(b)
        if r & (1 << 12):
            r ^= g
    for _ in range(12):
        r <<= 1
        if r & (1 << 12):
            r ^= g
    return [int(b) for b in msg34] + [(r >> (11 - k)) & 1 for k in range(12)]


def _meisei_subframe(hdr24: int, words12) -> list:
    """24 header bits + 6 blocks of (16 bits, parity, 16 bits, parity, 12 check bits); parity = 1 when the 16 bits hold an even number of ones"""
    bits = [(hdr24 >> (23 - k)) & 1 for k in range(24)]
    for blk in range(6):
        m = []
        for w in (words12[2 * blk], words12[2 * blk + 1]):
            wb = [(int(w) >> (15 - k)) & 1 for k in range(16)]
            m += wb + [1 ^ (sum(wb) & 1)]
        bits += _bch46(m)
    return bits


def _f32_words(x: float):
    w = int(np.frombuffer(np.float32(x).tobytes(), np.uint32)[0])
    return w & 0xFFFF, w >> 16


def meisei_config(variant: str = "ims100", sn: float = 4123456.0) -> list:
    """the 64 configuration floats a sonde cycles through: serial number at 0 / 16 / 32 / 48, transmit frequency at 15, temperature table at
    17.., resistance table, humidity polynomial 49..52 and the resistance polynomial (meisei100mod.c:841-848 RS-11G, :1092-1102 iMS-100)"""
    c = [0.0] * 64
    for k in (0, 16, 32, 48):
        c[k] = sn
    c[15] = 42.5 if variant == "ims100" else 13.0         # 400e3 + 100 c kHz / 403700 + 100 c kHz
    if variant == "ims100":
        for j in range(12):
            c[17 + j] = 40.0 - 10.0 * j
            c[33 + j] = 5.0 * 1.6 ** j
        c[53:57] = [0.5, 30.0, 2.0, 0.1]
    else:
        for j in range(11):
            c[17 + j] = 40.0 - 10.0 * j
            c[37 + j] = 5.0 * 1.6 ** j
        c[33:37] = [0.5, 30.0, 2.0, 0.1]
    c[49:53] = [-5.0, 40.0, 3.0, -0.2]
    return c


def meisei_frame_bits(counter: int, variant: str = "ims100", cfg=None) -> list:
    """600 bits of one half-second frame (two subframes, headers 0x049DCE / 0xFB6230) in the layout meisei100mod.c:24-86 documents and decodes"""
    cfg = cfg or meisei_config(variant)
    k = counter
    c = cfg[k % 64]
    ms = (1000 * (k // 2)) % 60000
    hh, mi = 12, (k // 120) % 60
    if variant == "ims100":
        lo, hi = _f32_words(c)
        w = [0] * 12
        w[0] = k & 0xFFFF
        w[1] = 32768                                     # reference frequency (read at counter % 4 == 0)
        w[2], w[3] = lo, hi
        w[5] = 9500 + 7 * (k % 50)                        # thermistor count
        w[6] = 32768 if k % 4 == 3 else 8000 + 11 * (k % 40)      # reference again / humidity count
        w[7] = 0x30C1 + ((k & 1) << 8)
        w[10] = ms if k % 2 == 0 else (0x400 + k) & 0xFFFF
        w[11] = (hh << 8) | mi
        v = [0] * 12
        if k % 2 == 0:
            v[0] = 15 * 1000 + 6 * 10 + 4                  # day 15, month 6, year digit 4
            lat = 3512_3456 + 3 * k; lon = 13945_6789 - 2 * k; alt = 1234567 + 250 * k      # NMEA ddmm.mmmm * 1e4, cm
            v[1], v[2] = lat >> 16, lat & 0xFFFF
            v[3], v[4] = lon >> 16, lon & 0xFFFF
            v[5], v[6] = (alt >> 8) & 0xFFFF, (alt & 0xFF) << 8
            v[9] = 23456                                  # course 234.56
            v[10] = 1944                                  # 10 m/s in 1/194.384 knots
        else:
            v[1] = 97                                     # 5 m/s
            v[2] = 0x1200
        v[11] = (w[10] + w[11] + sum(v[:11])) & 0xFFFF
    else:
        w32 = int(np.frombuffer(np.float32(c * 4.0).tobytes(), np.uint32)[0])
        e2 = ((w32 >> 31) << 23) | (((w32 >> 23) & 0xFF) << 24) | (w32 & 0x7FFFFF)         # inverse of f32e2 (:163-191)
        lo, hi = e2 & 0xFFFF, e2 >> 16
        sw = lambda x: ((x & 0xFF) << 8) | (x >> 8)
        w = [0] * 12
        w[0] = k & 0xFFFF
        w[1] = 32768
        w[2], w[3] = sw(lo), sw(hi)
        w[5] = 9500 + 7 * (k % 50)
        w[6] = 8000 + 11 * (k % 40)
        w[7] = 0x30A2 + ((k & 1) << 8)
        if k % 2 == 1:
            w[10] = sw(ms)
            w[11] = (hh << 8) | mi
        v = [0] * 12
        if k % 2 == 0:
            lat = int(35.123456 * 1e7) + 30 * k; lon = int(139.456789 * 1e7) - 20 * k; alt = 1234567 + 250 * k
            v[1], v[2] = lat >> 16, lat & 0xFFFF
            v[3], v[4] = lon >> 16, lon & 0xFFFF
            v[5], v[6] = alt >> 16, alt & 0xFFFF
            v[7], v[8], v[9] = 1000, 23456, 500
            v[10] = 2024 - 0x0700                          # year = low byte + 0x700
            v[11] = (6 << 8) | 15
    return _meisei_subframe(0x049DCE, w) + _meisei_subframe(0xFB6230, v)


def meisei_symbols(n_frames: int, variant: str = "ims100", k0: int = 0, cfg=None) -> np.ndarray:
    """half symbols (0 / 1, 2400 Bd) of n_frames back-to-back frames in biphase-S: a transition at every bit boundary, one more in the middle of a 0"""
    out, level = [], 0
    for k in range(k0, k0 + n_frames):
        for b in meisei_frame_bits(k, variant, cfg):
            level ^= 1
            out.append(level)
            if not b:
                level ^= 1
            out.append(level)
    return np.array(out, dtype=np.uint8)


def meisei_capture(sr: int = 48_000, seconds: float = 6.0, fq: float = 0.0, *, variant: str = "ims100", amp: float = 0.5, noise_sigma: float = 0.02, seed: int = 1,
                   k0: int = 0) -> np.ndarray:
    """Meisei GFSK capture (2400 Bd half symbols, h = 2.4, BT 1.2 as the decoder assumes, meisei100mod.c:640-641), continuous frames from t = 0"""
    sym = meisei_symbols(int(seconds * 2) + 2, variant, k0)
    n = int(seconds * sr)
    z = gfsk_baseband(sym, sr, 2400.0, dev_hz=2.4 * 2400.0 / 2, bt=1.2)[:n]
    if len(z) < n:
        z = np.concatenate([z, np.zeros(n - len(z), z.dtype)])
    rng = np.random.default_rng(seed)
    z = amp * z * np.exp(2j * np.pi * fq * np.arange(n)) + noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, np.int16)
    out[0::2] = np.clip(np.round(z.real * 32767), -32768, 32767)
    out[1::2] = np.clip(np.round(z.imag * 32767), -32768, 32767)
    return out


# ---------------------------------------------------------------------------------------------------------------- Meteosis MTS01
def mts01_frame_bits(k: int = 0, *, sn: str = "A2031234", lat=39.912345, lon=32.854321, alt=1234.0) -> np.ndarray:
    """1048 bits behind the header: 0x80, 128 bytes of comma-separated ASCII telemetry (zero padded; field order as mts01mod.c:180-230 reads
    it), CRC-16 poly 0x8005 init 0xFFFF bit-reversed, low byte first (:76-99,:159-161); bytes MSB first (:112)"""
    sec = k % 60
    txt = f"{sn},0,{k + 1},2406151200{sec:02d},{7412 - k},{lat + 1e-5 * k:.6f},{lon - 1e-5 * k:.6f},{alt + 5 * k:.1f},{123.4 + k:.1f},{12.3:.1f},0,{20.0 + 0.5 * k:.2f},{20.0 + 0.5 * k:.2f},{40 + k},0,0"
    dat = txt.encode("ascii")[:128].ljust(128, b"\0")
    rem = 0xFFFF
    for b in dat:
        rem ^= b << 8
        for _ in range(8):
            rem = ((rem << 1) ^ 0x8005) & 0xFFFF if rem & 0x8000 else (rem << 1) & 0xFFFF
    re = int(f"{rem:016b}"[::-1], 2)
    fr = bytes([0x80]) + dat + bytes([re & 0xFF, re >> 8])
    return np.unpackbits(np.frombuffer(fr, np.uint8))


def mts01_onair_bits(n_frames: int, gap_bits: int = 152) -> np.ndarray:
    """n_frames frames, one per second at 1200 Bd: preamble AA AA, sync B4 2B, the frame, idle 0101.. up to the next"""
    hdr = np.array([int(c) for c in FAMILY["mts01mod"]["header"]], np.uint8)
    idle = np.tile(np.array([1, 0], np.uint8), gap_bits // 2 - 16)
    return np.concatenate([np.concatenate([idle, hdr, mts01_frame_bits(k)]) for k in range(n_frames)])


def mts01_capture(sr: int = 48_000, seconds: float = 6.0, fq: float = 0.0, *, amp: float = 0.5, noise_sigma: float = 0.02, seed: int = 1, invert: bool = False) -> np.ndarray:
    """MTS01 GFSK capture (1200 Bd, h = 0.9, BT 1.5 as the decoder assumes, mts01mod.c:526-527)"""
    bits = mts01_onair_bits(int(seconds) + 2)
    if invert:
        bits = 1 - bits
    n = int(seconds * sr)
    z = gfsk_baseband(bits, sr, 1200.0, dev_hz=0.9 * 1200.0 / 2, bt=1.5)[:n]
    if len(z) < n:
        z = np.concatenate([z, np.zeros(n - len(z), z.dtype)])
    rng = np.random.default_rng(seed)
    z = amp * z * np.exp(2j * np.pi * fq * np.arange(n)) + noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, np.int16)
    out[0::2] = np.clip(np.round(z.real * 32767), -32768, 32767)
    out[1::2] = np.clip(np.round(z.imag * 32767), -32768, 32767)
    return out


# ---------------------------------------------------------------------------------------------------------------- InterMet iMet-54
_IMET54_HAM = [0x00, 0x87, 0x99, 0x1E, 0xAA, 0x2D, 0x33, 0xB4, 0x4B, 0xCC, 0xD2, 0x55, 0xE1, 0x66, 0x78, 0xFF]        # imet54mod.c:196-197


def _imet54_check_words(b: bytearray):
    """the reference's 32-bit frame check (imet54mod.c:229-284) run forward: returns (crc0, crc1) of the data positions"""
    poly0, poly1 = 0x0EDB, 0x8260
    n, bit = 104, 0
    c0, c1 = 0x48EB, 0x1ACA
    nx0, nx1 = c0, c1
    crc0 = crc1 = 0
    while n >= 0:
        if n < 100 or 101 < n < 106:
            if (b[n] >> bit) & 1:
                crc0 ^= c0; crc1 ^= c1
        if c1 & 0x8000:
            nx0 ^= poly0; nx1 ^= poly1
        nx0 <<= 1; nx1 <<= 1
        if c1 & 0x8000:
            nx0 |= 1
        if (c1 ^ c0) & 0x8000:
            nx1 |= 1
        nx0 &= 0xFFFF
        nx1 &= 0xFFFFFFFF
        c0, c1 = nx0, nx1
        if bit < 7:
            bit += 1
        else:
            bit = 0
            n = n - 7 if n % 4 == 3 else n + 1
    return crc0, crc1


def imet54_frame(k: int = 0, *, sn: int = 54012345, lat=52.123456, lon=13.654321, alt_m=2345.6, check: str = "std", imet50: bool = False) -> bytes:
    """108 frame bytes (field map imet54mod.c:330-345): SN, GPS time hhmmssmmm, lat / lon as ddmm.mmmm * 1e4, alt dm, PTU floats, status,
    0xF8 marker; check = "std": the 32-bit check of the standard frame (:229-284), "cont": CRC-32 at 0x34 (:350-360), "none": neither"""
    f = bytearray(108)
    f[0:4] = int(sn).to_bytes(4, "big")
    sec = k % 60
    f[4:8] = int(((12 * 100 + 34) * 100 + sec) * 1000 + 250).to_bytes(4, "big")

    def nmea(x):
        d = int(abs(x)); m = (abs(x) - d) * 60.0
        v = int(round((d * 100 + m) * 1e4))
        return -v if x < 0 else v
    f[8:12] = nmea(lat + 1e-4 * k).to_bytes(4, "big", signed=True)
    f[12:16] = nmea(lon - 1e-4 * k).to_bytes(4, "big", signed=True)
    f[16:20] = int(round((alt_m + 5.0 * k) * 10)).to_bytes(4, "big", signed=True)
    if imet50:
        for p in (0x1C, 0x20, 0x24):
            f[p:p + 4] = bytes.fromhex("4E6E6B28")
        f[0x2A:0x2C] = bytes([0x00, 0x30])
    else:
        f[0x1C:0x20] = np.float32(-12.5 - 0.1 * k).tobytes()[::-1]
        f[0x20:0x24] = np.float32(55.0 + k).tobytes()[::-1]
        f[0x24:0x28] = np.float32(-10.0 - 0.1 * k).tobytes()[::-1]
        f[0x2A:0x2C] = bytes([0x00, 0x3E])
        rng = np.random.default_rng(500 + k)
        f[0x38:0x52] = rng.integers(0, 256, 0x52 - 0x38, dtype=np.uint8).tobytes()
    f[0x52] = 0xF8
    f[0x5E] = k & 0xFF
    if check == "cont":
        m4 = bytearray(0x34)
        for i in range(0x34 // 4):
            for j in range(4):
                m4[4 * i + j] = f[4 * i + 3 - j]
        rem = 0
        for byte in m4:
            rem ^= byte << 24
            for _ in range(8):
                rem = ((rem << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if rem & 0x80000000 else (rem << 1) & 0xFFFFFFFF
        f[0x34:0x38] = (rem ^ 0x63D60875).to_bytes(4, "big")
    elif check == "std":
        c0, c1 = _imet54_check_words(f)
        f[100:102] = (((c0 ^ 0x5000) & 0xF000) | 0x0ABC & 0x0FFF).to_bytes(2, "big")
        f[106:108] = ((c1 ^ 0x1DAD) & 0xFFFF).to_bytes(2, "big")
    return bytes(f)


def imet54_frame_bits(frame108: bytes) -> np.ndarray:
    """2200 bits behind the header: 8N1 characters (start 0, 8 bits, stop 1) of 0x24 0x24 0x42 and of the 64-bit interleaved Hamming(8,4)
    codewords of the 216 nibbles, one more character at the end (imet54mod.c:626-646 backwards)"""
    cw = []
    for b in frame108:
        for nib in (b >> 4, b & 0xF):
            cw += [(_IMET54_HAM[nib] >> j) & 1 for j in range(8)]
    cw = np.array(cw, np.uint8).reshape(-1, 8, 8)
    inter = cw.transpose(0, 2, 1).reshape(-1)                         # out[8 j + i] = in[8 i + j], its own inverse
    sync = np.unpackbits(np.array([0x24, 0x24, 0x42], np.uint8))
    data = np.concatenate([sync, inter, np.zeros(8, np.uint8)])
    chars = data.reshape(-1, 8)
    out = np.zeros((len(chars), 10), np.uint8)
    out[:, 1:9] = chars
    out[:, 9] = 1
    return out.reshape(-1)


def imet54_onair_bits(n_frames: int, **kw) -> np.ndarray:
    """n_frames frames, one per second at 4798 Bd: preamble 0x00 0xAA x 9, the header characters 0x00 0xAA 0x24 0x24, the frame, idle ones"""
    pre = np.array([int(c) for c in ("0000000001" "0101010101") * 9], np.uint8)
    hdr = np.array([int(c) for c in FAMILY["imet54mod"]["header"]], np.uint8)
    idle = np.ones(4798 - len(pre) - len(hdr) - 2200, np.uint8)
    return np.concatenate([np.concatenate([pre, hdr, imet54_frame_bits(imet54_frame(k, **kw)), idle]) for k in range(n_frames)])


def imet54_capture(sr: int = 48_000, seconds: float = 5.0, fq: float = 0.0, *, amp: float = 0.5, noise_sigma: float = 0.02, seed: int = 1, invert: bool = False, **kw) -> np.ndarray:
    """iMet-54 GFSK capture (4798 Bd, h = 0.8, BT 1.0 as the decoder assumes, imet54mod.c:953-954)"""
    bits = imet54_onair_bits(int(seconds) + 2, **kw)
    if invert:
        bits = 1 - bits
    n = int(seconds * sr)
    z = gfsk_baseband(bits, sr, 4798.0, dev_hz=0.8 * 4798.0 / 2, bt=1.0)[:n]
    if len(z) < n:
        z = np.concatenate([z, np.zeros(n - len(z), z.dtype)])
    rng = np.random.default_rng(seed)
    z = amp * z * np.exp(2j * np.pi * fq * np.arange(n)) + noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, np.int16)
    out[0::2] = np.clip(np.round(z.real * 32767), -32768, 32767)
    out[1::2] = np.clip(np.round(z.imag * 32767), -32768, 32767)
    return out


# ---------------------------------------------------------------------------------------------------------------- Meteo-Radiy MRZ (MP3-H1)
def mrz_config() -> list:
    """the 16 configuration words a sonde cycles through (mp3h1mod.c:541-616): NTC A/B/C, ADC polynomials, serial numbers, dates"""
    f = lambda x: int(np.frombuffer(np.float32(x).tobytes(), np.uint32)[0])
    return [f(0.012), f(3450.0), f(1.5), f(1e-6), f(0.9), f(120.0), f(2e-6), f(0.8), f(50.0), 0x1234, 0x5678, 0, 21043, 18765, 150323, 150624]


def mrz_frame(k: int = 0, *, latlon: bool = False, lat=55.751244, lon=37.618423, alt_m=3456.7) -> bytes:
    """frame bytes AA BF 35 .. CRC (50 bytes ECEF, 47 lat / lon; field map mp3h1mod.c:246-275): sub-frame counter k % 16 with its configuration
    word, time, position (ECEF cm + velocities cm/s, or lat / lon 1e-6 deg + alt cm), PTU, CRC-16 0xA001 reflected, init 0xFFFF, low byte first"""
    cfgw = mrz_config()
    sub = k % 16
    sec = (k // 1) % 60
    f = bytearray(50 if not latlon else 47)
    f[0:3] = bytes([0xAA, 0xBF, 0x35])
    f[3] = 0x80 | sub
    f[4:7] = bytes([12, 34, sec])
    o = -3 if latlon else 0
    if latlon:
        f[7:11] = int(round((lat + 1e-4 * k) * 1e6)).to_bytes(4, "little", signed=True)
        f[11:15] = int(round((lon - 1e-4 * k) * 1e6)).to_bytes(4, "little", signed=True)
        f[15:19] = int(round((alt_m + 5 * k) * 100)).to_bytes(4, "little", signed=True)
        f[19:21] = int(1234).to_bytes(2, "little")
        f[21:23] = int(23456).to_bytes(2, "little")
        f[23] = 9
        f[30:32] = b"\xff\xff"
    else:
        a, e2 = 6378137.0, 6.69437999014e-3
        ph, la, h = np.radians(lat + 1e-4 * k), np.radians(lon - 1e-4 * k), alt_m + 5 * k
        N = a / np.sqrt(1 - e2 * np.sin(ph) ** 2)
        xyz = ((N + h) * np.cos(ph) * np.cos(la), (N + h) * np.cos(ph) * np.sin(la), (N * (1 - e2) + h) * np.sin(ph))
        for j, v in enumerate(xyz):
            f[8 + 4 * j:12 + 4 * j] = int(round(v * 100)).to_bytes(4, "little", signed=True)
        for j, v in enumerate((310, -420, 530)):
            f[20 + 2 * j:22 + 2 * j] = int(v).to_bytes(2, "little", signed=True)
        f[26] = 11
        f[33:35] = b"\x12\x34"
    f[29 + o:31 + o] = int(-1234 - k).to_bytes(2, "little", signed=True)
    f[31 + o:33 + o] = int(5678 + k).to_bytes(2, "little", signed=True)
    f[35 + o:39 + o] = int(1_500_000 + 100 * k).to_bytes(4, "little")
    f[39 + o:43 + o] = int(900_000 + 100 * k).to_bytes(4, "little")
    f[43 + o] = sub + 1
    f[44 + o:48 + o] = int(cfgw[sub]).to_bytes(4, "little")
    n = 42 if latlon else 45
    rem = 0xFFFF
    for b in f[3:3 + n]:
        rem ^= b
        for _ in range(8):
            rem = (rem >> 1) ^ 0xA001 if rem & 1 else rem >> 1
    f[3 + n:5 + n] = rem.to_bytes(2, "little")
    return bytes(f)


def mrz_symbols(n_seconds: int, repeats: int = 2, **kw) -> np.ndarray:
    """half symbols at 2399 Bd: per second `repeats` copies of the frame back to back, each as AA + frame bytes MSB first in Manchester (1 -> 10,
    0 -> 01, mp3h1mod.c:133-135), then AA AA and random idle symbols"""
    out = []
    for k in range(n_seconds):
        fr = mrz_frame(k, **kw)
        bits = []
        for _ in range(repeats):
            bits += list(np.unpackbits(np.frombuffer(bytes([0xAA]) + fr, np.uint8)))
        bits += [1, 0, 1, 0, 1, 0, 1, 0] * 2
        sym = []
        for b in bits:
            sym += [1, 0] if b else [0, 1]
        pad = max(0, 2399 - len(sym))
        sym += list(np.random.default_rng(7000 + k).integers(0, 2, pad))        # idle: anything that does not resemble the 1001 preamble
        out += sym
    return np.array(out, np.uint8)


def mrz_capture(sr: int = 48_000, seconds: float = 5.0, fq: float = 0.0, *, amp: float = 0.5, noise_sigma: float = 0.02, seed: int = 1, invert: bool = False, **kw) -> np.ndarray:
    """MRZ GFSK capture (2399 Bd half symbols, h = 2.0, BT 1.0 as the decoder assumes, mp3h1mod.c:1122-1123)"""
    sym = mrz_symbols(int(seconds) + 2, **kw)
    if invert:
        sym = 1 - sym
    n = int(seconds * sr)
    z = gfsk_baseband(sym, sr, 2399.0, dev_hz=2.0 * 2399.0 / 2, bt=1.0)[:n]
    if len(z) < n:
        z = np.concatenate([z, np.zeros(n - len(z), z.dtype)])
    rng = np.random.default_rng(seed)
    z = amp * z * np.exp(2j * np.pi * fq * np.arange(n)) + noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, np.int16)
    out[0::2] = np.clip(np.round(z.real * 32767), -32768, 32767)
    out[1::2] = np.clip(np.round(z.imag * 32767), -32768, 32767)
    return out
