"""Synthetic Vaisala RS92 material for the parity tests (test infrastructure): a GPS constellation with its RINEX navigation file and SEM
almanac, frames whose GPS block carries pseudo-range chips of a receiver at a chosen position / velocity, the 32 calibration rows
(RS92-SGP, or RS92-NGP under a 16-byte key), on-air symbols (8N1, Manchester) and IQ captures.

Frame layout as rs92mod.c reads it (:243-271): 2A 2A 2A 2A 2A 10 | 65 10 config(32) crc | 69 0C ptu(24) crc | 67 3D gps(122) crc | 68 05
aux(10) crc | 6 bytes | 24 bytes Reed-Solomon parity over bytes 6..215.  The compiled reference decoding these frames to the generator's
position (tests/test_rs92_native.py) pins the generator."""
from __future__ import annotations

import math

import numpy as np

from tools import synth

MU = 3.986005e14
OMEGA_E = 7.2921151467e-05
C = 299792458.0
F_REL = -4.442807633e-10
WEEK = 604800.0
DF = 299792.458 / 1023.0 / 1024.0          # metres per pseudo-range unit (rs92mod.c:971)
DL = 1575.42 / 1.023 / 4.0                 # delta-chip scale (:973)
CAL170 = bytes([0x36, 0x98, 0x92, 0x25, 0x6b, 0xb3, 0x99, 0xe1, 0x57, 0x05, 0x30, 0x9a, 0xfe, 0x51, 0xf4, 0xab])      # :339-340


# ------------------------------------------------------------------------------------------------------------------- orbits
def constellation(seed: int = 7, toe: float = 302400.0, week: int = 2100, simple: bool = False, prns=None) -> list[dict]:
    """Broadcast-ephemeris parameter sets of 6 planes x 5 slots + PRN 31, 32; `simple` leaves out everything an almanac cannot carry."""
    rng = np.random.default_rng(seed)
    prns = list(prns) if prns is not None else list(range(1, 33))
    out = []
    for n, prn in enumerate(prns):
        plane, slot = n % 6, n // 6
        e = dict(prn=prn, week=week, toe=toe,
                 sqrta=5153.6 + rng.uniform(-0.5, 0.5), e=rng.uniform(0.001, 0.015),
                 i0=0.30 * math.pi + rng.uniform(-0.01, 0.01) * math.pi, Omega0=(plane * 60.0 + rng.uniform(-2, 2)) * math.pi / 180.0 - math.pi,
                 OmegaDot=-2.6e-9 * math.pi + rng.uniform(-1e-10, 1e-10), w=rng.uniform(-math.pi, math.pi),
                 M0=((slot * 72.0 + plane * 25.0 + rng.uniform(-5, 5)) % 360.0 - 180.0) * math.pi / 180.0,
                 af0=rng.uniform(-3e-4, 3e-4), af1=rng.uniform(-5e-12, 5e-12), af2=0.0, health=0)
        if simple:
            e.update(delta_n=0.0, idot=0.0, cuc=0.0, cus=0.0, crc=0.0, crs=0.0, cic=0.0, cis=0.0, tgd=0.0)
        else:
            e.update(delta_n=rng.uniform(3e-9, 6e-9), idot=rng.uniform(-5e-10, 5e-10), cuc=rng.uniform(-5e-6, 5e-6), cus=rng.uniform(-5e-6, 5e-6),
                     crc=rng.uniform(150, 350), crs=rng.uniform(-100, 100), cic=rng.uniform(-2e-7, 2e-7), cis=rng.uniform(-2e-7, 2e-7),
                     tgd=rng.uniform(-2e-8, 2e-8))
        out.append(e)
    return out


def sat_state(e: dict, tow: float):
    """position / velocity (ECEF at the corrected time) and clock correction [m] of one satellite — IS-GPS-200 model, float64"""
    a = e["sqrta"] ** 2
    n = math.sqrt(MU / a ** 3) + e["delta_n"]

    def kepler(tk):
        M = e["M0"] + n * tk
        E = M
        for _ in range(12):
            E = M + e["e"] * math.sin(E)
        return E

    def wrap(t):                                                # the nearest week (rs92mod.c:915-918)
        return (t + WEEK / 2) % WEEK - WEEK / 2

    tk = wrap(tow - e["toe"])
    E = kepler(tk)
    cc = (e["af0"] + e["af1"] * tk + e["af2"] * tk * tk - e["tgd"]) * C + F_REL * e["e"] * e["sqrta"] * math.sin(E) * C
    tk = wrap(tow + cc / C - e["toe"])

    def pos(tk):
        E = kepler(tk)
        v = math.atan2(math.sqrt(1 - e["e"] ** 2) * math.sin(E), math.cos(E) - e["e"])
        u = v + e["w"]
        c2, s2 = math.cos(2 * u), math.sin(2 * u)
        r = a * (1 - e["e"] * math.cos(E)) + e["crc"] * c2 + e["crs"] * s2
        inc = e["i0"] + e["cic"] * c2 + e["cis"] * s2 + e["idot"] * tk
        u += e["cuc"] * c2 + e["cus"] * s2
        xo, yo = r * math.cos(u), r * math.sin(u)
        om = e["Omega0"] + e["OmegaDot"] * tk - OMEGA_E * (tk + e["toe"])
        return np.array([xo * math.cos(om) - yo * math.sin(om) * math.cos(inc), xo * math.sin(om) + yo * math.cos(om) * math.cos(inc), yo * math.sin(inc)])

    p = pos(tk)
    v = (pos(tk + 0.5) - pos(tk - 0.5))
    return p, v, cc


def llh2ecef(lat, lon, alt):
    a, b = 6378137.0, 6356752.31424518
    e2 = 1 - b * b / (a * a)
    la, lo = math.radians(lat), math.radians(lon)
    N = a / math.sqrt(1 - e2 * math.sin(la) ** 2)
    return np.array([(N + alt) * math.cos(la) * math.cos(lo), (N + alt) * math.cos(la) * math.sin(lo), (N * (1 - e2) + alt) * math.sin(la)])


def _dfield(x: float) -> str:
    return ("%19.12E" % x).replace("E", "D")


def rinex_nav(ephs: list[dict], extra_toe=()) -> bytes:
    """RINEX 2 navigation file (8 lines per entry, D exponents) of the sets, plus copies with another toe (same orbit, moved epoch)"""
    hdr = ["     2.10           N: GPS NAV DATA                         RINEX VERSION / TYPE",
           "synth_rs92          tests               20200408 000000 UTC PGM / RUN BY / DATE",
           "                                                            END OF HEADER"]
    lines = [h.ljust(80) for h in hdr]
    sets = []
    for dt in (0.0,) + tuple(extra_toe):
        for e in ephs:
            if dt == 0.0:
                sets.append(e)
            else:
                n = math.sqrt(MU / e["sqrta"] ** 6) + e["delta_n"]
                f = dict(e)
                f.update(toe=e["toe"] + dt, M0=e["M0"] + n * dt, Omega0=e["Omega0"] + e["OmegaDot"] * dt, i0=e["i0"] + e["idot"] * dt, af0=e["af0"] + e["af1"] * dt)
                sets.append(f)
    for e in sets:
        toe = e["toe"]
        day, sod = int(toe // 86400), toe % 86400
        hh, mi, ss = int(sod // 3600), int(sod % 3600 // 60), sod % 60
        lines.append("%2d %02d %2d %2d %2d %2d%5.1f%s%s%s" % (e["prn"], 20, 4, 5 + day, hh, mi, ss, _dfield(e["af0"]), _dfield(e["af1"]), _dfield(e["af2"])))
        rows = [(33.0, e["crs"], e["delta_n"], e["M0"]), (e["cuc"], e["e"], e["cus"], e["sqrta"]), (toe, e["cic"], e["Omega0"], e["cis"]),
                (e["i0"], e["crc"], e["w"], e["OmegaDot"]), (e["idot"], 1.0, float(e["week"]), 0.0), (2.0, float(e["health"]), e["tgd"], 33.0)]
        for r in rows:
            lines.append("   " + "".join(_dfield(v) for v in r))
        lines.append("   " + _dfield(toe - 7200.0 + 18.0) + _dfield(4.0))
    return ("\n".join(lines) + "\n").encode()


def sem_almanac(ephs: list[dict], gps_week: int) -> bytes:
    """SEM almanac of the same orbits (angles in semicircles, inclination relative to 0.30, 10-bit week)"""
    toa = int(ephs[0]["toe"])
    s = ["%d  SYNTH.ALM" % len(ephs), "%d %d" % (gps_week % 1024, toa), ""]
    for e in ephs:
        s += ["%d" % e["prn"], "%d" % (40 + e["prn"]), "0",
              "%.14E %.14E %.14E" % (e["e"], e["i0"] / math.pi - 0.30, e["OmegaDot"] / math.pi),
              "%.14E %.14E %.14E" % (e["sqrta"], e["Omega0"] / math.pi, e["w"] / math.pi),
              "%.14E %.14E %.14E" % (e["M0"] / math.pi, e["af0"], e["af1"]), "%d" % e["health"], "9", ""]
    return "\n".join(s).encode()


# ------------------------------------------------------------------------------------------------------------------- frame
def cal_rows(seed: int = 3, *, freq_khz: int = 402500, killtimer: int = 0xFFFF, ngp_key: bytes | None = None) -> bytes:
    """the 512 calibration bytes (32 rows of 16): row 0 carries frequency and kill timer, 66 coefficient records (index byte + float32)
    from 0x40 on with the fixed tail every RS92 has at 0x16C..0x17F; under `ngp_key` the record block is sent the RS92-NGP way (float bytes
    in the order 2, 0, 1, 3, XORed with the key)"""
    rng = np.random.default_rng(seed)
    cal = bytearray(rng.integers(0, 256, 512, dtype=np.uint8).tobytes())
    cal[0:2] = b"\x00\x00"
    f = (freq_khz - (1600000 if ngp_key else 400000)) // 10
    cal[2:4] = int(f).to_bytes(2, "little")
    cal[4:6] = int(killtimer).to_bytes(2, "little")
    coef = {}
    coef.update({30: -150.3, 31: 100.187, 32: 2.1, 33: -0.18, 34: 0.0, 35: 0.0, 37: 1.3})            # T = poly(x), x = 1 / (y0 - y)
    coef.update({40: -41.3, 41: 60.2, 42: -1.25, 43: 0.031, 44: 0.0, 45: 0.0, 47: 1.41})                     # U1
    coef.update({50: -40.9, 51: 59.7, 52: -1.21, 53: 0.029, 54: 0.0, 55: 0.0, 57: 1.40})                     # U2
    coef.update({10: -512.0, 11: 998.7, 12: 14.1, 13: -0.82, 14: 0.011, 15: 0.0, 17: 1.25})                  # P
    fixed = [(0x97, None), (0x98, bytes([0x92, 0x25, 0x6b, 0xb3])), (0x99, bytes([0xe1, 0x57, 0x05, 0x30])), (0x9a, bytes([0xfe, 0x51, 0xf4, 0xab])),
             (0x9d, np.float32(0.7).tobytes()), (0xa7, np.float32(0.7).tobytes())]
    recs = []
    for idx in sorted(coef):
        recs.append((idx, np.float32(coef[idx]).tobytes()))
    k = 0x60
    while len(recs) < 60:                                       # other coefficients the decoder does not look at
        recs.append((k, np.float32(rng.normal()).tobytes()))
        k += 1
    for idx, b in fixed:
        recs.append((idx, b if b is not None else bytes([0xac, 0x64, 0x9f, 0x36])))
    assert len(recs) == 66
    blk = bytearray()
    for idx, b in recs:
        blk.append(idx)
        blk += bytes([b[2], b[0], b[1], b[3]]) if ngp_key else b
    if ngp_key:
        blk = bytearray(v ^ ngp_key[j % 16] for j, v in enumerate(blk))
    cal[0x40:0x40 + 330] = blk
    if not ngp_key:
        assert bytes(cal[0x170:0x180]) == CAL170
    return bytes(cal)


def _xptu16(cal: bytes) -> bytes:
    """the measurement key an RS92-NGP derives from calibration bytes 0x24.. (rs92mod.c:367-419)"""
    out = bytearray(16)
    for j in range(8):
        a = 0x1d89
        for k in range(4):
            a = (a + cal[0x24 + j + k]) & 0xFFFFFFFF
            a = (a + (a << 10)) & 0xFFFFFFFF
            a ^= a >> 6
        a = (a + (a << 3)) & 0xFFFFFFFF
        a ^= a >> 11
        a = (a + (a << 15)) & 0xFFFFFFFF
        out[2 * j], out[2 * j + 1] = a & 0xFF, (a >> 8) & 0xFF
    return bytes(out)


def ptu_counts(k: int, *, y_t=0.46, y_u=0.55, y_p=0.37) -> bytes:
    ref1, ref3, ref4 = 560000, 389000, 352000
    t = ref1 - int((y_t + 0.001 * k) * (ref1 - ref4))
    u1 = ref1 - int((y_u - 0.002 * k) * (ref1 - ref3))
    u2 = ref1 - int((y_u - 0.002 * k + 0.004) * (ref1 - ref3))
    p = ref1 - int((y_p + 0.002 * k) * (ref1 - ref4))
    ch = [t, u1, u2, ref1, 455000, p, ref3, ref4]
    return b"".join(int(v).to_bytes(3, "little") for v in ch)


def gps_block(ephs: list[dict], tow_ms: int, lat: float, lon: float, alt: float, vel_enu=(0.0, 0.0, 0.0), *, seed: int = 1, noise_m: float = 1.5,
              min_elev_deg: float = 7.0, bias_m: float = -2.62e7, rate_bias: float = -6000.0, spoil: dict | None = None, order=None) -> bytes:
    """the 122 bytes of the GPS block for a receiver at lat / lon / alt moving with vel_enu [m/s]: time of week, PRNs (12 x 5 bits), status,
    pseudo-range chips / delta chips of the satellites above `min_elev_deg`.  `spoil`: {prn: metres} added to single ranges; `order`: PRNs
    in slot order (else by elevation)."""
    rng = np.random.default_rng(seed + tow_ms % 100000)
    rx = llh2ecef(lat, lon, alt)
    la, lo = math.radians(lat), math.radians(lon)
    E = np.array([-math.sin(lo), math.cos(lo), 0.0])
    N = np.array([-math.sin(la) * math.cos(lo), -math.sin(la) * math.sin(lo), math.cos(la)])
    U = np.array([math.cos(la) * math.cos(lo), math.cos(la) * math.sin(lo), math.sin(la)])
    vrx = vel_enu[0] * E + vel_enu[1] * N + vel_enu[2] * U
    tow = tow_ms / 1000.0
    vis = []
    for e in ephs:
        p, v, cc = sat_state(e, tow)
        d = p - rx
        el = math.degrees(math.asin(np.dot(d, U) / np.linalg.norm(d)))
        if el < min_elev_deg:
            continue
        tau = np.linalg.norm(d) / C
        ang = OMEGA_E * tau
        pr = np.array([math.cos(ang) * p[0] + math.sin(ang) * p[1], -math.sin(ang) * p[0] + math.cos(ang) * p[1], p[2]])
        rho = np.linalg.norm(pr - rx) + bias_m - cc + rng.normal(0.0, noise_m)
        if spoil and e["prn"] in spoil:
            rho += spoil[e["prn"]]
        los = (p - rx) / np.linalg.norm(p - rx)
        rate = float(np.dot(v - vrx, los)) + rate_bias + rng.normal(0.0, 0.02)
        vis.append((el, e["prn"], rho, rate))
    vis.sort(reverse=True)
    if order is not None:
        by = {v[1]: v for v in vis}
        vis = [by[p] for p in order if p in by]
    vis = vis[:12]
    prn = [0] * 12
    status = [0] * 12
    data = bytearray()
    for j in range(12):
        if j < len(vis):
            el, p, rho, rate = vis[j]
            prn[j] = p
            status[j] = 0x0F | (max(1, min(15, 4 + int(el / 8))) << 4)
            chips = int(round(-rho / DF)) & 0xFFFFFFFF
            delta = int(round(-rate * DL / DF)) & 0xFFFFFF
            data += chips.to_bytes(4, "little") + delta.to_bytes(3, "little") + bytes([0x30 + j])
        else:
            data += (0x7FFFFFFF).to_bytes(4, "little") + (0x555555).to_bytes(3, "little") + b"\x00"
    words = []
    for b in range(4):
        w = 0
        for c in range(3):
            p = prn[3 * b + c]
            if p == 32:                                         # 6th bit: spills into the next number's lowest bit / the word's spare bit
                w |= 1 << (5 * c + 5)
            else:
                w |= (p & 31) << (5 * c)
        words.append(w & 0xFFFF)
    out = int(tow_ms).to_bytes(4, "little") + b"\x00\x00" + b"".join(w.to_bytes(2, "little") for w in words) + bytes(status) + bytes(data)
    assert len(out) == 122
    return out


def _block(frame: bytearray, pos: int, blk_id: int, payload: bytes) -> None:
    frame[pos] = blk_id
    frame[pos + 1] = len(payload) // 2
    frame[pos + 2:pos + 2 + len(payload)] = payload
    crc = synth.crc16_ccitt_false(payload)
    frame[pos + 2 + len(payload)] = crc & 0xFF
    frame[pos + 3 + len(payload)] = crc >> 8


def rs92_frame(k: int, cal: bytes, gps: bytes, *, sonde_id: str = "K1234567", frame0: int = 2000, aux=(0, 0, 0, 0), ngp: bool = False,
               ptu: bytes | None = None) -> bytes:
    f = bytearray(240)
    f[0:6] = bytes([0x2A] * 5 + [0x10])
    frnr = (frame0 + k) & 0xFFFF
    row = (frame0 + k) % 32
    cfg = bytearray(32)
    cfg[0:2] = frnr.to_bytes(2, "little")
    cfg[2:4] = b"\x00\x00"
    cfg[4:12] = sonde_id.encode()[:8].ljust(8)
    cfg[12:14] = b"\x00\x61"
    cfg[14] = 0x00
    cfg[15] = row
    cfg[16:32] = cal[16 * row:16 * row + 16]
    _block(f, 6, 0x65, bytes(cfg))
    m = bytearray(ptu if ptu is not None else ptu_counts(k))
    if ngp:
        x = _xptu16(cal)
        for j in range(24):
            m[j] ^= cfg[j & 1] ^ x[j % 16]
    _block(f, 42, 0x69, bytes(m))
    _block(f, 70, 0x67, gps)
    a = b"\x03\x03" + b"".join(int(v).to_bytes(2, "little") for v in aux)
    _block(f, 196, 0x68, a)
    f[210:216] = bytes([0xFF, 0x02, 0x02, 0x00, 0x02, 0x00])
    f[216:240] = synth.rs255_231_parity(bytes(f[6:216]) + bytes(21)).tobytes()
    return bytes(f)


def flight(n: int, ephs, *, tow_ms: int = 302410_000, lat=47.7123, lon=8.9456, alt=14321.0, vel_enu=(12.5, -7.25, 5.1), seed: int = 1, cal: bytes | None = None,
           ngp: bool = False, aux=(0, 0, 0, 0), frame0: int = 2000, **gps_kw) -> list[bytes]:
    """n consecutive frames (1 s apart) of one ascent"""
    cal = cal if cal is not None else cal_rows()
    out = []
    for k in range(n):
        la = lat + vel_enu[1] * k / 111120.0
        lo = lon + vel_enu[0] * k / (111120.0 * math.cos(math.radians(lat)))
        g = gps_block(ephs, tow_ms + 1000 * k, la, lo, alt + vel_enu[2] * k, vel_enu, seed=seed, **gps_kw)
        out.append(rs92_frame(k, cal, g, ngp=ngp, aux=aux, frame0=frame0))
    return out


# ------------------------------------------------------------------------------------------------------------------- on air
def frame_symbols(frame: bytes) -> np.ndarray:
    """8N1 (start 0, LSB first, stop 1), each bit as a Manchester pair 0 -> 10, 1 -> 01 (rs92mod.c:180-196)"""
    bits = np.unpackbits(np.frombuffer(frame, np.uint8)[:, None], axis=1, bitorder="little")
    b10 = np.concatenate([np.zeros((len(frame), 1), np.uint8), bits, np.ones((len(frame), 1), np.uint8)], axis=1).ravel()
    sym = np.empty(2 * len(b10), np.uint8)
    sym[0::2] = 1 - b10
    sym[1::2] = b10
    return sym


def onair_symbols(frames: list[bytes], lead: int = 200, gap: int = 0) -> np.ndarray:
    idle = np.tile(np.array([1, 0], np.uint8), lead // 2)
    parts = [idle]
    for f in frames:
        parts.append(frame_symbols(f))
        if gap:
            parts.append(np.tile(np.array([1, 0], np.uint8), gap // 2))
    parts.append(idle)
    return np.concatenate(parts)


def rs92_capture(frames: list[bytes], sr: int = 48_000, fq: float = 0.0, *, amp: float = 0.5, noise_sigma: float = 0.02, seed: int = 1, invert: bool = False,
                 dev_hz: float = 2400.0, f_offset_hz: float = 0.0, tail_s: float = 0.3) -> np.ndarray:
    """GFSK capture (4800 symbols/s = 2400 bit/s Manchester, BT 0.5), interleaved int16 IQ"""
    sym = onair_symbols(frames, lead=int(0.2 * 4800))
    if invert:
        sym = 1 - sym
    z = synth.gfsk_baseband(sym, sr, 4800.0, dev_hz=dev_hz, bt=0.5)
    n = len(z) + int(tail_s * sr)
    z = np.concatenate([z, np.zeros(n - len(z), z.dtype)])
    rng = np.random.default_rng(seed)
    z = amp * z * np.exp(2j * np.pi * (fq + f_offset_hz / sr) * np.arange(n)) + noise_sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty(2 * n, np.int16)
    out[0::2] = np.clip(np.round(z.real * 32767), -32768, 32767)
    out[1::2] = np.clip(np.round(z.imag * 32767), -32768, 32767)
    return out
