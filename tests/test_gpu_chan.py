"""Polyphase channelizer (include/sonde_chan.h, BASELINE configs[2]): numerics against the defining sum, streaming invariance, and —
parity as SURVEY.md §7 states it for a component the reference does not have — detected type and decoded frame bytes: the
channels it produces are read by the REFERENCE's own tools (`dft_detect --iq`, `rs41mod --iq2`, float32 input) and by this repo's,
with identical results, and both find and decode every sonde of a 10 Msps stream."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
BIN = os.path.join(ROOT, "host", "bin")


def _direct(x, h, M, D, m_list, k_list):
    """y_k[m] = sum_n h[n] x[mD-n] exp(-2 pi i k (mD-n)/M) in float64"""
    out = np.zeros((len(k_list), len(m_list)), np.complex128)
    T = len(h)
    for j, m in enumerate(m_list):
        n = np.arange(T)
        idx = m * D - n
        xv = np.where(idx >= 0, x[np.clip(idx, 0, len(x) - 1)], 0)
        for i, k in enumerate(k_list):
            out[i, j] = np.sum(h * xv * np.exp(-2j * np.pi * k * idx / M))
    return out


def _prototype(M, P):
    T = M * P
    n = np.arange(T)
    t = n - 0.5 * (T - 1)
    fc = 0.5 / M
    h = 2 * fc * np.sinc(2 * fc * t) * (0.42 - 0.5 * np.cos(2 * np.pi * n / (T - 1)) + 0.08 * np.cos(4 * np.pi * n / (T - 1)))
    return h / h.sum()


@pytest.mark.parametrize("M,D,P", [(256, 200, 16), (64, 48, 8), (128, 128, 8)])
def test_channelizer_matches_defining_sum(M, D, P):
    import torch
    from radiosonde_auto_rx_amd.chan import Channelizer
    rng = np.random.default_rng(3)
    n = 40_000
    xi = rng.integers(-20000, 20000, size=2 * n).astype(np.int16)
    x = (xi[0::2].astype(np.float64) + 1j * xi[1::2]) / 32768.0
    ch = Channelizer(1_000_000, M, D, P, max_chunk=n)
    out = torch.zeros(M, ch.max_frames, 2, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()                     # the fill runs on torch's stream, the channelizer on its own: order them
    # the stream in three uneven calls: the filter state carries over
    got, pos = 0, 0
    for take in (12_345, 77, n - 12_345 - 77):
        k = ch.process_host(xi[2 * pos:2 * (pos + take)], out.data_ptr() + 8 * got, ch.max_frames)
        got += k; pos += take
    ch.sync()
    assert got == (n - 1) // D + 1
    y = out.cpu().numpy()
    y = y[..., 0] + 1j * y[..., 1]
    ms = [0, 1, 5, got // 2, got - 1]
    ks = [0, 1, 7, M // 2 - 1, M // 2, M - 3]
    want = _direct(x, _prototype(M, P), M, D, ms, ks)
    err = np.abs(y[np.ix_(ks, ms)] - want).max()
    assert err < 2e-6, err
    ch.close()


def _stream(sr, seconds, sondes):
    """sum of synthetic sondes at absolute offsets (Hz) in one cs16 stream"""
    from tools import synth
    n = int(sr * seconds)
    acc = np.zeros(2 * n, np.float64)
    for kind, f_hz, seed in sondes:
        fq = f_hz / sr
        if kind == "rs41":
            x = synth.rs41_capture(sr=sr, seconds=seconds, fq=fq, seed=seed, noise_sigma=0.0, amp=0.2)
        elif kind == "dfm":
            x = synth.dfm_capture(sr=sr, seconds=seconds, fq=fq, seed=seed, noise_sigma=0.0, amp=0.2)
        else:
            x = synth.m10_capture(sr=sr, seconds=seconds, fq=fq, seed=seed, noise_sigma=0.0, amp=0.2)
        acc[:len(x)] += x[:2 * n]
    rng = np.random.default_rng(99)
    acc += rng.normal(0.0, 60.0, size=2 * n)
    return np.clip(np.round(acc), -32768, 32767).astype(np.int16)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "rs41mod")), reason="compiled reference not present")
def test_wideband_stream_channels_detect_and_decode_like_reference():
    import torch
    from radiosonde_auto_rx_amd.chan import Channelizer
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    sr, M, D = 10_000_000, 256, 200
    spacing = sr / M
    sondes = [("rs41", 31 * spacing + 1500.0, 5), ("dfm", -17 * spacing - 900.0, 6), ("m10", 90 * spacing + 400.0, 7)]
    x = _stream(sr, 2.6, sondes)
    ch = Channelizer(sr, M, D, 16, max_chunk=sr)
    if_sr = int(ch.out_rate)
    out = torch.zeros(M, 3 * ch.max_frames, 2, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    got = 0
    n = len(x) // 2
    for pos in range(0, n, sr):
        take = min(sr, n - pos)
        got += ch.process_host(x[2 * pos:2 * (pos + take)], out.data_ptr() + 8 * got, 3 * ch.max_frames)
    ch.sync()
    y = out[:, :got].cpu().numpy()
    want_type = {"rs41": "RS41", "dfm": "DFM9", "m10": "M10"}
    for kind, f_hz, _ in sondes:
        k = ch.nearest_channel(f_hz)
        iq = np.ascontiguousarray(y[k]).astype(np.float32).tobytes()          # float32 IQ of that channel: `- <sr> 32`
        a = subprocess.run([os.path.join(BIN, "dft_detect"), "--iq", "--dc", "-", str(if_sr), "32"], input=iq, capture_output=True, timeout=120)
        b = subprocess.run([os.path.join(REF, "dft_detect"), "--iq", "--dc", "-", str(if_sr), "32"], input=iq, capture_output=True, timeout=120)
        assert a.stdout == b.stdout and a.returncode == b.returncode, (kind, a.stdout, b.stdout)
        assert want_type[kind] in a.stdout.decode(), (kind, a.stdout)
        # a neighbouring channel, one spacing away, sees nothing of it
        iq2 = np.ascontiguousarray(y[(k + 3) % M]).astype(np.float32).tobytes()
        c = subprocess.run([os.path.join(REF, "dft_detect"), "--iq", "--dc", "-t", "2", "-", str(if_sr), "32"], input=iq2, capture_output=True, timeout=120)
        assert want_type[kind] not in c.stdout.decode()
    # decoded frame bytes of the RS41 channel: reference decoder and this repo's, same channel samples
    k = ch.nearest_channel(sondes[0][1])
    iq = np.ascontiguousarray(y[k]).astype(np.float32).tobytes()
    # the residual offset inside the channel is what the detector reports and auto_rx tunes by (scan.py: detect_sonde -> decode.py --IQ)
    resid = (sondes[0][1] - ch.channel_freq(k)) / if_sr
    argv = ["rs41mod", "-r", "--ecc2", "--crc", "--IQ", repr(resid), "--lpIQ", "-", str(if_sr), "32"]
    a = subprocess.run([os.path.join(BIN, argv[0])] + argv[1:], input=iq, capture_output=True, timeout=120)
    b = subprocess.run([os.path.join(REF, argv[0])] + argv[1:], input=iq, capture_output=True, timeout=120)
    assert a.returncode == 0 and a.stdout == b.stdout
    lines = a.stdout.decode().splitlines()
    assert len(lines) >= 2 and all("[OK]" in l for l in lines)
    ch.close()


def test_scanner_ordered_behind_the_channelizer_on_the_device():
    """sonde_scan_wait_stream: the scanner's stream waits for the channelizer's instead of the host waiting in between (what bench.py's scan_wide step
    does) — same detections, chunk by chunk, as with sonde_chan_sync() in front of every scanner call"""
    import torch
    from tools import synth
    from radiosonde_auto_rx_amd.chan import Channelizer
    from radiosonde_auto_rx_amd.scan import Scanner, IFIQ
    sr, M, D, P = 2_400_000, 64, 48, 8
    spacing = sr / M
    k_rs41, k_dfm = 9, 52
    sig = [dict(kind="rs41", fq=(k_rs41 * spacing + 1500.0) / sr, t_first=0.05, amp=0.08),
           dict(kind="dfm", fq=((k_dfm - M) * spacing - 2000.0) / sr, t_first=0.1, amp=0.08)]
    x = synth.wideband_capture(sr, 1.6, sig, noise_sigma=0.01, seed=5)
    n_tot = len(x) // 2 // D * D
    wb = torch.from_numpy(x).to("cuda")
    chunk = sr // 4
    got = {}
    for ordered in (False, True):
        ch = Channelizer(sr, M, D, P, max_chunk=chunk)
        out = torch.zeros(M, ch.max_frames, 2, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        sc = Scanner(sr // D, n_channels=M, iq_mode=IFIQ, dc=True, cont=True, max_chunk=ch.max_frames, bits=32)
        dets = []
        for s0 in range(0, n_tot, chunk):
            take = min(chunk, n_tot - s0)
            nf = ch.process_device(wb.data_ptr() + 4 * s0, take, out.data_ptr(), ch.max_frames)
            if ordered:
                sc.wait_stream(ch.stream())
            else:
                ch.sync()
            if nf > 0:
                sc.process_device(out.data_ptr(), ch.max_frames, nf)
                dets += [(d["channel"], d["type"], d["sample"], d["line"]) for d in sc.fetch(verbose=True)]
            ch.sync()                                   # `out` is rewritten by the next channelizer call: the scanner call above has returned (it ends with a host wait)
        sc.close(); ch.close()
        got[ordered] = dets
    assert got[True] == got[False]
    assert {(c, t) for c, t, _, _ in got[True]} >= {(k_rs41, "RS41"), (k_dfm, "DFM9")}
