"""The CALLER side of boundary B1, executed: auto_rx's own Python (imported read-only from /root/reference/auto_rx — it cannot travel to the GPU
box, so this test runs where the reference is) parses what THIS repo's binaries print and must arrive at what it arrives at for the reference's
binaries:
  * autorx/scan.py:227 parse_dft_detect_output + the exit-code rules of detect_sonde (:622-656)   <- host/bin/dft_detect (stdout, exit code)
  * autorx/fsk_demod.py:14-97 FSKDemodStats.update                                                 <- host/bin/fsk_demod --stats=5 (stderr)
  * autorx/decode.py:1602-2003 SondeDecoder.handle_decoder_line (required fields, the "version" gate :1654, subtype / frequency handling,
    modem statistics merged into the telemetry, the exporter callback)                             <- host/bin/{rs41mod,dfm09mod,m10mod} --softin
with the argument lists auto_rx builds (tools/caller_cases.py).  dft_detect and fsk_demod need the GPU: their output was recorded on an MI355X by
tools/record_cli_outputs.py (tests/golden/cli_ours.npz; tests/test_gpu_cli_recorded.py keeps the recording honest on every GPU run).  The --softin
decoders are host code and run here, on the recorded soft bits.  The reference side is generated live from oracle/_ref on the same captures."""
import json
import os
import re
import subprocess
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUTORX = "/root/reference/auto_rx"
REF = os.path.join(ROOT, "oracle", "_ref")
OURS = os.path.join(ROOT, "tests", "golden", "cli_ours.npz")
BIN = os.path.join(ROOT, "host", "bin")
pytestmark = pytest.mark.skipif(not (os.path.isdir(AUTORX) and os.path.exists(os.path.join(REF, "v_rs41mod")) and os.path.exists(OURS) and os.path.exists(os.path.join(BIN, "rs41mod"))),
                                reason="needs /root/reference/auto_rx, oracle/_ref (make -C oracle ref), host/bin and the recorded outputs")


@pytest.fixture(scope="module")
def autorx():
    sys.modules.setdefault("semver", types.ModuleType("semver"))          # the one import of autorx.utils this image lacks (version check of its updater)
    if AUTORX not in sys.path:
        sys.path.insert(0, AUTORX)
    import autorx as pkg
    import autorx.decode
    import autorx.fsk_demod
    import autorx.scan
    return pkg


@pytest.fixture(scope="module")
def sides():
    """{'ours': recorded outputs of host/bin on the MI355X, 'ref': outputs of oracle/_ref generated now} for every GPU-backed stage"""
    sys.path.insert(0, ROOT)
    from tools import record_cli_outputs as rec
    return {"ours": rec.load(OURS), "ref": rec.run_all(bindir=REF)}


def _detect(autorx, out, rc):
    """detect_sonde's handling of the process result (scan.py:622-656): exit code 1 = nothing; >= 2 = parse the output; 0 = parse the output"""
    if rc == 1:
        return (None, 0.0)
    return autorx.scan.parse_dft_detect_output(out.decode("utf8"), "test")


def test_scanner_side_parses_our_dft_detect_like_the_references(autorx, sides):
    from tools import caller_cases as cc
    seen = set()
    for name in cc.DETECT:
        o = _detect(autorx, sides["ours"][name + ".stdout"], sides["ours"][name + ".rc"])
        r = _detect(autorx, sides["ref"][name + ".stdout"], sides["ref"][name + ".rc"])
        assert sides["ours"][name + ".rc"] == sides["ref"][name + ".rc"], name
        assert o == r, (name, o, r)
        seen.add(o[0])
    assert {"RS41", "DFM", "M10", None} <= seen                            # each branch of the parser was taken


class _Clock:
    def __init__(self):
        self.t = 1_700_000_000.0

    def time(self):
        self.t += 0.2
        return self.t


def _stats(autorx, stderr, monkeypatch):
    """every stderr line through FSKDemodStats.update (decode.py:1478-1500 feeds it line by line); the state after each line"""
    monkeypatch.setattr(autorx.fsk_demod.time, "time", _Clock().time)
    st = autorx.fsk_demod.FSKDemodStats(averaging_time=2.0, peak_hold=True)
    trace = []
    for line in stderr.split(b"\n"):
        # the one field of a statistics line the caller never reads (FSK_STATS_FIELDS, fsk_demod.py:23) is taken out before parsing: in some frames
        # the reference normalises its eye diagram by a maximum it read out of bounds (utils/fsk_demod.c eye print; nan or other values from run to
        # run — two runs of the reference on the same input differ there, and a line with `-nan` is not JSON at all), tests/test_gpu_fsk.py
        line = re.sub(rb'"eye_diagram":\[\[.*?\]\],?\s*', b"", line)
        st.update(line)
        trace.append((st.snr, st.ppm, tuple(st.fest), tuple(st.fft)))
    return st, trace


def test_modem_statistics_parser_reads_our_stderr_like_the_references(autorx, sides, monkeypatch):
    from tools import caller_cases as cc
    for name in cc.FSK:
        so, to = _stats(autorx, sides["ours"][name + ".stderr"], monkeypatch)
        sr, tr = _stats(autorx, sides["ref"][name + ".stderr"], monkeypatch)
        assert len(to) == len(tr) and len(to) > 8, name
        assert to == tr, name                                              # same SNR / ppm / tone estimates / spectrum after every line
        assert so.snr != -999.0 and len(so.fft) > 0                        # the parser did accept the lines


def _decoder(autorx, sonde_type, stats, sink):
    d = object.__new__(autorx.decode.SondeDecoder)                         # handle_decoder_line with the state __init__ would have set up, no SDR, no subprocess
    d.raw_file = None; d.udp_mode = False; d.sonde_type = sonde_type; d.sonde_freq = 402.5e6; d.rx_frequency = 402.5e6
    d.sdr_type = "RTLSDR"; d.rtl_device_idx = "0"; d.sdr_hostname = "localhost"; d.sdr_port = 5555
    d.close_on_encrypted = False; d.exporters = [sink.append]; d.demod_stats = stats
    d.telem_filter = None; d.enable_realtime_filter = False; d.last_positions = {}; d.max_velocity = 1000
    d.rs41_subframe_uploads = []; d.imet_type = None; d.imet_prev_frame = None; d.imet_prev_time = None; d.imet_id = []; d.imet_max_ids = 4
    d.exit_state = "OK"; d.decoder_running = True
    return d


def _telemetry(autorx, monkeypatch, stderr, decoder_stdout, sonde_type):
    stats, _ = _stats(autorx, stderr, monkeypatch)
    sink, rets = [], []
    d = _decoder(autorx, sonde_type, stats, sink)
    for line in decoder_stdout.split(b"\n"):
        if line:
            rets.append(d.handle_decoder_line(line + b"\n"))
    return sink, rets, d


def _run(exe, argv, data, version):
    env = dict(os.environ, SONDE_JSN_VERSION=version)
    r = subprocess.run([exe] + argv, input=data, capture_output=True, env=env, timeout=120)
    return r.stdout


def test_decoder_line_handler_accepts_our_chain_like_the_references(autorx, sides, monkeypatch):
    from tools import caller_cases as cc
    ver = autorx.__version__
    for name, (_cap, _fargv, dec, dargv, typ) in cc.FSK.items():
        # ours: this repo's decoder on this repo's modem output (recorded on the GPU); reference: both halves of the compiled reference
        ours_out = _run(os.path.join(BIN, dec), dargv, sides["ours"][name + ".stdout"], ver)
        ref_out = _run(os.path.join(REF, "v_" + dec), dargv, sides["ref"][name + ".stdout"], ver)
        to, ro, do = _telemetry(autorx, monkeypatch, sides["ours"][name + ".stderr"], ours_out, typ)
        tr, rr, dr = _telemetry(autorx, monkeypatch, sides["ref"][name + ".stderr"], ref_out, typ)
        assert ro == rr and do.exit_state == dr.exit_state == "OK", name
        assert len(to) == len(tr) and len(to) >= 2, (name, len(to), len(tr))
        for a, b in zip(to, tr):
            assert a == b, (name, {k: (a.get(k), b.get(k)) for k in set(a) | set(b) if a.get(k) != b.get(k)})
        t = to[-1]
        assert t["version"] == ver and "snr" in t and "f_centre" in t and t["freq_float"] == 402.5                 # passed the gate, statistics merged
        assert t["type"].startswith({"RS41": "RS41", "DFM": "DFM", "M10": "M10"}[typ])


def test_version_gate_rejects_a_decoder_of_another_version(autorx, sides, monkeypatch):
    """decode.py:1654: a decoder whose JSON carries another version string stops the SondeDecoder — for ours exactly as for the reference's"""
    from tools import caller_cases as cc
    name = "fsk_rs41"
    _cap, _fargv, dec, dargv, typ = cc.FSK[name]
    ours_out = _run(os.path.join(BIN, dec), dargv, sides["ours"][name + ".stdout"], "0.0.1")
    ref_out = subprocess.run([os.path.join(REF, dec)] + dargv, input=sides["ref"][name + ".stdout"], capture_output=True).stdout   # built with VER_JSN_STR "oracle"
    for out in (ours_out, ref_out):
        sink, rets, d = _telemetry(autorx, monkeypatch, sides["ours"][name + ".stderr"], out, typ)
        assert sink == [] and False in rets and d.exit_state == "Decoder Version Mismatch" and d.decoder_running is False


def test_batch_scanner_binding_equals_auto_rx_scanning_the_peaks_one_by_one(autorx, sides, monkeypatch, tmp_path):
    """§8f-3: auto_rx's OWN scan_peaks_concurrent (autorx/scan_async.py:298-378: one `dft_detect --iq` process per peak on a 48 kHz channel of the band —
    here cut out of the band by the reference's own iq_dec, the role the KA9Q server plays) against radiosonde_auto_rx_amd/scan_batch.py on the output of
    ONE batch process `dft_detect --IQ fq1,fq2,fq3,fq4` over the same band (recorded on the MI355X: tests/golden/cli_ours.npz `batch_band`, kept honest by
    tests/test_gpu_cli_recorded.py), every channel's lines parsed by auto_rx's own parse_dft_detect_output: the same [(frequency, type)] list."""
    import asyncio
    import autorx.scan_async as sa
    import autorx.sdr_wrappers as sw
    from tools import caller_cases as cc
    from radiosonde_auto_rx_amd import scan_batch as sb
    band = tmp_path / "band.cs16"
    band.write_bytes(cc.capture("band").tobytes())

    def iq_cmd(sdr_type=None, frequency=None, sample_rate=None, **kw):          # get_sdr_iq_cmd: "a pipeline that ends in | and delivers cs16 IQ of that channel"
        fq = (float(frequency) - cc.BATCH_CENTER) / cc.BATCH_SR
        assert sample_rate == 48000
        return "cat %s | %s --bo 16 --iq %.9f - %d 16 2>/dev/null |" % (band, os.path.join(REF, "iq_dec"), fq, cc.BATCH_SR)
    monkeypatch.setattr(sw, "get_sdr_iq_cmd", iq_cmd)
    monkeypatch.setattr(sw, "get_sdr_name", lambda *a, **k: "test")
    monkeypatch.setattr(sw, "shutdown_sdr", lambda *a, **k: None)
    ref = asyncio.run(sa.scan_peaks_concurrent(cc.BATCH_PEAKS, max_concurrent=2, rs_path=REF + "/", dwell_time=cc.BATCH_DWELL, sdr_type="KA9Q"))

    async def recorded(cmd):                                                     # the one batch process, as it ran on the GPU
        assert "--IQ " + cc.BATCH["batch_band"][1][3] in cmd and "-t %d" % cc.BATCH_DWELL in cmd
        return sides["ours"]["batch_band.stdout"], sides["ours"]["batch_band.rc"]
    ours = asyncio.run(sb.scan_peaks_batch(cc.BATCH_PEAKS, center_frequency=cc.BATCH_CENTER, sample_rate=cc.BATCH_SR, iq_cmd="cat %s |" % band,
                                           rs_path=BIN, dwell_time=cc.BATCH_DWELL, parse=autorx.scan.parse_dft_detect_output, sdr_name="test", run=recorded))
    assert sorted(ours) == sorted(ref) and len(ref) == 3
    assert {t for _, t in ref} == {"RS41", "DFM", "M10"}
    # and the synchronous wrapper (the counterpart of run_async_scan) on the same recording
    assert sorted(sb.run_batch_scan(cc.BATCH_PEAKS, center_frequency=cc.BATCH_CENTER, sample_rate=cc.BATCH_SR, iq_cmd="cat %s |" % band, rs_path=BIN,
                                    dwell_time=cc.BATCH_DWELL, parse=autorx.scan.parse_dft_detect_output, run=recorded)) == sorted(ref)
