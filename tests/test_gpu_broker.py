"""The resident broker (host/sonde_broker.c): N `fsk_demod` shim processes — what auto_rx spawns per sonde, decode.py:1489-1529 — share one
HIP context and one batched modem engine; every step demodulates the pending frame of all clients in one launch.  Their stdout / stderr must
be byte for byte what the same command prints on its own, with 64 of them running concurrently."""
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "bin")


def _streams():
    from tools import synth
    ecef = dict(ecef_cm=(418833319, 85974133, 473346430))
    out = []
    for k in range(8):            # 48 kHz RS41 IQ at different noise levels / offsets: one modem configuration -> one engine group
        x = synth.rs41_capture(sr=48_000, seconds=3.3, fq=0.0, noise_sigma=0.05 + 0.04 * k, frame_kw=ecef, n_frames=3, t_first=0.1 + 0.01 * k, seed=300 + k)
        out.append((["--cs16", "-b", "-12000", "-u", "12000", "-s", "--stats=5", "2", "48000", "4800", "-", "-"], x.tobytes()))
    for k in range(2):            # a second configuration (DFM, 50 kHz / 2500 Bd, hard decisions): its own group in the same broker
        x = synth.dfm_capture(sr=50_000, seconds=2.2, fq=0.0, noise_sigma=0.05 + 0.05 * k, seed=400 + k)
        out.append((["--cs16", "--nsym=25", "-p", "5", "2", "50000", "2500", "-", "-"], x.tobytes()))
    return out


@pytest.mark.gpu
def test_64_concurrent_shims_share_one_engine(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    streams = _streams()
    alone = [subprocess.run([os.path.join(BIN, "fsk_demod")] + a, input=x, capture_output=True, timeout=120) for a, x in streams]
    assert all(r.returncode == 0 and len(r.stdout) > 0 for r in alone)
    sock = str(tmp_path / "broker.sock")
    broker = subprocess.Popen([os.path.join(BIN, "sonde_broker"), "--socket", sock, "--slots", "64"], stderr=subprocess.PIPE)
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        assert os.path.exists(sock)
        env = dict(os.environ, SONDE_BROKER=sock)
        files = []
        for i, (a, x) in enumerate(streams):
            p = tmp_path / ("in%d.raw" % i)
            p.write_bytes(x)
            files.append(str(p))
        procs = []
        for i in range(64 + 8):                     # 64 clients on the RS41 group (8 per stream), 8 on the DFM group
            k = i % 8 if i < 64 else 8 + i % 2
            a = list(streams[k][0]); a[-2] = files[k]
            procs.append((k, subprocess.Popen([os.path.join(BIN, "fsk_demod")] + a, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
        for k, p in procs:
            out, err = p.communicate(timeout=300)
            assert p.returncode == 0, err[-300:]
            assert out == alone[k].stdout
            assert err == alone[k].stderr
        # a second wave on the same broker: the channels freed by the first are reset and handed out again
        procs = []
        for i in range(16):
            k = (3 * i) % 8
            a = list(streams[k][0]); a[-2] = files[k]
            procs.append((k, subprocess.Popen([os.path.join(BIN, "fsk_demod")] + a, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
        for k, p in procs:
            out, err = p.communicate(timeout=300)
            assert p.returncode == 0 and out == alone[k].stdout and err == alone[k].stderr
    finally:
        broker.send_signal(signal.SIGTERM)
        _, berr = broker.communicate(timeout=30)
    line = [l for l in berr.decode().splitlines() if l.startswith("broker: groups")][-1].split()
    st = {line[i]: int(line[i + 1]) for i in range(1, len(line), 2)}
    assert st["groups"] == 2 and st["clients"] == 72 + 16
    assert st["max_batch"] >= 48 and st["frames"] >= 8 * st["steps"], st          # one launch sequence per block of (nearly) all clients


def test_broker_without_gpu_reports_the_error_to_the_client(tmp_path):
    """CPU: the protocol end to end up to engine creation — without a HIP device the broker answers ERROR, the shim prints it and exits 1
    (there is no CPU fallback on either side)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    sock = str(tmp_path / "b.sock")
    broker = subprocess.Popen([os.path.join(BIN, "sonde_broker"), "--socket", sock], stderr=subprocess.PIPE)
    try:
        for _ in range(100):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        r = subprocess.run([os.path.join(BIN, "fsk_demod"), "--cs16", "-s", "2", "48000", "4800", "-", "-"], input=b"\0" * 8000,
                           env=dict(os.environ, SONDE_BROKER=sock), capture_output=True, timeout=60)
        assert r.returncode == 1 and b"Couldn't open files" in r.stderr and r.stdout == b""
        # a decoder shim on IF-rate IQ: the same protocol, the decoder's own error line and exit code (255 like the reference's `return -1`)
        r = subprocess.run([os.path.join(BIN, "rs41mod"), "-r", "--iq2", "--lpIQ", "-", "48000", "16"], input=b"\0" * 8000,
                           env=dict(os.environ, SONDE_BROKER=sock), capture_output=True, timeout=60)
        assert r.returncode == 255 and b"error: init buffers" in r.stderr and r.stdout == b""
        # the base-rate form is not served by the broker: the shim opens the GPU itself (and fails the same way without one)
        r = subprocess.run([os.path.join(BIN, "rs41mod"), "-r", "--IQ", "0.1", "-", "2400000", "16"], input=b"\0" * 8000,
                           env=dict(os.environ, SONDE_BROKER=sock), capture_output=True, timeout=60)
        assert r.returncode == 255 and b"error: init buffers" in r.stderr
    finally:
        broker.send_signal(signal.SIGTERM)
        _, berr = broker.communicate(timeout=30)
    assert b"broker: groups 0" in berr


def _decoder_streams():
    from tools import synth
    ecef = dict(ecef_cm=(418833319, 85974133, 473346430))
    out = []
    for k in range(4):            # RS41 on FM audio (WAV): what auto_rx's classic chain pipes into rs41mod (decode.py:375-417)
        x = synth.rs41_capture(sr=48_000, seconds=4.3, fq=0.0, noise_sigma=0.02 + 0.02 * k, frame_kw=ecef, n_frames=4, t_first=0.1 + 0.07 * k, seed=500 + k,
                               first_frame_no=200 + 10 * k)
        out.append(("rs41mod", ["--ptu2", "--json", "--jsnsubfrm1"], synth.wav_bytes(synth.fm_audio(x), 48_000)))
    for k in range(2):            # RS41 on IF-rate IQ, raw lines
        x = synth.rs41_capture(sr=48_000, seconds=3.3, fq=0.0, noise_sigma=0.05 + 0.05 * k, frame_kw=ecef, n_frames=3, t_first=0.2, seed=510 + k)
        out.append(("rs41mod", ["-r", "--ecc2", "--crc", "--iq2", "--lpIQ", "-", "48000", "16"], x.tobytes()))
    for k in range(2):            # DFM on FM audio, raw frames with the polarity found automatically
        x = synth.dfm_capture(sr=48_000, seconds=3.2, fq=0.0, noise_sigma=0.03 + 0.03 * k, seed=520 + k)
        out.append(("dfm09mod", ["-r", "--ecc", "--auto"], synth.wav_bytes(synth.fm_audio(x), 48_000)))
    for k in range(2):            # M10 on FM audio with auto_rx's options (decode.py:544)
        x = synth.m10_capture(sr=48_000, seconds=4.3, noise_sigma=0.03 + 0.02 * k, seed=530 + k, frame_fn=lambda j, k=k: synth.m10_frame(j, rng=np.random.default_rng(40 + 7 * k + j)))
        out.append(("m10mod", ["--json", "--ptu", "-vvv"], synth.wav_bytes(synth.fm_audio(x), 48_000)))
    x = synth.m10_capture(sr=48_000, seconds=4.3, noise_sigma=0.03, seed=540, baud=9600.0,       # M20 on IF-rate IQ
                          frame_fn=lambda j: synth.m20_frame(j, fw=8, pressure_hpa=700.0 - j, rng=np.random.default_rng(60 + j)))
    out.append(("m20mod", ["--json", "--ptu", "-vv", "--iq2", "-", "48000", "16"], x.tobytes()))
    return out


@pytest.mark.gpu
def test_decoder_shims_share_one_engine_per_configuration(tmp_path):
    """rs41mod / dfm09mod / m10mod on FM audio and IF-rate IQ behind the broker: every process owns a channel of one engine per configuration
    (sonde_engine_restart_channel / finish_channel), all channels advance in the same process call, and every process prints what it prints
    on its own — including a second wave of processes that reuses the channels of the first."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    streams = _decoder_streams()
    env0 = dict(os.environ, SONDE_JSN_VERSION="oracle")
    alone = [subprocess.run([os.path.join(BIN, b)] + a, input=x, capture_output=True, timeout=120, env=env0) for b, a, x in streams]
    assert all(r.returncode == 0 and len(r.stdout) > 0 for r in alone), [r.stderr[-200:] for r in alone]
    sock = str(tmp_path / "broker.sock")
    broker = subprocess.Popen([os.path.join(BIN, "sonde_broker"), "--socket", sock, "--slots", "32"], stderr=subprocess.PIPE)
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        env = dict(env0, SONDE_BROKER=sock)
        files = []
        for i, (b, a, x) in enumerate(streams):
            p = tmp_path / ("dec%d.bin" % i)
            p.write_bytes(x)
            files.append(str(p))
        for wave in range(2):
            procs = []
            for rep in range(3 if wave == 0 else 2):
                for k, (b, a, x) in enumerate(streams):
                    procs.append((k, subprocess.Popen([os.path.join(BIN, b)] + a, env=env, stdin=open(files[k], "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
            for k, p in procs:
                out, err = p.communicate(timeout=300)
                assert p.returncode == 0, (streams[k][0], err[-300:])
                assert out == alone[k].stdout, (wave, streams[k][0], streams[k][1], out[:200], alone[k].stdout[:200])
    finally:
        broker.send_signal(signal.SIGTERM)
        _, berr = broker.communicate(timeout=30)
    line = [l for l in berr.decode().splitlines() if l.startswith("broker: decoder_groups")][-1].split()
    st = {line[i]: int(line[i + 1]) for i in range(1, len(line), 2)}
    assert st["decoder_groups"] == 5 and st["max_batch"] >= 6 and st["records"] > 0, st


@pytest.mark.gpu
def test_a_paused_decoder_neither_stalls_its_peers_nor_dies(tmp_path):
    """Two rs41mod shims of one configuration behind the broker.  One of them gets its input in two parts with a pause longer than --stall-ms in
    between (an SDR hiccup): the other one still prints exactly what it prints on its own and ends long before the pause is over; the paused one
    is parked (its channel ended and re-armed), keeps running, decodes the frames behind the pause and exits 0 — it is not dropped."""
    from tools import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    args = ["-r", "--ecc2", "--crc", "--iq2", "--lpIQ", "-", "48000", "16"]
    xa = synth.rs41_capture(sr=48_000, seconds=3.3, fq=0.0, noise_sigma=0.05, n_frames=3, t_first=0.2, seed=610).tobytes()
    xb = synth.rs41_capture(sr=48_000, seconds=5.3, fq=0.0, noise_sigma=0.05, n_frames=5, t_first=0.2, seed=611).tobytes()
    alone_a = subprocess.run([os.path.join(BIN, "rs41mod")] + args, input=xa, capture_output=True, timeout=120)
    alone_b = subprocess.run([os.path.join(BIN, "rs41mod")] + args, input=xb, capture_output=True, timeout=120)
    assert alone_a.returncode == 0 and len(alone_a.stdout.splitlines()) == 3 and len(alone_b.stdout.splitlines()) == 5
    sock = str(tmp_path / "broker.sock")
    broker = subprocess.Popen([os.path.join(BIN, "sonde_broker"), "--socket", sock, "--slots", "4", "--stall-ms", "300"], stderr=subprocess.PIPE)
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        env = dict(os.environ, SONDE_BROKER=sock)
        pb = subprocess.Popen([os.path.join(BIN, "rs41mod")] + args, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        cut = 4 * 103200                                     # 2.15 s of the 5.3 s: the second frame is complete (2.12 s), the third starts at 2.2 s
        pb.stdin.write(xb[:cut]); pb.stdin.flush()
        time.sleep(0.5)                                      # b is connected and has delivered what it has
        t0 = time.time()
        ra = subprocess.run([os.path.join(BIN, "rs41mod")] + args, env=env, input=xa, capture_output=True, timeout=120)
        ta = time.time() - t0
        assert ra.returncode == 0 and ra.stdout == alone_a.stdout, ra.stderr[-300:]
        assert ta < 10.0                                     # it waited for b at most --stall-ms, not for b's input to come back
        time.sleep(1.0)
        pb.stdin.write(xb[cut:]); pb.stdin.close()
        out_b = pb.stdout.read(); err_b = pb.stderr.read()
        assert pb.wait(timeout=120) == 0, err_b[-300:]
        lines_b, want_b = out_b.decode().splitlines(), alone_b.stdout.decode().splitlines()
        assert lines_b[:2] == want_b[:2]                     # the frames before the pause
        assert lines_b[-2:] == want_b[-2:]                   # the channel started over: the frames well behind the pause are decoded again
    finally:
        broker.send_signal(signal.SIGTERM)
        _, berr = broker.communicate(timeout=30)
    line = [l for l in berr.decode().splitlines() if l.startswith("broker: decoder_groups")][-1].split()
    st = {line[i]: int(line[i + 1]) for i in range(1, len(line), 2)}
    assert st["parked"] >= 1, st
