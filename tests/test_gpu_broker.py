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
    finally:
        broker.send_signal(signal.SIGTERM)
        _, berr = broker.communicate(timeout=30)
    assert b"broker: groups 0" in berr
