#!/usr/bin/env python3
"""Runs ON THE GPU BOX (gpurun): the binaries of this repo (host/bin) on the pipelines of tools/caller_cases.py, stdout / stderr / exit code of
every GPU-backed stage into one npz (default gpurun_out/cli_ours.npz; the committed copy is tests/golden/cli_ours.npz).  tests/test_caller_contract.py
feeds these bytes to auto_rx's own parsers in the container that has /root/reference; tests/test_gpu_cli_recorded.py checks on every GPU run that
the binaries still print exactly this."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import caller_cases as cc  # noqa: E402

BIN = os.path.join(ROOT, "host", "bin")


def run_all(bindir=BIN, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    out = {}
    for name, (cap, argv) in cc.DETECT.items():
        r = subprocess.run([os.path.join(bindir, "dft_detect")] + argv, input=cc.capture(cap).tobytes(), capture_output=True, env=env, timeout=300)
        out[name + ".stdout"], out[name + ".stderr"], out[name + ".rc"] = r.stdout, r.stderr, r.returncode
    for name, (cap, argv) in cc.BATCH.items():
        r = subprocess.run([os.path.join(bindir, "dft_detect")] + argv, input=cc.capture(cap).tobytes(), capture_output=True, env=env, timeout=300)
        out[name + ".stdout"], out[name + ".stderr"], out[name + ".rc"] = r.stdout, r.stderr, r.returncode
    for name, (cap, fargv, _dec, _dargv, _typ) in cc.FSK.items():
        r = subprocess.run([os.path.join(bindir, "fsk_demod")] + fargv, input=cc.capture(cap).tobytes(), capture_output=True, env=env, timeout=300)
        out[name + ".stdout"], out[name + ".stderr"], out[name + ".rc"] = r.stdout, r.stderr, r.returncode
    return out


def save(out, path):
    np.savez_compressed(path, **{k: (np.frombuffer(v, np.uint8) if isinstance(v, bytes) else np.array(v)) for k, v in out.items()})


def load(path):
    z = np.load(path)
    return {k: (int(z[k]) if k.endswith(".rc") else z[k].tobytes()) for k in z.files}


if __name__ == "__main__":
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "cli_ours.npz")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    o = run_all()
    save(o, dst)
    for k in sorted(o):
        print(k, o[k] if k.endswith(".rc") else len(o[k]))
