"""Multi-rank readiness on ONE GPU: `bench.py --gpus 2` launched the way the driver launches it (torch.distributed.run, one process per
rank), with SONDE_DIST_BACKEND=gloo so that both ranks may share device 0 — exercises Dist, the per-rank engines, the per-step summary
all_gather (pipelined: from the engine's snapshots), rank_ms_per_step, sum_ints and the per-rank oracle verification end to end.  No
scaling figure is taken from this; the driver measures N = 1, 2, 4, 8 over RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("lag", [1, 0])
def test_bench_two_ranks_on_one_device(lag):
    env = dict(os.environ, SONDE_DIST_BACKEND="gloo", SONDE_BENCH_NO_REPEAT="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SONDE_BENCH_VERBOSE="1")      # (the full object, not the compact line)
    port = 29600 + (os.getpid() + lag) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--channels", "40", "--lag", str(lag)]
    r = subprocess.run(cmd, capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints, nobody else
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and cfg["channels_per_gpu"] == 40
    assert len(cfg["rank_ms_per_step"]) == 2 and all(t > 0 for t in cfg["rank_ms_per_step"])
    assert abs(d["ms_per_step"] - max(cfg["rank_ms_per_step"])) < 1e-3                 # max over ranks
    assert abs(d["value"] - 2 * 40 * 2.4e6 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]        # whole-job samples / max time
    # summed over ranks; the bench bank carries an error mix (bench.ERROR_MIX: 2 of 20 captures beyond the code): 36 of every 40 channels decode
    assert cfg["frames_decoded"] >= 2 * 40 * 5 and cfg["frames_ecc_ok"] == cfg["frames_decoded"] * 36 // 40
    assert cfg["frames_ecc_failed"] == cfg["frames_decoded"] - cfg["frames_ecc_ok"] and cfg["frames_repaired"] > 0
    assert cfg["frames_decoded_by_host_rs"] == 0                                                          # the Reed-Solomon decoder ran on the device
    assert cfg["verified_channels"] == 80 and not cfg.get("verify_failed")                                # both ranks' channels against the oracle
    assert cfg["frame_fetch_lag"] == lag
    assert "cpu_baseline" not in d and "detect_in_step" not in d                                          # single-GPU extras stay out of N > 1 lines


@pytest.mark.parametrize("config", ["scan_wide", "fsk_mixed", "mixed_2400k"])
def test_other_configs_two_ranks_on_one_device(config):
    """BASELINE configs[2] and [3] the way the driver would launch them on N GPUs (configs[3] is quoted on four): the N-rank line executes, every rank does its own
    full workload (weak scaling: independent streams / channels per GPU, no data-path collective), rank 0 prints one line with the job's aggregate"""
    env = dict(os.environ, SONDE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SONDE_BENCH_VERBOSE="1")
    port = 29300 + (os.getpid() + len(config)) % 300
    extra = ["--channels", "48"] if config == "fsk_mixed" else ["--channels", "20"] if config == "mixed_2400k" else []
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--config", config, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert len(cfg["rank_ms_per_step"]) == 2 and all(t > 0 for t in cfg["rank_ms_per_step"])
    assert abs(d["ms_per_step"] - max(cfg["rank_ms_per_step"])) < 1e-3
    assert "cpu_baseline" not in d
    if config == "fsk_mixed":
        assert cfg["channels_per_gpu"] == 48 and cfg["checked_channels"] == 48 and cfg["verified_channels"] == 48          # rank 0's channels against the reference modem
        assert abs(d["value"] - 2 * (16 * 48000 + 16 * 50000 + 16 * 48080) / (d["ms_per_step"] * 1e-3) / 1e6) < 0.02 * d["value"]
    elif config == "mixed_2400k":
        # ONE mixed-type engine per rank; both ranks' channels against the reference decoders (summed over ranks), the summaries gathered from snapshots every step
        assert cfg["channels"] == {"rs41": 10, "dfm": 6, "m10": 4} and cfg["verified_channels"] == cfg["checked_channels"] == {"rs41": 20, "dfm": 12, "m10": 8}
        assert abs(d["value"] - 2 * 20 * 2.4e6 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.02 * d["value"]
        assert d["roofline"]["kernel"] == "k_mix_decimate50" and d["roofline"]["launches"] == d["steps"]            # one decimator launch per step for all types
        assert "host_decode_ab" not in d and "detect_in_step" not in d                                            # single-GPU extras stay out of N > 1 lines
    else:
        assert cfg["channels"] == 256 and len(cfg["detections_last_step"]) >= 10                                           # the dozen planted sondes
        assert abs(d["value"] - 2 * 10e6 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.02 * d["value"]
