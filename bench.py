#!/usr/bin/env python3
"""bench.py — throughput of the radiosonde hot path on MI355X (BASELINE.json metric), one JSON line per run.

  --config demod (default, the headline; BASELINE configs[1] batched to the per-GPU share of configs[4])
      One step = the whole RS41 `--IQ fq --lpIQ` path (mix + decimate -> IF chain -> header correlation -> frame sync -> frame fetch
      + RS ECC) over CHANNELS channels x 1 s of 2.4 Msps cs16 IQ per GPU, resident in HBM.  Channels shard across ranks with no
      data-path collective; the per-channel detection summaries (32 B, written by the frame-sync kernel) are all_gathered from device
      memory once per step.  After the timed loop every channel's last frame is compared with the CPU oracle's frame for
      the stream that channel saw (`config.verified_channels`).  Extra objects of the default single-GPU run, none of them part of `value`:
      `detect_in_step` (BASELINE configs[4]: the same steps with the dft_detect scanner re-scanning a rotating 1/16 of the channels inside
      every step), `pcie_inclusive` (the same steps fed from pinned host memory), `scan_wide` and `fsk_mixed` (BASELINE configs[2] and
      configs[3], each with its own ms_per_step / roofline / cpu_baseline — the objects `--config scan_wide|fsk_mixed` print on their own).
  --config scan_wide (BASELINE configs[2]): 256 channels out of ONE 10 Msps stream -> dft_detect scanner, per 0.2 s of stream.
  --config fsk_mixed (BASELINE configs[3]): 1024 mixed RS41 / DFM09 / M10 channels through the 2-FSK modem (fsk_demod path).
  --config mixed_2400k (BASELINE configs[4] at configs[3]'s type mix): 512 channels at 2.4 Msps, 50 % RS41 / 30 % DFM09 / 20 % M10, demodulated and block-decoded on the device.

  python bench.py --gpus 1 [--config demod] [--steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue wait for each other's kernels.  The mixed
# configurations drive 6 - 8 streams (three modems + their three device decoders; decimator, IF-rate stages, decoder, record copies): with four queues a modem's launch
# sat behind another family's decoder kernel (profiles/r6c_fsk_mixed_hw_queues.txt: fsk_mixed 1.80 -> 1.63 ms).  Read once, when the runtime initialises; a
# deployment knob like any other, left alone if the caller has set it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

SR = 2_400_000
BANK = 20                # unique synthetic captures tiled over the channels
# on-air bit flips per frame of each capture (tools/synth.rs41_capture bit_errors; the flips fall on both interleaved RS(255,231) codewords):
# 50 % of the channels clean, 30 % with 3 .. 10 symbol errors, 15 % at the code's limit (t = 12 per codeword: some frames need the 2nd pass
# of --ecc2, some fail), 5 % beyond it.  What the decoder makes of each class is reported in config.error_mix.
ERROR_MIX = [0] * 10 + [3, 4, 5, 6, 8, 10] + [20, 22, 24] + [48]
ERROR_CLASSES = ["clean"] * 10 + ["3-10 symbol errors"] * 6 + ["near t = 12 per codeword"] * 3 + ["uncorrectable"]
PROFILE_TAG = "r6"       # profiles/<tag>_*_traffic.json: HBM traffic of the dominant kernel from rocprofv3 --pmc passes of this command


def make_bank(seconds: float = 1.0):
    from tools import synth
    rng = np.random.default_rng(2024)
    fqs, caps = [], []
    for b in range(BANK):
        fq = synth.snap_fq(float(rng.uniform(-0.4, 0.4)), SR)
        caps.append(synth.rs41_capture(sr=SR, seconds=seconds, fq=fq, n_frames=1, t_first=0.15, seed=100 + b,
                                       first_frame_no=1000 * (b + 1), sonde_id="T%07d" % b, noise_sigma=0.01, bit_errors=ERROR_MIX[b]))
        fqs.append(fq)
    return fqs, caps


def _time_reference(cmds, inputs, units_per_pass, unit, what, budget_s=15.0):
    """ncores concurrent reference processes (oracle/_ref) fed from files, repeated for ~budget_s"""
    ncores = len(cmds)
    reps, t0, total, out_bytes = 0, time.perf_counter(), 0.0, 0
    while time.perf_counter() - t0 < budget_s:
        procs = [subprocess.Popen(c, stdin=open(i, "rb"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for c, i in zip(cmds, inputs)]
        for pr in procs:
            out, _ = pr.communicate()
            out_bytes += len(out)
        total += ncores * units_per_pass
        reps += 1
    dt = time.perf_counter() - t0
    return dict(value=total / dt / 1e6, unit=unit, cores=ncores, kind="reference", per_core=total / dt / 1e6 / ncores,
                sample=f"{ncores} concurrent reference {what} x {reps} passes", stdout_bytes=out_bytes)


def cpu_baseline_demod(fqs, caps, budget_s: float = 15.0):
    """Reference rs41mod (oracle/_ref, built from /root/reference) timed on this host's cores; falls back to the
    single-threaded CPU restatement (oracle/liboracle.so) when the compiled reference is absent."""
    from oracle import bind
    ncores = max(1, min(os.cpu_count() or 1, 8))
    secs = 20
    if bind.have_ref():
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for b in range(min(BANK, ncores)):
                p = os.path.join(td, f"c{b}.cs16")
                with open(p, "wb") as f:
                    for _ in range(secs):
                        f.write(caps[b].tobytes())
                paths.append(p)
            exe = os.path.join(bind.REFDIR, "rs41mod")
            for p in paths:                      # page cache
                open(p, "rb").read()
            cmds = [[exe, "-r", "--ecc2", "--crc", "--IQ", repr(fqs[k % len(paths)]), "--lpIQ", "-", str(SR), "16"] for k in range(ncores)]
            r = _time_reference(cmds, [paths[k % len(paths)] for k in range(ncores)], secs * SR, "Msamples/s",
                                f"rs41mod processes (-O3, demod_mod.o -Ofast) over {secs} s of the same 2.4 Msps cs16 RS41 captures", budget_s)
        return r
    t0 = time.perf_counter()
    n, total = 0, 0
    while time.perf_counter() - t0 < budget_s:
        bind.ora_rs41_decode(caps[n % BANK], SR, fq=fqs[n % BANK], want_soft=False)
        total += len(caps[n % BANK]) // 2
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=total / dt / 1e6, unit="Msamples/s", cores=1, kind="port", per_core=total / dt / 1e6,
                sample=f"{n} x 1 s 2.4 Msps cs16 RS41 captures through the single-threaded CPU restatement (oracle/)")


def _traffic(name, algorithmic_bytes):
    """HBM traffic of the dominant kernel: it cannot be sampled from inside the process — quoted from the committed rocprofv3 --pmc
    summary of this same command (tools/profile_round.sh) when that was taken on the same launch geometry"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{name}_traffic.json")))
        if tj["algorithmic_bytes"] == algorithmic_bytes:
            return round(tj["traffic_bytes"] / 1e9, 3), f"GB per launch, (2 x FETCH_SIZE + WRITE_SIZE) from profiles/{PROFILE_TAG}_{name}_traffic.json (rocprofv3 --pmc)"
    except Exception:
        pass
    return None, None


class Dist:
    """torch.distributed as the driver launches it (one rank per GPU, RCCL), or a single process"""

    def __init__(self):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (libsonde_hip has no CPU fallback)")
        # SONDE_DIST_BACKEND=gloo: the multi-rank code path on a box with fewer GPUs than ranks (tests/test_gpu_multirank.py): ranks share
        # the devices round robin and the collectives go through host memory; the driver's runs use nccl (= RCCL), one rank per GPU
        self.backend = os.environ.get("SONDE_DIST_BACKEND", "nccl")
        if self.backend != "nccl":
            self.local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def finish_times(self, dt):
        """(max over ranks, per-rank list)"""
        from radiosonde_auto_rx_amd import shard
        if not self.dist:
            return dt, [dt]
        per = shard.gather_floats(self.dist, dt, self.world, self.dev)
        return max(per), per

    def sum_ints(self, *vals):
        if not self.dist:
            return vals
        t = self.torch.tensor(list(vals), device=self.dev if self.backend == "nccl" else "cpu", dtype=self.torch.int64)
        self.dist.all_reduce(t)
        return tuple(int(v) for v in t)

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def _lead_in(eng, ptr, stride):
    """Align the call boundaries with the reference's IQ-DC segments (75000 * 2^k samples, then every 2.4 M): untimed lead-in calls up
    to the last short boundary, after which every 1 s step is exactly one segment.  Returns the lengths fed (each from the row start)."""
    fed = []
    while eng.samples_to_dc_boundary() < SR:
        n = eng.samples_to_dc_boundary()
        eng.process_device(ptr, stride, n)
        fed.append(n)
    eng.fetch_frames_np()
    return fed


def _oracle_last_frames(fqs, caps, fed):
    """What the CPU oracle (oracle/, the checker) decodes last from the stream a channel of bank b saw: the lead-in pieces, then whole
    seconds of the capture until the IQ-DC mean, the filter histories and the sync state repeat (4 s).  -> per bank (frame bytes, len, ecc)"""
    from oracle import bind
    out = []
    for b in range(BANK):
        cap = caps[b]
        x = np.concatenate([cap[:2 * n] for n in fed] + [cap] * 4)
        o = bind.ora_rs41_decode(x, SR, fq=fqs[b], want_soft=False)
        # [-1] is the frame in progress at the end of the oracle's input (emitted with the bits that exist); [-2] is the last whole one
        out.append((bytes(o["frames"][-2]), int(o["flen"][-2]), int(o["ecc"][-2])) if o["n"] >= 2 else None)
    return out


def bench_demod(args, D: Dist):
    torch = D.torch
    from radiosonde_auto_rx_amd.engine import Engine
    from radiosonde_auto_rx_amd import shard
    C = args.channels or int(os.environ.get("SONDE_BENCH_CHANNELS", "512"))
    steps = args.steps or 400
    warmup = 5 if args.warmup is None else args.warmup
    fqs, caps = make_bank()
    ch_bank = [(c + D.rank) % BANK for c in range(C)]
    ch_fq = [fqs[b] for b in ch_bank]
    bank_t = torch.from_numpy(np.stack(caps)).to(D.dev)                     # [BANK, 2*SR] int16
    idx = torch.tensor(ch_bank, device=D.dev)
    iq = bank_t.index_select(0, idx).contiguous()                         # [C, 2*SR] resident input
    STRIDE = SR + int(os.environ.get("SONDE_BENCH_PAD", "0"))              # experiments: channel rows padded apart (samples)
    if STRIDE != SR:
        padded = torch.zeros((C, 2 * STRIDE), dtype=iq.dtype, device=D.dev)
        padded[:, :2 * SR] = iq
        iq = padded
    del bank_t
    torch.cuda.synchronize()

    lag = args.lag
    eng = Engine(ch_fq, SR, device=D.local_rank, lp_iq=True, ecc=2, max_chunk=SR, max_frames=4 * C, pipeline=(lag > 0 and args.two_streams))
    summary = shard.summary_buffer(C, D.dev)                              # written by the frame-sync kernel, gathered from device memory
    eng.set_summary(summary.data_ptr(), D.rank * C)
    # pipelined steps (lag 1): the all_gather of step k reads the snapshot the engine took at the end of call k-1 while call k runs
    snaps = shard.summary_buffer(2 * C, D.dev).view(2, C, shard.SUMMARY_BYTES) if (D.dist and lag) else None
    snap_next = eng.set_summary_snapshots(snaps.data_ptr()) if snaps is not None else 0
    gathered = [torch.empty_like(summary) for _ in range(D.world)] if D.dist else None
    fed = _lead_in(eng, iq.data_ptr(), STRIDE)
    calls = [0]
    last = {}                                                             # channel -> its most recent frame record (verification, untimed)

    def step(lag=lag):
        # lag = 0: every step waits for its own frames.  lag = 1: the frames of call k-1 are fetched (D2H of the records, host RS ECC) while
        # call k runs — the host work leaves the critical path; kernels of one engine stream still run one after the other, so the HIP-event
        # time of k_mix_decimate50 is that kernel alone.  --two-streams: the IF-rate kernels of call k-1 also overlap the decimator of call k.
        eng.process_device(iq.data_ptr(), STRIDE, SR)
        calls[0] += 1
        frames = eng.fetch_frames_np(lag=lag)
        if D.dist:                                                        # 32 B per channel over RCCL, device to device
            src = summary if snaps is None else snaps[(snap_next + calls[0] - 2) & 1]
            if snaps is None or calls[0] >= 2:
                shard.gather_summaries(D.dist, src, D.world, gathered)
        return frames

    for _ in range(warmup):
        step()
    eng.fetch_frames_np(lag=0)
    # at least 2 s of timed work (2.2 s aimed at: the probe runs a few per cent slow) whatever K is (the driver's GPU-busy sampler needs to see the run): the K steps are repeated R times and
    # every figure below is over all K x R steps.  R comes from a steady-state probe (the warm-up itself carries first-call costs)
    eng.sync(); torch.cuda.synchronize()
    t_p = time.perf_counter()
    for _ in range(10):
        step()
    eng.fetch_frames_np(lag=0)
    eng.sync(); torch.cuda.synchronize()
    est = max((time.perf_counter() - t_p) / 10, 1e-4)
    repeats = 1 if os.environ.get("SONDE_BENCH_NO_REPEAT") else max(1, int(np.ceil(2.2 / (est * steps))))
    total_steps = steps * repeats
    # what this box's HBM gives a plain read stream over the very buffer the decimator reads (best of five passes), and the clocks around the timed loop
    stream_gbps = _stream_probe(iq.data_ptr(), int(iq.numel()) * iq.element_size())
    clocks0 = _device_clocks()
    eng.profile(1)                      # timed region: HIP events around the dominant kernel only (2 events per step)
    D.barrier()
    t0 = time.perf_counter()
    nframes, nok, nfixed, nsym = 0, 0, 0, 0
    host_ecc0 = eng.host_ecc_frames()
    fr_tail = [None, None, None]                                          # the last fetches (every channel ends a frame once per step)

    def tally(fr):
        nonlocal nframes, nok, nfixed, nsym
        e = fr["ecc"]
        nframes += len(fr)
        nok += int((e >= 0).sum())
        nfixed += int((e > 0).sum())
        nsym += int(e[e > 0].sum())

    for k in range(total_steps):
        fr = step()
        fr_tail[k % 3] = fr
        tally(fr)
    fr_tail = [fr_tail[(total_steps + i) % 3] for i in range(3)] + [eng.fetch_frames_np(lag=0)]     # oldest first; drain: all work is inside the timed region
    tally(fr_tail[-1])
    eng.sync()
    D.barrier()
    dt_local = time.perf_counter() - t0
    clocks1 = _device_clocks()
    dt, per_rank = D.finish_times(dt_local)
    host_ecc = eng.host_ecc_frames() - host_ecc0
    nframes, nok, nfixed, nsym, host_ecc = D.sum_ints(nframes, nok, nfixed, nsym, host_ecc)
    md_ms, md_n = eng.kernel_ms("mix_decimate")

    # untimed: every channel's last frame against the CPU oracle's last frame of the same stream (frame bytes, length, ECC verdict)
    for arr in fr_tail:
        for f in (arr if arr is not None else ()):
            last[int(f["channel"])] = f
    verified, mismatched = 0, []
    want = _oracle_last_frames(fqs, caps, fed) if not args.no_verify else None
    if want is not None:
        for c in range(C):
            f, w = last.get(c), want[ch_bank[c]]
            ok = f is not None and w is not None and int(f["len"]) == w[1] and int(f["ecc"]) == w[2] and bytes(f["frame"][:w[1]]) == w[0][:w[1]]
            verified += int(ok)

            if not ok and len(mismatched) < 8:
                mismatched.append(c)
        verified, = D.sum_ints(verified)

    error_mix = {"bit_flips_per_frame_by_capture": ERROR_MIX, "classes": {k: ERROR_CLASSES.count(k) / BANK for k in dict.fromkeys(ERROR_CLASSES)},
                 "rs41_ecc_value_by_capture": [(want[b][2] if want is not None and want[b] is not None else None) for b in range(BANK)],
                 "note": "channel c decodes capture (c + rank) mod %d; rs41_ecc value = repaired symbols of both codewords, negative = codeword 1 / 2 / both "
                         "unrepairable (CPU oracle's verdict; every channel's device result is compared with it: verified_channels)" % BANK}
    # untimed: the same steps with the Reed-Solomon decoder on the host (the round-3 arrangement: k_framesync leaves first-pass syndromes, the
    # fetch runs rs41_ecc on one host thread) — what the device decoder replaces
    ab = None
    if D.world == 1 and not args.no_extras:
        eng.set_device_ecc(False)
        for _ in range(3):
            step()
        eng.fetch_frames_np(lag=0); eng.sync()
        h0, t_ab, n_ab = eng.host_ecc_frames(), time.perf_counter(), 60
        for _ in range(n_ab):
            step()
        eng.fetch_frames_np(lag=0); eng.sync()
        t_ab = (time.perf_counter() - t_ab) / n_ab
        ab = dict(ms_per_step=round(t_ab * 1e3, 3), frames_decoded_by_host_rs_per_step=round((eng.host_ecc_frames() - h0) / n_ab, 1), steps=n_ab,
                  note="SONDE_HOST_ECC form: syndromes on the device, rs_decode of every damaged codeword on one host thread inside the fetch")
        eng.set_device_ecc(True)
        for _ in range(3):
            step()
        eng.fetch_frames_np(lag=0); eng.sync()

    # untimed: the dominant kernel on a stream of its own order — a second engine with ONE stream (every kernel of a call behind the other), same channels, same
    # input: what k_mix_decimate50 does when no IF-rate kernel of the call before holds CU slots beside it.  `roofline.frac` above is the timed run's figure
    # (two streams: the tail runs beside the decimator and stretches it); this one goes beside it as `frac_kernel_alone`
    alone, kern_serial = None, None
    if D.world == 1 and not args.no_extras and lag > 0 and args.two_streams:
        # (the Reed-Solomon kernel too: on the timed engine it has a stream of its own beside the next call's decimator; SONDE_ECC_INLINE is read when an engine is made)
        had_inline = os.environ.get("SONDE_ECC_INLINE")
        os.environ["SONDE_ECC_INLINE"] = "1"
        try:
            e1 = Engine(ch_fq, SR, device=D.local_rank, lp_iq=True, ecc=2, max_chunk=SR, max_frames=4 * C, pipeline=False)
        finally:
            if had_inline is None:
                del os.environ["SONDE_ECC_INLINE"]
        _lead_in(e1, iq.data_ptr(), STRIDE)
        for _ in range(5):
            e1.process_device(iq.data_ptr(), STRIDE, SR); e1.fetch_frames_np(lag=lag)
        e1.fetch_frames_np(lag=0); e1.sync()
        e1.profile(1)
        n1, t1 = 80, time.perf_counter()
        for _ in range(n1):
            e1.process_device(iq.data_ptr(), STRIDE, SR); e1.fetch_frames_np(lag=lag)
        e1.fetch_frames_np(lag=0); e1.sync()
        t1 = (time.perf_counter() - t1) / n1
        ms1, k1 = e1.kernel_ms("mix_decimate")
        if ms1 > 0 and k1 > 0:
            alone = dict(avg_launch_ms=round(ms1, 4), frac=round(C * SR * 4 / (ms1 * 1e-3) / 1e9 / 8000.0, 4), launches=int(k1), ms_per_step_one_stream=round(t1 * 1e3, 3))
        # the per-kernel table of a step as EXECUTION times: on this one-stream engine every kernel of a call runs behind the other, so the HIP events around a kernel
        # bracket its work and nothing else (on the two-stream engine of the timed loop they also bracket its wait for CU slots the decimator holds: kernels_two_streams)
        e1.profile(2)
        nser = 20
        for _ in range(nser):
            e1.process_device(iq.data_ptr(), STRIDE, SR); e1.fetch_frames_np(lag=lag)
        e1.fetch_frames_np(lag=0); e1.sync()
        kern_serial = {}
        for k in ("mix_decimate", "if_chain", "header_corr", "framesync", "rs_ecc"):
            ms, n = e1.kernel_ms(k)
            kern_serial[k] = dict(ms_per_step=round(ms * n / nser, 4), launches_per_step=n / nser)
        e1.profile(0)
        e1.close()

    # untimed: per-kernel table of a step (events around every kernel cost ~0.1 ms of host time per step, so not in the timed region)
    # pipelined like the timed loop (a step that waits for its own frames lets the clocks drop between steps and reads 10 % slow)
    eng.profile(2)
    nprof = 20
    for _ in range(nprof):
        step()
    eng.fetch_frames_np(lag=0)
    kern = {}
    for k in ("mix_decimate", "if_chain", "header_corr", "framesync", "rs_ecc"):
        ms, n = eng.kernel_ms(k)
        kern[k] = dict(ms_per_step=round(ms * n / nprof, 4), launches_per_step=n / nprof)
    eng.profile(0)
    rec = shard.decode_summaries(summary)
    total_samples = D.world * C * SR * total_steps
    value = total_samples / dt / 1e6
    # dominant kernel = k_mix_decimate: algorithmic bytes = 4 B per complex cs16 sample;
    # per-step launches may differ in size (IQ-DC segment edges) so the rate is (bytes of all launches)/(time of all launches)
    md_total_s = md_ms * md_n / 1e3
    achieved = (C * SR * total_steps * 4) / md_total_s / 1e9 if md_total_s > 0 else 0.0
    traffic, traffic_src = _traffic("mix_decimate", C * SR * 4)
    out = None
    if D.rank == 0:
        out = {
            "metric": "IQ Msamples/s (RS41 --IQ --lpIQ demod + framesync + ECC), concurrent real-time 2.4 Msps channels = value/2.4",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": D.world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / total_steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RS41 2.4 Msps cs16 IQ, rs41mod --ecc2 --IQ fq --lpIQ (BASELINE configs[1]) x %d channels per GPU "
                                   "(per-GPU share of configs[4]), 1 s of signal per channel per step, input resident in HBM" % C,
                       "channels_per_gpu": C, "samples_per_channel_per_step": SR, "realtime_channels": round(value / 2.4, 1),
                       "repeats": repeats, "timed_steps": total_steps, "timed_seconds": round(dt, 3),
                       "frame_fetch_lag": lag, "two_streams": bool(lag > 0 and args.two_streams),
                       "frames_decoded": nframes, "frames_ecc_ok": nok, "frames_ecc_failed": nframes - nok, "frames_repaired": nfixed,
                       "symbols_repaired": nsym, "frames_decoded_by_host_rs": host_ecc,
                       "ecc": "rs41_ecc (--ecc2: RS(255,231) Euclid / Chien / Forney per codeword, 2nd pass with the known block ids) inside k_framesync on the "
                              "device for every frame with non-zero syndromes; the host formats only (frames_decoded_by_host_rs = 0)",
                       "error_mix": error_mix,
                       "verified_channels": verified if want is not None else None, "verify_mismatch_channels": mismatched,
                       "verify_note": "untimed, after the loop: the last frame of every channel (bytes, length, ECC verdict) equals the CPU oracle's last frame "
                                      "of the stream that channel saw (oracle/ora_rs41_decode on lead-in + 4 s of its capture)",
                       "rank_ms_per_step": [round(t / total_steps * 1e3, 3) for t in per_rank],
                       "summary_records": {"bytes_per_channel": shard.SUMMARY_BYTES, "channels_with_frames": int((rec["frames"] > 0).sum()),
                                           "frames_clean_on_device": int(rec["frames_clean"].sum())},
                       "kernels": kern_serial if kern_serial else kern, "kernels_two_streams": kern if kern_serial else None,
                       "kernels_note": "kernels: execution time per step of every kernel, HIP events on a ONE-stream engine (same channels, same input; untimed, 20 steps) — each "
                                       "kernel runs alone behind the one before, so the column adds up to that engine's step; kernels_two_streams: the same events on the timed "
                                       "engine, where a kernel's span includes its wait for CU slots the decimator on the other stream holds"},
            "roofline": {"bound": "hbm", "kernel": "k_mix_decimate50", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 4), "frac_kernel_alone": alone["frac"] if alone else None, "kernel_alone": alone,
                         "measured_stream_GBps": round(stream_gbps, 1) if stream_gbps else None,
                         "frac_of_measured": round(achieved / stream_gbps, 4) if stream_gbps else None,
                         "measured_note": "a plain read stream over the same 4.9 GB buffer in this process just before the timed loop (16 B per lane, non-temporal, no arithmetic; "
                                          "best of five passes): what THIS box's HBM delivers — boxes of the pool differ by several per cent",
                         "clocks_mhz": {"before": clocks0, "after": clocks1},
                         "traffic": traffic, "traffic_note": traffic_src,
                         "algorithmic_gb_per_launch": round(C * SR * 4 / 1e9, 3), "avg_launch_ms": round(md_ms, 4), "launches": md_n,
                         "step_frac": round(C * SR * 4 / (dt / total_steps) / 8e12, 4),
                         "note": "achieved = 4 B x complex samples of all timed k_mix_decimate50 launches / their HIP-event time on the engine stream; "
                                 "step_frac = the same bytes over the whole step time (all kernels + host); frac_kernel_alone = the same kernel on a one-stream engine "
                                 "(no IF-rate kernel beside it), untimed A/B in the same process: kernel_alone"},
        }
        if want is not None and verified != D.world * C:
            out["config"]["verify_failed"] = True
    # ---- extras on one GPU (never part of `value`)
    if D.world == 1 and not args.no_extras:
        out["host_ecc_ab"] = ab
        out["detect_in_step"] = detect_in_step_extra(D, eng, iq, ch_fq, C, STRIDE, lag)
        if not args.no_configs:                                       # the same at higher scan duties (BASELINE configs[4] "full detect" = duty 1)
            out["detect_in_step"]["duty_1_4"] = detect_in_step_extra(D, eng, iq, ch_fq, C, STRIDE, lag, groups=4)
            out["detect_in_step"]["duty_1_1"] = detect_in_step_extra(D, eng, iq, ch_fq, C, STRIDE, lag, groups=1)
        out["pcie_inclusive"] = pcie_extra(D, eng, iq, C)
    eng.set_summary(0)
    eng.close()
    del iq
    torch.cuda.empty_cache()
    if D.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_demod(fqs, caps)
    if D.world == 1 and not args.no_extras and not args.no_configs:
        import bench_configs
        try:
            out["side_paths"] = bench_configs.side_paths(D, caps, fqs)
        except Exception as exc:
            out["side_paths"] = {"error": repr(exc)}
        for name in ("scan_wide", "fsk_mixed", "mixed_2400k"):            # BASELINE configs[2], [3] and [4] at [3]'s type mix in the same line (short runs)
            sub = argparse.Namespace(**vars(args))
            sub.config, sub.steps, sub.warmup, sub.channels, sub.cpu_budget = name, None, None, 0, 6.0
            try:
                out[name] = bench_configs.run(sub, D, short=True)
            except Exception as exc:                                      # the headline must survive a failure beside it — and say so
                out[name] = {"error": repr(exc)}
    return out


def detect_in_step_extra(D: Dist, eng, iq, ch_fq, C, STRIDE, lag, groups=16, engine_step=None, engine_drain=None):
    """BASELINE configs[4] "full detect -> demod -> ECC" as one step: the demodulator step over all C channels with the dft_detect scanner
    (`--IQ fq --dc`, front end + 14 templates) re-scanning a rotating 1/16 of the channels over the same second, inside the step — the
    auto_rx duty cycle: decoders run on the channels that were found while the scanner keeps sweeping (scan.py:948, decode.py:869-913).
    The scanner works on its own (high-priority) stream beside the engine's and is driven from a host thread of its own; the step ends when the
    engine's frames of the previous call AND the scanner's detections of this one have been fetched.  Not part of `value`."""
    torch = D.torch
    from radiosonde_auto_rx_amd.scan import Scanner
    if engine_step is None:                                               # (another engine's step / drain: the mixed-type configuration, bench_configs.py)
        def engine_step():
            eng.process_device(iq.data_ptr(), STRIDE, SR)                          # asynchronous on the engine's stream(s)
            return eng.fetch_frames_np(lag=lag)

        def engine_drain():
            eng.fetch_frames_np(lag=0)
            eng.sync()
    per = max(1, C // groups)
    scs = [Scanner(SR, fq=ch_fq[g * per:(g + 1) * per], dc=True, cont=True, max_chunk=SR, device=D.local_rank) for g in range(groups)]
    found = [0]

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1)

    def scan(g):                                                          # on a host thread of its own (the C call releases the GIL): the scanner's
        sc = scs[g]                                                       # call waits for its kernels between stages, on its own high-priority stream
        sc.process_device(iq.data_ptr() + 4 * STRIDE * g * per, STRIDE, SR)
        return sum(1 for d in sc.fetch() if d["type"] in ("RS41", "DFM9", "M10", "M20"))

    def step(k):
        job = pool.submit(scan, k % groups)
        fr = engine_step()
        found[0] += job.result()                                                    # the step is over when both are
        return fr

    for k in range(max(groups, 3)):                                       # every scanner has seen a second (allocations, first windows)
        step(k)
    engine_drain()
    torch.cuda.synchronize()
    found[0] = 0
    n = max(12, 4 * groups)
    t0 = time.perf_counter()
    for k in range(n):
        step(k)
    engine_drain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    pool.shutdown()
    for sc in scs:
        sc.close()
    return dict(ms_per_step=round(dt * 1e3, 3), value=round(C * SR / dt / 1e6, 1), unit="Msamples/s", realtime_channels=round(C * SR / dt / 2.4e6, 1),
                channels_scanned_per_step=per, scan_duty=("1/%d of the channels per step, rotating" % groups) if groups > 1 else "every channel every step", steps=n,
                detections_per_scanned_channel=round(found[0] / float(n * per), 3),
                note="demodulator step over all channels + dft_detect scanner (front end, 14 templates) over 1 s of %s, inside the step" % (("a rotating 1/%d of them" % groups) if groups > 1 else "all of them"))


def pcie_extra(D: Dist, eng, iq, C):
    """the same step fed from pinned host memory (process_host: H2D copy, then the kernels): bounded by PCIe, never `value`"""
    torch = D.torch
    nch = C
    host = torch.empty((nch, 2 * SR), dtype=torch.int16).pin_memory()
    host.copy_(iq[:nch])
    arr = host.numpy()
    eng.process_host(arr)
    eng.fetch_frames_np()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        eng.process_host(arr)
        eng.fetch_frames_np()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # double-buffered: the copy of second k+1 (its own stream, pinned source) runs while the kernels of second k do
    bufs = [torch.empty_like(iq[:nch]), torch.empty_like(iq[:nch])]
    cs = torch.cuda.Stream()
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    with torch.cuda.stream(cs):
        bufs[0].copy_(host, non_blocking=True); evs[0].record(cs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps2 = 4
    for k in range(reps2):
        with torch.cuda.stream(cs):
            bufs[(k + 1) & 1].copy_(host, non_blocking=True); evs[(k + 1) & 1].record(cs)
        evs[k & 1].synchronize()
        eng.process_device(bufs[k & 1].data_ptr(), iq.shape[1] // 2, SR)
        eng.fetch_frames_np()
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / reps2
    del bufs
    return dict(value=round(nch * SR / dt2 / 1e6, 1), unit="Msamples/s", ms_per_step=round(dt2 * 1e3, 3), realtime_channels=round(nch * SR / dt2 / 2.4e6, 1),
                host_to_device_GBps=round(nch * SR * 4 / dt2 / 1e9, 1),
                sequential=dict(value=round(nch * SR / dt / 1e6, 1), ms_per_step=round(dt * 1e3, 3)),
                note="samples arrive in pinned host memory; double-buffered: the host-to-device copy of second k+1 on its own stream beside the kernels of second k "
                     "(process_device on the buffer that has landed) — bounded by the link, never `value`; `sequential` = process_host, copy and kernels in turn")


def _device_clocks(index=0):
    """current shader / memory clock of the device in MHz from sysfs (the line pp_dpm_* marks with '*'), or None where the box does not expose them"""
    import glob
    out = {}
    for name, key in (("pp_dpm_sclk", "sclk_mhz"), ("pp_dpm_mclk", "mclk_mhz")):
        val = None
        for path in sorted(glob.glob("/sys/class/drm/card*/device/" + name)):
            try:
                for line in open(path).read().splitlines():
                    if line.rstrip().endswith("*"):
                        val = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
                break
            except Exception:
                continue
        out[key] = val
    return out if any(v is not None for v in out.values()) else None


def _stream_probe(ptr, nbytes):
    """GB/s of a plain read stream over `nbytes` of device memory on this box, now (libsonde_hip sonde_probe_read_gbps): the measured figure beside the nominal peak"""
    import ctypes as C
    from radiosonde_auto_rx_amd.engine import lib
    L = lib()
    L.sonde_probe_read_gbps.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_double)]
    g = C.c_double(0)
    rc = L.sonde_probe_read_gbps(C.c_void_p(ptr), C.c_size_t(nbytes), 5, C.byref(g))
    return float(g.value) if rc == 0 and g.value > 0 else None


# ---- the printed line.  The driver keeps the last ~8 KB of stdout: the line must fit with every object in it.  What is printed is the numbers; the prose
# (notes, method descriptions) stays in the full object, written next to it (gpurun_out/bench_full.json, or $SONDE_BENCH_FULL), and in DESIGN.md §5 / INTEGRATION.md.
_DROP_KEYS = {"kernels_two_streams", "ecc", "error_mix", "frames_ok_means", "kernel_alone", "clocks_mhz", "rank_ms_per_step", "verify_mismatch_channels", "stdout_bytes", "per_core",
              "samples_per_channel_per_step", "repeats", "scan_duty", "sequential", "summary_records", "algorithmic_gb_per_launch", "launches", "unique_captures_per_family",
              "timed_seconds", "two_streams", "frame_fetch_lag", "fsk_demod_args", "what", "step", "soft_decisions", "detections_last_step", "stages", "stream_rate",
              "frames_dropped", "symbols_or_codewords_repaired", "per_type", "launches_per_step"}
_KEEP_LONG = {"workload": 230, "sample": 170, "metric": 130}
LINE_LIMIT = 7600


def compact(obj, top=True, head=True):
    """the object as it is printed: no notes, no long strings, sub-objects reduced to their figures"""
    if isinstance(obj, dict):
        out = {}
        for k, v in obj.items():
            if k.endswith("note") or k.endswith("notes") or k in _DROP_KEYS:
                continue
            if not top and k in ("higher_is_better", "scaling", "vs_baseline", "data", "n_gpus", "warmup"):     # (said once, by the headline)
                continue
            if isinstance(v, str):
                lim = _KEEP_LONG.get(k, 90) if head or k not in ("workload", "metric") else {"workload": 130, "metric": 70}[k]
                v = v if len(v) <= lim else v[:lim - 3] + "..."
            else:
                v = compact(v, False, head and (not top or k in ("config", "roofline", "cpu_baseline")))
            out[k] = v
        return out
    if isinstance(obj, list):
        return [compact(v, False, head) for v in obj] if len(obj) <= 24 else None
    if isinstance(obj, float):
        return float("%.5g" % obj)
    return obj


def emit(out):
    full = os.environ.get("SONDE_BENCH_FULL") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, "w") as f:
            json.dump(out, f)
    except OSError:
        pass
    if os.environ.get("SONDE_BENCH_VERBOSE"):
        print(json.dumps(out), flush=True)
        return
    c = compact(out)
    line = json.dumps(c, separators=(",", ":"))
    # should an object have grown: shed tables until the line fits the driver's tail, the figures of every object stay
    for path in (("fsk_mixed", "config", "consumers"), ("scan_wide", "config", "brute_force"), ("mixed_2400k", "kernels"), ("config", "kernels"), ("scan_wide", "config", "kernels_ms_per_launch"),
                 ("detect_in_step", "duty_1_4"), ("mixed_2400k", "config", "frames_ok_per_step")):
        if len(line) <= LINE_LIMIT:
            break
        d = c
        for k in path[:-1]:
            d = d.get(k, {}) if isinstance(d, dict) else {}
        if isinstance(d, dict) and d.pop(path[-1], None) is not None:
            line = json.dumps(c, separators=(",", ":"))
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="demod", choices=["demod", "scan_wide", "fsk_mixed", "mixed_2400k"])
    ap.add_argument("--channels", type=int, default=0, help="channels per GPU (0 = the configuration's own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="demod: skip the detect / PCIe-inclusive extras")
    ap.add_argument("--no-configs", action="store_true", help="demod: skip the scan_wide / fsk_mixed objects")
    ap.add_argument("--no-verify", action="store_true", help="demod: skip the oracle check of every channel's last frame")
    ap.add_argument("--lag", type=int, default=1, help="demod: 1 = frame fetch one step behind (default), 0 = every step waits for its own frames")
    ap.add_argument("--two-streams", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--one-stream", action="store_true", help="demod: decimator and IF-rate kernels on ONE in-order stream (the round-3 default; the A/B switch)")
    args = ap.parse_args()
    args.two_streams = not args.one_stream
    D = Dist()
    if args.config == "demod":
        out = bench_demod(args, D)
    else:
        import bench_configs
        out = bench_configs.run(args, D)
    if D.rank == 0:
        emit(out)
    D.close()


if __name__ == "__main__":
    main()
