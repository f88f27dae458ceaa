"""One scanner process for ALL the peaks of a scan step — the multi-channel counterpart of auto_rx's per-peak detection
(auto_rx/autorx/scan_async.py:40-295 `detect_sonde_async`, :298-378 `scan_peaks_concurrent`, :381-420 `run_async_scan`).

auto_rx's KA9Q path starts one `dft_detect -t <dwell> --iq --bw 15 --dc - 48000 16` per peak frequency on a channel the SDR server cuts out of
the band for it, a few at a time (`max_concurrent`).  With the whole band as ONE wideband IQ stream and the batched scanner of this repo, a
single `dft_detect -t <dwell> --IQ fq1,fq2,... --bw 15 --dc - <sr> 16` (host/dft_detect.c: every listed offset of the stream on stdin in the same
GPU launches) answers all peaks in one dwell.  This module is the caller-side binding:

    detections = run_batch_scan(peak_frequencies, center_frequency=..., sample_rate=..., iq_cmd="... |", rs_path=..., dwell_time=...)

returns what `run_async_scan(peak_frequencies, max_concurrent, **detect_kwargs)` returns — `[(frequency_quantised_to_1_kHz, sonde_type), ...]` —
so `SondeScanner.sonde_search` (auto_rx/autorx/scan.py, the `run_async_scan` call) can switch on one line.  Every channel's lines go through
auto_rx's OWN `parse_dft_detect_output` (scan.py:227) with the exit-code rule of `detect_sonde_async` (:236-262): that function is passed in
(`parse=`) or imported from `autorx.scan` when auto_rx is importable — nothing of it is restated here.

tests/test_caller_contract.py runs auto_rx's `scan_peaks_concurrent` over N reference `dft_detect` processes and this binding over the recorded
output of ONE batch process on the same band and asserts equal results.  Not needed by the engine itself (no GPU code in here)."""
from __future__ import annotations

import asyncio
import os
import shlex
from typing import Callable, Iterable, Optional

DETECTION_TIMEOUT_MULTIPLIER = 2.5            # as detect_sonde_async: the subprocess gets dwell_time x this


def batch_command(frequencies: Iterable[float], *, center_frequency: float, sample_rate: int, rs_path: str = "./", dwell_time: int = 10,
                  if_bw: int = 15, bits: int = 16) -> str:
    """the scanner half of the pipeline: `<rs_path>/dft_detect -t D --IQ fq1,fq2,... --bw B --dc - sr bits`; fq_k = (f_k - centre) / sr"""
    fqs = []
    for f in frequencies:
        fq = (float(f) - float(center_frequency)) / float(sample_rate)
        if not -0.5 < fq < 0.5:
            raise ValueError("peak %.0f Hz is outside the band of the stream (centre %.0f Hz, %d samples/s)" % (f, center_frequency, sample_rate))
        fqs.append("%.9f" % fq)
    return shlex.quote(os.path.join(rs_path, "dft_detect")) + " -t %d --IQ %s --bw %d --dc - %d %d 2>/dev/null" % (dwell_time, ",".join(fqs), if_bw, sample_rate, bits)


def split_batch_output(text: str, n: int):
    """stdout of the batch form -> per channel (the lines a single `dft_detect` would have printed, the exit code it would have returned).
    Lines: `<index> <fq> <reference line>` per detection, `# <index> <fq> <code>` at the end (host/dft_detect.c)."""
    lines = [[] for _ in range(n)]
    codes = [1] * n
    seen = [False] * n
    for raw in text.splitlines():
        parts = raw.split(" ", 3 if raw.startswith("#") else 2)
        try:
            if raw.startswith("#"):
                k, code = int(parts[1]), int(parts[3])
                if 0 <= k < n:
                    codes[k], seen[k] = code & 0xFF, True
            else:
                k = int(parts[0])
                if 0 <= k < n and len(parts) == 3:
                    lines[k].append(parts[2])
        except (ValueError, IndexError):
            continue
    return [("\n".join(l) + ("\n" if l else ""), c if s else None) for l, c, s in zip(lines, codes, seen)]


def _default_parser():
    from autorx.scan import parse_dft_detect_output           # auto_rx's own (scan.py:227); the caller's environment has it
    return parse_dft_detect_output


def interpret(per_channel, parse: Optional[Callable] = None, sdr_name: str = "batch"):
    """per channel -> (type | None, offset) exactly as detect_sonde_async treats one process: exit code 0 or >= 2: parse the output; 1: nothing"""
    parse = parse or _default_parser()
    out = []
    for text, code in per_channel:
        if code is None or code == 1:
            out.append((None, 0.0))
        else:
            out.append(parse(text, sdr_name))
    return out


async def detect_sondes_batch_async(frequencies, *, center_frequency: float, sample_rate: int, iq_cmd: str, rs_path: str = "./", dwell_time: int = 10,
                                    wideband_sondes: bool = False, parse: Optional[Callable] = None, sdr_name: str = "batch", run=None):
    """All `frequencies` in one process.  iq_cmd: the shell pipeline that writes the band as cs16 IQ to stdout, ending in `|` (what
    get_sdr_iq_cmd returns for one channel, here for the whole band).  run: test hook — an async callable(cmd) -> (stdout bytes, returncode)
    instead of the subprocess.  -> [(type | None, offset_hz), ...] in the order of `frequencies`."""
    freqs = [float(f) for f in frequencies]
    if not freqs:
        return []
    cmd = iq_cmd + " " + batch_command(freqs, center_frequency=center_frequency, sample_rate=sample_rate, rs_path=rs_path, dwell_time=dwell_time,
                                       if_bw=64 if wideband_sondes else 15)
    if run is not None:
        stdout, _rc = await run(cmd)
    else:
        proc = await asyncio.create_subprocess_shell(cmd, stdout=asyncio.subprocess.PIPE, stderr=asyncio.subprocess.PIPE)
        try:
            stdout, _ = await asyncio.wait_for(proc.communicate(), timeout=dwell_time * DETECTION_TIMEOUT_MULTIPLIER)
        except asyncio.TimeoutError:
            proc.kill()
            await proc.wait()
            return [(None, 0.0)] * len(freqs)
    return interpret(split_batch_output(stdout.decode("utf8", "replace"), len(freqs)), parse, sdr_name)


async def scan_peaks_batch(peak_frequencies, **kw):
    """what scan_peaks_concurrent returns: [(frequency + offset quantised to 1 kHz, type), ...] for the peaks where something was found"""
    freqs = [float(f) for f in peak_frequencies]
    res = await detect_sondes_batch_async(freqs, **kw)
    return [(round((f + off) / 1000.0) * 1000.0, typ) for f, (typ, off) in zip(freqs, res) if typ]


def run_batch_scan(peak_frequencies, **kw):
    """synchronous wrapper, the counterpart of run_async_scan (scan_async.py:381)"""
    try:
        asyncio.get_running_loop()
    except RuntimeError:
        return asyncio.run(scan_peaks_batch(peak_frequencies, **kw))
    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(max_workers=1) as ex:          # called from inside a running loop: a loop of its own on a worker thread
        return ex.submit(lambda: asyncio.run(scan_peaks_batch(peak_frequencies, **kw))).result()
