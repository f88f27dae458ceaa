"""Start-up latency of the CLI front ends on the GPU box (start emitting within the decoder timeout): time from process start to
the first stdout line and to exit, for a 3.3 s 48 kHz IQ capture piped in at once.  Run from the repo root: python tools/cli_latency.py"""
import sys, subprocess, time, os
sys.path.insert(0, os.getcwd())
from tools import synth
x = synth.rs41_capture(sr=48000, seconds=3.3, fq=0.0, n_frames=3, t_first=0.15, seed=5).tobytes()
for name, cmd in (("rs41mod 48k IQ", ["host/bin/rs41mod", "--ptu2", "--json", "--IQ", "0.0", "--lpIQ", "-", "48000", "16"]),
                  ("dft_detect 48k IQ", ["host/bin/dft_detect", "-t", "2", "--iq", "--bw", "15", "--dc", "-", "48000", "16"])):
    for rep in range(3):
        t0 = time.perf_counter()
        p = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        try:
            p.stdin.write(x); p.stdin.close()
        except BrokenPipeError:
            pass                                  # dft_detect exits as soon as it has decided
        first = p.stdout.readline(); t1 = time.perf_counter()
        p.stdout.read(); p.wait(); t2 = time.perf_counter()
        print(name, "first line after %.3f s, exit after %.3f s" % (t1 - t0, t2 - t0), first[:40])
