"""GPU-box micro-benchmark of the batched 2-FSK modem and the scanner (not the headline bench)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from tools import synth
from radiosonde_auto_rx_amd.fsk import FskModem
from radiosonde_auto_rx_amd.scan import Scanner, IFIQ

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Fs = 48000
caps = [synth.rs41_capture(sr=Fs, seconds=1.0, fq=0.0, n_frames=1, t_first=0.05, noise_sigma=0.02, seed=s, f_offset_hz=200.0 * s) for s in range(4)]
X = np.stack([caps[c % 4] for c in range(C)])
d = torch.from_numpy(X).cuda()
md = FskModem(Fs, 4800, n_channels=C, P=5, nsym=300, mask=5000, lower=-20000, upper=20000, max_chunk=Fs)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    md.process_device(d.data_ptr(), Fs, Fs)
    t1 = time.time()
    print("fsk step", it, "wall ms", (t1 - t0) * 1e3, "kernel ms", md.kernel_ms(), "Msps", C * Fs / (t1 - t0) / 1e6)
sc = Scanner(Fs, n_channels=C, iq_mode=IFIQ, dc=True, bw_khz=15.0, cont=True, max_chunk=Fs)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    sc.process_device(d.data_ptr(), Fs, Fs)
    t1 = time.time()
    print("scan step", it, "wall ms", (t1 - t0) * 1e3, sc.kernel_ms("front_end"), sc.kernel_ms("scan_if"), sc.kernel_ms("scan_corr"), "Msps", C * Fs / (t1 - t0) / 1e6)

# BASELINE config 3: 256 channels on a 10 kHz raster mixed out of ONE 10 Msps stream, scanned for one second
from radiosonde_auto_rx_amd.scan import BBIQ
sr = 10_000_000
sig = [dict(kind="rs41", fq=synth.snap_fq(0.0128 * (k % 7) + 0.01, sr), t_first=0.04 + 0.01 * k, amp=0.05) for k in range(4)]
wb = torch.from_numpy(synth.wideband_capture(sr, 0.2, sig, noise_sigma=0.01, seed=3)).cuda()
fqs = [synth.snap_fq(-0.128 + 0.001 * k, sr) for k in range(256)]
sw = Scanner(sr, fq=fqs, iq_mode=BBIQ, dc=True, cont=True, max_chunk=2_000_000)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    sw.process_device(wb.data_ptr(), 0, 2_000_000)
    t1 = time.time()
    print("wideband step", it, "wall ms", (t1 - t0) * 1e3, sw.kernel_ms("front_end"), sw.kernel_ms("scan_if"), sw.kernel_ms("scan_corr"),
          "realtime factor", 0.2 / (t1 - t0))
