"""GPU-box micro-benchmark of ONE modem configuration (not the headline bench): python tools/fsk_alone.py rs41|dfm|m10 [channels] [launches]"""
import sys, time
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth
from radiosonde_auto_rx_amd.fsk import FskModem

kind = sys.argv[1] if len(sys.argv) > 1 else "rs41"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 342
L = int(sys.argv[3]) if len(sys.argv) > 3 else 6
Fs, Rs, P, nsym, mask, lim = {"rs41": (48000, 4800, 5, 300, 5000, 5000), "dfm": (50000, 2500, 10, 50, 0, 5000), "m10": (48080, 9616, 5, 50, 0, 10000)}[kind]
caps = []
for s in range(4):
    if kind == "rs41":
        caps.append(synth.rs41_capture(sr=Fs, seconds=1.0, fq=0.0, n_frames=1, t_first=0.05, noise_sigma=0.02, seed=s, f_offset_hz=150.0 * s))
    elif kind == "dfm":
        caps.append(synth.dfm_capture(sr=Fs, seconds=1.0, fq=0.0, noise_sigma=0.02, seed=10 + s))
    else:
        caps.append(synth.m10_capture(sr=Fs, seconds=1.0, fq=0.0, noise_sigma=0.02, seed=20 + s, baud=float(Rs), dev_hz=Rs / 2.0))
n = min(len(c) for c in caps)
X = torch.from_numpy(np.stack([caps[c % 4][:n] for c in range(C)])).cuda()
md = FskModem(Fs, Rs, n_channels=C, P=P, nsym=nsym, mask=mask, lower=-lim, upper=lim, max_chunk=Fs)
for it in range(L):
    torch.cuda.synchronize(); t0 = time.time()
    md.process_device(X.data_ptr(), n // 2, n // 2)
    t1 = time.time()
    print(kind, C, "step", it, "wall ms %.3f" % ((t1 - t0) * 1e3), "kernel ms", md.kernel_ms())
