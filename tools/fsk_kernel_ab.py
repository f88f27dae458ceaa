"""GPU-box A/B of the modem kernel alone: one modem family at a time (no other kernel on the device), kernel ms per launch of 1 s per channel."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tools import synth
from radiosonde_auto_rx_amd.fsk import FskModem

C = int(sys.argv[1]) if len(sys.argv) > 1 else 342
out = []
for kind, Fs, Rs in (("rs41", 48000, 4800), ("dfm", 50000, 2500), ("m10", 48080, 9616)):
    caps = []
    for s in range(4):
        if kind == "rs41":
            caps.append(synth.rs41_capture(sr=Fs, seconds=1.0, fq=0.0, n_frames=1, t_first=0.05, noise_sigma=0.02, seed=s, f_offset_hz=150.0 * s))
        elif kind == "dfm":
            caps.append(synth.dfm_capture(sr=Fs, seconds=1.0, fq=0.0, noise_sigma=0.02, seed=10 + s))
        else:
            caps.append(synth.m10_capture(sr=Fs, seconds=1.0, fq=0.0, noise_sigma=0.02, seed=20 + s, baud=float(Rs)))
    L = min(len(c) for c in caps)
    X = np.stack([caps[c % 4][:L] for c in range(C)])
    d = torch.from_numpy(X).cuda()
    n = L // 2
    md = FskModem(Fs, Rs, n_channels=C, P=5, nsym=300 if kind == "rs41" else 150, mask=5000 if kind == "rs41" else 0, max_chunk=n)
    ms = []
    for it in range(5):
        k0 = md.kernel_ms()
        md.process_device(d.data_ptr(), n, n)
        k1 = md.kernel_ms()
        ms.append(k1[0] * k1[1] - k0[0] * k0[1] if isinstance(k1, tuple) else k1)
    out.append((kind, [round(float(x), 3) for x in ms[1:]]))
    md.close()
print(out)
