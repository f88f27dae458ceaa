"""GPU-box micro-benchmark of the 2-FSK modem kernel alone (not the bench line): python tools/fsk_multi.py rs41,m10,dfm [channels per configuration]
One engine per listed configuration (auto_rx's argument sets), all submitted, then all waited for, ten times: step time and each launch's own event time.
What tools/profile_round.sh runs under rocprofv3 for the SQ counters of k_fsk_wave."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth
from radiosonde_auto_rx_amd.fsk import FskModem, SoftinDev
kinds = sys.argv[1].split(",")
C = int(sys.argv[2]) if len(sys.argv) > 2 else 342
G = {"rs41": (48000, 4800, 5, 300, 5000, 5000), "dfm": (50000, 2500, 10, 50, 0, 5000), "m10": (48080, 9616, 5, 50, 0, 10000)}
E = []
for i, kind in enumerate(kinds):
    Fs, Rs, P, nsym, mask, lim = G[kind]
    caps = []
    for s in range(4):
        if kind == "rs41": caps.append(synth.rs41_capture(sr=Fs, seconds=1.0, fq=0.0, n_frames=1, t_first=0.05, noise_sigma=0.02, seed=s, f_offset_hz=150.0 * s, bit_errors=4 * s))
        elif kind == "dfm": caps.append(synth.dfm_capture(sr=Fs, seconds=1.0, fq=0.0, noise_sigma=0.02, seed=10 + s))
        else: caps.append(synth.m10_capture(sr=Fs, seconds=1.0, fq=0.0, noise_sigma=0.02, seed=20 + s, baud=float(Rs), dev_hz=Rs / 2.0, frame_fn=lambda k, s=s: synth.m10_frame(k, rng=np.random.default_rng(900 + 10 * s + k))))
    L = min(len(c) for c in caps)
    X = torch.from_numpy(np.stack([caps[c % 4][:L] for c in range(C)])).cuda()
    md = FskModem(Fs, Rs, n_channels=C, P=P, nsym=nsym, mask=mask, lower=-lim, upper=lim, max_chunk=Fs)
    E.append((kind, X, md))
def step():
    for kind, X, md in E: md.submit_device(X.data_ptr(), X.shape[1] // 2, X.shape[1] // 2)
    for kind, X, md in E: md.wait()
for _ in range(3): step()
k0 = [md.kernel_ms() for _, _, md in E]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
k1 = [md.kernel_ms() for _, _, md in E]
per = [(b[0] * b[1] - a[0] * a[1]) / (b[1] - a[1]) for a, b in zip(k0, k1)]
print(sys.argv[1], C, "step %.3f ms; kernels in submission order:" % dt, " ".join("%.3f" % p for p in per))
