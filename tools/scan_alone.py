import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth
from radiosonde_auto_rx_amd.scan import Scanner
SR = 2_400_000
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fqs = [synth.snap_fq(-0.4 + 0.8 * k / NCH, SR) for k in range(NCH)]
caps = [synth.rs41_capture(sr=SR, seconds=1.0, fq=f, n_frames=1, t_first=0.05, noise_sigma=0.02, seed=k) for k, f in enumerate(fqs[:4])]
X = torch.from_numpy(np.stack([caps[k % 4] for k in range(NCH)])).cuda()
sc = Scanner(SR, fq=fqs, dc=True, cont=True, max_chunk=SR)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sc.process_device(X.data_ptr(), SR, SR)
    d = sc.fetch()
    t1 = time.perf_counter()
    km = {k: sc.kernel_ms(k) for k in ("front_end", "scan_if", "scan_pre", "scan_corr")}
    print("call", it, "channels", NCH, "wall ms %.3f" % ((t1 - t0) * 1e3), km, "front end Gsamples/s %.0f" % (NCH * SR / (km["front_end"][0] * 1e-3) / 1e9 if km["front_end"][0] else 0))
