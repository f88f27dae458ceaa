import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from tools import synth
from radiosonde_auto_rx_amd.scan import Scanner
SR = 2_400_000
fqs = [synth.snap_fq(0.05 + 0.01 * k, SR) for k in range(32)]
caps = [synth.rs41_capture(sr=SR, seconds=1.0, fq=f, n_frames=1, t_first=0.05, noise_sigma=0.02, seed=k) for k, f in enumerate(fqs[:4])]
X = torch.from_numpy(np.stack([caps[k % 4] for k in range(32)])).cuda()
sc = Scanner(SR, fq=fqs, dc=True, cont=True, max_chunk=SR)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sc.process_device(X.data_ptr(), SR, SR)
    d = sc.fetch()
    t1 = time.perf_counter()
    print("call", it, "wall ms %.3f" % ((t1 - t0) * 1e3), {k: sc.kernel_ms(k) for k in ("front_end", "scan_if", "scan_pre", "scan_corr")})
