"""Step time of the mixed-type engine (BASELINE configs[4] at configs[3]'s type mix, 512 channels x 2.4 Msps: the bench's mixed_2400k step without its extras) and the host
time inside it (enqueue / fetch), for frame fetches 1, 2, ... steps behind (LAGS=1,2).  Environment switches of the engine apply (SONDE_MIXED_SPLIT=1: a stream per
type group; SONDE_ECC_INLINE=1; SONDE_B_PRIO=0; SONDE_SMALL_TAIL=1).  -> profiles/r6a_mixed_lag_ab.txt (tools/ab_mixed.sh)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import bench_configs
from radiosonde_auto_rx_amd.engine import MixedEngine
SR=2400000; C=512
dev=torch.device("cuda:0"); torch.cuda.init()
kinds=[bench_configs.MIX_PATTERN[c%10] for c in range(C)]
bank=bench_configs.mixed_bank(SR)
seen={k:0 for k in bank}; chb=[]
for kd in kinds: chb.append(seen[kd]%16); seen[kd]+=1
fq=[bank[kd][0][b] for kd,b in zip(kinds,chb)]
order={"rs41":0,"dfm":1,"m10":2}
allc=torch.from_numpy(np.stack([cap for k in ("rs41","dfm","m10") for cap in bank[k][1]])).to(dev)
X=allc.index_select(0, torch.tensor([order[kd]*16+b for kd,b in zip(kinds,chb)],device=dev)).contiguous()
torch.cuda.synchronize()
for lag in [int(v) for v in os.environ.get("LAGS","1,2").split(",")]:
    eng=MixedEngine(fq,kinds,SR,max_chunk=SR,max_frames=8*C)
    while eng.samples_to_dc_boundary()<SR: eng.process_device(X.data_ptr(),SR,eng.samples_to_dc_boundary())
    def fetch(l):
        eng.fetch_frames_np(lag=l); eng.fetch_dfm_raw(lag=l); eng.fetch_m10_raw(lag=l)
    for _ in range(5):
        eng.process_device(X.data_ptr(),SR,SR); fetch(lag)
    fetch(0); eng.sync()
    tp=tf=0; n=300
    t0=time.perf_counter()
    for _ in range(n):
        a=time.perf_counter(); eng.process_device(X.data_ptr(),SR,SR); b=time.perf_counter(); fetch(lag); c=time.perf_counter()
        tp+=b-a; tf+=c-b
    fetch(0); eng.sync()
    dt=(time.perf_counter()-t0)/n
    print("lag",lag,"step %.3f ms  host: process %.3f fetch %.3f"%(dt*1e3,tp/n*1e3,tf/n*1e3), flush=True)
    eng.close()
