"""debug: header hits of the reference (libref_demod.so harness) and of the engine for the MRZ capture of tests/test_gpu_mrz.py with --dc"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tools import synth
from oracle import bind
from radiosonde_auto_rx_amd.engine import Engine

sr = 48_000
x = synth.mrz_capture(sr=sr, seconds=18.5, noise_sigma=0.05, seed=71)[:2 * sr * 7]
hdr = b"100110011001100110011001100110011001" b"10101010"
for afc in (False, True):
    r = bind.ref_softframes(x, sr, iq_mode=5, fq=0.0, lp_iq=True, afc=afc, baud=2399.0, bt=1.0, h=2.0, lpiq_bw=9000, lpfm_bw=6000, hdr=hdr, symlen=2, symhd=2,
                            thres=0.76, hdmax=2, bitofs=2, l=2.0, nbits=386, max_hits=64)
    gen = dict(header=hdr.decode(), baud=2399.0, bt=1.0, h=2.0, symlen=2, symhd=2, hdmax=2, bitofs=2, nbits=386, l_win=2.0, lpiq_bw=9000, lpfm_bw=6000)
    eng = Engine([0.0], sr, sonde="generic", generic=gen, thres=0.76, max_chunk=sr, max_frames=64, lp_iq=True, lp_fm=afc, opt_dc=afc, keep_soft=True)
    hits = []
    n = len(x) // 2
    for s0 in range(0, n, sr // 10):
        s1 = min(n, s0 + sr // 10)
        eng.process_host(np.ascontiguousarray(x[None, 2 * s0:2 * s1]))
        hits += eng.fetch_hits(finish=s1 >= n)
    eng.close()
    print("afc", afc, "ref", r["n"], "eng", len(hits))
    print(" ref:", [(int(p), round(float(m), 4)) for p, m in zip(r["mv_pos"], r["mv"])])
    print(" eng:", [(int(h["mv_pos"]), round(float(h["mv"]), 4)) for h in hits])
