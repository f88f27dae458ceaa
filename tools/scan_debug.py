"""GPU-box debug: scanner windows vs the reference harness (oracle/_ref/libref_scan.so)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tools import synth
from radiosonde_auto_rx_amd.scan import Scanner, BBIQ, IFIQ, TYPES
from oracle import bind

np.set_printoptions(precision=4, linewidth=220, suppress=True)


def run(name, iq, sr, mode, fq, dc, bw, chunk):
    print("====", name)
    r = bind.ref_scan_windows(iq, sr, iq_mode=mode, fq=fq, dc=dc, bw_khz=bw, want_fm=20000)
    sc = Scanner(sr, fq=[fq], iq_mode=mode, dc=dc, bw_khz=bw, cont=True, max_chunk=chunk)
    print(sc.info, r["consts"])
    D = sc.info["decM"]
    n = len(iq) // 2
    wins = []
    for s0 in range(0, n - n % D, chunk):
        s1 = min(n - n % D, s0 + chunk)
        sc.process_host(iq[2 * s0:2 * s1])
        wins += sc.last_windows()
        if s0 == 0:
            for st in range(4):
                m = min(20000, (s1 - s0) // D)
                g = sc.read_fm(0, st, 0, m)
                d = g - r["fm"][st][:m]
                print("fm stream", st, "rms diff", float(np.sqrt(np.mean(d * d))), "max", float(np.abs(d).max()), "rms ref", float(np.sqrt(np.mean(r["fm"][st][:m] ** 2))))
    print("windows", len(wins), r["n"])
    worst = 0
    for w, g in enumerate(wins[:r["n"]]):
        ok = g["pos"] == r["pos"][w]
        act = r["mp"][w] != 0
        both = (g["mp"] > 0) & (r["mp"][w] > 0)
        dmv = np.abs(g["mv"] - r["mv"][w])[both]
        worst = max(worst, dmv.max() if len(dmv) else 0)
        bad = (~ok) or np.any((g["mp"] != r["mp"][w])[act]) or np.any((g["mpos"] != r["mpos"][w])[both]) or np.any((g["herrs"] != r["herrs"][w])[both]) or np.any(g["m10"] != r["m10"][w])
        if bad or w < 2:
            print(w, "pos", g["pos"], r["pos"][w], "BAD" if bad else "")
            print("  mp ", g["mp"]); print("  ref", r["mp"][w])
            print("  mv ", g["mv"]); print("  ref", r["mv"][w])
            print("  mpos", g["mpos"]); print("  ref ", r["mpos"][w])
            print("  dc ", g["dc"]); print("  ref", r["dc"][w])
            print("  herrs", g["herrs"], r["herrs"][w])
    print("worst |mv diff|", worst, "max dc diff", max(float(np.abs(g["dc"] - r["dc"][w]).max()) for w, g in enumerate(wins[:r["n"]])))
    dets = sc.fetch(verbose=True)
    for d in dets:
        print(repr(d["line"]))
    print("result", sc.result(0), sc.kernel_ms("front_end"), sc.kernel_ms("scan_if"), sc.kernel_ms("scan_corr"))
    args = (["--IQ", str(fq)] if mode == BBIQ else ["--iq"]) + (["--bw", str(bw)] if bw else []) + (["--dc"] if dc else []) + ["-v", "-c", "-", str(sr), "16"]
    print(bind.ref_run("dft_detect", args, iq))


iq = synth.rs41_capture(sr=2400000, seconds=1.5, fq=0.1, n_frames=1, t_first=0.3, noise_sigma=0.01, seed=5, f_offset_hz=-400)
run("rs41 2.4M --IQ 0.1 --dc", iq, 2400000, BBIQ, 0.1, True, 0.0, 1200000)
iq = synth.rs41_capture(sr=48000, seconds=3.0, fq=0.0, n_frames=2, t_first=0.4, noise_sigma=0.02, seed=7, f_offset_hz=900)
run("rs41 48k --iq --bw 15 --dc", iq, 48000, IFIQ, 0.0, True, 15.0, 48000)
iq = synth.dfm_capture(sr=2400000, seconds=1.0, fq=synth.snap_fq(-0.2, 2400000), noise_sigma=0.01, seed=3)
run("dfm 2.4M --IQ -0.2", iq, 2400000, BBIQ, synth.snap_fq(-0.2, 2400000), False, 0.0, 2400000)
