"""debugging aid: where product and reference differ in the low-SNR sweep of tests/test_gpu_lowsnr.py (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_lowsnr as T
for kind in ("rs41", "dfm", "m10"):
    x, baud = T._clean(kind)
    argv, _ = T.CASES[kind]
    for k, ebno in enumerate(T.EBNO):
        data = T.add_noise(x, baud, ebno, 1000 + k).tobytes()
        a = T._run([os.path.join(T.BIN, argv[0])] + argv[1:], data).decode().splitlines()
        b = T._run([os.path.join(T.REF, argv[0])] + argv[1:], data).decode().splitlines()
        nd = sum(1 for la, lb in zip(a, b) if la != lb)
        print(kind, ebno, len(a), len(b), "differing lines", nd)
        for i, (la, lb) in enumerate(zip(a, b)):
            if la != lb:
                ha, hb = la.split(" ")[0], lb.split(" ")[0]
                try:
                    nb = bin(int(ha, 16) ^ int(hb, 16)).count("1") if len(ha) == len(hb) else -1
                except ValueError:
                    nb = -2
                print("   line", i, "bits", nb, "|", la[-30:], "|", lb[-30:])
