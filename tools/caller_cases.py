"""The pipelines auto_rx itself builds around the binaries of this repo's boundary B1 (test infrastructure for tests/test_caller_contract.py):
seeded captures at the sample rates auto_rx asks its SDR for, and the argument lists of auto_rx/autorx/scan.py and decode.py, verbatim.

  detect  scan.py:541-547   dft_detect -t <dwell> --iq --bw 15 --dc - 48000 16            (400 MHz band, IQ mode; exit code + one text line)
  fsk     decode.py:901     fsk_demod --cs16 -b -5000 -u 5000 -s --mask 5000 --nsym=300 -p 5 --stats=5 2 48000 4800 - -   | rs41mod --ptu2 --json --jsnsubfrm1 --softin -i
          decode.py:1067    fsk_demod --cs16 -b -5000 -u 5000 -s -i --stats=5 2 50000 2500 - -                          | dfm09mod -vv --ecc --json --dist --auto --softin
          decode.py:1120    fsk_demod --cs16 -b -10000 -u 10000 -s -p 5 --stats=5 2 48080 9616 - -                      | m10mod --json --ptu -vvv --softin -i
  audio   decode.py:396-417 (rtl_fm | sox ->) rs41mod --ptu2 --json --jsnsubfrm1           on 48 kHz FM audio (WAV)

Only dft_detect, fsk_demod and the FM-audio decoder need the GPU; the --softin decoders behind fsk_demod are host code on both sides."""
import numpy as np

from tools import synth

ECEF = (418833319, 85974133, 473346430)          # a position the decoders accept (48.1 N 11.6 E 12300 m)


def _iq16(x, noise_sigma, seed):
    rng = np.random.default_rng(seed)
    x = x + noise_sigma * (rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x)))
    out = np.empty(2 * len(x), dtype=np.int16)
    out[0::2] = np.clip(np.round(x.real * 32767 * 0.9), -32768, 32767).astype(np.int16)
    out[1::2] = np.clip(np.round(x.imag * 32767 * 0.9), -32768, 32767).astype(np.int16)
    return out


def _dfm_telemetry(sr, seed):
    """a DFM09 that sends real telemetry (configuration cycle with its serial number, GPS packets 0..8: tools/make_golden.dfm_field_symbols),
    as GFSK: the random payload of synth.dfm_capture never gets past dfm09mod --dist / --json"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import make_golden
    sym = (make_golden.dfm_field_symbols(make_golden.DFM_FIELD_SCENARIOS["dfmf_09_40"]) > 0).astype(np.uint8)
    burst = synth.gfsk_baseband(sym, sr, 2500.0, 2400.0, bt=0.5)
    x = np.zeros(len(burst) + sr // 5, np.complex128)
    x[sr // 10:sr // 10 + len(burst)] = 0.5 * burst
    return _iq16(x, 0.02, seed)


def capture(name):
    if name == "rs41":
        return synth.rs41_capture(sr=48000, seconds=5.3, fq=0.0, seed=41, noise_sigma=0.02, f_offset_hz=350.0, frame_kw=dict(ecef_cm=ECEF))
    if name == "rs41_weak":
        return synth.rs41_capture(sr=48000, seconds=4.3, fq=0.0, seed=42, noise_sigma=0.12, f_offset_hz=-800.0, bit_errors=5, frame_kw=dict(ecef_cm=ECEF))
    if name == "dfm":
        return _dfm_telemetry(50000, 43)
    if name == "dfm48":
        return synth.dfm_capture(sr=48000, seconds=4.0, fq=0.0, noise_sigma=0.02, seed=44)
    if name == "m10":
        return synth.m10_capture(sr=48080, seconds=5.0, fq=0.0, noise_sigma=0.02, seed=45, baud=9616.0, dev_hz=4808.0,     # tone spacing = Rs: what fsk_demod's peak estimator assumes (utils/fsk.c:478-520)
                                 frame_fn=lambda k: synth.m10_frame(k, rng=np.random.default_rng(900 + k)))
    if name == "m10_48":
        return synth.m10_capture(sr=48000, seconds=4.0, fq=0.0, noise_sigma=0.02, seed=46)
    if name == "band":
        sig = [dict(kind=k, fq=(f - BATCH_CENTER) / BATCH_SR, t_first=0.15 + 0.1 * i, amp=0.06) for i, (k, f) in enumerate(BATCH_TRUE)]
        return synth.wideband_capture(BATCH_SR, 3.6, sig, noise_sigma=0.01, seed=48)
    if name == "noise":
        rng = np.random.default_rng(47)
        return np.clip(np.round(rng.standard_normal(2 * 48000 * 3) * 900), -32768, 32767).astype(np.int16)
    raise KeyError(name)


DETECT = {  # case -> (capture, argv behind the binary's name)
    "detect_rs41": ("rs41", ["-t", "10", "--iq", "--bw", "15", "--dc", "-", "48000", "16"]),
    "detect_dfm": ("dfm48", ["-t", "10", "--iq", "--bw", "15", "--dc", "-", "48000", "16"]),
    "detect_m10": ("m10_48", ["-t", "10", "--iq", "--bw", "15", "--dc", "-", "48000", "16"]),
    "detect_noise": ("noise", ["-t", "2", "--iq", "--bw", "15", "--dc", "-", "48000", "16"]),
}
# one band, several sondes: the batch form `dft_detect --IQ fq1,fq2,...` (radiosonde_auto_rx_amd/scan_batch.py) against auto_rx's per-peak scan
BATCH_SR, BATCH_CENTER, BATCH_DWELL = 960_000, 403_000_000, 4
BATCH_PEAKS = [403_120_000, 402_800_000, 403_310_000, 402_950_000]        # what the spectrum peak search hands to the detector (the last one is empty)
BATCH_TRUE = [("rs41", 403_120_300), ("dfm", 402_799_600), ("m10", 403_310_450)]        # where the sondes really are (a few hundred Hz off their peaks)
BATCH = {"batch_band": ("band", ["-t", str(BATCH_DWELL), "--IQ", ",".join("%.9f" % ((f - BATCH_CENTER) / BATCH_SR) for f in BATCH_PEAKS), "--bw", "15", "--dc", "-", str(BATCH_SR), "16"])}
FSK = {     # case -> (capture, fsk_demod argv, decoder binary, decoder argv, auto_rx sonde type)
    "fsk_rs41": ("rs41", ["--cs16", "-b", "-5000", "-u", "5000", "-s", "--mask", "5000", "--nsym=300", "-p", "5", "--stats=5", "2", "48000", "4800", "-", "-"],
                 "rs41mod", ["--ptu2", "--json", "--jsnsubfrm1", "--softin", "-i"], "RS41"),
    "fsk_rs41_weak": ("rs41_weak", ["--cs16", "-b", "-5000", "-u", "5000", "-s", "--mask", "5000", "--nsym=300", "-p", "5", "--stats=5", "2", "48000", "4800", "-", "-"],
                      "rs41mod", ["--ptu2", "--json", "--jsnsubfrm1", "--softin", "-i"], "RS41"),
    "fsk_dfm": ("dfm", ["--cs16", "-b", "-5000", "-u", "5000", "-s", "-i", "--stats=5", "2", "50000", "2500", "-", "-"],
                "dfm09mod", ["-vv", "--ecc", "--json", "--dist", "--auto", "--softin"], "DFM"),
    "fsk_m10": ("m10", ["--cs16", "-b", "-10000", "-u", "10000", "-s", "-p", "5", "--stats=5", "2", "48080", "9616", "-", "-"],
                "m10mod", ["--json", "--ptu", "-vvv", "--softin", "-i"], "M10"),
}
