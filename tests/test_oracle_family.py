"""The CPU restatement (oracle/ora_dsp.c) pinned against the reference's own demod_mod.o on the generic decoder family, incl. the scenario that exposed
the engine's header-check bug of round 1: MRZ on IQ with --dc — the FM-stream fallback finds a header before the correlation window and the
reference's header check reads ring slots the newest samples have already overwritten (demod_mod.c:268,850).  Header hits (positions exact, scores)
and soft bits of both sides; no GPU."""
import numpy as np
import pytest

from tools import synth

def _rs92_capture():
    from tools import synth_rs92 as R
    return R.rs92_capture(R.flight(4, R.constellation()), sr=48_000, noise_sigma=0.08, seed=81)


FAM = {
    "mrz": dict(cap=lambda: synth.mrz_capture(sr=48_000, seconds=18.5, noise_sigma=0.05, seed=71)[:2 * 48_000 * 7],
                kw=dict(baud=2399.0, bt=1.0, h=2.0, lpiq_bw=9000, lpfm_bw=6000, hdr=b"100110011001100110011001100110011001" b"10101010", symlen=2, symhd=2,
                        thres=0.76, hdmax=2, bitofs=2, l=2.0, nbits=386)),
    "lms6": dict(cap=lambda: synth.lms6_capture(sr=48_000, seconds=4.0, noise_sigma=0.08, seed=22),
                 kw=dict(baud=4800.0, bt=1.2, h=0.9, lpiq_bw=16000, lpfm_bw=6000, hdr=b"0101011000001000" b"0001110010010111" b"0001101010100111" b"0011110100111110",
                         symlen=1, symhd=1, thres=0.65, hdmax=10, bitofs=0, l=-1.0, nbits=4096)),
    "imet54": dict(cap=lambda: synth.imet54_capture(sr=48_000, seconds=4.5, noise_sigma=0.08, seed=62),
                   kw=dict(baud=4798.0, bt=1.0, h=0.8, lpiq_bw=7400, lpfm_bw=6000, hdr=b"0000000001" b"0101010101" b"0001001001" b"0001001001", symlen=1, symhd=1,
                           thres=0.7, hdmax=4, bitofs=1, l=2.0, nbits=2200)),
    "meisei": dict(cap=lambda: synth.meisei_capture(sr=48_000, seconds=5.0, noise_sigma=0.1, seed=41),
                   kw=dict(baud=2400.0, bt=1.2, h=2.4, lpiq_bw=16000, lpfm_bw=4000, hdr=b"101010101011010100101011001101001100101011001101", symlen=1, symhd=1,
                           thres=0.7, hdmax=1, bitofs=0, l=-1.0, nbits=1152)),
    "rs92": dict(cap=lambda: _rs92_capture(),
                 kw=dict(baud=4800.0, bt=0.5, h=0.8, lpiq_bw=8000, lpfm_bw=6000, hdr=b"10100110011001101001" b"1010011001100110100110101010100110101001", symlen=2, symhd=2,
                         thres=0.7, hdmax=3, bitofs=2, l=4.0, nbits=2340)),
    "mts01": dict(cap=lambda: synth.mts01_capture(sr=48_000, seconds=5.5, noise_sigma=0.1, seed=51),
                  kw=dict(baud=1200.0, bt=1.5, h=0.9, lpiq_bw=4000, lpfm_bw=4000, hdr=b"10101010" b"10101010" b"10110100" b"00101011", symlen=1, symhd=1,
                          thres=0.76, hdmax=2, bitofs=0, l=2.0, nbits=1048)),
}


@pytest.mark.parametrize("afc", [False, True], ids=["plain", "dc"])
@pytest.mark.parametrize("name", sorted(FAM))
def test_restatement_matches_reference_on_family_hits(oracle, name, afc):
    if not oracle.have_ref():
        pytest.skip("compiled reference not present")
    x = FAM[name]["cap"]()
    kw = dict(FAM[name]["kw"], iq_mode=5, fq=0.0, lp_iq=True, afc=afc, max_hits=64)
    r = oracle.ref_softframes(x, 48_000, libname="libref_demod_O2.so", **kw)
    o = oracle.ora_softframes(x, 48_000, **kw)
    assert r["n"] == o["n"] and r["n"] >= 3
    assert list(r["mv_pos"]) == list(o["mv_pos"]) and list(r["nbits"]) == list(o["nbits"])
    tol = 2e-5 if afc else 1e-6              # after an AFC step the two sides rotate the IF samples with phasors built in a different order: a few 1e-6
    assert np.max(np.abs(r["mv"] - o["mv"])) < tol
    for h in range(r["n"]):
        nb = int(r["nbits"][h])
        scale = float(np.sqrt(np.mean(r["soft"][h][:nb] ** 2))) + 1e-12
        assert np.max(np.abs(r["soft"][h][:nb] - o["soft"][h][:nb])) < (5e-4 if afc else 1e-5) * scale + 1e-7, (name, afc, h)      # --dc: single soft bits up to a few 1e-4 apart after AFC steps
    if name == "mrz" and afc:
        # the candidate the round-1 engine accepted: its header ends at 144992, 39 samples before the window; both CPU sides reject it
        assert 144992 not in list(r["mv_pos"]) and 129565 in list(r["mv_pos"])
