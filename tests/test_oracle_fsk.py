"""CPU: the numpy restatement of the 2-FSK modem (oracle/ora_fsk.py) against recordings of the compiled reference
(tests/golden/fsk_*.npz from oracle/ref_fsk_harness.c over the reference's own utils/fsk.c).

Pins the algorithm the GPU kernel implements: estimator bins, oscillator recurrence with its float drift, integrator
windows, timing estimate / nin control, interpolation.  Tolerances: nin sequence and tone estimates exact, hard decisions
exact; soft decisions 2e-6 of their RMS (numpy's cosf/sinf/atan2f may differ from libm by an ulp); timing 1e-6."""
import numpy as np
import pytest
from golden_cases import load_fsk, fsk_capture

CASES = ["fsk_rs41_48k_mask", "fsk_dfm_50k", "fsk_m10_48080", "fsk_rs41_48k_real"]


@pytest.mark.parametrize("name", CASES)
def test_modem_restatement_matches_reference(name):
    from oracle import ora_fsk
    g = load_fsk(name)
    x, case = fsk_capture(name)
    nfr = min(len(g["nin"]), 12)                                   # a dozen frames keep the pure-Python recurrence short
    md = ora_fsk.FskModem(case["cap"]["sr"], case["Rs"], P=case["P"], nsym=case["nsym"], lower=case["lower"], upper=case["upper"],
                          mask=case["mask"], fmt=case["fmt"])
    assert dict(Ts=md.Ts, N=md.N, Ndft=md.Ndft, Nmem=md.Nmem) == g["consts"]
    per = 1 if case["fmt"] == 1 else 2
    need = int(g["nin"][:nfr].sum())
    sd, recs = md.run(x[:per * need])
    assert len(recs) == nfr
    assert [r["nin"] for r in recs] == g["nin"][:nfr].tolist()
    assert [r["nin_next"] for r in recs] == g["nin_next"][:nfr].tolist()
    assert np.array_equal(np.array([r["f_est"] for r in recs], np.float32), g["f_est"][:nfr])
    ref = g["sd"][:nfr]
    rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    assert np.sqrt(np.mean((sd - ref).astype(np.float64) ** 2)) < 2e-6 * rms
    assert np.array_equal(sd < 0, ref < 0)
    assert np.abs(np.array([r["norm_rx_timing"] for r in recs]) - g["norm_rx_timing"][:nfr]).max() < 1e-6
    assert np.abs(np.array([r["EbNodB"] for r in recs]) - g["EbNodB"][:nfr]).max() < 1e-2
