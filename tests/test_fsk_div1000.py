"""The modem kernel converts int16 samples with x / 1000.f (fsk_demod.c:283-311).  sonde_fsk.hip evaluates the quotient as
q = x r, q + fma(-q, 1000, x) r with r = fl(1/1000): this checks, for every 16-bit input, that the result is the correctly
rounded float division (the fused multiply-adds are emulated in float64, which holds these products and sums exactly)."""
import numpy as np


def test_div1000_is_the_correctly_rounded_quotient_for_every_int16():
    x = np.arange(-32768, 32768, dtype=np.int32).astype(np.float32)
    want = (x / np.float32(1000.0)).astype(np.float32)
    r = np.float32(1.0) / np.float32(1000.0)
    q = (x * r).astype(np.float32)
    e = (-q.astype(np.float64) * 1000.0 + x.astype(np.float64)).astype(np.float32)       # fma(-q, 1000, x): exact here
    got = (e.astype(np.float64) * np.float64(r) + q.astype(np.float64)).astype(np.float32)
    assert (q != want).any()                      # the bare reciprocal product is NOT enough
    assert np.array_equal(got, want)
