"""What the receivers do with an LMS6 that turns out to be an LMS-X, on the CPU: header hits and soft bits of the CPU restatement of the demodulator
(oracle/, test infrastructure) under the LMS6 description, then under the LMS-X description with another sample clock and one block lost — the way a
receiver's engines deliver them around the change — through one family.FamilyDecoder object.  The first block makes the decoder report the other type;
every block behind the change decodes as LMS-X."""
import numpy as np

from radiosonde_auto_rx_amd import family as F
from tools import synth

HDR = b"0101011000001000" b"0001110010010111" b"0001101010100111" b"0011110100111110"


def test_decoder_object_across_the_change_of_engine(oracle):
    x = synth.lms6_capture(sr=48_000, seconds=9.0, noise_sigma=0.05, seed=31, baud=4797.8, lmsx=True)
    kw = dict(iq_mode=5, fq=0.0, lp_iq=True, bt=1.2, h=0.9, lpiq_bw=16000, lpfm_bw=6000, hdr=HDR, symlen=1, symhd=1, thres=0.65, hdmax=10, bitofs=0, l=-1.0, max_hits=64)
    h6 = oracle.ora_softframes(x, 48_000, baud=4800.0, nbits=F.FAMILY["LMS6"]["generic"]["nbits"], **kw)
    hx = oracle.ora_softframes(x, 48_000, baud=4797.8, nbits=F.FAMILY["LMSX"]["generic"]["nbits"], **kw)
    assert h6["n"] >= 8 and hx["n"] >= 8 and np.all(h6["mv"] > 0)

    def feed(d, H, i, shift=0):
        nb = int(H["nbits"][i])
        return d.hit(dict(soft=H["soft"][i][:nb], mv=float(H["mv"][i]), mv_pos=int(H["mv_pos"][i]) - shift), 48_000)

    d = F.FamilyDecoder("LMS6", version="x")
    assert d.lms_type() == ("LMS6", False)
    feed(d, h6, 0)
    assert d.lms_type() == ("LMSX", True)                       # -> the receiver moves the sonde (wideband.py _follow_lms, sonde_wideband.c move_sondes)
    d.moved = True
    base = int(hx["mv_pos"][2]) - 3000                          # the new channel's clock starts behind the second block: that one is lost
    ids = []
    for i in range(2, hx["n"]):
        js = d.json_objects(feed(d, hx, i, shift=base))
        assert d.lms_type() == ("LMSX", False)
        ids += [(j["id"], j["frame"]) for j in js]
    assert ids == [("LMSX-8123456", 100 + i) for i in range(2, hx["n"])]
    d.close()
