"""radiosonde_auto_rx_amd/family.py on the CPU: the ctypes bindings of the bit-rate tiers take one fetch_hits() dict each (soft bits in the
engine's conventions) and return the reference's text; descriptors agree with the C front ends."""
import os
import re

import numpy as np

from radiosonde_auto_rx_amd import family as F
from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _soft(bits):
    return (2.0 * np.asarray(bits, np.float64) - 1.0).astype(np.float32)


def test_decoders_on_clean_hits():
    d = F.FamilyDecoder("LMS6", version="x", freq_khz=403000)
    s = _soft(synth.lms6_onair_bits(3))
    t = d.hit(dict(soft=s[80:80 + 4096], mv=0.9, mv_pos=1000))
    js = d.json_objects(t)
    assert "[OK]" in t and len(js) == 1 and js[0]["id"] == "LMS6-8123456" and js[0]["freq"] == 403000 and js[0]["version"] == "x"
    # an inverted signal: the engine stores the bits in the polarity in effect (= as sent) and reports a negative score; the decoder wants them raw
    assert "[OK]" in d.hit(dict(soft=s[4160 + 80:4160 + 80 + 4096], mv=-0.9, mv_pos=70000))
    d.close()
    d = F.FamilyDecoder("IMET5")
    t = d.hit(dict(soft=_soft(synth.imet54_frame_bits(synth.imet54_frame(1))), mv=0.9, mv_pos=0))
    assert "[OK]" in t and d.json_objects(t)[0]["id"] == "IMET5-54012345"
    d = F.FamilyDecoder("MTS01")
    t = d.hit(dict(soft=-_soft(synth.mts01_frame_bits(1)), mv=-0.9, mv_pos=0))
    assert "[OK]" in t and d.json_objects(t)[0]["id"] == "MTS01-A2031234"
    d = F.FamilyDecoder("MEISEI")
    sy = synth.meisei_symbols(2)
    t = d.hit(dict(soft=_soft(sy[48:1200]), mv=0.9, mv_pos=0)) + d.hit(dict(soft=-_soft(sy[1200 + 48:2400]), mv=-0.9, mv_pos=0))
    assert t.count("(ok)[OK]") == 2 and d.json_objects(t)[0]["subtype"] == "IMS100"
    d = F.FamilyDecoder("MRZ", uniq=0)
    bb = np.unpackbits(np.frombuffer(bytes([0xAA]) + synth.mrz_frame(3), np.uint8))[22:]
    assert "[OK]" in d.hit(dict(soft=-_soft(bb), mv=0.9, mv_pos=0))                                     # engine convention: second half symbol minus first


def test_rs92_decoder_solves_the_position(tmp_path):
    from tools import synth_rs92 as R
    eph = R.constellation()
    E = tmp_path / "brdc.nav"
    E.write_bytes(R.rinex_nav(eph))
    d = F.FamilyDecoder("RS92", version="x", freq_khz=402500, ephemeris=str(E))
    fr = R.flight(2, eph)
    for k, f in enumerate(fr):
        sym = R.frame_symbols(f)[2 * 10 * 6:]                                            # behind the header bytes; one value per Manchester pair: second minus first
        t = d.hit(dict(soft=(sym[1::2].astype(np.float32) - sym[0::2].astype(np.float32)), mv=0.9, mv_pos=0))
        js = d.json_objects(t)
        assert len(js) == 1 and js[0]["id"] == "K1234567" and js[0]["frame"] == 2000 + k and abs(js[0]["lat"] - 47.7123) < 3e-4 and abs(js[0]["alt"] - 14321.0) < 30.0
    d.close()
    assert "lat" not in F.FamilyDecoder("RS92").hit(dict(soft=(sym[1::2].astype(np.float32) - sym[0::2].astype(np.float32)), mv=0.9, mv_pos=0))     # no orbit data: no position


def test_descriptors_match_the_c_front_ends():
    """header, baud, BT, h, hdmax, frame bits, filter bandwidths: the same numbers as host/<decoder>.c"""
    files = {"LMS6": "lms6Xmod.c", "MEISEI": "meisei100mod.c", "IMET5": "imet54mod.c", "MRZ": "mp3h1mod.c", "MTS01": "mts01mod.c", "RS92": "rs92mod.c"}
    for typ, fn in files.items():
        src = open(os.path.join(ROOT, "host", fn)).read()
        g = F.FAMILY[typ]["generic"]
        hdr = "".join(re.findall(r'"([01]+)"', re.search(r"kHeader\[\] = ([^;]+);", src).group(1)))
        assert hdr == g["header"], typ
        assert re.search(r"g\.bt = %sf" % g["bt"], src) and re.search(r"g\.h = %sf" % g["h"], src), typ
        assert re.search(r"g\.hdmax = %d;" % g["hdmax"], src), typ
        assert re.search(r"g\.symlen = %d; g\.symhd = %d;" % (g["symlen"], g["symhd"]), src), typ
        assert re.search(r"lpfm_bw = %d;" % g["lpfm_bw"], src) and re.search(r"cli_in_init\(&in, %d," % g["lpiq_bw"], src), typ
