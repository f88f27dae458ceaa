"""One wideband IQ stream -> every RS41, DFM, M10, M20 (and LMS6, iMet-54, Meisei, MRZ, MTS01) in it, in one process on one GPU.

The reference handles a wideband source by starting one detector process per candidate peak (auto_rx/autorx/scan.py:413-656:
rtl_power peaks -> `dft_detect` per peak) and then one decoder pipeline per sonde (decode.py).  Here the same two steps run
batched over a frequency raster: a Scanner whose channels all mix their own fq out of the shared stream (`dft_detect --IQ fq
--dc` per raster point, channel stride 0), and one demodulator Engine per detected sonde fed from the same chunks, followed by the
telemetry tier.  Everything sample-rate runs on the GPU; this module is the orchestration auto_rx does in Python.

    python -m radiosonde_auto_rx_amd.wideband --cfreq 403000000 --raster 10000 - 2400000 16 < capture.cs16

prints one JSON object per decoded frame (the reference's `rs41mod --json` object, "freq" = channel frequency in kHz).
"""
from __future__ import annotations

import json
import sys

import numpy as np

from .engine import Engine, snap_fq
from .scan import Scanner
from .family import FAMILY, LMS_BASE, FamilyDecoder
from .telemetry import DfmTelemetry, M10Telemetry, M20Telemetry, Rs41Telemetry


class WidebandReceiver:
    def __init__(self, sample_rate: int, *, cfreq_hz: int = 0, raster_hz: int = 10_000, span: float = 0.45, chunk: int | None = None,
                 merge_hz: float = 6_000.0, version: str = "sonde_hip", idle_s: float = 30.0):
        """idle_s: a decoder that has produced no frame for that long (a false detection, a sonde that has landed) is closed and its engine
        freed; the scanner starts a new one if the signal comes back."""
        self.sr, self.cfreq, self.merge_hz, self.version, self.idle_s = sample_rate, cfreq_hz, merge_hz, version, idle_s
        self.t = 0.0                           # stream time in seconds
        kmax = int(span * sample_rate / raster_hz)
        self.raster = [snap_fq(k * raster_hz / sample_rate, sample_rate) for k in range(-kmax, kmax + 1)]
        self.chunk = chunk or sample_rate // 4
        self.scanner = Scanner(sample_rate, fq=self.raster, dc=True, cont=True, max_chunk=self.chunk)
        # the demodulators decimate to the reference's IF rate (48 kHz, raised until it divides the sample rate, demod_mod.c:1229-1236);
        # pushes are cut at multiples of both decimation factors, the rest of a push waits for the next one
        if_sr = min(48000, sample_rate)
        while sample_rate % if_sr:
            if_sr += 1
        D = int(np.lcm(self.scanner.info["decM"], sample_rate // if_sr))
        self.align, self._rest = D, None
        self.chunk -= self.chunk % D
        self.sondes: list[dict] = []           # {fq, engine, telemetry, type, frames}
        self.log: list[dict] = []

    def _start(self, fq: float, typ: str):
        for s in self.sondes:
            if abs(s["fq"] - fq) * self.sr < self.merge_hz * (3 if typ in ("M10", "M20") else 1):    # 9.6 kBd: seen from neighbouring raster points too
                return
        fq = snap_fq(fq, self.sr)
        khz = int(round((self.cfreq + fq * self.sr) / 1000.0)) if self.cfreq else 0
        if typ in FAMILY:                                      # generic sonde description + the type's bit-rate tier (family.py)
            eng = self._family_engine(typ, fq)
            tel = FamilyDecoder(typ, freq_khz=khz, version=self.version)
        elif typ == "DFM":
            eng = Engine([fq], self.sr, sonde="dfm", ecc=1, auto=True, max_chunk=self.chunk, max_frames=8)
            tel = DfmTelemetry(freq_khz=khz, version=self.version)
        elif typ in ("M10", "M20"):
            eng = Engine([fq], self.sr, sonde=typ.lower(), max_chunk=self.chunk, max_frames=8)
            tel = (M10Telemetry if typ == "M10" else M20Telemetry)(freq_khz=khz, version=self.version)
        else:
            eng = Engine([fq], self.sr, max_chunk=self.chunk, max_frames=8)
            tel = Rs41Telemetry(freq_khz=khz, version=self.version)
        self.sondes.append(dict(fq=fq, type=typ, engine=eng, telemetry=tel, frames=0, khz=khz, t_last=self.t))
        self.log.append(dict(event="detected", type=typ, fq=fq, freq_khz=khz))

    def _family_engine(self, typ: str, fq: float):
        f = FAMILY[typ]
        return Engine([fq], self.sr, sonde="generic", generic=f["generic"], thres=f["thres"], auto=f["auto"], keep_soft=True, lp_iq=True,
                      max_chunk=self.chunk, max_frames=8)

    def _follow_lms(self, s):
        """an LMS6 whose decoder found LMS-X blocks (or the reverse): from the next chunk on its samples go through an engine of the other description
        (the stand-alone decoder replays its input from the end of the block instead, host/lms6Xmod.c; here the stream is live and that block's successor is lost)"""
        want, changed = s["telemetry"].lms_type()
        if changed and want != s["type"]:
            s["engine"].close()
            s["engine"] = self._family_engine(want, s["fq"])
            self.log.append(dict(event="retuned", type=want, was=s["type"], fq=s["fq"], freq_khz=s["khz"]))
            s["type"], s["telemetry"].moved = want, True

    def push(self, iq: np.ndarray, finish: bool = False):
        """iq: interleaved int16 I/Q, a whole number of chunks is not required; returns the JSON objects of this call."""
        out = []
        if self._rest is not None and len(self._rest):       # samples that did not fill a decimation block last time
            iq = np.concatenate([self._rest, np.asarray(iq, np.int16)])
        n = len(iq) // 2
        D = self.align                                        # lcm of the scanner's and the demodulators' decimation factors
        self._rest = np.array(iq[2 * (n - n % D):2 * n], np.int16)
        for s0 in range(0, n - n % D, self.chunk):
            x = iq[2 * s0:2 * min(n - n % D, s0 + self.chunk)]
            self.scanner.process_host(x, shared=True)
            for d in self.scanner.fetch():
                if d["type"] == "RS41" and d["score"] > 0:
                    self._start(self.raster[d["channel"]] + d["df"], "RS41")
                elif d["type"] == "DFM9":                                   # either polarity: the decoder runs with --auto
                    self._start(self.raster[d["channel"]] + d["df"], "DFM")
                elif d["type"] in ("M10", "M20"):                           # differential code: polarity does not matter
                    self._start(self.raster[d["channel"]] + d["df"], d["type"])
                elif d["type"] in FAMILY and (d["score"] > 0 or FAMILY[d["type"]]["auto"]):
                    self._start(self.raster[d["channel"]] + d["df"], d["type"])
            self.t += (len(x) // 2) / self.sr
            for s in list(self.sondes):
                s["engine"].process_host(x)
                before, hits = s.get("good", 0), s["frames"]
                out += self._drain(s, False)
                if s["frames"] != hits and s["type"] in ("LMS6", "LMSX"):     # any block tells what the sonde is, accepted or not
                    self._follow_lms(s)
                if s.get("good", 0) != before:      # only frames that passed their check keep a channel alive
                    s["t_last"] = self.t
                elif self.t - s["t_last"] > self.idle_s:      # silent for too long: give the engine back
                    out += self._drain(s, True)
                    s["engine"].close(); s["telemetry"].close()
                    self.sondes.remove(s)
                    self.log.append(dict(event="released", type=s["type"], fq=s["fq"], freq_khz=s["khz"], frames=s["frames"]))
        if finish:
            for s in self.sondes:
                out += self._drain(s, True)
        return out

    @staticmethod
    def _drain(s, finish):
        e = s["engine"]
        if s["type"] in FAMILY:
            out = []
            for h in e.fetch_hits(finish=finish):
                s["frames"] += 1
                js = FamilyDecoder.json_objects(s["telemetry"].hit(h, e.info["if_sr"]))
                s["good"] = s.get("good", 0) + (1 if js else 0)
                out += js
            return out
        frames = e.fetch_dfm(finish=finish) if s["type"] == "DFM" else e.fetch_mxx(finish=finish) if s["type"] in ("M10", "M20") else e.fetch_frames(finish=finish)
        out = []
        for fr in frames:
            js = s["telemetry"].json(fr)
            s["frames"] += 1
            s["good"] = s.get("good", 0) + int(_frame_valid(s["type"], fr))
            if js is not None:
                out.append(js)
        return out

    def close(self):
        self.scanner.close()
        for s in self.sondes:
            s["engine"].close(); s["telemetry"].close()



def _frame_valid(typ, fr):
    """a frame that passed its own check — the Reed-Solomon code (RS41), all three Hamming blocks (DFM), the checksum (M10 / M20): only such
    frames count as a sign of life when a receiver decides whether a channel has gone silent (the reference's decoders time out on the absence
    of VALID telemetry; a false detection that keeps producing garbage hits must not hold a channel)"""
    if typ == "DFM":
        return all(e >= 0 for e in fr["ecc"])
    if typ in ("M10", "M20"):
        return bool(fr["cs_ok"])
    return fr["ecc"] >= 0


class ChannelizedReceiver:
    """The same job for wide streams (BASELINE configs[2]: 10 Msps and 256 channels) without re-reading the stream once per channel:

        stream -> polyphase channelizer (chan.py: M channels at sr / D, one pass) -> Scanner on all M channels (`dft_detect --iq --dc` each)
               -> per sonde type ONE demodulator engine whose channels are handed out at run time: a detection takes a free channel
                  (sonde_engine_restart_channel), is fine-tuned to the offset the scanner measured inside its channelizer channel
                  (cfg.if_tune / sonde_engine_tune_channel) and is fed that channel's samples from then on -> telemetry JSON.

    Every block costs one channelizer launch, one scanner step and one launch sequence per sonde type, whatever the number of sondes."""

    TYPES = {"RS41": ("rs41", Rs41Telemetry), "DFM": ("dfm", DfmTelemetry), "M10": ("m10", M10Telemetry), "M20": ("m20", M20Telemetry)}
    # + the scanner's RS92 / LMS6 / MEISEI / IMET5 / MRZ / MTS01: generic sonde descriptions and the bit-rate tiers of family.py

    def __init__(self, sample_rate: int, *, M: int = 256, D: int = 200, P: int = 16, cfreq_hz: int = 0, slots: int = 16, chunk: int | None = None,
                 version: str = "sonde_hip", device: int = 0, idle_s: float = 30.0):
        """idle_s: a decoder channel that has produced no frame for that long (a false detection, a sonde that has landed) is ended
        (sonde_engine_finish_channel) and handed to the next detection of its type."""
        import torch
        self.idle_s, self.t = idle_s, 0.0
        from .chan import Channelizer
        from .scan import IFIQ
        self.torch, self.sr, self.cfreq, self.version, self.slots, self.device = torch, sample_rate, cfreq_hz, version, slots, device
        self.chunk = chunk or sample_rate // 4
        self.chunk -= self.chunk % D
        self.ch = Channelizer(sample_rate, M, D, P, max_chunk=self.chunk, device=device)
        self.M, self.if_sr, self.nmax = M, int(self.ch.out_rate), self.ch.max_frames
        dev = torch.device("cuda", device)
        self.out = torch.zeros(M, self.nmax, 2, dtype=torch.float32, device=dev)
        self.scanner = Scanner(self.if_sr, n_channels=M, iq_mode=IFIQ, dc=True, cont=True, max_chunk=self.nmax, device=device, bits=32)
        self.groups: dict[str, dict] = {}       # type -> {engine, stage, owner[slot] = sonde dict or None}
        self.sondes: list[dict] = []
        self.log: list[dict] = []
        self._rest = None
        torch.cuda.synchronize(dev)

    def _group(self, typ: str):
        g = self.groups.get(typ)
        if g is None:
            kw = dict(bits=32, iq_mode=3, if_tune=True, lp_iq=True, max_chunk=self.nmax, max_frames=8 * self.slots, device=self.device)
            if typ in FAMILY:
                f = FAMILY[typ]
                kw.update(sonde="generic", generic=f["generic"], thres=f["thres"], auto=f["auto"], keep_soft=True)
            else:
                kw.update(sonde=self.TYPES[typ][0])
            if typ == "DFM":
                kw.update(ecc=1, auto=True)
            eng = Engine([0.0] * self.slots, self.if_sr, **kw)
            stage = self.torch.zeros(self.slots, self.nmax, 2, dtype=self.torch.float32, device=self.out.device)
            self.torch.cuda.synchronize(self.out.device)
            g = self.groups[typ] = dict(engine=eng, stage=stage, owner=[None] * self.slots, calls=0)
        return g

    def _start(self, k: int, typ: str, df: float):
        f_hz = self.ch.channel_freq(k) + df * self.if_sr
        for s in self.sondes:                                    # the neighbouring channel sees a strong signal too
            if LMS_BASE.get(s["type"], s["type"]) == typ and abs(s["f_hz"] - f_hz) < (FAMILY[typ]["sep_hz"] if typ in FAMILY else 20_000.0 if typ in ("M10", "M20") else 8_000.0):
                return
        g = self._group(typ)
        if None not in g["owner"]:
            self.log.append(dict(event="no free channel", type=typ, f_hz=f_hz))
            return
        slot = g["owner"].index(None)
        if g["calls"]:
            g["engine"].restart_channel(slot)
        g["engine"].tune_channel(slot, df)
        khz = int(round((self.cfreq + f_hz) / 1000.0)) if self.cfreq else 0
        tel = FamilyDecoder(typ, freq_khz=khz, version=self.version) if typ in FAMILY else self.TYPES[typ][1](freq_khz=khz, version=self.version)
        s = dict(type=typ, f_hz=f_hz, chan=k, slot=slot, telemetry=tel, frames=0, khz=khz, t_last=self.t, seen=0)
        g["owner"][slot] = s
        self.sondes.append(s)
        self.log.append(dict(event="detected", type=typ, f_hz=f_hz, channel=k, slot=slot, freq_khz=khz))

    def push(self, iq: np.ndarray, finish: bool = False):
        """iq: interleaved int16 I/Q of the wide stream; returns the JSON objects of this call."""
        torch = self.torch
        out = []
        if self._rest is not None and len(self._rest):
            iq = np.concatenate([self._rest, np.asarray(iq, np.int16)])
        n = len(iq) // 2
        D = self.sr // self.if_sr
        self._rest = np.array(iq[2 * (n - n % D):2 * n], np.int16)
        for s0 in range(0, n - n % D, self.chunk):
            x = np.ascontiguousarray(iq[2 * s0:2 * min(n - n % D, s0 + self.chunk)])
            m = self.ch.process_host(x, self.out.data_ptr(), self.nmax)           # m IF samples per channel
            self.ch.sync()
            self.scanner.process_device(self.out.data_ptr(), self.nmax, m)
            for d in self.scanner.fetch():
                if d["type"] == "RS41" and d["score"] > 0:
                    self._start(d["channel"], "RS41", d["df"])
                elif d["type"] == "DFM9":
                    self._start(d["channel"], "DFM", d["df"])
                elif d["type"] in ("M10", "M20"):
                    self._start(d["channel"], d["type"], d["df"])
                elif d["type"] in FAMILY and (d["score"] > 0 or FAMILY[d["type"]]["auto"]):
                    self._start(d["channel"], d["type"], d["df"])
            for typ, g in list(self.groups.items()):
                for slot, s in enumerate(g["owner"]):
                    if s is not None:
                        g["stage"][slot, :m] = self.out[s["chan"], :m]
                torch.cuda.synchronize(self.out.device)                           # the copies ran on torch's stream, the engine has its own
                g["engine"].process_device(g["stage"].data_ptr(), self.nmax, m)
                g["calls"] += 1
                out += self._drain(typ, g, False)
                for slot, s in enumerate(g["owner"]):                             # channels that have gone silent go back to the pool
                    if s is None:
                        continue
                    if s.get("good", 0) != s["seen"]:                              # only frames that passed their check keep a channel
                        s["seen"], s["t_last"] = s.get("good", 0), self.t
                    elif self.t - s["t_last"] > self.idle_s:
                        g["engine"].finish_channel(slot)
                        out += self._drain(typ, g, False)                         # what the end of its stream still gave
                        g["owner"][slot] = None
                        s["telemetry"].close()
                        self.sondes.remove(s)
                        self.log.append(dict(event="released", type=typ, f_hz=s["f_hz"], slot=slot, frames=s["frames"]))
            for s in [s for s in self.sondes if s["type"] in ("LMS6", "LMSX")]:     # after every engine has had this block: sondes whose decoder changed type
                self._follow_lms(s)
            self.t += m / self.if_sr
        if finish:
            for typ, g in self.groups.items():
                out += self._drain(typ, g, True)
        return out

    def _follow_lms(self, s):
        """an LMS6 whose decoder found LMS-X blocks (or the reverse): the sonde moves to a channel of the engine of the other description, decoder object and
        all; the next block of samples is the first it sees there"""
        want, changed = s["telemetry"].lms_type()
        if not changed or want == s["type"]:
            return
        g0 = self.groups[s["type"]]
        g0["engine"].finish_channel(s["slot"])
        g0["owner"][s["slot"]] = None
        g = self._group(want)
        if None not in g["owner"]:
            s["telemetry"].close()
            self.sondes.remove(s)
            self.log.append(dict(event="no free channel", type=want, f_hz=s["f_hz"]))
            return
        slot = g["owner"].index(None)
        if g["calls"]:
            g["engine"].restart_channel(slot)
        g["engine"].tune_channel(slot, (s["f_hz"] - self.ch.channel_freq(s["chan"])) / self.if_sr)
        self.log.append(dict(event="retuned", type=want, was=s["type"], f_hz=s["f_hz"], slot=slot))
        s["type"], s["slot"], s["telemetry"].moved = want, slot, True
        g["owner"][slot] = s

    def _drain(self, typ, g, finish):
        e = g["engine"]
        if typ in FAMILY:                                          # header hits -> the type's bit-rate tier (family.py), one decoder object per sonde
            out = []
            for h in e.fetch_hits(finish=finish):
                s = g["owner"][h["channel"]]
                if s is None:
                    continue
                s["frames"] += 1
                js = FamilyDecoder.json_objects(s["telemetry"].hit(h, self.if_sr))
                s["good"] = s.get("good", 0) + (1 if js else 0)
                out += js
            return out
        frames = e.fetch_dfm(finish=finish) if typ == "DFM" else e.fetch_mxx(finish=finish) if typ in ("M10", "M20") else e.fetch_frames(finish=finish)
        out = []
        for fr in frames:
            s = g["owner"][fr["channel"]]
            if s is None:
                continue
            js = s["telemetry"].json(fr)
            s["frames"] += 1
            s["good"] = s.get("good", 0) + int(_frame_valid(typ, fr))
            if js is not None:
                out.append(js)
        return out

    def close(self):
        self.scanner.close(); self.ch.close()
        for g in self.groups.values():
            g["engine"].close()
        for s in self.sondes:
            s["telemetry"].close()


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--cfreq", type=int, default=0, help="centre frequency of the stream in Hz (for the JSON freq field)")
    ap.add_argument("--raster", type=int, default=10_000, help="scanner raster in Hz")
    ap.add_argument("--channelize", action="store_true", help="polyphase channelizer front end (256 channels at sr / 200): for streams of several Msps")
    ap.add_argument("--rs92-ephem", help="RINEX navigation file for RS92 positions (rs92mod -e)")
    ap.add_argument("--rs92-alm", help="SEM almanac for RS92 positions (rs92mod -a)")
    ap.add_argument("dash"); ap.add_argument("sr", type=int); ap.add_argument("bits", type=int)
    a = ap.parse_args(argv)
    if a.rs92_ephem or a.rs92_alm:
        from .family import set_rs92_orbits
        set_rs92_orbits(ephemeris=a.rs92_ephem, almanac=a.rs92_alm)
    if a.dash != "-" or a.bits != 16:
        ap.error("input is `- <sr> 16` (cs16 on stdin)")
    rx = ChannelizedReceiver(a.sr, cfreq_hz=a.cfreq) if a.channelize else WidebandReceiver(a.sr, cfreq_hz=a.cfreq, raster_hz=a.raster)
    inp = sys.stdin.buffer
    while True:
        buf = inp.read(rx.chunk * 4)
        last = len(buf) < rx.chunk * 4                        # also an EMPTY read: the frame in progress at EOF is still due
        for js in rx.push(np.frombuffer(buf[:len(buf) // 4 * 4], np.int16), finish=last):
            print(json.dumps(js), flush=True)
        if last:
            break
    rx.close()


if __name__ == "__main__":
    main()
