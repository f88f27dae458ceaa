"""The line bench.py prints must fit the driver's record of it (the last ~8 KB of stdout) with every object in it: `compact` keeps the figures and drops the prose
(which stays in the full object bench.py writes beside it, and in DESIGN.md §5).  Checked on the committed full objects of earlier runs (CPU, no GPU needed)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _walk(o):
    if isinstance(o, dict):
        for k, v in o.items():
            yield k, v
            yield from _walk(v)
    elif isinstance(o, list):
        for v in o:
            yield from _walk(v)


def test_compact_line_fits_and_keeps_every_object():
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files
    for path in files[-2:]:
        full = json.load(open(path))
        c = bench.compact(full)
        line = json.dumps(c, separators=(",", ":"))
        assert len(line) <= bench.LINE_LIMIT, (path, len(line))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in c and (c[k] == full[k] or isinstance(full[k], (dict, float, str))), k          # the contract's keys, unchanged where they are plain values
        assert c["config"]["workload"] == full["config"]["workload"] and c["roofline"]["frac"] == full["roofline"]["frac"]
        for name, sub in full.items():                                     # every sub-object keeps its own figures
            if isinstance(sub, dict) and "ms_per_step" in sub:
                assert c[name]["ms_per_step"] == sub["ms_per_step"], name
                for r in ("roofline", "cpu_baseline"):
                    if r in sub:
                        assert r in c[name] and c[name][r].get("frac", c[name][r].get("value")) is not None
        assert not any(str(k).endswith("note") for k, _ in _walk(c))
        assert all(len(v) <= 230 for _, v in _walk(c) if isinstance(v, str))
