"""The generated instruction streams of the decimator (radiosonde_auto_rx_amd/csrc/md_fast_gen.h) are committed; this checks that
the file is what tools/gen_md_fast.py produces today (no hand edits, no stale copy)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_md_fast_gen_in_sync():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_md_fast.py"), "--print"], capture_output=True, text=True, check=True).stdout
    cur = open(os.path.join(ROOT, "radiosonde_auto_rx_amd", "csrc", "md_fast_gen.h")).read()
    assert out == cur, "run `python3 tools/gen_md_fast.py` (without --experiments) and rebuild"


def test_md_loop_structure():
    """every wait that names a count is justified in the generator; here: the stream only uses the counts the schedule was derived for"""
    cur = open(os.path.join(ROOT, "radiosonde_auto_rx_amd", "csrc", "md_fast_gen.h")).read()
    waits = {l.split('"')[1].split("\\n")[0] for l in cur.splitlines() if "s_waitcnt" in l}
    assert waits <= {"s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(13)", "s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(0) lgkmcnt(0)"}, waits
    # 13 loads per staging set and per fetch: vmcnt(13) relies on it
    body = cur[cur.index("#define MD50_LOOP_1"):]
    first_fetch = body[:body.index("s_cmp_gt_i32")]
    assert first_fetch.count("global_load_dwordx4") == 12 and first_fetch.count("global_load_dwordx2") == 1
