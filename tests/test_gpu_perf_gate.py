"""The measured perf gate must catch what it claims to catch (round 3's ceilings -- never under 13 us -- would not have
caught a 2-3 x regression of any short chain).  tools/perf_gate.py --quick times K1's one-launch headline and the whole-frame up-scaling
kernels against the committed ceilings (max(1.25 x, + 1.5 us) over the median of three graph-replayed measurements):
  * the tree as built passes;
  * with deliberately bad knobs -- K1 forced to 4 rows per wave at 50 crops (CVGS_K1_RPW=4) and the four-pixels-per-lane up-scaling kernel
    switched off (CVGS_K1_X4=0: the one-pixel kernel, the round-2 state) -- it FAILS, and the 1.5 x slip of the 540p -> 1080p resize
    (6.0 -> 8.9 us) is among the chains it names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_gate(extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "perf_gate.py"), "--quick"], capture_output=True, text=True, env=env, timeout=600)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return p.returncode, json.loads(line)


def test_the_gate_passes_the_tree_and_fails_a_deliberately_slow_build():
    rc, v = run_gate({})
    if rc != 0:  # one retry: a cold box's first seconds (clock ramp) are the only legitimate reason
        rc, v = run_gate({})
    assert rc == 0 and v["pass"] and v["checked"] >= 4 and not v["new"], v
    rc, v = run_gate({"CVGS_K1_RPW": "4", "CVGS_K1_X4": "0"})
    assert rc == 1 and not v["pass"], v
    over = {o["chain"]: o["us"] / o["ceiling_us"] for o in v["over"]}
    assert "resize packed 1920x1080 -> 3840x2160 u8c3" in over and "resize packed 960x540 -> 1920x1080 u8c3" in over, v
    slow = {r["chain"]: r["us"] for r in v["rows"]}
    table = json.load(open(os.path.join(ROOT, "tools", "perf_ceilings_measured.json")))
    # the smallest slip it caught is about 1.5 x of the recorded median
    ratio = slow["resize packed 960x540 -> 1920x1080 u8c3"] / table["resize packed 960x540 -> 1920x1080 u8c3"]
    assert 1.25 < ratio < 2.0, ratio
