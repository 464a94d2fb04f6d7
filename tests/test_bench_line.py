"""The driver's contract on bench.py's output, held on CPU: stdout carries ONE compact strict-JSON line (< 4 KB) with the headline,
`roofline` and `cpu_baseline`; everything else lives in bench_extra.json.  (Round 3's 20.9 KB line was not parsed by the driver:
BENCH_r03.json `parsed: null`.)  The protocol mirrored: one small row per test, tests/testsCommon.cuh:128-195 of the reference."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config")


def _strict(text):
    def no_constants(c):
        raise ValueError("non-strict JSON constant " + c)
    return json.loads(text, parse_constant=no_constants)


def _check(text):
    assert "\n" not in text
    assert len(text) < 4096, len(text)
    j = _strict(text)
    for k in CONTRACT:
        assert k in j, k
    assert "workload" in j["config"] and "model" not in j["config"]
    return j


@pytest.mark.parametrize("name", ["r03_k_bench_default.json", "r03_j_bench_default.json"])
def test_round3_full_records_shrink_to_a_parseable_line(name):
    """The 20.9 KB records of round 3, through the formatter: contract keys, roofline and cpu_baseline survive, the sweeps do not."""
    full = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 8192  # the input really is the oversized record
    j = _check(bench.compact_line(full))
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1 and j["roofline"]["peak"] == 8000.0
    assert j["roofline"]["achieved"] > 0 and "traffic" in j["roofline"]
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
    assert j["value"] == full["value"] and j["ms_per_step"] == full["ms_per_step"]
    assert "extra" not in j and "timing" not in j
    assert j["one_launch_per_step"]["frac"] == full["one_launch_per_step"]["frac"]
    assert j["perf_gate"]["pass"] in (True, False)
    assert j["extra_file"] == bench.EXTRA_FILE


def test_a_pathological_record_still_fits():
    """Strings of any length and a huge extras block cannot push the line over the limit; the contract keys always survive."""
    fake = {"metric": bench.baseline_metric(), "value": 1.0, "unit": "Mpix/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 0.002,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w" * 5000, "exchange": "e" * 5000, "submission": "s" * 5000, "parallelism": "p" * 5000, "junk": ["x"] * 1000},
            "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.000125, "traffic": None, "kernel": "k", "junk": "j" * 9000},
            "cpu_baseline": {"value": 1.0, "unit": "Mpix/s", "cores": 1, "kind": "port", "sample": "s" * 9000},
            "n1_same_workload": {"Mpix_per_s": 1.0}, "legs": {"compute_only_us": 1.0}, "xgmi_probe": {"junk": "x" * 3000},
            "extra": {"perf_gate": {"pass": False, "checked": 70, "over": [{"test": "t" * 500}] * 70, "new": []}, "sweeps": ["y" * 100] * 500}}
    j = _check(bench.compact_line(fake))
    assert j["roofline"]["frac"] == 0.000125 and j["cpu_baseline"]["value"] == 1.0 and j["n_gpus"] == 8


def test_error_line_is_compact_and_parseable():
    class A:
        gpus, steps, warmup = 8, 20, 5
    j = _check(bench.error_line(A, "--gpus 8 but this box has 1 visible GPU(s)"))
    assert j["value"] is None and "error" in j


def test_only_the_result_line_reaches_stdout():
    """guard_stdout(): Python prints and the C runtime's stdio (RCCL's banner) end up on stderr; stdout holds exactly the line."""
    code = ("import sys, os, ctypes; sys.path.insert(0, %r); import bench; bench.guard_stdout(); print('python noise'); "
            "ctypes.CDLL(None).puts(b'C stdio noise'); "
            "bench.EXTRA_FILE = os.devnull; "
            "bench.write_line(bench.compact_line({'metric': 'm', 'value': 1, 'config': {'workload': 'w'}}))" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["value"] == 1
    assert "python noise" in p.stderr and "C stdio noise" in p.stderr


def test_gpus_n_without_the_gpus_prints_a_parseable_error_line():
    """`python bench.py --gpus 8` on a box with fewer GPUs (here: none): a compact JSON error line as the last stdout line, exit != 0
    -- never a SystemExit message the driver cannot parse (round 3: bench.py:402-403)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")})
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("this box really has 8 GPUs")
    assert p.returncode != 0
    j = _check(p.stdout.strip().splitlines()[-1])
    assert j["value"] is None and j["n_gpus"] == 8 and "visible GPU" in j["error"]
