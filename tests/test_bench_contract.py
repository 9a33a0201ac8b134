"""The committed bench line (profiles/r01_bench.json, written by `python bench.py` on the GPU box) carries every field of the
driver's contract, and bench.py's argument surface is the contracted one."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    lines = open(os.path.join(ROOT, "profiles", "r01_bench.json")).read().strip().splitlines()
    d = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "mel-frames/sec" and d["unit"] == "mel-frames/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"] and "B=16" in d["config"]["workload"]
    assert abs(d["value"] - 16 * 800 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    x = d["roofline_cross_attention"]
    assert x["bound"] == "hbm" and x["peak"] == 8000.0 and abs(x["frac"] - x["achieved"] / x["peak"]) < 1e-9
    assert abs(x["achieved"] - x["algorithmic_bytes_per_launch"] / (x["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * x["achieved"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    assert d["parity"]["max_abs_mel_err"] < d["parity"]["tolerance"] == 1e-3


def test_bench_cli_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--streams"):
        assert flag in out.stdout
