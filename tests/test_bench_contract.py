"""The committed bench line (profiles/r01_bench.json, written by `python bench.py` on the GPU box) carries every field of the
driver's contract, and bench.py's argument surface is the contracted one."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_bench_line():
    pdir = os.path.join(ROOT, "profiles")
    f = sorted(x for x in os.listdir(pdir) if x.endswith("_bench.json"))[-1]
    return f, json.loads(open(os.path.join(pdir, f)).read().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_fields():
    fname, d = _latest_bench_line()
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
    if "fused_launch" in x:          # round 4 on: no stand-alone cross-attention kernel; the fused chain launch reports its own time and bytes
        fl = x["fused_launch"]
        assert fl["launches_per_step"] >= 1 and fl["avg_launch_us"] > 0 and fl["alignment_bytes_per_launch"] == 4.0 * 16 * 4 * 400 * 128
        assert abs(fl["attention_core_bytes_per_launch"] - 30408704.0) < 1.0
    else:
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
    for flag in ("--gpus", "--steps", "--warmup", "--streams", "--in-flight"):
        assert flag in out.stdout


def test_round2_line_is_coherent():
    """From round 2 on: `value` is the single-stream schedule the roofline blocks were profiled on, the roofline fraction is
    taken against the pipe the dominant kernel really uses, and a PMC traffic figure is only printed when its record carries
    the digest of the running kernel sources."""
    fname, d = _latest_bench_line()
    if fname < "r02":
        return
    assert d["config"]["batches_in_flight_per_gpu"] == 1
    r = d["roofline"]
    assert r["class"] in ("chain", "gemm", "gemm_fp32")
    assert r["peak"] == (157.3 if r["class"] == "gemm_fp32" else 2500.0)
    if r["class"] != "gemm_fp32":
        assert abs(r["achieved"] - 3 * r["algorithmic_tflops"]) < 1e-6 * r["achieved"]
    # same schedule: the kernel time of the profiled pass matches the step (the event-instrumented launches run a few per cent
    # slower than the plain ones of the timed region; with several batches in flight the sum would be 1.5x the step)
    assert 0.85 * d["ms_per_step"] <= d["end_to_end"]["kernel_ms_sum"] <= 1.08 * d["ms_per_step"]
    if "profiled_pass_ms_per_step" in d["end_to_end"]:      # round 4 on: the pass the per-kernel numbers come from is timed itself
        assert d["end_to_end"]["kernel_ms_sum"] <= 1.005 * d["end_to_end"]["profiled_pass_ms_per_step"]
    if fname >= "r04":
        assert "exact_fp32" in d and d["exact_fp32"]["ms_per_step"] > d["ms_per_step"] and d["exact_fp32"]["max_abs_mel_err"] < 1e-3
        t = d["training"]
        assert t["rccl_ranks"] == 1 and t["roofline"]["peak"] == 2500.0 and 0 < t["roofline"]["frac"] < 1
        assert abs(t["roofline"]["frac"] - t["roofline"]["achieved"] / 2500.0) < 1e-9
    if r["traffic"] is not None:
        assert d["kernel_source_digest"] in r["traffic_source"] or "same kernel sources" in r["traffic_source"]


def test_launcher_spawns_n_ranks_with_torchrun_style_env(tmp_path):
    """`bench.py --gpus N` without WORLD_SIZE: the parent starts N fresh processes (fake worker here: no GPU), each with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, and relays rank 0's stdout."""
    sys.path.insert(0, ROOT)
    import bench
    worker = tmp_path / "fake_worker.py"
    worker.write_text(
        "import json, os, sys\n"
        "e = {k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}\n"
        "e['argv'] = sys.argv[1:]\n"
        "open(os.path.join(%r, 'rank%%s.json' %% e['RANK']), 'w').write(json.dumps(e))\n"
        "if e['RANK'] == '0': print(json.dumps({'n_gpus': int(e['WORLD_SIZE'])}))\n" % str(tmp_path))
    rc, out0 = bench.launch_ranks(3, ["--gpus", "3", "--steps", "2"], worker=[sys.executable, str(worker)], timeout=60)
    assert rc == 0 and json.loads(out0.strip())["n_gpus"] == 3
    envs = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and all(e["argv"] == ["--gpus", "3", "--steps", "2"] for e in envs)


def test_launcher_propagates_a_failing_rank(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    worker = tmp_path / "w.py"
    worker.write_text("import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(30)\n")
    t0 = __import__("time").time()
    rc, _ = bench.launch_ranks(2, [], worker=[sys.executable, str(worker)], timeout=60)
    assert rc == 7 and __import__("time").time() - t0 < 20      # the surviving rank was terminated, not waited for


def test_watchdog_fires_while_the_main_thread_sits_in_a_ctypes_call(tmp_path):
    """The training-block watchdog must end a rank that is blocked INSIDE a foreign call (a stalled RCCL exchange under
    hipStreamSynchronize): a timer thread, not a Python signal handler; the headline line still comes out; exit code non-zero."""
    worker = tmp_path / "blocked.py"
    worker.write_text(
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "bench.start_watchdog(1.0, lambda: '{\"value\": 1, \"training\": {\"error\": \"watchdog\"}}')\n"
        "ctypes.CDLL(None).sleep(60)\n"               # one blocking C call: no bytecode boundary for a signal handler
        "print('not reached')\n" % ROOT)
    t0 = __import__("time").time()
    out = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=50)
    assert __import__("time").time() - t0 < 30
    assert out.returncode == 3
    assert json.loads(out.stdout.strip().splitlines()[-1])["training"]["error"] == "watchdog" and "not reached" not in out.stdout


def test_plan_of_a_two_rank_strong_scaled_run_in_fresh_child_processes():
    """VERDICT round 5 "next round" #7: `bench.py --gpus 2 --global-batch 32` deals 16 utterances to each rank.  `--plan` runs the real
    launcher (two fresh child processes with torchrun-style environments, parent makes no GPU call), the real control plane (TCP star,
    barrier / gather / max) and the same arithmetic as the measuring path -- without a GPU."""
    import json
    import bench
    rc, out = bench.launch_ranks(2, ["--gpus", "2", "--global-batch", "32", "--plan"], timeout=120.0)
    assert rc == 0, out
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["error"] is None and d["max_rank_seen"] == 1.0
    assert d["training"] == {"batch_per_rank": 16, "global_batch": 32, "collective": "RCCL all-reduce of the flat gradient (4 buckets)"}
    assert [(r["rank"], r["device"], r["train_batch"], r["inference_batch"]) for r in d["ranks"]] == [(0, 0, 16, 16), (1, 1, 16, 16)]
    assert len({r["batch_seed"] for r in d["ranks"]}) == 2          # the ranks draw different utterances
    # weak scaling (no --global-batch): B = 32 on every rank; an uneven deal is refused by name
    rc, out = bench.launch_ranks(2, ["--gpus", "2", "--plan"], timeout=120.0)
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert rc == 0 and d["training"]["batch_per_rank"] == 32 and d["training"]["global_batch"] == 64 and d["inference"]["global_batch"] == 32
    assert bench.train_batch_per_rank(33, 2) == (0, "--global-batch 33 is not a multiple of the 2 ranks")
    assert bench.train_batch_per_rank(32, 8) == (4, None) and bench.train_batch_per_rank(0, 8) == (32, None)
