"""bench.py contract pieces that do not need a GPU: the algorithmic FLOP count of SURVEY 8d, the
`cpu_baseline` record (the oracle timed on the host, bounded sample) and the command-line defaults."""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_flop_count_and_peak_match_the_survey():
    import bench
    per_token_fwd = 6 * (4 * 2 * 288 ** 2 + 3 * 2 * 288 * 768 + 2 * 2 * 256 * 288) + 2 * 288 * 32000
    assert per_token_fwd == 32_145_408
    assert bench.FLOP_PER_SAMPLE == 3 * 256 * per_token_fwd              # 24.688 GFLOP per sample fwd+bwd
    assert abs(bench.FLOP_PER_SAMPLE / 1e9 - 24.688) < 1e-3
    assert bench.PEAK_FP32_MFMA == 157.3e12


def test_cpu_baseline_record_shape():
    import bench
    rec = bench.cpu_baseline(seconds_budget=0.0)                         # warm-up + the minimum of 2 steps
    assert set(rec) == {"value", "unit", "cores", "kind", "sample", "product_numpy_device"}
    assert set(rec["product_numpy_device"]) == {"batch_1", "batch_8"}
    assert rec["unit"] == "samples/s" and rec["kind"] == "port" and rec["value"] > 0
    assert rec["cores"] == os.cpu_count() and "batch 1" in rec["sample"]


def test_bench_uses_no_pytorch():
    """Device memory, streams, events and RCCL come from libpdnhip.so: bench.py imports no torch."""
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    mods = {a.name.split(".")[0] for n in ast.walk(tree) if isinstance(n, ast.Import) for a in n.names}
    mods |= {n.module.split(".")[0] for n in ast.walk(tree) if isinstance(n, ast.ImportFrom) and n.module}
    assert "torch" not in mods


def test_command_line_defaults_and_json_keys():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    defaults = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            flag = node.args[0].value
            for kw in node.keywords:
                if kw.arg == "default" and isinstance(kw.value, ast.Constant):
                    defaults[flag] = kw.value.value
    assert defaults["--gpus"] == 1 and defaults["--steps"] == 10 and defaults["--warmup"] == 3
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"',
                '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"',
                '"roofline"', '"cpu_baseline"', '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"',
                '"parity_gate"'):
        assert key in src, key


def test_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus N` with WORLD_SIZE unset must start its own N ranks (one per GPU), each with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; only rank 0 owns stdout.  The probe switch makes
    every rank report its environment instead of touching a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["PDN_BENCH_SPAWN_PROBE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    err = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["probe"]["RANK"] == "0"                  # rank 0's line is THE output
    ranks = sorted(int(x["probe"]["RANK"]) for x in out + err)
    assert ranks == [0, 1, 2, 3]
    for x in out + err:
        pr = x["probe"]
        assert pr["WORLD_SIZE"] == "4" and pr["LOCAL_RANK"] == pr["RANK"] and pr["MASTER_ADDR"] == "127.0.0.1"
        assert pr["MASTER_PORT"] == out[0]["probe"]["MASTER_PORT"]
        assert x["argv"] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]


def test_launcher_propagates_a_failing_rank():
    import bench
    import subprocess
    # a rank that exits non-zero makes the whole launch fail (and the surviving ranks are terminated)
    code = ("import os, sys, time\n"
            "sys.exit(3) if os.environ['RANK'] == '1' else time.sleep(30)\n")
    path = os.path.join(ROOT, "tests", "_rank_probe_tmp.py")
    open(path, "w").write(code)
    real = bench.__file__
    try:
        bench.__file__ = path
        import time
        t0 = time.monotonic()
        rc = bench.launch_ranks(2, [], timeout=60)
        assert rc == 3 and time.monotonic() - t0 < 20
    finally:
        bench.__file__ = real
        os.remove(path)


COMPACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_gate", "batch_gate", "detail"}
ROOFLINE_KEYS = {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us",
                 "algorithmic_flop_per_launch", "traffic_source"}


def test_compact_line_of_the_round5_record_fits_the_driver_tail():
    """Round 5's full record (20 KB; the driver's 8 KB stdout tail lost its head) through compact_line: < 4 KB, the
    contract's keys, the same headline numbers."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    assert len(json.dumps(full)) > 16000
    c = bench.compact_line(full, "bench_detail.json")
    line = json.dumps(c)
    assert len(line) < bench.COMPACT_LIMIT == 4096, len(line)
    assert COMPACT_KEYS <= set(c) and ROOFLINE_KEYS <= set(c["roofline"])
    assert {"workload", "seq_len", "per_gpu_batch", "global_batch", "parallelism"} <= set(c["config"])
    assert {"value", "unit", "cores", "kind"} <= set(c["cpu_baseline"])
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"]
    assert c["roofline"]["frac"] == full["roofline"]["frac"] and c["roofline"]["kernel"] == full["roofline"]["kernel"]
    assert set(c["other_configs"]) == set(full["other_configs"])
    assert c["other_configs"]["lenet_b4096"]["ms"] == round(full["other_configs"]["lenet_b4096"]["ms_per_step"], 4)
    assert c["batch_gate"]["worst_grad_rel_err"] == full["batch_gate"]["worst_grad_rel_err"]
    # an over-long addition is dropped in favour of the headline, never the other way round
    full["other_configs"] = {f"cfg{i}": {"value": 1.0, "unit": "x" * 200, "ms_per_step": 1.0} for i in range(40)}
    c = bench.compact_line(full, "bench_detail.json")
    assert len(json.dumps(c)) < 4096 and isinstance(c["other_configs"], str) and c["value"] == full["value"]


def test_whole_bench_main_prints_one_compact_line_and_a_detail_file(emulated_hip, monkeypatch, tmp_path, capsys):
    """`bench.main()` at world 1 on the emulated ABI (model shrunk through the module constants; the CPU baseline and the
    other configs answered by canned records of REAL size): stdout is exactly one line, < 4 KB, with the key set the
    driver parses; the full record -- family tables, other configs, memory -- is in the side file the line names."""
    import json
    import bench
    full5 = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    for k, v in dict(V=192, D=96, H=2, F_=128, L=32, LAYERS=2).items():
        monkeypatch.setattr(bench, k, v)
    monkeypatch.setattr(bench, "cpu_baseline", lambda *a, **k: full5["cpu_baseline"])
    monkeypatch.setattr(bench, "other_configs", lambda *a, **k: full5["other_configs"])
    detail = tmp_path / "detail.json"
    monkeypatch.setenv("PDN_BENCH_DETAIL", str(detail))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-parity-gate"])
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(lines) == 1
    assert len(lines[0]) < 4096, len(lines[0])
    c = json.loads(lines[0])
    assert COMPACT_KEYS <= set(c) and ROOFLINE_KEYS <= set(c["roofline"])
    assert c["n_gpus"] == 1 and c["steps"] == 2 and c["warmup"] == 1 and c["config"]["per_gpu_batch"] == 2
    assert c["value"] > 0 and abs(c["value"] - 2 / (c["ms_per_step"] * 1e-3)) <= 1e-6 * c["value"]
    assert 0 < c["executed_flops_frac"] <= c["model_flops_frac_of_fp32_mfma_peak"]
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] > 0
    assert c["roofline"]["adam_hbm"]["avg_launch_us"] >= 0.0          # timed inside the steps (one event pair per step)
    assert set(c["other_configs"]) == set(full5["other_configs"])
    d = json.load(open(detail))
    assert d["value"] == c["value"] and "other_gemm_families" in d["roofline"] and "memory" in d
    assert d["roofline"]["hbm_bound_kernels"]["adam_multi_kernel"]["launches_timed"] == 2
    assert d["other_configs"]["lenet_b4096"]["roofline"]


def test_gpus_n_beyond_the_visible_devices_fails_fast():
    """`bench.py --gpus 8` on a box with fewer GPUs: refused before any rank is started (no RCCL peer left hanging)."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PDN_BENCH_SPAWN_PROBE")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and time.monotonic() - t0 < 10
    assert "--gpus 64 but only" in r.stderr and r.stdout.strip() == ""


def test_transformer_config_runs_on_the_emulated_device(emulated_hip):
    """`bench.py --config transformer` (the example of the reference's third published benchmark at its own width): gate
    against the reference-generated fixture, the plain-operator attention chain on the RESIDENT head-dim-128 kernels."""
    import argparse
    import bench_other
    a = argparse.Namespace(config="transformer", batch=4, steps=2, warmup=1, no_graph=True, no_cpu_baseline=True, gpus=1)   # (no hipGraphs on the emulated device)
    r = bench_other.run_transformer(a)
    assert r["attention_kernel"] == "resident" and r["value"] > 0 and r["unit"] == "samples/s"
    assert r["parity_gate"]["worst_loss_rel_err"] <= 1e-4 and r["config"]["per_gpu_batch"] == 4
