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
