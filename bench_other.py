"""`bench.py --config {mlp, lenet, gru, decode}`: the other workloads of BASELINE.json (configs 2 and 3), the GRU of
examples/pydynet/ts_prediction.py and the KV-cache greedy decode of llm/llama/infer.py:46-63, each as ONE JSON line
of the same shape as the headline line (metric / value / unit / ... / roofline of the dominant kernel timed live with
HIP events / cpu_baseline = the oracle on this host's cores).  One GPU; no PyTorch anywhere.

Launch-bound training configs (MLP and LeNet at the reference's default batch 256, mnist.py:106-109) replay the whole
step -- forward, backward, Adam -- as ONE hipGraph by default (`--no-graph` times the eager launches instead).
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

PEAK_FP32_MFMA = 157.3e12
PEAK_HBM = 8.0e12
ROOT = os.path.dirname(os.path.abspath(__file__))

MLP_FLOP = 9_564_160           # 3 x 2 x (784*1024 + 1024*1024 + 1024*10) - dX of layer 1
LENET_FLOP = 25_665_840
LENET_CONV_BYTES_FWD = (12 + 80 + 20 + 50 + 12.5) * 1024     # x, y1, pooled, y2, pooled per sample


def _timed_cpu(step, budget, max_steps):
    step()
    t0, n = time.perf_counter(), 0
    while n < 2 or (time.perf_counter() - t0 < budget and n < max_steps):
        step()
        n += 1
    return n, time.perf_counter() - t0


def _models():
    import pydynet_amd.nn as nn
    import pydynet_amd.nn.functional as F

    class MLP(nn.Module):                      # examples/pydynet/mnist.py:65-79
        def __init__(self):
            super().__init__()
            self.layer1 = nn.Linear(784, 1024, dtype=np.float32)
            self.layer2 = nn.Linear(1024, 1024, dtype=np.float32)
            self.layer3 = nn.Linear(1024, 10, dtype=np.float32)

        def forward(self, x):
            x = x.reshape(x.shape[0], -1)
            return self.layer3(F.relu(self.layer2(F.relu(self.layer1(x)))))

    class LeNet(nn.Module):                    # mnist.py:82-98, shape-adapted to 3x32x32
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 20, 3, 1, 1, dtype=np.float32)
            self.conv2 = nn.Conv2d(20, 50, 3, 1, 1, dtype=np.float32)
            self.fc1 = nn.Linear(8 * 8 * 50, 500, dtype=np.float32)
            self.fc2 = nn.Linear(500, 10, dtype=np.float32)

        def forward(self, x):
            x = F.max_pool2d(F.relu(self.conv1(x)), 2, 2)
            x = F.max_pool2d(F.relu(self.conv2(x)), 2, 2)
            return self.fc2(F.relu(self.fc1(x.reshape(-1, 8 * 8 * 50))))
    return MLP, LeNet


def _time_steps(hp, step, steps, warmup, use_graph):
    """(seconds for `steps` steps, graph nodes or None).  Barrier = stream synchronisation on both sides.
    The steps of these workloads are 0.2 - 10 ms long and the parity gate before them leaves the GPU idle for a second
    or two (its clocks drop): besides the W warm-up steps, warm-up goes on until 0.3 s have passed, or a short run
    measures the clock ramp (seen: 8.8 instead of 1.05 ms per MLP step at batch 8192, at random)."""
    t0 = time.perf_counter()
    done = 0
    while done < max(warmup, 1) or time.perf_counter() - t0 < 0.3:
        step()
        done += 1
        if done % 8 == 0:
            hp.synchronize()
    hp.synchronize()
    g = None
    if use_graph:
        g = hp.Graph()
        g.capture(step)
        g.replay()
        hp.synchronize()
    run = g.replay if g is not None else step
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    hp.synchronize()
    dt = time.perf_counter() - t0
    nodes = g.nodes if g is not None else None
    return dt, nodes, g


def _gemm_roofline(lib, step, hp, n=5):
    """Dominant GEMM family of `n` eager steps: algorithmic 2MNK of its launches / HIP-event time around them."""
    ms, fl, cnt = (ctypes.c_double * 5)(), (ctypes.c_double * 5)(), (ctypes.c_int64 * 5)()
    hp.synchronize()
    lib.call("pdn_gemm_prof_enable", 1)
    for _ in range(n):
        step()
    hp.synchronize()
    lib.call("pdn_gemm_prof_enable", 0)
    lib.call("pdn_gemm_prof_collect_families", ms, fl, cnt)
    names = ("gemm_f32_mfma_kernel", "gemm_tn_stream_dma_kernel", "gemm_rowres_kernel", "gemm_outres_kernel",
             "gemm_outres_tn_kernel")
    i = max(range(5), key=lambda j: ms[j])
    tot_ms, tot_fl = sum(ms), sum(fl)
    ach = fl[i] / (ms[i] * 1e-3) / 1e12 if ms[i] > 0 else 0.0
    return {"bound": "mfma", "kernel": names[i], "achieved": ach, "peak": PEAK_FP32_MFMA / 1e12, "unit": "TFLOP/s",
            "frac": ach / (PEAK_FP32_MFMA / 1e12), "traffic": None, "launches_per_step": cnt[i] / n,
            "avg_launch_us": 1e3 * ms[i] / max(cnt[i], 1), "algorithmic_flop_per_launch": fl[i] / max(cnt[i], 1),
            "all_gemm": {"achieved": tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0,
                         "ms_per_step": tot_ms / n}}


def _launch_floor_us(hp, lib, nodes=64, replays=20):
    """What ONE dependent kernel launch costs when there is nothing to compute: a chain of `nodes` one-element fills on the
    compute stream, captured and replayed as a hipGraph (how the launch-bound steps here are issued).  A step of n
    launches cannot take less than n x this, whatever its kernels do -- the roof of a latency-bound line."""
    buf = hp.zeros((64,), np.float32)
    shape, strides = (ctypes.c_int64 * 1)(1), (ctypes.c_int64 * 1)(1)

    def chain():
        for _ in range(nodes):
            lib.call("pdn_fill", 0, 1.0, 1, shape, buf._ptr, strides, hp.stream())
    chain()
    hp.synchronize()
    g = hp.Graph()
    g.capture(chain)
    g.replay()
    hp.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        g.replay()
    hp.synchronize()
    us = (time.perf_counter() - t0) / (replays * nodes) * 1e6
    g.destroy()
    return us


def _latency_roof(hp, lib, launches, step_us, what):
    """Roofline block of a step that is bound by its chain of dependent launches, not by a throughput roof."""
    floor = _launch_floor_us(hp, lib)
    return {"bound": "latency", "what": what, "launches_per_step": launches, "per_launch_floor_us": floor,
            "floor_us_per_step": launches * floor, "measured_us_per_step": step_us, "unit": "us",
            "achieved": step_us, "peak": launches * floor, "frac": launches * floor / max(step_us, 1e-9), "traffic": None,
            "note": "frac = (launches x measured cost of an empty dependent launch) / measured step: 1.0 would be a step "
                    "whose kernels cost nothing; a throughput roof (HBM, MFMA) says nothing about these lines"}


def _event_time_us(hp, fn, n=10):
    fn()
    hp.synchronize()
    with hp.Timer() as t:
        for _ in range(n):
            fn()
    return max(t.ms / n * 1e3, 1e-3)


def run_train(args, which):
    import pydynet_amd as pdn
    import pydynet_amd.nn.functional as F
    from pydynet_amd import hipnp as hp, _lib
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    lib = _lib.lib()
    hp.set_device(0)
    MLP, LeNet = _models()
    B = args.batch if args.batch else 256                     # the reference's default batch (mnist.py:106-109)
    use_graph = (not args.no_graph) and B <= 1024
    Graph.clear()
    np.random.seed(42)
    net = (MLP if which == "mlp" else LeNet)().to("hip:0")
    opt = Adam(net.parameters(), lr=1e-4)
    opt.flatten_grads()                                       # one flat gradient buffer: zero_grad is a single fill (as bench.py)
    shape = (1, 28, 28) if which == "mlp" else (3, 32, 32)
    X = pdn.Tensor(np.random.rand(B, *shape).astype(np.float32), device="hip:0")
    y = pdn.Tensor(np.random.randint(0, 10, B), dtype=np.int64, device="hip:0")

    def step():
        loss = F.cross_entropy_loss(net(X), y)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss

    # parity before timing: the first steps against the oracle on the same inputs and initial weights
    gate = _train_gate(which, net, X, y, step)
    dt, nodes, g = _time_steps(hp, step, args.steps, args.warmup, use_graph)
    if g is not None:
        g.destroy()
    flop = MLP_FLOP if which == "mlp" else LENET_FLOP
    value = B * args.steps / dt
    roof = _gemm_roofline(lib, step, hp)
    if which == "lenet":
        roof = dict(_conv_roofline(lib, hp, B), gemm=roof)
    if use_graph and nodes:
        # launch-bound at this batch: the informative roof is the launch chain (the GEMM block stays beside it)
        roof = dict(_latency_roof(hp, lib, int(nodes), 1e6 * dt / args.steps, "whole step replayed as one hipGraph"),
                    dominant_gemm=roof)
    out = {
        "metric": f"training-step samples/sec ({'3-layer MLP 784-1024-1024-10' if which == 'mlp' else 'LeNet, 3x32x32 inputs'})",
        "value": value, "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("examples/pydynet/mnist.py MLP (Flatten, Linear 784-1024-1024-10, ReLU)" if which == "mlp" else
                                "examples/pydynet/mnist.py LeNet shape-adapted to 3x32x32 (Conv 3-20-50, k 3, pad 1; 2x2 max-pool; FC 3200-500-10)")
                               + ", cross entropy, Adam lr 1e-4, fwd+bwd+Adam",
                   "per_gpu_batch": B, "global_batch": B, "parallelism": "dp1",
                   "step_launch": f"hipGraph replay ({nodes} nodes)" if use_graph else "eager launches"},
        "algorithmic_tflops": flop * value / 1e12,
        "model_flops_frac_of_fp32_mfma_peak": flop * value / PEAK_FP32_MFMA,
        "parity_gate": gate, "roofline": roof,
    }
    if which == "lenet":
        out["conv_algorithmic_GBps_over_step"] = 3 * LENET_CONV_BYTES_FWD * value / 1e9
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = _cpu_train(which)
    return out


def _train_gate(which, net, X, y, step, rtol=1e-4):
    """Three optimisation steps against the oracle (NumPy restatement of the reference's op sequence) started from
    the same weights on the same batch: losses within 1e-4 relative.  Raises instead of timing a wrong path."""
    from oracle import llama as ollama, nn as onn, tape as otape
    from pydynet_amd.optim import Adam
    otape.reset_tape()
    ref = ollama.MLP() if which == "mlp" else ollama.LeNet(3, 32)
    mine = dict(net.named_parameters())
    names = (["layer1", "layer2", "layer3"] if which == "mlp" else ["conv1", "conv2", "fc1", "fc2"])
    refs = ([ref.l1, ref.l2, ref.l3] if which == "mlp" else [ref.c1, ref.c2, ref.f1, ref.f2])
    for n, r in zip(names, refs):
        r["weight"].value[...] = mine[n + ".weight"].numpy()
        r["bias"].value[...] = mine[n + ".bias"].numpy().reshape(r["bias"].value.shape)
    nb = min(X.shape[0], 64)                                 # a bounded sample keeps the oracle's share to seconds
    import pydynet_amd as pdn
    import pydynet_amd.nn.functional as F
    xs, ys = X.numpy()[:nb], y.numpy()[:nb]
    Xs, Ys = pdn.Tensor(xs, device="hip:0"), pdn.Tensor(ys, dtype=np.int64, device="hip:0")
    keep = {n: p.numpy() for n, p in mine.items()}
    popt = Adam(net.parameters(), lr=1e-4)
    ropt = onn.Adam(ref.parameters(), lr=1e-4)
    worst = 0.0
    from bench import kernel_counters
    from pydynet_amd import _lib
    kernel_counters(_lib.lib(), reset=True)
    from pydynet_amd.core import fused
    fused_path = X.shape[0] >= fused.linear_relu.min_rows      # the timed batch takes Linear + ReLU as one product:
    saved_rows, fused.linear_relu.min_rows = fused.linear_relu.min_rows, (1 if fused_path else fused.linear_relu.min_rows)
    for _ in range(3):                                        # ... then so does the (smaller) gate batch
        loss = F.cross_entropy_loss(net(Xs), Ys)
        popt.zero_grad(); loss.backward(); popt.step()
        want = ollama.train_step(ref, otape.Var(xs), otape.Var(ys, dtype=np.int64), ropt)
        err = abs(loss.item() - want) / abs(want)
        worst = max(worst, err)
        if err > rtol:
            raise SystemExit(f"bench.py --config {which}: parity gate FAILED: loss {loss.item()!r} vs oracle {want!r}")
    for n, p in mine.items():                                # timing starts from the initial weights again
        p.data[...] = keep[n]
    otape.reset_tape()
    fused.linear_relu.min_rows = saved_rows
    launched = {k: v for k, v in kernel_counters(_lib.lib(), reset=True).items() if v}
    if which == "mlp" and fused_path and (launched.get("linear_relu_fwd", 0) < 6 or launched.get("linear_dx_masked", 0) < 6):
        # Linear -> ReLU as one product, the relu gradient applied in the consumer's input-gradient product: a dispatch
        # that falls back to separate relu passes would keep the losses right and only show as a slower number
        raise SystemExit(f"bench.py --config mlp: the fused Linear + ReLU products were not launched ({launched})")
    if which == "lenet" and (launched.get("conv_quad_fwd", 0) < 6 or launched.get("conv_quad_dgrad", 0) < 3 or
                             launched.get("conv_quad_wgrad", 0) < 6):
        # both layers' conv + relu + pool forwards and weight gradients, the second layer's data gradient, three steps:
        # the kernels of csrc/conv_quad.hip the roofline block below prices (a fall-back to conv_direct.hip is 30 % slower)
        raise SystemExit(f"bench.py --config lenet: the conv_quad.hip kernels were not launched ({launched})")
    return {"steps": 3, "batch": nb, "worst_loss_rel_err": worst, "rtol": rtol, "against": "oracle (NumPy port of the reference)",
            "kernel_launches": launched}


def _cpu_train(which, budget=12.0):
    from oracle import llama as ollama, nn as onn, tape as otape
    otape.reset_tape()
    np.random.seed(42)
    ref = ollama.MLP() if which == "mlp" else ollama.LeNet(3, 32)
    B = 256
    x = otape.Var(np.random.rand(B, *((1, 28, 28) if which == "mlp" else (3, 32, 32))).astype(np.float32))
    y = otape.Var(np.random.randint(0, 10, B), dtype=np.int64)
    opt = onn.Adam(ref.parameters(), lr=1e-4)
    n, dt = _timed_cpu(lambda: ollama.train_step(ref, x, y, opt), budget, 20)
    otape.reset_tape()
    return {"value": B * n / dt, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} steps at batch {B} of the oracle (NumPy / BLAS default threads), after 1 warm-up step"}


def _conv_pmc():
    """HBM bytes per dispatch of the convolution kernels from the newest committed counter summary of the LeNet step
    (profiles/r*_pmc_lenet_b4096.json: tools/pmc_cmd.sh + tools/stamp_pmc.py; FETCH_SIZE x 2 -- the gfx950 correction --
    + WRITE_SIZE, KiB).  Keys: rocprofv3 kernel names (template arguments included)."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_lenet_b4096.json")))
    if not files:
        return {}, None
    raw = open(files[-1], "rb").read()
    rows = json.loads(raw)
    meta = rows.pop("_meta", {})
    stale = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import stamp_pmc
        if meta.get("kernel_sources_sha16"):
            stale = meta["kernel_sources_sha16"] != stamp_pmc.sources_sha()
    except Exception:
        pass
    out = {}
    for name, r in rows.items():
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            out[name] = (2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0
    src = {"file": os.path.relpath(files[-1], ROOT), "sha12": hashlib.sha256(raw).hexdigest()[:12], "stale": stale,
           "batch": 4096}
    return out, src


def _conv_roofline(lib, hp, B):
    """The convolution kernels the LeNet step really launches -- conv + bias + relu + 2x2 max-pool in one forward kernel,
    data / weight gradients that expand the pooled gradient through the hit map while they stage it -- each timed live
    with HIP events at the step's shapes.  Every kernel gets BOTH roofs: algorithmic FLOPs (2 O C k^2 OH OW per image)
    against the fp32-MFMA peak and algorithmic bytes (operands read once, results written once) against HBM; `bound`
    is the nearer roof (arithmetic intensity against the ridge 157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B: conv2 at 64 FLOP/B
    is MFMA-bound, conv1 at ~12 is HBM-bound).  `traffic` = counter bytes per dispatch from the committed PMC summary of
    the batch-4096 step, when the kernel name is in it."""
    rng = np.random.default_rng(0)
    out = {}
    pmc, pmc_src = _conv_pmc()
    ridge = PEAK_FP32_MFMA / PEAK_HBM

    def traffic_of(parts):                                 # kernel template + geometry arguments, as rocprofv3 prints them
        hits = [v for k, v in pmc.items() if all(p in k for p in parts)]
        return hits[0] if len(hits) == 1 and B == (pmc_src or {}).get("batch") else None

    for tag, (C, H, O) in (("conv1", (3, 32, 20)), ("conv2", (20, 16, 50))):
        sup = lib.query("pdn_conv2d_relu_pool_supported", C, H, H, O, 3, 1, 1)
        if sup & 5 != 5 or (C != 3 and not sup & 2):
            continue
        P = H // 2
        x = hp.from_numpy(rng.standard_normal((B, C, H, H), dtype=np.float32))
        w = hp.from_numpy((0.1 * rng.standard_normal((O, C, 3, 3))).astype(np.float32))
        b = hp.from_numpy(rng.standard_normal((O,), dtype=np.float32))
        pooled = hp.empty((B, O, P, P), np.float32)
        maskp = hp.empty((B, O, H * H // 32), np.int32)          # (the hit map: one bit per conv output position)
        dpool = hp.from_numpy(rng.standard_normal((B, O, P, P), dtype=np.float32))
        dx = hp.empty((B, C, H, H), np.float32)
        dw, db = hp.empty((O, C, 3, 3), np.float32), hp.empty((O,), np.float32)
        wsb = lib.query("pdn_conv2d_bwd_weight_workspace_bytes", B, C, H, H, O, 3, 1, 1)
        flop = 2.0 * B * O * C * 9 * H * H
        xb, pb, mb = 4.0 * B * C * H * H, 4.0 * B * O * P * P, 4.0 * B * O * H * H / 32
        lib.call("pdn_conv2d_relu_pool_fwd_f32", x._ptr, w._ptr, b._ptr, pooled._ptr, maskp._ptr, B, C, H, H, O, 3, 1, 1,
                 hp.stream())
        cases = [("fwd_relu_pool", lambda: lib.call("pdn_conv2d_relu_pool_fwd_f32", x._ptr, w._ptr, b._ptr, pooled._ptr,
                                                    maskp._ptr, B, C, H, H, O, 3, 1, 1, hp.stream()), xb + pb + mb)]
        if C != 3:                                            # the first layer's input has no gradient
            cases.append(("bwd_data", lambda: lib.call("pdn_conv2d_relu_pool_bwd_data_f32", dpool._ptr, maskp._ptr, w._ptr,
                                                       dx._ptr, B, C, H, H, O, 3, 1, 1, hp.stream()), pb + mb + xb))

        def wgrad():
            ws, n = hp.workspace(wsb)
            lib.call("pdn_conv2d_relu_pool_bwd_weight_f32", x._ptr, dpool._ptr, maskp._ptr, dw._ptr, db._ptr, 0, B, C, H, H,
                     O, 3, 1, 1, ws, n, hp.stream())
        cases.append(("bwd_weight", wgrad, xb + pb + mb))
        for kind, fn, nbytes in cases:
            us = _event_time_us(hp, fn)
            tf, gb = flop / (us * 1e-6) / 1e12, nbytes / (us * 1e-6) / 1e9
            mfma_bound = flop / nbytes > ridge
            out[f"{tag}_{kind}"] = {
                "bound": "mfma" if mfma_bound else "hbm",
                "achieved": tf if mfma_bound else gb, "peak": (PEAK_FP32_MFMA / 1e12) if mfma_bound else PEAK_HBM / 1e9,
                "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": tf / (PEAK_FP32_MFMA / 1e12) if mfma_bound else gb / (PEAK_HBM / 1e9),
                "mfma_frac": tf / (PEAK_FP32_MFMA / 1e12), "hbm_frac": gb / (PEAK_HBM / 1e9),
                "flop_per_byte": flop / nbytes, "algorithmic_flop_per_launch": flop,
                "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": us, "traffic": None}
    # counter traffic: kernel names of the committed summary (template arguments identify layer and direction)
    for key, parts in (("conv1_fwd_relu_pool", ("conv_quad_fwd_kernel", "<3, 32, 32, 20>")),
                       ("conv2_fwd_relu_pool", ("conv_quad_fwd_kernel", "<20, 16, 16, 50>")),
                       ("conv2_bwd_data", ("conv_quad_dgrad_kernel", "<20, 16, 16, 50>")),
                       ("conv1_bwd_weight", ("conv_quad_wgrad_kernel", "<3, 32, 32, 20>")),
                       ("conv2_bwd_weight", ("conv_quad_wgrad_kernel", "<20, 16, 16, 50>"))):
        if key in out:
            out[key]["traffic"] = traffic_of(parts)
    dom_key = max(out, key=lambda k: out[k]["avg_launch_us"])
    dom = out[dom_key]
    return {"bound": dom["bound"], "kernel": dom_key + " (conv_quad.hip)", "achieved": dom["achieved"], "peak": dom["peak"],
            "unit": dom["unit"], "frac": dom["frac"], "traffic": dom["traffic"], "avg_launch_us": dom["avg_launch_us"],
            "algorithmic_flop_per_launch": dom["algorithmic_flop_per_launch"],
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "traffic_source": pmc_src,
            "conv_kernels": out}



# ---------------------------------------------------------------------------------------------------------------
def run_transformer(args):
    """examples/pydynet/transformer.py:53-230 at its own shape: 1-layer Transformer classifier, dim 512, 4 heads (head dim
    128), feed-forward expansion 3, 44 positions, batch 128, right-padded ids with the (B, 1, 1, L) padding mask, logistic
    loss on +-1 labels, Adam lr 5e-4 -- the third benchmark the reference's README publishes (README.md:153: 1.075 s per
    CoLA epoch on an RTX 4090 with CuPy, 17.5 s on NumPy).  The model is the example's own code written with plain
    operators (tests/models_transformer.py, the definition the reference-generated fixture was produced with); on the HIP
    device its score chain is recognised and runs on the resident head-dim-128 attention kernels.
    Gate: the tiny configuration of tests/golden/transformer_example.npz (vectors of the REAL reference) on this device."""
    import pydynet_amd as pdn
    import pydynet_amd.nn as nn
    import pydynet_amd.nn.functional as F
    from pydynet_amd import hipnp as hp, _lib
    from pydynet_amd.optim import Adam
    from pydynet_amd.core import fused
    from pydynet_amd.core.tensor import Graph
    from tests import models_transformer as mt
    lib = _lib.lib()
    hp.set_device(0)
    Transformer, loss_fn = mt.build(pdn, nn, F)
    # ---- gate: three steps of the fixture's tiny model against the reference's losses -------------------------
    d = np.load(os.path.join(ROOT, "tests", "golden", "transformer_example.npz"))
    c = mt.CFG
    ids_s, labels_s, emb_s = mt.make_inputs()
    Graph.clear()
    np.random.seed(11)
    net = Transformer(c["embed"], c["layers"], c["heads"], c["expansion"], c["vocab"], c["max_len"])
    net.word_embedding.weight.data[...] = emb_s
    net.to("hip:0")
    opt = Adam(net.parameters(), lr=c["lr"])
    net.train()
    worst = 0.0
    for s_ in range(c["steps"]):
        loss = loss_fn(net, pdn.Tensor(ids_s, device="hip:0"), pdn.Tensor(labels_s, device="hip:0"))
        opt.zero_grad(); loss.backward(); opt.step()
        want = float(d["losses"][s_])
        err = abs(loss.item() - want) / abs(want)
        worst = max(worst, err)
        if err > 1e-4:
            raise SystemExit(f"bench.py --config transformer: parity gate FAILED: loss {loss.item()!r} vs reference {want!r}")
    gate = {"steps": c["steps"], "worst_loss_rel_err": worst, "rtol": 1e-4,
            "against": "tests/golden/transformer_example.npz (the real reference, tools/gen_golden.py)"}
    # ---- the example's own shape ------------------------------------------------------------------------------
    B, L, V, D, H, E = (args.batch or 128), 44, 6000, 512, 4, 3
    rng = np.random.default_rng(0)
    ids_np = rng.integers(1, V, (B, L))
    for i, n in enumerate(rng.integers(6, L + 1, B)):
        ids_np[i, n:] = 0
    labels_np = rng.choice([-1.0, 1.0], B).astype(np.float32)
    Graph.clear()
    np.random.seed(0)
    net = Transformer(D, 1, H, E, V, L)
    net.word_embedding.weight.data[...] = (0.1 * rng.standard_normal((V, D))).astype(np.float32)
    net.to("hip:0")
    opt = Adam(net.parameters(), lr=5e-4)
    opt.flatten_grads()                                       # one flat gradient buffer: zero_grad is a single fill (as bench.py)
    net.train()
    ids, labels = pdn.Tensor(ids_np, dtype=np.int64, device="hip:0"), pdn.Tensor(labels_np, device="hip:0")
    kinds = []
    orig = fused.attention.forward_

    def spy(node, *a):
        out = orig(node, *a)
        kinds.append(node._kind)
        return out

    def step():
        loss = loss_fn(net, ids, labels)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss

    fused.attention.forward_ = spy
    try:
        step()
    finally:
        fused.attention.forward_ = orig
    if kinds != ["resident"]:
        raise SystemExit(f"bench.py --config transformer: the attention chain did not run on the resident kernels ({kinds})")
    # the whole step replayed as one hipGraph, as the other small configurations (tests/test_graph_gpu.py pins the replayed
    # steps of this model to the reference's losses); --no-graph times the eager launches: ~1.9 ms of host work for ~1.5 ms
    # of kernels at this shape
    use_graph = not args.no_graph
    dt, nodes, g = _time_steps(hp, step, args.steps, args.warmup, use_graph)
    if g is not None:
        g.destroy()
    value = B * args.steps / dt
    # algorithmic FLOPs per sample, forward x 3: four 512 x 512 projections, two 512 x 1536 feed-forward products, scores
    flop = 3 * (L * (4 * 2 * D * D + 2 * 2 * D * E * D) + 2 * 2 * L * L * D)
    launches = int(nodes or 0)
    if not launches and not os.environ.get("PDN_BENCH_NO_GRAPH_PROBE"):   # (rocprofv3 crashes on graph launches: profile runs set it)
        try:
            gcount = hp.Graph()
            gcount.capture(step)
            launches = int(gcount.nodes)
            gcount.destroy()
        except Exception:
            launches = 0
    roof = _latency_roof(hp, lib, launches, 1e6 * dt / args.steps,
                         "whole step replayed as one hipGraph" if use_graph else "eager launches of one training step") if launches else None
    out = {"metric": "training-step samples/sec (1-layer Transformer classifier, dim 512, 4 heads, 44 positions; transformer.py)",
           "value": value, "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "examples/pydynet/transformer.py: Transformer(512, 1 layer, 4 heads, expansion 3), vocab 6000, 44 positions, "
                                  "padding mask, logistic loss, Adam lr 5e-4, fwd+bwd+Adam; plain-operator model code",
                      "per_gpu_batch": B, "global_batch": B, "parallelism": "dp1", "step_launch": f"hipGraph replay ({nodes} nodes)" if use_graph else "eager launches"},
           "algorithmic_tflops": flop * value / 1e12, "model_flops_frac_of_fp32_mfma_peak": flop * value / PEAK_FP32_MFMA,
           "attention_kernel": kinds[0], "parity_gate": gate, "roofline": roof,
           "reference_published": "README.md:153: 1.075 s per CoLA epoch (CuPy, RTX 4090), 17.5 s (NumPy) -- an epoch is 54 training "
                                  "batches of 128 plus forward-only accuracy passes; context, not the same unit"}
    return out

# ---------------------------------------------------------------------------------------------------------------
def run_gru(args):
    """examples/pydynet/ts_prediction.py: GRU(1 -> 32) over T = 40 steps + Linear head, MSE, Adam."""
    import pydynet_amd as pdn
    import pydynet_amd.nn as nn
    import pydynet_amd.nn.functional as F
    from pydynet_amd import hipnp as hp, _lib
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    lib = _lib.lib()
    hp.set_device(0)
    T_, Hd = 40, 32
    B = args.batch if args.batch else 1568
    Graph.clear()
    np.random.seed(0)
    gru = nn.GRU(1, Hd, dtype=np.float32).to("hip:0")
    head = nn.Linear(Hd, 1, dtype=np.float32).to("hip:0")
    params = list(gru.parameters()) + list(head.parameters())
    opt = Adam(params, lr=1e-3)
    xs_np, ys_np = np.random.rand(T_, B, 1).astype(np.float32), np.random.rand(B, 1).astype(np.float32)
    xs, ys = pdn.Tensor(xs_np, device="hip:0"), pdn.Tensor(ys_np, device="hip:0")

    def step():
        out, hn = gru(xs)
        loss = F.mse_loss(head(hn[0]), ys)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss

    gate = _gru_gate(gru, head, xs_np, ys_np, Hd)
    # the whole step replayed as one hipGraph, as the other small configurations; --no-graph times the eager launches
    use_graph = not args.no_graph
    dt, nodes, g = _time_steps(hp, step, args.steps, args.warmup, use_graph)
    if g is not None:
        g.destroy()
    value = B * args.steps / dt
    # dominant kernel: the persistent sequence kernel (hidden state in MFMA accumulators), forward launch
    rng = np.random.default_rng(0)
    g1, g2 = hp.from_numpy(rng.standard_normal((T_, B, 2 * Hd), dtype=np.float32)), hp.from_numpy(rng.standard_normal((T_, B, Hd), dtype=np.float32))
    h0 = hp.zeros((B, Hd), np.float32)
    wh1, wh2 = hp.from_numpy(0.1 * rng.standard_normal((Hd, 2 * Hd), dtype=np.float32)), hp.from_numpy(0.1 * rng.standard_normal((Hd, Hd), dtype=np.float32))
    z, r, rh, nn_, outb = (hp.empty((T_, B, Hd), np.float32) for _ in range(5))
    us = _event_time_us(hp, lambda: lib.call("pdn_gru_seq_fwd_f32", g1._ptr, g2._ptr, h0._ptr, wh1._ptr, wh2._ptr, z._ptr, r._ptr,
                                             rh._ptr, nn_._ptr, outb._ptr, T_, B, Hd, hp.stream()))
    nbytes = 4.0 * T_ * B * Hd * (3 + 5)                      # reads the hoisted projections (3 H), writes z, r, rh, n, out
    flops = 2.0 * T_ * B * Hd * 3 * Hd
    seq = {"kernel": "gru_seq_fwd_kernel (gru_seq.hip)", "avg_launch_us": us, "algorithmic_bytes_per_launch": nbytes,
           "achieved_GBps": nbytes / (us * 1e-6) / 1e9,
           "note": f"a {T_}-step recurrence ({flops / (us * 1e-6) / 1e12:.2f} TFLOP/s of recurrent MFMA work): {T_} dependent "
                   "steps of 48 MFMAs inside ONE launch -- neither an HBM nor an MFMA roof applies"}
    launches = int(nodes or 0)
    if not launches:
        try:
            gcount = hp.Graph()
            gcount.capture(step)
            launches = int(gcount.nodes)
            gcount.destroy()
        except Exception:
            launches = 0
    roof = dict(_latency_roof(hp, lib, launches, 1e6 * dt / args.steps,
                              "whole step replayed as one hipGraph" if use_graph else "eager launches of one training step"),
                sequence_kernel=seq)
    out = {"metric": "training-step sequences/sec (GRU 1->32, T=40, ts_prediction.py)", "value": value, "unit": "sequences/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "examples/pydynet/ts_prediction.py GRU(1->32), T = 40, Linear(32,1) head, MSE, Adam lr 1e-3, fwd+bwd+Adam",
                      "per_gpu_batch": B, "global_batch": B, "parallelism": "dp1",
                      "step_launch": f"hipGraph replay ({nodes} nodes)" if use_graph else "eager launches"},
           "parity_gate": gate, "roofline": roof}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = _cpu_gru(T_, Hd)
    return out


def _oracle_gru(gru, head, Hd):
    from oracle import nn as onn
    names = dict(gru.named_parameters())
    hn = dict(head.named_parameters())
    p = onn.gru_cell_params(1, Hd, True, np.float32)
    got = {k: v for k, v in names.items()}
    # the product registers the cell's tensors under the reference's names (rnn.py:546-554)
    for key in list(p):
        src = [n for n in got if n.endswith(key)]
        assert len(src) == 1, (key, list(got))
        p[key].value[...] = got[src[0]].numpy().reshape(p[key].value.shape)
    hw = onn.linear_params(Hd, 1, True, np.float32)
    hw["weight"].value[...] = hn["weight"].numpy()
    hw["bias"].value[...] = hn["bias"].numpy().reshape(hw["bias"].value.shape)
    return p, hw


def _gru_gate(gru, head, xs, ys, Hd, rtol=1e-4):
    from oracle import nn as onn, tape as otape
    import pydynet_amd as pdn
    import pydynet_amd.nn.functional as F
    otape.reset_tape()
    p, hw = _oracle_gru(gru, head, Hd)
    nb = 64
    x, y = xs[:, :nb], ys[:nb]
    out, hn = gru(pdn.Tensor(x, device="hip:0"))
    loss = F.mse_loss(head(hn[0]), pdn.Tensor(y, device="hip:0"))
    loss.backward()
    _, h = onn.gru_sequence(p, otape.Var(x), otape.Var(np.zeros((nb, Hd), np.float32)))
    ref = onn.mse_loss(onn.linear(h, hw["weight"], hw["bias"]), otape.Var(y))
    ref.backward()
    err = abs(loss.item() - ref.item()) / abs(ref.item())
    worst = 0.0
    names = dict(gru.named_parameters())
    for key in p:
        g = [v for n, v in names.items() if n.endswith(key)][0].grad.get().reshape(p[key].grad.shape)
        worst = max(worst, float(np.abs(g - p[key].grad).max() / max(np.abs(p[key].grad).max(), 1e-30)))
    for q in list(gru.parameters()) + list(head.parameters()):
        q.zero_grad()
    otape.reset_tape()
    if err > rtol or worst > rtol:
        raise SystemExit(f"bench.py --config gru: parity gate FAILED (loss rel err {err:.2e}, worst grad rel err {worst:.2e})")
    return {"batch": nb, "loss_rel_err": err, "worst_grad_rel_err": worst, "rtol": rtol,
            "against": "oracle (NumPy port of the reference's GRU cell loop)"}


def _cpu_gru(T_, Hd, budget=10.0):
    from oracle import nn as onn, tape as otape
    otape.reset_tape()
    np.random.seed(0)
    B = 256
    p = onn.gru_cell_params(1, Hd, True, np.float32)
    hw = onn.linear_params(Hd, 1, True, np.float32)
    params = list(p.values()) + [hw["weight"], hw["bias"]]
    opt = onn.Adam(params, lr=1e-3)
    x, y = otape.Var(np.random.rand(T_, B, 1).astype(np.float32)), otape.Var(np.random.rand(B, 1).astype(np.float32))

    def step():
        _, h = onn.gru_sequence(p, x, otape.Var(np.zeros((B, Hd), np.float32)))
        loss = onn.mse_loss(onn.linear(h, hw["weight"], hw["bias"]), y)
        opt.zero_grad(); loss.backward(); opt.step()
    n, dt = _timed_cpu(step, budget, 30)
    otape.reset_tape()
    return {"value": B * n / dt, "unit": "sequences/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} steps at batch {B} (T = {T_}) of the oracle, after 1 warm-up step"}


# ---------------------------------------------------------------------------------------------------------------

def _decode_step_launch(model):
    """How a decode step was issued, from the plan the model actually used (graphs captured per key-range count)."""
    st = getattr(model, "_decode_st", None) or {}
    graphs = st.get("graphs") or {}
    if graphs and not st.get("nograph"):
        nodes = sorted({int(g.nodes) for g in graphs.values() if g})
        return f"hipGraph replay, {'/'.join(str(n) for n in nodes)} kernel nodes per step ({len(graphs)} captured range count{'s' if len(graphs) != 1 else ''})"
    return "eager launches"

def run_decode(args):
    """Greedy KV-cache decoding of the 6-layer Llama at batch 1 (llm/llama/infer.py:46-63 prints tokens/s this
    way; the reference's README quotes 300 tok/s): a "step" is one generated token, host read-back included."""
    import pydynet_amd as pdn
    from pydynet_amd import hipnp as hp, _lib
    from pydynet_amd.llm.llama import Llama
    from pydynet_amd.core.tensor import Graph
    lib = _lib.lib()
    hp.set_device(0)
    V, D, H, F_, LAYERS = 32000, 288, 6, 768, 6
    B = args.batch if args.batch else 1
    prompt_len = 8
    Graph.clear()
    np.random.seed(0)
    model = Llama(V, D, H, F_, 1024, B, LAYERS, np.float32)
    model.tok_embedding.weight.data[...] = (0.02 * np.random.randn(V, D)).astype(np.float32)
    host_w = {n: p.numpy() for n, p in model.named_parameters()}
    model = model.to("hip:0")
    model.eval()
    ids = np.random.randint(0, V, (B, prompt_len))
    total = prompt_len + args.warmup + args.steps + 1
    if total > 1024:
        raise SystemExit("bench.py --config decode: prompt + warmup + steps must fit max_seq_len 1024")
    toks, gate = [], None
    try:
        with pdn.no_grad():
            n, t0 = 0, None
            for tok in model.generate(ids, total):
                toks.append(tok.numpy())                      # host read-back per token, as infer.py does
                n += 1
                if n == 1 + args.warmup:
                    hp.synchronize()
                    t0 = time.perf_counter()                  # the prompt pass and the warm-up tokens are not timed
            hp.synchronize()
            dt = time.perf_counter() - t0
    finally:
        model.train(True)
        pdn.autograd.set_grad_enabled(True)
    produced = n - 1 - args.warmup
    assert produced == args.steps, (produced, args.steps)
    toks = np.concatenate(toks, axis=1)
    gate = _decode_gate(host_w, ids, toks, (V, D, H, F_, LAYERS))
    value = B * produced / dt
    # algorithmic bytes per token: every weight matrix read once (the embedding contributes B rows), + the KV cache
    wbytes = 4.0 * (LAYERS * (4 * D * D + 3 * D * F_ + 2 * D) + D + D * V + V + B * D)
    # dominant kernel: the vocabulary projection (norm + skinny product), timed live
    x = hp.from_numpy(np.random.default_rng(0).standard_normal((B, D), dtype=np.float32))
    lg = hp.empty((B, V), np.float32)
    head = model.lm_head
    us = _event_time_us(hp, lambda: lib.call("pdn_decode_gemv_f32", x._ptr, D, model.norm.weight.data._ptr, 1e-6,
                                             head.weight.data._ptr, V, V, 0, head.bias.data._ptr, None, 0, lg._ptr, V,
                                             B, D, V, 0, 0, 0, None, None, hp.stream()), n=50)
    hb = 4.0 * (D * V + V + B * D + B * V)
    st = getattr(model, "_decode_st", None) or {}
    graphs = st.get("graphs") or {}
    per_token = max([int(g.nodes) for g in graphs.values() if g] or [0])
    roof = dict(_latency_roof(hp, lib, per_token, 1e6 * dt / max(produced, 1),
                              "one token = a chain of dependent launches replayed as one hipGraph"),
                vocabulary_projection={"kernel": "decode_gemv_kernel<64, 1, 18, true> (lm_head, decode.hip)", "avg_launch_us": us,
                                       "algorithmic_bytes_per_launch": hb, "achieved_GBps": hb / (us * 1e-6) / 1e9},
                whole_step={"algorithmic_weight_bytes_per_token": wbytes, "achieved_GBps": wbytes * value / B / 1e9,
                            "frac_of_hbm_peak": wbytes * value / B / PEAK_HBM,
                            "note": "the 61 MB of weights sit in the 256 MiB Infinity Cache after the first token; a "
                                    "token is 2 launches per block + 2 = 14 dependent launches replayed as one hipGraph "
                                    "(latency chain: DESIGN.md 4.10, profiles/*_decode_trace.txt), the next step queued "
                                    "while the host polls the mapped mailbox slot the pick kernel stored the token into"})
    out = {"metric": "greedy decode tokens/sec (6L Llama3, KV cache, batch 1)", "value": value, "unit": "tokens/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / produced,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "llm/llama 6-layer Llama3 (dim 288, 6 heads, ffn 768, vocab 32000) greedy generate with KV cache, "
                                  "random init, one token read back to the host per step (infer.py:46-63)",
                      "batch": B, "prompt_len": prompt_len, "parallelism": "dp1",
                      "step_launch": _decode_step_launch(model)},
           "parity_gate": gate, "roofline": roof}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = _cpu_decode(host_w, ids, (V, D, H, F_, LAYERS))
    return out


def _oracle_llama(host_w, cfg, B):
    from oracle import llama as ollama
    V, D, H, F_, LAYERS = cfg
    m = ollama.Llama(V, D, H, F_, 1024, B, LAYERS, np.float32)
    for k, v in host_w.items():
        if k in m.params:
            m.params[k].value[...] = v
    m.reset_cache(B, 1024)
    return m


def _decode_gate(host_w, ids, toks, cfg, n_check=24):
    """The first tokens against the oracle's KV-cache generate on the same weights and prompt.  Greedy decoding
    amplifies round-off at near-ties, so a token may differ only where the oracle's top-2 logit margin is below
    1e-4 of the logit scale; checking stops there (everything after depends on that pick)."""
    m = _oracle_llama(host_w, cfg, ids.shape[0])
    k = 0
    for (t, lg), mine in zip(m.generate(ids, ids.shape[1] + n_check), toks.T):
        if not np.array_equal(t[:, 0], mine):
            top = np.sort(lg[:, -1, :], axis=-1)[:, -2:]
            margin = float((top[:, 1] - top[:, 0]).min())
            if margin > 1e-4 * float(np.abs(lg).max()):
                raise SystemExit(f"bench.py --config decode: parity gate FAILED at token {k}: {mine} vs oracle {t[:, 0]} "
                                 f"(top-2 margin {margin:.3e})")
            break
        k += 1
    return {"tokens_equal_to_oracle": k, "checked": n_check, "against": "oracle KV-cache generate (pinned to the reference's generate.npz)"}


def _cpu_decode(host_w, ids, cfg, budget=10.0):
    m = _oracle_llama(host_w, cfg, ids.shape[0])
    it = m.generate(ids, 1024)
    next(it)                                                  # prompt pass (fills the cache): not timed, as on the GPU
    t0, n = time.perf_counter(), 0
    while n < 4 or (time.perf_counter() - t0 < budget and n < 200):
        next(it)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": ids.shape[0] * n / dt, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{n} greedy tokens at batch {ids.shape[0]} of the oracle's KV-cache generate, after the prompt pass"}


def run(args):
    if args.gpus != 1:
        raise SystemExit("bench.py: --config mlp / lenet / gru / decode are single-GPU lines (the data-parallel path is --config llama)")
    out = {"mlp": lambda: run_train(args, "mlp"), "lenet": lambda: run_train(args, "lenet"),
           "gru": lambda: run_gru(args), "decode": lambda: run_decode(args),
           "transformer": lambda: run_transformer(args)}[args.config]()
    from pydynet_amd import hipnp
    out["memory"] = hipnp.memory_stats()
    print(json.dumps(out), flush=True)
