"""Throughput of the DROP-IN formulation: the full-size 6-layer Llama written with plain operators (tests/models_plain_llama.py =
what a user of the reference's own llm/llama/model.py gets) beside the fused-node model the headline is quoted on.
Measured round 6 (MI355X): batch 64: 2697 vs 4153 samples/s (0.65; the plain step is host-bound there: ~430 nodes per step);
batch 256: 3891 vs 5139 (0.76) with the attention chain recognised (core/fused/chain.py), 3361 (0.65) without; 3983 vs 5027 (0.79)
once the strided elementwise kernels divide by multiply-high (csrc/elementwise.hip: the strided kernels of a step 6.8 -> 4.3 ms under rocprofv3);
4198 vs 5118 (0.82) with the loss of model.py:239-249 taken as one linear_cross_entropy node (core/fused/chain.py);
4334 vs 5070 (0.85) with the backward of the rotary embedding as one node (chain.rope_chain);
4524 vs 5111 (0.885) with silu(gate) * up as one swiglu node (chain.swiglu_chain); 4663 vs 5018-5130 (0.91-0.93) with the rotary
embedding's operators as pending links that end in one fused.rope node (chain.rope_link); batch 64: 3346 vs 4163 (0.80).
usage: python tools/plain_llama_bench.py [batch=64] [steps=5] [plain|fused: only that model, e.g. under rocprofv3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(batch=64, steps=5, warmup=2, which=("plain", "fused")):
    import pydynet_amd as pdn
    from pydynet_amd import hipnp as hp, _lib
    from pydynet_amd.core.tensor import Graph
    from pydynet_amd.llm.llama import Llama
    from pydynet_amd.optim import Adam
    from tests.models_plain_llama import PlainLlama
    import bench
    lib = _lib.lib()
    V, D, H, F_, L, LAYERS = 32000, 288, 6, 768, 256, 6
    rng = np.random.default_rng(1)
    ids_np, tgt_np = rng.integers(0, V, (batch, L)), rng.integers(0, V, (batch * L,))
    out = {}
    for kind in which:
        Graph.clear()
        np.random.seed(0)
        m = PlainLlama(V, D, H, F_, 1024, LAYERS, np.float32) if kind == "plain" else Llama(V, D, H, F_, 1024, 1, LAYERS, np.float32)
        m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(V, D)).astype(np.float32)
        m.to("hip:0")
        opt = Adam(m.parameters(), lr=1e-4)
        ids = pdn.Tensor(ids_np, dtype=np.int64, device="hip:0")
        tgt = pdn.Tensor(tgt_np, dtype=np.int64, device="hip:0")
        m.train(True)

        def step():
            opt.zero_grad()
            loss = m.loss(ids, tgt) if kind == "fused" else m.loss(ids, tgt_np)
            loss.backward()
            opt.step()
            return loss
        for _ in range(warmup):
            loss = step()
        hp.synchronize()
        bench.kernel_counters(lib, reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        hp.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[kind] = {"samples_per_s": batch / dt, "ms_per_step": 1e3 * dt, "loss": float(loss.item()),
                     "mfma_frac": bench.FLOP_PER_SAMPLE * batch / dt / bench.PEAK_FP32_MFMA,
                     "kernel_launches_per_step": {k: v / steps for k, v in bench.kernel_counters(lib, reset=True).items() if v}}
        del m, opt, ids, tgt, step, loss
        Graph.clear()
    if "plain" in out and "fused" in out:
        out["plain_over_fused"] = out["plain"]["samples_per_s"] / out["fused"]["samples_per_s"]
    return out


if __name__ == "__main__":
    import json
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    which = (sys.argv[3],) if len(sys.argv) > 3 else ("plain", "fused")
    print(json.dumps(run(b, s, which=which), indent=1))
