"""The WHOLE `bench.main()` with two ranks, without a GPU: the real `RcclComm` class, the TCP rendezvous and
`DataParallel` run on the emulated C ABI (tests/abi_emulator/: `pdn_comm_*` answered by gloo on host buffers), the
model shrunk through bench.py's module constants.  Checks what the driver's multi-GPU run relies on: rank 0 prints
exactly ONE JSON line and the other rank none, `value` is the whole job's, `per_rank_samples_per_s` has one entry per
rank, the `comm` block describes the buckets actually reduced (the embedding table alone in the last one), and the
two ranks end with identical parameters.  (BASELINE config 5 itself -- 8 x MI355X over RCCL -- needs the hardware.)"""
import io
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")                      # (the emulated library has one device)
    from tests import abi_emulator
    abi_emulator.install(_Patch())
    import bench
    bench.V, bench.D, bench.H, bench.F_, bench.L, bench.LAYERS = 192, 96, 2, 128, 32, 2
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "2", "--no-parity-gate",
                "--no-cpu-baseline"]
    captured = io.StringIO()
    real, sys.stdout = sys.stdout, captured
    model_box = {}
    from pydynet_amd import distributed as pdist
    DP = pdist.DataParallel

    class SpyDP(DP):                                       # keep a handle on the model bench.main() builds
        def __init__(self, module, *a, **k):
            k.setdefault("bucket_mb", 0.05)                # several buckets on the tiny model
            super().__init__(module, *a, **k)
            model_box["m"], model_box["dp"] = module, self
    pdist.DataParallel = SpyDP
    try:
        bench.main()
    finally:
        sys.stdout = real
        pdist.DataParallel = DP
    open(os.path.join(out_dir, f"stdout{rank}.txt"), "w").write(captured.getvalue())
    m = model_box["m"]
    np.savez(os.path.join(out_dir, f"params{rank}.npz"), **{n: p.data.get() for n, p in m.named_parameters()})
    json.dump([[int(v) for v in b] for b in model_box["dp"].buckets], open(os.path.join(out_dir, f"buckets{rank}.json"), "w"))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_whole_bench_main_n_ranks_emulated(tmp_path, world, monkeypatch):
    port = _free_port()
    monkeypatch.setenv("PDN_BENCH_DETAIL", str(tmp_path / "detail.json"))
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    out0 = [l for l in open(tmp_path / "stdout0.txt").read().splitlines() if l.strip()]
    for r in range(1, world):
        outr = [l for l in open(tmp_path / f"stdout{r}.txt").read().splitlines() if l.strip()]
        assert outr == [], outr                            # only rank 0 speaks
    assert len(out0) == 1, out0                            # ... exactly one line
    assert len(out0[0]) < 4096, len(out0[0])               # ... that fits the driver's stdout tail
    d = json.loads(out0[0])
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["per_gpu_batch"] == 2 and d["config"]["global_batch"] == 2 * world
    assert d["config"]["parallelism"] == f"dp{world}"
    assert len(d["per_rank_samples_per_s"]) == world and all(v > 0 for v in d["per_rank_samples_per_s"])
    # whole-job rate over the slowest rank's time (all_reduce_scalar(MAX) of the per-rank times)
    assert d["value"] > 0 and abs(d["value"] - world * 2 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    assert d["ms_per_step"] * 2e-3 >= max(2 * 2 / v for v in d["per_rank_samples_per_s"]) * (1 - 1e-4)
    # the whole job is as fast as its slowest rank: value <= sum of the per-rank rates
    assert d["value"] <= sum(d["per_rank_samples_per_s"]) * (1 + 1e-6)
    comm = d["comm"]
    buckets = json.load(open(tmp_path / "buckets0.json"))
    assert comm["buckets"] == len(buckets) >= 3 and len(comm["bucket_MB"]) == comm["buckets"]
    assert abs(sum(comm["bucket_MB"]) - comm["payload_MB_per_step"]) < 0.05
    assert comm["exposed_ms_per_step"] >= 0.0
    # the last bucket is the embedding table alone (the last gradient of backward): nothing waits behind it
    lo, hi, plo, phi = buckets[-1]
    assert phi - plo == 1 and hi - lo == 192 * 96
    assert np.isfinite(d["final_loss"])
    assert comm["ranks"] == world and comm["backend"] == "rccl"
    p0 = np.load(tmp_path / "params0.npz")
    for r in range(1, world):
        p1 = np.load(tmp_path / f"params{r}.npz")
        assert sorted(p0.files) == sorted(p1.files) and len(p0.files) > 10
        for n in p0.files:
            assert np.array_equal(p0[n], p1[n]), n         # same reduced gradients -> same Adam step on every rank
    full = json.load(open(tmp_path / "detail.json"))
    assert full["value"] == d["value"] and full["comm"]["bucket_MB"] == comm["bucket_MB"]


def test_rendezvous_bind_host(monkeypatch):
    """Rank 0 listens on MASTER_ADDR only when that is safe: a name that resolves to loopback HERE while the job spans
    nodes (the `127.0.1.1 <hostname>` line of a container's /etc/hosts) must not hide rank 0 from the other nodes."""
    from pydynet_amd import rendezvous
    fake = {"node0": [(2, 1, 6, "", ("127.0.1.1", 0))], "10.0.0.5": [(2, 1, 6, "", ("10.0.0.5", 0))],
            "127.0.0.1": [(2, 1, 6, "", ("127.0.0.1", 0))]}

    def gai(host, *a, **k):
        if host not in fake:
            raise socket.gaierror("unknown")
        return fake[host]
    monkeypatch.setattr(socket, "getaddrinfo", gai)
    for k in ("LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "NNODES", "PDN_RDZV_BIND_ALL"):
        monkeypatch.delenv(k, raising=False)
    assert rendezvous._bind_host("127.0.0.1", 8) == "127.0.0.1"          # one node: loopback is right
    assert rendezvous._bind_host("node0", 8) == "node0"
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert rendezvous._bind_host("node0", 16) == ""                      # two nodes, the name is loopback here: all interfaces
    assert rendezvous._bind_host("10.0.0.5", 16) == "10.0.0.5"           # a routable address: bind it
    assert rendezvous._bind_host("nowhere", 16) == ""                    # unresolvable: all interfaces
    monkeypatch.setenv("PDN_RDZV_BIND_ALL", "1")
    assert rendezvous._bind_host("10.0.0.5", 16) == ""
