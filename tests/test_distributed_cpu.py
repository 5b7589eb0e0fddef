"""Data-parallel layer on CPU with the gloo backend, world_size 2 (the N>1 path of bench.py).

Contract: an N-rank step on N shards == a 1-rank step on the concatenated batch.
Token ids are a permutation (duplicate-free) because the embedding gradient is a scatter-ASSIGN
in the reference, for which shard-then-sum differs from the single-process result on duplicates.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build(seed=1234):
    import pydynet_amd as pdn  # noqa: F401
    from pydynet_amd.llm.llama import Llama
    np.random.seed(seed)
    m = Llama(64, 48, 2, 96, 64, 2, 2, np.float32)
    m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(64, 48)).astype(np.float32)
    return m


def _data():
    rng = np.random.default_rng(5)
    ids = rng.permutation(64)[:32].reshape(2, 16)
    tgt = rng.integers(0, 64, (2, 16))
    return ids, tgt


def _worker(rank, world, port, out_dir, bucket_mb):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from pydynet_amd.optim import Adam
    from pydynet_amd.distributed import DataParallel, init_process_group, shard_batch
    init_process_group("gloo")
    m = _build(seed=1234 + 7 * rank)          # ranks start DIFFERENT: the wrapper must broadcast rank 0's weights
    opt = Adam(m.parameters(), lr=1e-3)
    dp = DataParallel(m, opt, bucket_mb=bucket_mb)
    ids, tgt = _data()
    lo, hi = shard_batch(2, rank, world)
    losses = []
    for _ in range(2):
        m.train(True)
        opt.zero_grad()
        loss = m.loss(ids[lo:hi], tgt[lo:hi])
        loss.backward()
        dp.finish()
        if _ == 0:
            grads = {n: p.grad.copy() / world for n, p in m.named_parameters()}
        opt.step()
        losses.append(loss.item())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=np.array(losses),
             **{"g/" + n: g for n, g in grads.items()},
             **{"p/" + n: p.data for n, p in m.named_parameters()})
    assert len(dp.buckets) >= (2 if bucket_mb < 0.1 else 1)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb", [0.02, 25.0])
def test_two_rank_step_equals_single_process_step(tmp_path, bucket_mb):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), bucket_mb), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # single process on the concatenated batch
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    m = _build()
    opt = Adam(m.parameters(), lr=1e-3)
    ids, tgt = _data()
    ref_losses = []
    for s in range(2):
        m.train(True)
        opt.zero_grad()
        loss = m.loss(ids, tgt)
        loss.backward()
        if s == 0:
            ref_g = {n: p.grad.copy() for n, p in m.named_parameters()}
        opt.step()
        ref_losses.append(loss.item())
    for n, p in m.named_parameters():
        g = r0["g/" + n]
        assert np.array_equal(g, r1["g/" + n]), n                  # both ranks hold the same reduced gradient
        scale = max(np.abs(ref_g[n]).max(), 1e-12)
        assert np.abs(g - ref_g[n]).max() <= 1e-5 * scale + 1e-9, n
        assert np.array_equal(r0["p/" + n], r1["p/" + n]), n       # replicas stay in lock-step
        assert np.allclose(r0["p/" + n], p.data, rtol=1e-5, atol=1e-7), n
    assert abs((r0["losses"][0] + r1["losses"][0]) / 2 - ref_losses[0]) < 1e-6


def test_shard_batch_rules():
    from pydynet_amd.distributed import shard_batch
    assert [shard_batch(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    with pytest.raises(ValueError):
        shard_batch(7, 0, 2)


# ---- the same contract on the (emulated) HIP device: hipnp buffers, fused nodes, flat buckets -------
class _Patch:                       # minimal stand-in for pytest's monkeypatch inside a spawned worker
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def _build_hip(seed=77):
    """Flash-compatible shape (head_dim 48, L 32): Attention takes the fused qkv_attention node."""
    from pydynet_amd.llm.llama import Llama
    np.random.seed(seed)
    m = Llama(128, 96, 2, 128, 64, 2, 2, np.float32)
    m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(128, 96)).astype(np.float32)
    return m


def _data_hip():
    rng = np.random.default_rng(9)
    ids = rng.permutation(128)[:64].reshape(2, 32)
    tgt = rng.integers(0, 128, (2, 32))
    return ids, tgt


def _hip_step_loop(m, dp, ids, tgt, world, steps=2):
    from pydynet_amd.optim import Adam
    opt = dp["opt"]
    losses, grads = [], None
    for s in range(steps):
        m.train(True)
        opt.zero_grad()
        loss = m.loss(ids, tgt)
        loss.backward()
        if dp["dp"] is not None:
            dp["dp"].finish()
        if s == 0:
            grads = {n: p.grad.get() / world for n, p in m.named_parameters()}
        opt.step()
        losses.append(loss.item())
    return losses, grads


def _worker_hip(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tests import abi_emulator
    abi_emulator.install(_Patch())
    from pydynet_amd.optim import Adam
    from pydynet_amd.core import fused
    from pydynet_amd.distributed import DataParallel, init_process_group, shard_batch
    init_process_group("gloo")
    m = _build_hip(seed=77 + 5 * rank).to("hip:0")
    opt = Adam(m.parameters(), lr=1e-3)
    dp = DataParallel(m, opt, bucket_mb=0.05)
    assert len(dp.buckets) > 2
    ids, tgt = _data_hip()
    lo, hi = shard_batch(2, rank, world)
    from pydynet_amd import _lib
    losses, grads = _hip_step_loop(m, {"opt": opt, "dp": dp}, ids[lo:hi], tgt[lo:hi].reshape(-1), world)
    assert "pdn_attention_fwd_f32" in _lib.lib().calls and "pdn_adam_multi_f32" in _lib.lib().calls
    np.savez(os.path.join(out_dir, f"hrank{rank}.npz"), losses=np.array(losses),
             **{"g/" + n: g for n, g in grads.items()},
             **{"p/" + n: p.data.get() for n, p in m.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_on_emulated_hip_device(tmp_path, emulated_hip):
    port = _free_port()
    mp.spawn(_worker_hip, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "hrank0.npz"), np.load(tmp_path / "hrank1.npz")
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    m = _build_hip().to("hip:0")
    opt = Adam(m.parameters(), lr=1e-3)
    ids, tgt = _data_hip()
    ref_losses, ref_g = _hip_step_loop(m, {"opt": opt, "dp": None}, ids, tgt.reshape(-1), 1)
    for n, p in m.named_parameters():
        g = r0["g/" + n]
        assert np.array_equal(g, r1["g/" + n]), n
        scale = max(np.abs(ref_g[n]).max(), 1e-12)
        assert np.abs(g - ref_g[n]).max() <= 2e-5 * scale + 1e-9, n
        assert np.array_equal(r0["p/" + n], r1["p/" + n]), n
    assert abs((r0["losses"][0] + r1["losses"][0]) / 2 - ref_losses[0]) < 2e-6
