"""Data-parallel layer on CPU with the gloo backend, world sizes 2, 4 and 8 (the N>1 path of bench.py).

Contract: an N-rank step on N shards == a 1-rank step on the concatenated batch.
The embedding gradient is a scatter-ASSIGN in the reference; the wrapper keeps that across ranks
(an all-reduce(MAX) picks the rank holding the last occurrence of every token id), so the contract
holds with duplicate token ids too -- both cases are run.
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


def _data(dups=False, nseq=2):
    rng = np.random.default_rng(5)
    ids = rng.permutation(64)[:32].reshape(2, 16)
    if nseq > 2:                                   # (more tokens than ids: some repeat even without `dups`)
        ids = np.stack([rng.permutation(64)[:16] for _ in range(nseq)])
    if dups:
        # token ids repeated inside a shard AND across shards: the embedding gradient is a scatter-ASSIGN
        # (last occurrence of the concatenated batch wins, tensor.py:937-940), which a plain sum of
        # per-rank gradients would get wrong.  With 8 sequences every id below 12 is held by ~ 6 ranks.
        ids = rng.integers(0, 12, (nseq, 16))
    tgt = rng.integers(0, 64, (nseq, 16))
    return ids, tgt


def _worker(rank, world, port, out_dir, bucket_mb, dups=False, nseq=2):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from pydynet_amd.optim import Adam
    from pydynet_amd import distributed as pdist
    from pydynet_amd.distributed import DataParallel, init_process_group, shard_batch
    from tests import gloo_comm
    gloo_comm.install()
    init_process_group("gloo")
    m = _build(seed=1234 + 7 * rank)          # ranks start DIFFERENT: the wrapper must broadcast rank 0's weights
    opt = Adam(m.parameters(), lr=1e-3)
    dp = DataParallel(m, opt, bucket_mb=bucket_mb)
    ids, tgt = _data(dups, nseq)
    lo, hi = shard_batch(nseq, rank, world)
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
    pdist.get_group().barrier()
    pdist.destroy_process_group()


def _owners_of_duplicates(ids, world):
    """How many ranks hold the most widely shared token id (the owner vote's fan-in)."""
    per = ids.shape[0] // world
    holders = {}
    for r in range(world):
        for t in np.unique(ids[r * per:(r + 1) * per]):
            holders[int(t)] = holders.get(int(t), 0) + 1
    return max(holders.values())


# world 2 as before; world 4 (two sequences per rank) and world 8 (one per rank, the node BASELINE config 5 names) on
# a global batch of 8: bucket cut over more ranks, owner vote with up to 8 contributors per token id
@pytest.mark.parametrize("world,nseq,bucket_mb,dups", [(2, 2, 0.02, False), (2, 2, 25.0, False), (2, 2, 0.02, True),
                                                       (4, 8, 0.02, True), (8, 8, 0.02, False), (8, 8, 0.02, True)])
def test_n_rank_step_equals_single_process_step(tmp_path, world, nseq, bucket_mb, dups):
    port = _free_port()
    if dups and world > 2:
        assert _owners_of_duplicates(_data(dups, nseq)[0], world) >= 3      # duplicate ids across >= 3 ranks
    mp.spawn(_worker, args=(world, port, str(tmp_path), bucket_mb, dups, nseq), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    r0, r1 = ranks[0], ranks[-1]
    for r in ranks[1:-1]:
        for k in r0.files:
            if k != "losses":
                assert np.array_equal(r0[k], r[k]), k
    # single process on the concatenated batch
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    m = _build()
    opt = Adam(m.parameters(), lr=1e-3)
    ids, tgt = _data(dups, nseq)
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
        # two Adam steps: entries whose gradient sits at round-off level may take a different step
        # direction (u = lr * g / (|g| + eps)); everything else agrees to 1e-5
        err = np.abs(r0["p/" + n] - p.data)
        bad = err > 1e-7 + 1e-5 * np.abs(p.data)
        assert bad.sum() <= max(1, p.size // 500) and err.max() <= 2 * 1e-3 * 2, (n, int(bad.sum()), float(err.max()))
    assert abs(np.mean([r["losses"][0] for r in ranks]) - ref_losses[0]) < 1e-6


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


def _data_hip(dups=False, nseq=2):
    rng = np.random.default_rng(9)
    ids = rng.permutation(128)[:64].reshape(2, 32)
    if nseq > 2:
        ids = np.stack([rng.permutation(128)[:32] for _ in range(nseq)])
    if dups:
        ids = rng.integers(0, 20, (nseq, 32))
    tgt = rng.integers(0, 128, (nseq, 32))
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


def _worker_hip(rank, world, port, out_dir, dups=False, nseq=2):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from tests import abi_emulator
    abi_emulator.install(_Patch())
    from pydynet_amd.optim import Adam
    from pydynet_amd.core import fused
    from pydynet_amd import distributed as pdist
    from pydynet_amd.distributed import DataParallel, init_process_group, shard_batch
    # the RCCL communicator class itself (unique-id rendezvous over TCP, communication stream, events);
    # the emulator answers its pdn_comm_* calls with gloo on the host buffers
    init_process_group("rccl", 0)
    assert isinstance(pdist.get_group(), pdist.RcclComm)
    m = _build_hip(seed=77 + 5 * rank).to("hip:0")
    opt = Adam(m.parameters(), lr=1e-3)
    dp = DataParallel(m, opt, bucket_mb=0.05)
    assert len(dp.buckets) > 2
    ids, tgt = _data_hip(dups, nseq)
    lo, hi = shard_batch(nseq, rank, world)
    from pydynet_amd import _lib
    losses, grads = _hip_step_loop(m, {"opt": opt, "dp": dp}, ids[lo:hi], tgt[lo:hi].reshape(-1), world)
    assert "pdn_attention_fwd_f32" in _lib.lib().calls and "pdn_adam_multi_f32" in _lib.lib().calls
    np.savez(os.path.join(out_dir, f"hrank{rank}.npz"), losses=np.array(losses),
             **{"g/" + n: g for n, g in grads.items()},
             **{"p/" + n: p.data.get() for n, p in m.named_parameters()})
    assert "pdn_comm_allreduce_f32" in _lib.lib().calls and "pdn_comm_broadcast" in _lib.lib().calls
    pdist.get_group().barrier()
    pdist.destroy_process_group()


@pytest.mark.parametrize("world,nseq,dups", [(2, 2, False), (2, 2, True), (4, 8, True), (8, 8, True)])
def test_n_rank_step_on_emulated_hip_device(tmp_path, emulated_hip, world, nseq, dups):
    port = _free_port()
    mp.spawn(_worker_hip, args=(world, port, str(tmp_path), dups, nseq), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"hrank{r}.npz") for r in range(world)]
    r0, r1 = ranks[0], ranks[-1]
    for r in ranks[1:-1]:
        for k in r0.files:
            if k != "losses":
                assert np.array_equal(r0[k], r[k]), k
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    m = _build_hip().to("hip:0")
    opt = Adam(m.parameters(), lr=1e-3)
    ids, tgt = _data_hip(dups, nseq)
    ref_losses, ref_g = _hip_step_loop(m, {"opt": opt, "dp": None}, ids, tgt.reshape(-1), 1)
    for n, p in m.named_parameters():
        g = r0["g/" + n]
        assert np.array_equal(g, r1["g/" + n]), n
        scale = max(np.abs(ref_g[n]).max(), 1e-12)
        assert np.abs(g - ref_g[n]).max() <= 2e-5 * scale + 1e-9, n
        assert np.array_equal(r0["p/" + n], r1["p/" + n]), n
    assert abs(np.mean([r["losses"][0] for r in ranks]) - ref_losses[0]) < 2e-6
