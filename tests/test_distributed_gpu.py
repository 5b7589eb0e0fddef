"""The data-parallel wrapper on real GPUs with the RCCL communicator of the C ABI (pdn_comm_*):
  * single rank: the bucketed all-reduce of HBM gradient slices is issued from the grad-ready hooks on
    the communication stream and must leave a one-rank step unchanged;
  * two ranks (skipped unless >= 2 GPUs are visible): N-rank step == 1-rank step on the concatenated
    batch, token ids with duplicates across shards included.
(Multi-GPU scaling itself is measured by the driver; the same contract is covered on CPU with gloo
and on the emulated device in test_distributed_cpu.py.)"""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _model(seed=7):
    from pydynet_amd.llm.llama import Llama
    np.random.seed(seed)
    m = Llama(64, 96, 2, 128, 64, 2, 2, np.float32)
    m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(64, 96)).astype(np.float32)
    return m


def _steps(m, opt, dp, ids, tgt, n=3):
    losses = []
    for _ in range(n):
        m.train(True)
        opt.zero_grad()
        loss = m.loss(ids, tgt)
        loss.backward()
        if dp:
            dp.finish()
        opt.step()
        losses.append(loss.item())
    return losses


def test_single_rank_rccl_allreduce_path(hip):
    from pydynet_amd import distributed as pdist
    from pydynet_amd.optim import Adam
    from pydynet_amd.distributed import DataParallel
    from pydynet_amd.core.tensor import Graph
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    pdist.init_process_group("rccl", 0)
    try:
        rng = np.random.default_rng(3)
        ids, tgt = rng.integers(0, 64, (2, 32)), rng.integers(0, 64, (2, 32))

        def run(use_dp):
            Graph.clear()
            m = _model().to("hip:0")
            opt = Adam(m.parameters(), lr=1e-3)
            dp = DataParallel(m, opt, bucket_mb=0.05, always_reduce=True) if use_dp else None
            losses = _steps(m, opt, dp, ids, tgt)
            if dp:
                assert len(dp.buckets) > 2
            return losses, {n: p.numpy() for n, p in m.named_parameters()}

        l0, p0 = run(False)
        l1, p1 = run(True)
        assert np.allclose(l0, l1, rtol=1e-6)
        for n in p0:
            assert np.allclose(p0[n], p1[n], rtol=1e-6, atol=1e-7), n
        # collectives of the communicator itself
        g = pdist.get_group()
        a = hip.from_numpy(np.arange(8, dtype=np.float32))
        g.all_reduce(a, pdist.SUM); g.broadcast(a, 0)
        out = hip.empty((8,), np.float32)
        g.all_gather(a, out)
        g.wait()
        assert np.array_equal(out.get(), np.arange(8, dtype=np.float32))
        assert g.all_reduce_scalar(3.5, pdist.MAX) == 3.5
    finally:
        pdist.destroy_process_group()


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from pydynet_amd import distributed as pdist
    from pydynet_amd.optim import Adam
    from pydynet_amd.distributed import DataParallel, shard_batch
    pdist.init_process_group("rccl", rank)
    m = _model(seed=7 + 3 * rank).to(f"hip:{rank}")          # ranks start different: rank 0 is broadcast
    opt = Adam(m.parameters(), lr=1e-3)
    dp = DataParallel(m, opt, bucket_mb=0.05)
    rng = np.random.default_rng(3)
    ids, tgt = rng.integers(0, 20, (2, 32)), rng.integers(0, 64, (2, 32))     # duplicates across shards
    lo, hi = shard_batch(2, rank, world)
    losses = _steps(m, opt, dp, ids[lo:hi], tgt[lo:hi], n=1)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=np.array(losses),
             **{n: p.numpy() for n, p in m.named_parameters()})
    pdist.get_group().barrier()
    pdist.destroy_process_group()


def test_two_rank_rccl_step_equals_single_process_step(hip, tmp_path):
    from pydynet_amd import cuda
    if cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    from pydynet_amd.optim import Adam
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    m = _model().to("hip:0")
    opt = Adam(m.parameters(), lr=1e-3)
    rng = np.random.default_rng(3)
    ids, tgt = rng.integers(0, 20, (2, 32)), rng.integers(0, 64, (2, 32))
    ref = _steps(m, opt, None, ids, tgt, n=1)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert abs((r0["loss"][0] + r1["loss"][0]) / 2 - ref[0]) < 1e-5
    for n, p in m.named_parameters():
        assert np.array_equal(r0[n], r1[n]), n
        err = np.abs(r0[n] - p.numpy())
        assert (err > 1e-6 + 1e-4 * np.abs(p.numpy())).sum() <= max(1, p.size // 500) and err.max() <= 2e-3, n
