"""The data-parallel wrapper on a real GPU with the RCCL ("nccl") backend, single rank: the
bucketed all-reduce of HBM gradient views is issued from the grad-ready hooks and must leave a
one-rank step unchanged.  (Multi-GPU scaling itself is measured by the driver; the N-rank ==
1-rank contract is covered on CPU with gloo in test_distributed_cpu.py.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_nccl_allreduce_path(hip):
    import torch.distributed as dist
    from pydynet_amd.llm.llama import Llama
    from pydynet_amd.optim import Adam
    from pydynet_amd.distributed import DataParallel
    from pydynet_amd.core.tensor import Graph
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(3)
        ids, tgt = rng.integers(0, 64, (2, 32)), rng.integers(0, 64, (2, 32))

        def run(use_dp):
            Graph.clear()
            np.random.seed(7)
            m = Llama(64, 96, 2, 128, 64, 2, 2, np.float32)
            m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(64, 96)).astype(np.float32)
            m.to("hip:0")
            opt = Adam(m.parameters(), lr=1e-3)
            dp = DataParallel(m, opt, bucket_mb=0.05, always_reduce=True) if use_dp else None
            losses = []
            for _ in range(3):
                m.train(True)
                opt.zero_grad()
                loss = m.loss(ids, tgt)
                loss.backward()
                if dp:
                    dp.finish()
                opt.step()
                losses.append(loss.item())
            if dp:
                assert len(dp.buckets) > 2
            return losses, {n: p.numpy() for n, p in m.named_parameters()}

        l0, p0 = run(False)
        l1, p1 = run(True)
        assert np.allclose(l0, l1, rtol=1e-6)
        for n in p0:
            assert np.allclose(p0[n], p1[n], rtol=1e-6, atol=1e-7), n
    finally:
        dist.destroy_process_group()
