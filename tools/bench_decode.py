"""Greedy decode throughput of the 6-layer Llama with the KV cache (reference infer.py:46-63 prints
tokens/s the same way; README.md:23 quotes 300 tok/s for the reference).  Random weights.
usage: python tools/bench_decode.py [new_tokens] [prompt_len]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as _hpsync
import pydynet_amd as pdn
from pydynet_amd.llm.llama import Llama

new_tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prompt_len = int(sys.argv[2]) if len(sys.argv) > 2 else 8
np.random.seed(0)
model = Llama(32000, 288, 6, 768, 1024, 1, 6, np.float32)
model.tok_embedding.weight.data[...] = (0.02 * np.random.randn(32000, 288)).astype(np.float32)
model = model.to("cuda")
model.eval()
ids = np.random.randint(0, 32000, (1, prompt_len))
with pdn.no_grad():
    for warm in range(2):
        n, t0 = 0, None
        for tok in model.generate(ids, prompt_len + new_tokens):
            out = tok[0].numpy().tolist()          # host read-back per token, as infer.py does
            n += 1
            if n == 1:
                _hpsync.synchronize(); t0 = time.perf_counter()   # exclude the prompt pass
        _hpsync.synchronize()
        dt = time.perf_counter() - t0
print(f"decode: {n - 1} tokens in {dt:.3f} s -> {(n - 1) / dt:.0f} tokens/s (batch 1, greedy, KV cache, fp32)")
