"""bench.py's gates say what ran: the parity gate counts the fused tape nodes it built and reads
the library's per-kernel launch counters (`pdn_kernel_counters`), and REFUSES to go on when the step was not made of the
nodes / kernels the roofline block prices -- a dispatch regression to the unfused composition must not stay green and
show up only as a slower number.  Runs the real `bench.parity_gate` on the emulated C ABI (tests/abi_emulator/, which
restates the library's dispatch rules for its counters) with the full-size model of BASELINE.json config 4 at one
sequence: reference loss 11.395865 (tests/golden/llama_full.json, generated from /root/reference)."""
import numpy as np
import pytest


def _model():
    from pydynet_amd.llm.llama import Llama
    from pydynet_amd.core.tensor import Graph
    Graph.clear()
    np.random.seed(0)
    m = Llama(32000, 288, 6, 768, 1024, 1, 6, np.float32)
    m.tok_embedding.weight.data[...] = (0.02 * np.random.randn(32000, 288)).astype(np.float32)
    return m.to("hip:0")


def test_parity_gate_reports_nodes_and_kernels(emulated_hip):
    import bench
    import pydynet_amd as pdn
    rec = bench.parity_gate(_model(), "hip:0", pdn)
    assert rec["fused_nodes_built"] == {"qkv_attention": 6, "ffn_swiglu": 6, "linear_cross_entropy": 1}
    k = rec["kernel_launches"]
    assert k["attention_p_fwd"] == 6 and k["attention_p_bwd"] == 6
    assert k["lm_head_dx_sumexp"] == 1 and k["lm_head_dw_ce"] == 1 and k["rowtile_rowmax"] == 1
    assert k["rowres_chunk_epilogue"] == 18            # gate | up, dh, q | k | v of six blocks (256 tokens: the chunk kernel)
    assert abs(rec["loss"] - bench.GATE_LOSS) <= 1e-4 * bench.GATE_LOSS


@pytest.mark.parametrize("switch,match", [("ffn", "fused node ffn_swiglu"), ("qkv", "fused node qkv_attention"),
                                          ("persistent", "kernel attention_p_fwd"), ("lince", "fused node linear_cross_entropy")])
def test_parity_gate_refuses_a_fallback_path(emulated_hip, switch, match):
    """Each switch sends one part of the step down its unfused / slower path; the numbers still match the reference, the
    gate must raise all the same."""
    import bench
    import pydynet_amd as pdn
    from pydynet_amd.core import fused
    m = _model()
    cls, attr = {"ffn": (fused.ffn_swiglu, "enabled"), "qkv": (fused.qkv_attention, "enabled"),
                 "persistent": (fused.attention, "use_resident"), "lince": (fused.linear_cross_entropy, "enabled")}[switch]
    old = getattr(cls, attr)
    setattr(cls, attr, False)
    try:
        with pytest.raises(SystemExit, match=match):
            bench.parity_gate(m, "hip:0", pdn)
    finally:
        setattr(cls, attr, old)


def test_require_path_wants_the_tile_piece_kernels_at_full_size():
    import bench
    ok = dict.fromkeys(bench.COUNTER_SLOTS, 0)
    ok.update(attention_p_fwd=6, attention_p_bwd=6, lm_head_dx_sumexp=1, lm_head_dw_ce=1, rowtile_swiglu_fwd=6,
              rowtile_swiglu_bwd=6, rowtile_rope=6, rowtile_rowmax=1)
    bench.require_path("batch gate", None, ok, 6, full_size=True)
    for k in ("rowtile_swiglu_bwd", "rowtile_rope", "rowtile_rowmax", "attention_p_bwd", "lm_head_dw_ce"):
        bad = dict(ok)
        bad[k] = 0
        with pytest.raises(SystemExit, match=k):
            bench.require_path("batch gate", None, bad, 6, full_size=True)
