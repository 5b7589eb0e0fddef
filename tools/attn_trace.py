"""Reads the per-wave timestamps an -DATT_TRACE build of the attention forward kernel leaves (tools/attn_trace.sh):
phase durations by query tile, the life of a workgroup, and how busy a CU is between the first start and the last end
of the workgroups it hosted.  Benchmark shape: 1536 heads, L = 256, head dim 48, causal, RoPE."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydynet_amd import hipnp as hp, _lib
hp.set_device(0)
Lb = _lib.lib()
B, H, L, hd = 256, 6, 256, 48
D = H * hd
rng = np.random.default_rng(0)
qkv = hp.from_numpy(rng.standard_normal((B * L, 3 * D), dtype=np.float32))
inv = 1.0 / (10000 ** (np.arange(0, hd, 2) / hd))
fr = np.outer(np.arange(L), inv).astype(np.float32)
C, S = hp.from_numpy(np.cos(fr)), hp.from_numpy(np.sin(fr))
o, lse = hp.empty((B, L, H, hd)), hp.empty((B, H, L))
q, k, v = qkv._ptr, qkv._ptr + 4 * D, qkv._ptr + 8 * D
st = hp.stream()


def fwd():
    Lb.call("pdn_attention_fwd_f32", q, k, v, o._ptr, lse._ptr, B, H, L, hd, 3 * D, L * 3 * D, D, L * D, 1, C._ptr, S._ptr, st)


SLOTS = 1 << 17


def dump():
    buf = np.zeros((SLOTS, 10), np.uint64)
    n = ctypes.c_uint(0)
    Lb.cdll.pdn_att_trace_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    return buf[:min(n.value, SLOTS)]


for _ in range(20):
    fwd()
hp.synchronize(); dump()
fwd(); hp.synchronize()
r = dump()
t = r[:, :7].astype(np.int64) * 0.01                      # us
tile = (r[:, 8] & 0xff).astype(int)
blk = (r[:, 8] >> 8).astype(int)
hwid = (r[:, 9] >> 32).astype(np.int64)
xcc = (r[:, 9] & 0xf).astype(int)
cu = ((hwid >> 8) & 0xf) | (((hwid >> 12) & 1) << 4) | (((hwid >> 13) & 7) << 5) | (xcc << 8)     # cu_id | sh_id | se_id | xcc
t0 = t[:, 0].min()
print(f"{len(r)} waves, kernel span {t[:, 6].max() - t0:.1f} us (first wave start -> last wave end)")
names = ["preload + staging issue/stores", "barrier wait", "S^T phase", "softmax", "P.V phase", "normalise + store"]
d = np.diff(t, axis=1)
print("phase means by tile (key tiles = tile + 1):")
for tl in range(8):
    sel = tile == tl
    print(f"  tile {tl}: " + ", ".join(f"{n} {x:.2f}" for n, x in zip(names, d[sel].mean(0))) + f"  | wave life {(t[sel, 6] - t[sel, 0]).mean():.2f}")
# per workgroup
life = []
for b_ in np.unique(blk)[:4000]:
    s_ = blk == b_
    life.append((t[s_, 0].min(), t[s_, 6].max()))
life = np.array(life)
print(f"workgroup life: mean {np.mean(life[:, 1] - life[:, 0]):.2f} us, min {np.min(life[:, 1] - life[:, 0]):.2f}, max {np.max(life[:, 1] - life[:, 0]):.2f}")
# per CU
busy, span, nw = [], [], []
for c in np.unique(cu):
    s_ = cu == c
    bs = np.unique(blk[s_])
    iv = sorted((t[blk == b_, 0].min(), t[blk == b_, 6].max()) for b_ in bs)
    span.append(iv[-1][1] - iv[0][0]); busy.append(sum(e - a for a, e in iv)); nw.append(len(bs))
    gaps = [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
print(f"{len(span)} CUs: workgroups per CU {np.mean(nw):.2f} (min {min(nw)}, max {max(nw)}), span {np.mean(span):.1f} us, "
      f"sum of workgroup lives {np.mean(busy):.1f} us")
starts = np.sort(life[:, 0]) - t0
print("workgroup start times (us), every 128th:", np.round(starts[::128], 1))
