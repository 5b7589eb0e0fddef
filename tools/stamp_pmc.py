"""stamp_pmc.py <summary.json from tools/pmc_cmd.sh> <out.json>: copies the per-kernel counter summary and records,
under "_meta", a hash of the kernel sources it was measured on (csrc/*.hip + common.h).  bench.py recomputes that hash
and reports `traffic_stale: true` next to `roofline.traffic` when the sources have changed since the counters were
collected -- the committed summary cannot silently describe kernels that no longer exist."""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha():
    d = os.path.join(ROOT, "pydynet_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    rows = json.load(open(sys.argv[1]))
    rows["_meta"] = {"kernel_sources_sha16": sources_sha(),
                     "per_gpu_batch": int(sys.argv[3]) if len(sys.argv) > 3 else 256,
                     "collected_with": "tools/pmc_cmd.sh (rocprofv3 --kernel-trace --pmc, five separate passes)"}
    json.dump(rows, open(sys.argv[2], "w"), indent=1)
    print("stamped", sys.argv[2], rows["_meta"]["kernel_sources_sha16"])
