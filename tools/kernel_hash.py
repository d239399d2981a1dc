"""Identity of the traversal kernel's source: bench.py only trusts PMC numbers (profiles/pmc_bench_latest.json, written by tools/pmc_summary.py)
that were collected for exactly this source; anything else is reported as stale (traffic: null)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["embree_amd/csrc/trace.hip", "embree_amd/csrc/bvh_common.h", "embree_amd/csrc/internal.h", "embree_amd/build.py"]


def trace_kernel_hash():
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(trace_kernel_hash())
