"""Identity of the kernel sources a measurement belongs to: a hash over csrc/ (kernels, headers, host code, Makefile).
tools/pmc_summary.py stores it with the rocprofv3 counter summary, bench.py compares it with the tree it runs from and
reports counters of another build as stale instead of printing them beside its own timing."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_source_hash():
    h = hashlib.sha256()
    files = sorted(f for pat in ("*.hip", "*.h", "*.hpp", "*.cpp", "Makefile") for f in glob.glob(os.path.join(_CSRC, pat)))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
