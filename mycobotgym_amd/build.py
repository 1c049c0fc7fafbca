"""In-tree build of the HIP library (gfx950 only)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "mcg_hip.hip")
DEPS = sorted(glob.glob(os.path.join(_HERE, "csrc", "*"))) + [os.path.join(ROOT, "include", "mcg.h")]      # every source and header
OUT = os.path.join(_HERE, "libmycobot_hip.so")


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*, include/mcg.h): stamps counter measurements, so that bench.py can tell when
    profiles/pmc_latest.json was collected on other kernels than the ones it is running."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode()); h.update(open(d, "rb").read())
    return h.hexdigest()


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(_HERE, "csrc"), SRC, "-o", OUT]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    if verbose:
        print(r.stderr)
    return OUT
