"""Build libmmdistill_hip.so (gfx950) in-tree with hipcc.  `python -m mm_distillnet_amd.build`."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmmdistill_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-fPIC", "-std=c++17", "-Wno-unused-value",
         "-Wno-unused-result"] + os.environ.get("MMD_EXTRA_HIPCC_FLAGS", "").split()


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs = []
    os.makedirs(os.path.join(PKG, "build"), exist_ok=True)
    procs = []
    hdr_t = max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")))
    for s in srcs:
        o = os.path.join(PKG, "build", os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(s), hdr_t):
            continue            # object newer than its source and every header
        procs.append((s, subprocess.Popen([hipcc, *FLAGS, "-c", s, "-o", o], stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {s}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
