"""Builds libmuninn_hip.so (gfx950) in-tree with hipcc.  No JIT cache, no torch extension machinery:
the product is a plain C-ABI shared library (include/muninn_hip.h)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmuninn_hip.so")
SOURCES = ["mn_kernels.hip", "mn_build.hip", "mn_seq.hip", "mn_spec.hip", "mn_index.hip", "mn_brute.hip", "mn_graph.hip", "mn_n2v.hip", "mn_comm.hip", "mn_shards.hip", "mn_graph_algo.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "muninn_hip.h")]
# -ffp-contract=off: the reference's distance loops use separate mul/add (src/vec_math.c:85,106);
# the wave-order kernels call fmaf explicitly where fusion is intended.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-value", "-ldl"]


def _hipcc() -> str:
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        p = shutil.which(c)
        if p:
            return p
    raise RuntimeError("hipcc not found: cannot build libmuninn_hip.so")


OBJ = os.path.join(HERE, "csrc", "_obj")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS + ["exports.map"]] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False) -> str:
    """One object per translation unit, compiled side by side (the kernels instantiate many templates: ~3 minutes serially,
    under a minute on 8 cores), then one link.  No device code crosses a translation unit, so plain -c objects suffice."""
    if not (force or stale()):
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OBJ, exist_ok=True)
    cc = _hipcc()
    cflags = [f for f in FLAGS if f not in ("-shared", "-ldl")]
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + [os.path.basename(__file__)] if os.path.exists(os.path.join(CSRC, h)))
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))

    def one(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        sp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), hdr_t):
            return obj
        subprocess.run([cc] + cflags + ["-c", "-o", obj, sp], check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, SOURCES))
    # exports.map: the C-ABI (mn_*) is the only thing this library exports
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
                    "-o", LIB] + objs + ["-ldl"], check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
