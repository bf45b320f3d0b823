"""Builds scripts/_phase/libmuninn_hip.so = the product sources with -DMN_PHASE_TIMING (per-phase timers of one search's latency
chain, mn_beam.hpp).  For scripts/probe_phases.py only; the product library never carries the timers."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("mn_build", os.path.join(ROOT, "sqlite-muninn_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
out = os.path.join(ROOT, "scripts", "_phase"); os.makedirs(out, exist_ok=True)
cflags = [f for f in b.FLAGS if f not in ("-shared", "-ldl")] + ["-DMN_PHASE_TIMING"]
def one(src):
    obj = os.path.join(out, src.replace(".hip", ".o"))
    subprocess.run([b._hipcc()] + cflags + ["-c", "-o", obj, os.path.join(b.CSRC, src)], check=True, cwd=b.CSRC)
    return obj
with ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(one, b.SOURCES))
subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libmuninn_hip.so")] + objs + ["-ldl"], check=True)
print(os.path.join(out, "libmuninn_hip.so"))
