"""Builds build/ab/<name>.so = the product sources with extra -D flags, for same-box A/B runs (MN_AB_LIB / scripts/ab_*.sh).
usage: build_variant_lib.py <name> <-Dflag> [<-Dflag> ...]   (never the product library)"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("mn_build", os.path.join(ROOT, "sqlite-muninn_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "build", "ab", "_obj_" + name); os.makedirs(out, exist_ok=True)
cflags = [f for f in b.FLAGS if f not in ("-shared", "-ldl")] + extra
def one(src):
    obj = os.path.join(out, src.replace(".hip", ".o"))
    subprocess.run([b._hipcc()] + cflags + ["-c", "-o", obj, os.path.join(b.CSRC, src)], check=True, cwd=b.CSRC)
    return obj
with ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(one, b.SOURCES))
lib = os.path.join(ROOT, "build", "ab", name + ".so")
subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(b.CSRC, "exports.map"),
                "-o", lib] + objs + ["-ldl"], check=True)
print(lib)
