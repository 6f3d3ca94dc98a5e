"""dev: build gloc3d_amd/lib/libgloc3d_<tag>.so = the library with reg.hip compiled with extra flags (A/B through
GLOC3D_LIB_PATH / tools/dev_sweep.sh).  usage: build_variant.py tag -DFLAG[=v] ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gloc3d_amd import build as b
b.build()
tag, flags = sys.argv[1], sys.argv[2:]
obj = os.path.join(b.LIBDIR, f"reg_{tag}.o")
subprocess.check_call([b.HIPCC] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, "reg.hip"), "-o", obj])
objs = [os.path.join(b.LIBDIR, s.replace(".hip", ".o")) for s in b.SOURCES if s != "reg.hip"] + [obj]
out = os.path.join(b.LIBDIR, f"libgloc3d_{tag}.so")
subprocess.check_call(["g++", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
print(out)
