"""Build a variant of libpmgt_hip.so with ONE source recompiled under extra flags (ablation / profiling builds), the other objects taken from
the product build.  Usage: python tools/prof/build_variant.py qkvc_attn.hip out.so -DPMGT_QA3_NO_ATTN [...]"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pmgt_amd import _build
src, out, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
_build.build_hip()
obj = out + ".o"
subprocess.run([_build.HIPCC] + _build.HIP_FLAGS + flags + ["-c", os.path.join(_build.CSRC, src), "-o", obj], check=True)
objs = [obj if s == src else os.path.join(_build.OBJ, s.replace(".hip", ".o")) for s in _build.HIP_SOURCES]
subprocess.run([_build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
os.remove(obj)
print(out)
