"""Build an A/B variant of the WHOLE engine (one translation unit per source file) into ab_libs/<name>.so (git-ignored AND listed in .gpurunignore:
take it out of that file for the call that runs the A/B; select with SOLORL_LIB).
usage: build_variant.py NAME [-DDEFINE ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solorl_amd import build as b
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "ab_libs"); os.makedirs(out, exist_ok=True)
tmp = os.path.join("/tmp", "abobj_" + name); os.makedirs(tmp, exist_ok=True)
flags = b._flags() + extra
objs = []
for unit, deps in b.UNITS.items():
    o = os.path.join(tmp, unit.replace(".hip", ".o"))
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", "-o", o, deps[0]], stderr=subprocess.DEVNULL)
    objs.append(o)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, name + ".so")] + objs)
print(os.path.join(out, name + ".so"))
