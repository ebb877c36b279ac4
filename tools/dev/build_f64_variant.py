"""Build an A/B variant of the engine whose fp64 translation units (solorl_hip.hip parts 2 and 3: the double-precision step kernels) are
compiled with extra defines; the other objects are taken from the regular build (solorl_amd/_lib/obj).  -> ab_libs/<name>.so, select with
SOLORL_LIB (tools/dev/ab_f64.py).   usage: build_f64_variant.py NAME -DSOLO_WAVES_PER_SIMD=1 -DSOLO_SETUP_GROUP_F64=3 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solorl_amd import build as b
name, extra = sys.argv[1], sys.argv[2:]
b.build()
out = os.path.join(ROOT, "ab_libs"); os.makedirs(out, exist_ok=True)
tmp = os.path.join("/tmp", "abobj_" + name); os.makedirs(tmp, exist_ok=True)
src = b.UNITS["solorl_hip.hip"][0]
procs, objs = [], []
for part in range(5):
    if part in (2, 3):
        o = os.path.join(tmp, "part%d.o" % part)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + b._flags() + extra + ["-DSOLO_TU_PART=%d" % part, "-c", "-o", o, src], stderr=subprocess.DEVNULL))
    else:
        o = os.path.join(b.OBJ, "solorl_hip.part%d.o" % part)
    objs.append(o)
objs.append(os.path.join(b.OBJ, "solorl_ppo.o"))
assert all(p.wait() == 0 for p in procs)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, name + ".so")] + objs)
print(os.path.join(out, name + ".so"))
