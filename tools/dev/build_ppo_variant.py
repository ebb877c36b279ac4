"""A/B variant of the PPO kernels only: solorl_ppo.hip recompiled with extra defines and linked with the engine objects of the current
build (solorl_amd/_lib/obj) into ab_libs/<name>.so (git-ignored AND in .gpurunignore: take it out for the call that runs the A/B; select
with SOLORL_LIB).   usage: build_ppo_variant.py NAME [-DDEFINE ...]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solorl_amd import build as b
b.build()
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "ab_libs"); os.makedirs(out, exist_ok=True)
o = os.path.join("/tmp", "abppo_%s.o" % name)
subprocess.check_call(["/opt/rocm/bin/hipcc"] + b._flags() + extra + ["-c", "-o", o, b.UNITS["solorl_ppo.hip"][0]], stderr=subprocess.DEVNULL)
objs = sorted(glob.glob(os.path.join(b.OBJ, "solorl_hip.part*.o"))) + [o]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, name + ".so")] + objs)
print(os.path.join(out, name + ".so"))
