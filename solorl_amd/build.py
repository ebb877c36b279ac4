"""In-tree build of the HIP engine: ``hipcc --offload-arch=gfx950`` -> solorl_amd/_lib/libsolorl_hip.so.

The .so is git-ignored but travels with the gpurun snapshot; ``build()`` is also what
``__graft_entry__.build()`` calls (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("solorl_hip.hip", "dynamics.hpp", "spatial.hpp")]
DEPS = SRC + [os.path.join(ROOT, "include", f) for f in ("solorl.h", "solorl_model_data.h")]
LIB = os.path.join(HERE, "_lib", "libsolorl_hip.so")


def needs_build():
    return not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: packed fp32 operations appear only where the source writes 2-vectors (the PGS sweep, dynamics.hpp)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize",
           "-I" + os.path.join(ROOT, "include"), "-o", LIB + ".tmp", SRC[0]]
    cmd[1:1] = ["-D" + d for d in os.environ.get("SOLORL_BUILD_DEFINES", "").split() if d]   # dev instrumentation
    cmd[1:1] = os.environ.get("SOLORL_BUILD_FLAGS", "").split()                                 # dev experiments
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="-f" in sys.argv, verbose="-v" in sys.argv))
