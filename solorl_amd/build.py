"""In-tree build of the HIP engine: ``hipcc --offload-arch=gfx950`` -> solorl_amd/_lib/libsolorl_hip.so.

The .so is git-ignored but travels with the gpurun snapshot; ``build()`` is also what
``__graft_entry__.build()`` calls (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INC = [os.path.join(ROOT, "include", f) for f in ("solorl.h", "solorl_model_data.h")]
# translation units -> what each depends on (the env engine takes ~2.5 min to compile, the PPO kernels seconds: separate objects)
UNITS = {
    "solorl_hip.hip": [os.path.join(HERE, "csrc", f) for f in ("solorl_hip.hip", "dynamics.hpp", "spatial.hpp")] + INC,
    "solorl_ppo.hip": [os.path.join(HERE, "csrc", "solorl_ppo.hip"), INC[0]],
}
SRC = UNITS["solorl_hip.hip"][:3]
LIB = os.path.join(HERE, "_lib", "libsolorl_hip.so")
OBJ = os.path.join(HERE, "_lib", "obj")


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def _flags():
    # -fno-slp-vectorize: packed fp32 operations appear only where the source writes 2-vectors (the PGS sweep, dynamics.hpp)
    # -disable-promote-alloca-to-lds: the step kernels must have NO static LDS -- their phase functions address the dynamic LDS from a
    #   constant base (dynamics.hpp, SOLO_LDS_BASE); left alone the compiler moved a private array of the fp64 team kernel into 3 KB of it
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize", "-mllvm", "-disable-promote-alloca-to-lds",
             "-I" + os.path.join(ROOT, "include")]
    flags += ["-D" + d for d in os.environ.get("SOLORL_BUILD_DEFINES", "").split() if d]   # dev instrumentation
    flags += os.environ.get("SOLORL_BUILD_FLAGS", "").split()                                 # dev experiments
    return flags


TAG = os.path.join(OBJ, "flags.txt")      # the flag string the objects (and the library) were built with


def _same_flags():
    return os.path.exists(TAG) and open(TAG).read() == " ".join(_flags())


def needs_build():
    """Stale sources OR a library built with other flags (an instrumented SOLORL_BUILD_DEFINES / SOLORL_BUILD_FLAGS build must
    never be reused by a plain run: the bench and the tests would measure the wrong binary).  A missing flag record with a
    library present (e.g. a prebuilt .so shipped to a box without its objects) counts as matching only for the plain flags."""
    if any(_stale(LIB, deps) for deps in UNITS.values()):
        return True
    if os.path.exists(TAG):
        return not _same_flags()
    return bool(os.environ.get("SOLORL_BUILD_DEFINES", "").split() or os.environ.get("SOLORL_BUILD_FLAGS", "").split())


# solorl_hip.hip is compiled as five objects in parallel (its SOLO_TU_PART switch: one step-kernel instantiation per part 0..3 =
# 2 * f64 + solo12, part 4 = the C ABI), ~2 min each instead of ~6 min in one piece; a -DSOLO_WAVE_TIMING dev build keeps ONE
# translation unit (its device-side counters are a single global array).
def _jobs(flags):
    single = any(f.startswith("-DSOLO_WAVE_TIMING") for f in flags)
    jobs = []
    for name, deps in UNITS.items():
        if name == "solorl_hip.hip" and not single:
            for part in range(5):
                jobs.append((os.path.join(OBJ, "solorl_hip.part%d.o" % part), deps, ["-DSOLO_TU_PART=%d" % part]))
        else:
            jobs.append((os.path.join(OBJ, name.replace(".hip", ".o")), deps, []))
    return jobs


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = _flags()
    same = _same_flags()
    objs, running = [], []
    for o, deps, extra in _jobs(flags):
        objs.append(o)
        if force or not same or _stale(o, deps):
            cmd = [hipcc] + flags + extra + ["-c", "-o", o, deps[0]]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
    failed = [cmd for cmd, p in running if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    open(TAG, "w").write(" ".join(flags))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="-f" in sys.argv, verbose="-v" in sys.argv))
