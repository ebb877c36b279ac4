"""Static inspection of the built gfx950 code object (no GPU needed): per-function counts of scratch (spill),
FLAT and global memory instructions.  Used by tests/test_abi.py to pin two properties that were each worth a
regression hunt on the GPU: the fp32 team-mode phases must not spill (101 spilled VGPRs in one phase
multiplied the kernel's HBM traffic by eight) and must not reach their context through FLAT instructions."""
import os
import re
import subprocess
import tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def disassemble(lib_path):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path, os.path.join(d, "x.so")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)


def function_stats(lib_path):
    """{mangled name: {"insts", "scratch", "flat", "global"}}"""
    out, cur = {}, None
    for line in disassemble(lib_path).splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), dict(insts=0, scratch=0, flat=0, **{"global": 0}))
            continue
        if cur is None or "\t" not in line:
            continue
        op = line.split("\t")[1].strip().split(" ")[0] if len(line.split("\t")) > 1 else ""
        if not op:
            continue
        cur["insts"] += 1
        if op.startswith("scratch_"): cur["scratch"] += 1
        elif op.startswith("flat_"): cur["flat"] += 1
        elif op.startswith("global_"): cur["global"] += 1
    return out


if __name__ == "__main__":
    import sys
    from .build import LIB
    for name, s in sorted(function_stats(sys.argv[1] if len(sys.argv) > 1 else LIB).items()):
        if s["insts"] > 100:
            print("%-120s insts %6d scratch %4d flat %4d global %4d" % (name[:120], s["insts"], s["scratch"], s["flat"], s["global"]))
