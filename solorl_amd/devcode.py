"""Static inspection of the built gfx950 code object (no GPU needed): per-function counts of scratch (spill),
FLAT and global memory instructions.  Used by tests/test_abi.py to pin two properties that were each worth a
regression hunt on the GPU: the fp32 team-mode phases must not spill (101 spilled VGPRs in one phase
multiplied the kernel's HBM traffic by eight) and must not reach their context through FLAT instructions."""
import os
import re
import subprocess
import tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(lib_path, workdir):
    """gfx950 code objects of the library: its .hip_fatbin section holds one offload bundle per translation unit (the engine is built
    from several, solorl_amd/build.py), back to back -- split at the bundle magic, unbundle each."""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path, os.path.join(workdir, "x.so")])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for k, a in enumerate(starts):
        part = os.path.join(workdir, "fat%d.bin" % k)
        open(part, "wb").write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        co = os.path.join(workdir, "dev%d.co" % k)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        if os.path.getsize(co) > 0:
            out.append(co)
    return out


def disassemble(lib_path):
    """Disassembly of every code object, concatenated.  A function template instantiated in several translation units (the sweeps
    depend on the arithmetic type only, the units are split by type AND robot) appears once per unit: later copies are renamed
    `<name>.dupN` so that per-function statistics are never mixed (the copies are identical code)."""
    out, seen = [], {}
    with tempfile.TemporaryDirectory() as d:
        for co in _code_objects(lib_path, d):
            for line in subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True).splitlines():
                m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
                if m:
                    k = seen.get(m.group(2), 0)
                    seen[m.group(2)] = k + 1
                    if k:
                        line = "%s <%s.dup%d>:" % (m.group(1), m.group(2), k)
                out.append(line)
    return "\n".join(out)


def kernel_static_lds(lib_path):
    """{kernel name: static LDS bytes (.group_segment_fixed_size of the code object's metadata)}"""
    with tempfile.TemporaryDirectory() as d:
        notes = "\n".join(subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True) for co in _code_objects(lib_path, d))
    out, size = {}, None
    for line in notes.splitlines():       # per kernel the keys come in alphabetical order: the size line precedes the name line
        m = re.search(r"\.group_segment_fixed_size:\s*(\d+)", line)
        if m:
            size = int(m.group(1))
        m = re.search(r"^\s*\.name:\s*(\S+)", line)
        if m and size is not None:
            out[m.group(1)] = size
            size = None
    return out


def function_stats(lib_path):
    """{mangled name: {"insts", "scratch", "flat", "global"}}"""
    out, cur = {}, None
    for line in disassemble(lib_path).splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), dict(insts=0, scratch=0, flat=0, **{"global": 0})) if ".dup" not in m.group(1) else None
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


def loop_stats(lib_path, name_filter):
    """For every function whose mangled name contains `name_filter`: the widest backward branch (spanning more than
    200 bytes) is taken as THE loop; returns {name: {"insts", "scratch", "scratch_in_loop", "valu_in_loop", "lds_in_loop"}}.
    Used to pin that the PGS sweep loops touch no memory at all (callee-saved registers are saved around them)."""
    funcs, cur = {}, None
    for line in disassemble(lib_path).splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is not None and "\t" in line:
            cur.append(line)
    out = {}
    off_re = re.compile(r"//\s*([0-9A-Fa-f]+):")
    for name, lines in funcs.items():
        if name_filter not in name or not lines or ".dup" in name:
            continue
        base = int(off_re.search(lines[0]).group(1), 16)
        rows = []
        for l in lines:
            mo = off_re.search(l)
            if mo:
                rows.append((int(mo.group(1), 16) - base, l.split("\t")[1].strip() if len(l.split("\t")) > 1 else ""))
        lo = hi = None
        for off, op in rows:
            if op.startswith(("s_cbranch", "s_branch")):
                mt = re.search(r"<[^>]*\+0x([0-9a-f]+)>", op)
                if mt and int(mt.group(1), 16) < off and off - int(mt.group(1), 16) > 200:
                    if lo is None or off - int(mt.group(1), 16) > hi - lo:         # the widest backward branch = the sweep loop
                        lo, hi = int(mt.group(1), 16), off
        inl = [op for off, op in rows if lo is not None and lo <= off <= hi]
        out[name] = dict(insts=len(rows), scratch=sum(op.startswith("scratch_") for _, op in rows), loop=(lo, hi),
                         scratch_in_loop=sum(op.startswith("scratch_") for op in inl), valu_in_loop=sum(op.startswith("v_") for op in inl),
                         lds_in_loop=sum(op.startswith("ds_") for op in inl), lds_reads_in_loop=sum(op.startswith("ds_read") for op in inl),
                         vmem_in_loop=sum(op.startswith(("global_", "flat_", "buffer_")) for op in inl))
    return out


if __name__ == "__main__":
    import sys
    from .build import LIB
    for name, s in sorted(function_stats(sys.argv[1] if len(sys.argv) > 1 else LIB).items()):
        if s["insts"] > 100:
            print("%-120s insts %6d scratch %4d flat %4d global %4d" % (name[:120], s["insts"], s["scratch"], s["flat"], s["global"]))
