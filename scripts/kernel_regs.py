#!/usr/bin/env python3
"""Register / LDS / spill figures of the gfx950 kernels in libredgpu.so, read from the code
objects' metadata notes (no GPU needed).  usage: kernel_regs.py [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(os.path.dirname(HERE), "one_amd", "csrc", "obj")


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as tmp:
        co = os.path.join(tmp, "k.co")
        fat = os.path.join(tmp, "k.fatbin")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section",
                            ".hip_fatbin=" + fat, obj], capture_output=True, text=True)
        if r.returncode or not os.path.exists(fat):
            return []
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                            "--output=" + co], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co):
            return []
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co],
                               capture_output=True, text=True).stdout
    out = []
    for block in notes.split("- .agpr_count:")[1:]:
        def field(name):
            m = re.search(r"\." + name + r":\s+(\S+)", block)
            return m.group(1) if m else "?"
        sym = field("symbol").replace(".kd", "")
        name = subprocess.run(["c++filt", sym], capture_output=True,
                              text=True).stdout.strip()
        out.append((name, field("vgpr_count"), field("sgpr_count"), field("vgpr_spill_count"),
                    field("sgpr_spill_count"), field("group_segment_fixed_size")))
    return out


def main():
    want = sys.argv[1:]
    for f in sorted(os.listdir(OBJ)):
        if not f.endswith(".o") or "-" in f:
            continue
        for name, v, s, vs, ss, lds in kernels_of(os.path.join(OBJ, f)):
            if want and not all(w in name for w in want):
                continue
            print("%-14s vgpr %3s sgpr %3s spill v%s s%s lds %6s  %s" % (f, v, s, vs, ss, lds,
                                                                         name[:150]))


if __name__ == "__main__":
    main()
