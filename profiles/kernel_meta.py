#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in libsubzero_hip.so (the gfx950 code object's metadata).
usage: python profiles/kernel_meta.py [pattern ...]   (substring filters on the demangled kernel name)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.environ.get('SUBZERO_HIP_LIB') or os.path.join(ROOT, 'subzero_amd', 'csrc', 'libsubzero_hip.so')
pats = sys.argv[1:]
with tempfile.TemporaryDirectory() as d:
    # every object of the library carries its own bundle: list the fat binary sections through the bundler
    out = os.path.join(d, "co")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + os.path.join(d, "fat.bin"), so], check=True)
    blob = open(os.path.join(d, "fat.bin"), "rb").read()
    # the section is a concatenation of bundles, each starting with the magic string
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    rows = []
    for k, s0 in enumerate(starts):
        e0 = starts[k + 1] if k + 1 < len(starts) else len(blob)
        bp = os.path.join(d, "b%d.bin" % k)
        open(bp, "wb").write(blob[s0:e0])
        cp = os.path.join(d, "c%d.co" % k)
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + bp,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + cp], capture_output=True, text=True)
        if r.returncode or not os.path.exists(cp) or os.path.getsize(cp) == 0:
            continue
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", cp], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda key, blk=blk: (re.search(r"\.%s:\s*(\S+)" % key, blk) or [None, "?"])[1]
            name = g("name")
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", ""))
            rows.append((dem, blk.split()[0], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                         g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("kernarg_segment_size")))
    print("%-60s %5s %5s %5s %6s %6s %8s %7s %8s" % ("kernel", "agpr", "vgpr", "sgpr", "vspill", "sspill", "scratch", "lds", "kernarg"))
    for r in sorted(rows):
        if pats and not any(p in r[0] for p in pats):
            continue
        print("%-60s %5s %5s %5s %6s %6s %8s %7s %8s" % r)
