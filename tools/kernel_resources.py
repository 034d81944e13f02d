#!/usr/bin/env python3
"""Per-kernel resource table of the gfx950 code object (VGPRs, spills, scratch bytes, static LDS) read from the
AMDGPU metadata notes: tools/kernel_resources.py [umi_engine.hip].  A kernel with scratch > 0 needs the runtime's
scratch allocation at its first dispatch."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fastf_amd/csrc/umi_engine.hip")
with tempfile.TemporaryDirectory() as d:
    co, elf = os.path.join(d, "k.co"), os.path.join(d, "k.elf")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + ROOT + "/include",
                           "-I" + ROOT + "/fastf_amd/csrc", "-Wno-pass-failed", "--cuda-device-only", "-c", src, "-o", co] + sys.argv[2:])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + co,
                           "--targets=hip-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", elf], capture_output=True, text=True, check=True).stdout
print("%8s %5s %6s %7s  %s" % ("scratch", "vgpr", "spill", "lds", "kernel"))
for blk in notes.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "-"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    print("%8s %5s %6s %7s  %s" % (g("private_segment_fixed_size"), g("vgpr_count"), g("vgpr_spill_count"),
                                   g("group_segment_fixed_size"), name[:120]))
