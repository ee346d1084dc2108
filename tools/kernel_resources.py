#!/usr/bin/env python3
"""Registers, LDS and occupancy of every kernel of the library, from the compiler's own remarks
(hipcc -Rpass-analysis=kernel-resource-usage, device code only; no GPU needed).
usage: tools/kernel_resources.py [file.hip ...]   (default: the kernels of a frame)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cudagaussianrenderer_amd", "csrc")
files = sys.argv[1:] or ["gs_clip.hip", "gs_radix.hip", "gs_tiles_fast.hip", "gs_raster.hip"]
LDS, OCC, SCR = r"LDS Size \[bytes/block\]", r"Occupancy \[waves/SIMD\]", r"ScratchSize \[bytes/lane\]"
for f in files:
    with tempfile.TemporaryDirectory() as tmp:
        p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                            "-Wno-unused-function", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
                            os.path.join(CSRC, f), "-o", os.path.join(tmp, "x.o")] + os.environ.get("EXTRA", "").split(), capture_output=True, text=True)
    for b in re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]:
        name = b.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", "")).replace("void ", "").replace("gs::", "")
        g = lambda k: re.search(k + r": (\d+)", b).group(1)
        print("%-62s VGPR %4s AGPR %3s SGPR %3s LDS %6s occ %s scratch %s" % (dem[:62], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(LDS), g(OCC), g(SCR)))
