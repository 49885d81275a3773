#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of a csrc/*.hip file (device-only compile for gfx950 + the code
object's metadata notes).  No GPU needed.

    python tools/kernel_resources.py gemm [filter]
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    name = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    src = os.path.join(REPO, "dreamerv3-torch_amd", "csrc", name + ".hip")
    with tempfile.TemporaryDirectory() as d:
        co, elf = os.path.join(d, "a.co"), os.path.join(d, "a.elf")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.dirname(src),
                        "-I", os.path.join(REPO, "include"), "--cuda-device-only", "-c", src, "-o", co], check=True,
                       capture_output=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={co}",
                        "--targets=hip-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"], check=True, capture_output=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        cur[k] = v
        if k == "wavefront_size":
            dem = subprocess.run(["c++filt", cur.get("name", "?")], capture_output=True, text=True).stdout
            dem = re.sub(r"\(.*", "", dem.strip()).replace("void ", "")
            if filt in dem:
                vg, ag = int(cur.get("vgpr_count", 0)), int(cur.get("agpr_count", 0))
                tot = -(-(vg + ag) // 8) * 8 if ag else vg  # unified file: arch VGPRs rounded up, then AGPRs
                print(f"{dem[:78]:78s} vgpr {vg:4d} agpr {ag:4d} lds {int(cur.get('group_segment_fixed_size', 0)):6d} "
                      f"scratch {cur.get('private_segment_fixed_size'):>4} spill {cur.get('vgpr_spill_count')} "
                      f"waves/SIMD<= {min(8, 512 // max(tot, 1))}")
            cur = {}


if __name__ == "__main__":
    main()
