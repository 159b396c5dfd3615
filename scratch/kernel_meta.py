#!/usr/bin/env python3
"""Code-object table of every kernel of libmocap_hip.so: VGPRs, AGPRs, SGPRs, static LDS, scratch per lane, as the compiler's
metadata states them.  Compiles each csrc/*.hip for the device only, with the Makefile's flags, and reads the note section
(runs without a GPU).   python scratch/kernel_meta.py [file.hip ...] > profiles/rNN_code_objects.md"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mocapv2_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function".split()
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"


def kernels_of(src):
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "k.co")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "--no-gpu-bundle-output", "-c", src, "-o", co], check=True, cwd=CSRC)
        notes = subprocess.run([READELF, "--notes", co], check=True, capture_output=True, text=True).stdout
    out = []
    for block in notes.split("- .agpr_count:")[1:]:
        block = ".agpr_count: " + block.split("amdhsa.target")[0]
        get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, block)
        name = re.search(r"\n\s+\.name:\s+(\S+)", block).group(1)
        out.append(dict(name=name, vgpr=int(get("vgpr_count").group(1)), agpr=int(get("agpr_count").group(1)),
                        sgpr=int(get("sgpr_count").group(1)), lds=int(get("group_segment_fixed_size").group(1)),
                        scratch=int(get("private_segment_fixed_size").group(1)),
                        spill=int(get("vgpr_spill_count").group(1)), wg=int(get("max_flat_workgroup_size").group(1))))
    return out


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f != "abi.hip")
    print("| file | kernel | VGPR | AGPR | SGPR | static LDS B | scratch B/lane | spilled VGPRs | max wg |")
    print("|---|---|---|---|---|---|---|---|---|")
    for f in files:
        ks = kernels_of(os.path.join(CSRC, f))
        names = subprocess.run([FILT], input="\n".join(k["name"] for k in ks), capture_output=True, text=True).stdout.split("\n")
        for k, nm in zip(ks, names):
            nm = re.sub(r"^void mocap::(\(anonymous namespace\)::)?", "", nm)
            nm = re.sub(r"\(.*\)$", "", nm)
            print(f"| {f} | `{nm}` | {k['vgpr']} | {k['agpr']} | {k['sgpr']} | {k['lds']} | {k['scratch']} | {k['spill']} | {k['wg']} |")


if __name__ == "__main__":
    main()
