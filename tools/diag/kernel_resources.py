"""registers, scratch and LDS of every kernel of the product library, from the compiler's own metadata (no GPU needed): each .hip
source compiled to device assembly with the library's flags, the .amdgpu_metadata records listed.

usage: python tools/diag/kernel_resources.py [out.txt]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "strelka_amd")


def main():
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for src in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))):
            asm = os.path.join(d, os.path.basename(src) + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm",
                            "-disable-promote-alloca-to-lds", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"),
                            "--cuda-device-only", "-S", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
            text = open(asm).read()
            meta = text[text.find(".amdgpu_metadata"):]
            for block in meta.split("  - .agpr_count:")[1:]:
                block = "  - .agpr_count:" + block

                def field(name, default="0"):
                    m = re.search(r"\.%s:\s+(\S+)" % name, block)
                    return m.group(1) if m else default
                name = field("name", "?")
                demangled = subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip()
                if "rocprim" in demangled or "hipcub" in demangled:
                    continue  # (library scan / sort helpers)
                short = demangled[5:] if demangled.startswith("void ") else demangled
                short = re.sub(r"\((anonymous namespace|[^()]*)\)$", "", re.sub(r"\(anonymous namespace\)::", "", short))
                short = short.split("(")[0] if "<" not in short.split("(")[0] else short[:short.find(">") + 1]
                rows.append((os.path.basename(src), short, int(field("vgpr_count")), int(field("agpr_count")), int(field("sgpr_count")),
                             int(field("private_segment_fixed_size")), int(field("group_segment_fixed_size")), int(field("max_flat_workgroup_size")),
                             int(field("vgpr_spill_count")), int(field("sgpr_spill_count"))))
    out = ["# kernel resources from the compiler's metadata (gfx950, the library's flags): VGPRs decide the waves per SIMD (512 / vgprs,",
           "# at most 8), scratch = private memory per lane in bytes (spills and private arrays live there), LDS = bytes per workgroup",
           "%-24s %-52s %5s %5s %5s %8s %7s %6s %6s %6s" % ("source", "kernel", "vgpr", "agpr", "sgpr", "scratch", "LDS", "maxwg", "vspill", "sspill")]
    for r in rows:
        out.append("%-24s %-52s %5d %5d %5d %8d %7d %6d %6d %6d" % (r[0], r[1][:52], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
