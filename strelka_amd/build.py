"""Build the native pieces of strelka_amd for gfx950 (in-tree, so the .so travels with the repo snapshot).

  libstrelka_amd.so   HIP kernels + C-ABI + host adapter   (hipcc --offload-arch=gfx950)

`python -m strelka_amd.build` or `strelka_amd.build.build_all()`.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "strelka_amd")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libstrelka_amd.so")

HIP_SOURCES = [
    "csrc/sk_context.hip",
    "csrc/score_alignments.hip",
    "csrc/germline_site.hip",
    "csrc/germline_fused.hip",
    "csrc/somatic_site.hip",
    "csrc/indel_lhood.hip",
    "csrc/pileup.hip",
    "csrc/global_align.hip",
    "csrc/read_enumerate.hip",
    "csrc/bam_feed.hip",
    "csrc/gvcf_block.hip",
]
HOST_SOURCES = [
    "host/align_flatten.cpp",
    "host/read_realign.cpp",
    "host/active_region.cpp",
    "host/bam_feed.cpp",
]


def _sources():
    return [os.path.join(PKG, s) for s in HIP_SOURCES + HOST_SOURCES]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_all(force=False, verbose=True):
    os.makedirs(LIB_DIR, exist_ok=True)
    import glob
    deps = _sources() + glob.glob(os.path.join(PKG, "csrc", "*.h")) + [os.path.join(ROOT, "include", "strelka_amd.h"),
                                                                       os.path.abspath(__file__)]
    if not force and not _stale(LIB_PATH, deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           # no FMA contraction on host or device: the reference is plain x86-64 arithmetic, and the order/rounding of
           # every add is part of the result
           "-ffp-contract=off", "-mllvm", "-disable-promote-alloca-to-lds", "-Wall",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")]
    cmd += os.environ.get("SK_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DSOM_WPE=3)
    for s in HOST_SOURCES:
        cmd += ["-x", "c++", os.path.join(PKG, s)]
    for s in HIP_SOURCES:
        cmd += ["-x", "hip", os.path.join(PKG, s)]
    cmd += ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print("[strelka_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print(LIB_PATH)
