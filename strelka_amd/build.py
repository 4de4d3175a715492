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
BROKER_PATH = os.path.join(LIB_DIR, "sk_broker")  # the per-GPU server of the broker mode (csrc/sk_rt.h), beside the library

HIP_SOURCES = [
    "csrc/sk_context.hip",
    "csrc/sk_rt.hip",
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


def _compile_flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            # no FMA contraction on host or device: the reference is plain x86-64 arithmetic, and the order/rounding of
            # every add is part of the result
            "-ffp-contract=off", "-mllvm", "-disable-promote-alloca-to-lds", "-Wall",
            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")] + \
        os.environ.get("SK_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DSOM_WPE=3)


def build_all(force=False, verbose=True):
    """One object per source under strelka_amd/_build/ (compiled in parallel, recompiled only when the source, a header or the flags
    changed), linked into the one shared library."""
    import glob
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(PKG, "_build")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = _compile_flags()
    headers = glob.glob(os.path.join(PKG, "csrc", "*.h")) + [os.path.join(ROOT, "include", "strelka_amd.h"), os.path.abspath(__file__)]
    stamp = hashlib.sha256(" ".join([hipcc] + flags).encode()).hexdigest()[:12]
    jobs = []
    objs = []
    for s in HOST_SOURCES + HIP_SOURCES:
        src = os.path.join(PKG, s)
        obj = os.path.join(obj_dir, os.path.basename(s).replace(".", "_") + "." + stamp + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            lang = "hip" if s.endswith(".hip") else "c++"
            jobs.append([hipcc] + flags + ["-x", lang, "-c", src, "-o", obj])
    broker_src = os.path.join(PKG, "broker", "sk_broker_main.cpp")

    def link_broker():
        if force or _stale(BROKER_PATH, [broker_src, LIB_PATH]):
            run([hipcc, "-O2", "-std=c++17", broker_src, "-L" + LIB_DIR, "-lstrelka_amd", "-Wl,-rpath,$ORIGIN", "-lpthread", "-o", BROKER_PATH + ".tmp"])
            os.replace(BROKER_PATH + ".tmp", BROKER_PATH)

    def run(cmd):
        if verbose:
            print("[strelka_amd.build]", " ".join(cmd[-4:]), file=sys.stderr)
        subprocess.run(cmd, check=True)
    if not jobs and not force and not _stale(LIB_PATH, objs):
        link_broker()
        return LIB_PATH

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    for old in glob.glob(os.path.join(obj_dir, "*.o")):
        if old not in objs:
            os.remove(old)
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + ".tmp"])
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    link_broker()
    return LIB_PATH


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print(LIB_PATH)
