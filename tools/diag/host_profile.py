"""sampling profile of the HOST code that remains when the hot path is routed: the adapter programs relinked with -pg (histogram
only -- the objects keep their normal code) over the CPU double of the C-ABI.  The double and libz are shared objects and are not
sampled, so the flat profile is the main program text = what a process still spends on its core beside the device.

usage: python tools/diag/host_profile.py germline|somatic [LENGTH] [out.txt] [RUNS=3]      (build container: needs /root/reference)
(the histogram ticks at 100 Hz: the runs' histograms are summed, the header gives the sampled seconds per run)"""
import glob
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, ".")
from strelka_amd import farm
from tests import e2e_util as E

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "oracle", "_ref")


def link_profiled(program):
    """the `make -C adapter double` link line with -pg; returns the binary's path"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "adapter"), "double"], check=True, stdout=subprocess.DEVNULL)
    objs = sorted(glob.glob(OUT + "/obj/adapter/hooked/*/*.o") + glob.glob(OUT + "/obj/adapter/hooked/*/*/*.o") + glob.glob(OUT + "/obj/adapter/sk_adapter_*.o"))
    L = REF + "/src/c++/lib"
    hts = OUT + "/redist/htslib-1.7-6-g6d2bfb7"
    inc = ["-I" + p for p in (OUT + "/adapter_src", ROOT + "/adapter", ROOT + "/include", L, L + "/starling_common", L + "/applications/starling",
                              L + "/applications/strelka", ROOT + "/oracle/ref/gen", ROOT + "/oracle/boost_shim", hts,
                              OUT + "/redist/rapidjson-1.1.0/include")]
    binary = os.path.join(tempfile.gettempdir(), program + "_prof")  # (not beside the product binaries: oracle/_ref/ travels to the GPU box)
    subprocess.run(["g++", "-std=c++11", "-O3", "-w", "-fPIC", "-ffp-contract=off", "-pg"] + inc + [REF + "/src/c++/bin/%s.cpp" % program] + objs +
                   [OUT + "/libreftus.a", hts + "/libhts.a", "-lm", "-lz", "-lpthread", "-L" + ROOT + "/oracle", "-lstrelka_amd_double",
                    "-Wl,-rpath," + ROOT + "/oracle", "-o", binary], check=True)
    return binary


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "germline"
    length = int(sys.argv[2]) if len(sys.argv) > 2 else (1000000 if mode == "germline" else 400000)
    out = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
    runs = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    region = "chrW:1-%d" % length
    with tempfile.TemporaryDirectory() as o:
        if mode == "germline":
            binary = link_profiled("starling2")
            d = E.wgs_dataset(length)
            md = os.path.join(o, "models")
            os.makedirs(md)
            subprocess.run([sys.executable, os.path.join(ROOT, "tools/make_dummy_germline_models.py"), md], check=True)
            argv = farm.germline_segment_argv(binary, o + "/", [os.path.join(d, "wgs.bam")], [region], os.path.join(d, "wgs.fa"),
                                              chrom_depth=os.path.join(d, "chrom_depth.txt"), evs_models=(md + "/germlineSNVScoringModels.json", md + "/germlineIndelScoringModels.json"))
        else:
            binary = link_profiled("strelka2")
            d = farm.wgs_somatic_dataset(length)
            argv = farm.somatic_segment_argv(binary, o + "/", os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam"), [region],
                                             os.path.join(d, "normal.fa"), chrom_depth=os.path.join(d, "chrom_depth.txt"), callable_regions=True)
        env = dict(os.environ, GMON_OUT_PREFIX=os.path.join(o, "gmon"))
        for _ in range(runs):
            subprocess.run(argv, check=True, cwd=o, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
        text = subprocess.run(["gprof", "-b", "-p", binary] + sorted(glob.glob(os.path.join(o, "gmon.*"))), check=True, stdout=subprocess.PIPE).stdout.decode()
    head = ("# %s drop-in over the CPU double of the C-ABI, %d bp WGS-like segment, build container: sampling profile of the main\n"
            "# program text (the double and libz are shared objects and are not sampled) = the host code that remains\n" % (mode, length))
    lines = [l[:200] for l in text.splitlines()]
    total = 0.0
    for l in lines:
        f = l.split()
        if len(f) > 3 and f[0].replace(".", "").isdigit() and f[1].replace(".", "").isdigit():
            total = max(total, float(f[1]))
    head += "# %d runs summed: %.2f s sampled per run\n" % (runs, total / runs)
    text = head + "\n".join(lines[:70]) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
