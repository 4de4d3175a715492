"""the host remainder of two builds of the adapter, side by side: tools/diag/host_profile.py's measurement (the adapter program relinked
with -pg over the CPU double, main program text only) for THIS tree and for another checkout of the repository, the runs alternating
so that both see the same machine.

usage: python tools/diag/host_profile_ab.py OTHER_ROOT germline|somatic [LENGTH] [RUNS=3]
(OTHER_ROOT: a worktree of an earlier commit with `make -C oracle double` and `make -C adapter double` done; its oracle/_ref may
link to this tree's reference objects and synthetic data)"""
import glob
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, ".")
from strelka_amd import farm
from tests import e2e_util as E

REF = os.environ.get("REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def link_profiled(root, program, tag):
    out = os.path.join(root, "oracle", "_ref")
    shared = os.path.join(HERE, "oracle", "_ref")
    objs = sorted(glob.glob(out + "/obj/adapter/hooked/*/*.o") + glob.glob(out + "/obj/adapter/hooked/*/*/*.o") + glob.glob(out + "/obj/adapter/sk_adapter_*.o"))
    L = REF + "/src/c++/lib"
    hts = shared + "/redist/htslib-1.7-6-g6d2bfb7"
    inc = ["-I" + p for p in (out + "/adapter_src", root + "/adapter", root + "/include", L, L + "/starling_common", L + "/applications/starling",
                              L + "/applications/strelka", HERE + "/oracle/ref/gen", HERE + "/oracle/boost_shim", hts, shared + "/redist/rapidjson-1.1.0/include")]
    binary = os.path.join(tempfile.gettempdir(), "%s_prof_%s" % (program, tag))
    subprocess.run(["g++", "-std=c++11", "-O3", "-w", "-fPIC", "-ffp-contract=off", "-pg"] + inc + [REF + "/src/c++/bin/%s.cpp" % program] + objs +
                   [shared + "/libreftus.a", hts + "/libhts.a", "-lm", "-lz", "-lpthread", "-L" + root + "/oracle", "-lstrelka_amd_double",
                    "-Wl,-rpath," + root + "/oracle", "-o", binary], check=True)
    return binary


def sampled_seconds(binary, gmons):
    text = subprocess.run(["gprof", "-b", "-p", binary] + gmons, check=True, stdout=subprocess.PIPE).stdout.decode()
    total = 0.0
    for l in text.splitlines():
        f = l.split()
        if len(f) > 3 and f[0].replace(".", "").isdigit() and f[1].replace(".", "").isdigit():
            total = max(total, float(f[1]))
    return total


def main():
    other = os.path.abspath(sys.argv[1])
    mode = sys.argv[2] if len(sys.argv) > 2 else "germline"
    length = int(sys.argv[3]) if len(sys.argv) > 3 else (1000000 if mode == "germline" else 400000)
    runs = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    program = "starling2" if mode == "germline" else "strelka2"
    builds = [("other", link_profiled(other, program, "other")), ("this", link_profiled(HERE, program, "this"))]
    region = "chrW:1-%d" % length
    with tempfile.TemporaryDirectory() as o:
        if mode == "germline":
            d = E.wgs_dataset(length)
            md = os.path.join(o, "models")
            os.makedirs(md)
            subprocess.run([sys.executable, os.path.join(HERE, "tools/make_dummy_germline_models.py"), md], check=True)

            def argv(binary, out):
                return farm.germline_segment_argv(binary, out, [os.path.join(d, "wgs.bam")], [region], os.path.join(d, "wgs.fa"),
                                                  chrom_depth=os.path.join(d, "chrom_depth.txt"),
                                                  evs_models=(md + "/germlineSNVScoringModels.json", md + "/germlineIndelScoringModels.json"))
        else:
            d = farm.wgs_somatic_dataset(length)

            def argv(binary, out):
                return farm.somatic_segment_argv(binary, out, os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam"), [region],
                                                 os.path.join(d, "normal.fa"), chrom_depth=os.path.join(d, "chrom_depth.txt"), callable_regions=True)
        outputs = {}
        for r in range(runs):
            for tag, binary in builds:
                w = os.path.join(o, "%s_%d" % (tag, r))
                os.makedirs(w)
                env = dict(os.environ, GMON_OUT_PREFIX=os.path.join(o, "gmon_" + tag))
                if tag == "other":  # ($OTHER_ENV="NAME=V,NAME=V": the same build under another setting, e.g. a stage window size)
                    env.update(dict(kv.split("=") for kv in os.environ.get("OTHER_ENV", "").split(",") if kv))
                subprocess.run(argv(binary, w + "/"), check=True, cwd=w, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
                body = {f: E.vcf_body(os.path.join(w, f), keep_header=True) for f in sorted(os.listdir(w)) if f.endswith(".vcf") or f.endswith(".bed")}
                outputs.setdefault(tag, body)
        print("%s, %d bp, %d alternating runs each; main program text, sampled seconds per run" % (mode, length, runs))
        for tag, binary in builds:
            s = sampled_seconds(binary, sorted(glob.glob(os.path.join(o, "gmon_" + tag + ".*"))))
            print("  %-6s %-40s %.2f s" % (tag, other if tag == "other" else HERE, s / runs))
        print("  outputs identical: %s" % (outputs["other"] == outputs["this"]))


if __name__ == "__main__":
    main()
