"""wall time of the reference's germline caller, unmodified and through the adapter, on a WGS-like synthetic sample
(tools/make_wgs_bam.py: 40x, 150 bp, human variant density) with the command line the workflow builds for a genome segment
(tests/e2e_util.germline_wgs_argv).  One process each; prints the adapter's hook timers.

usage: python tools/diag/e2e_wgs.py [LENGTH=1000000] [variant=amd|dbl] [windows "R:S,R:S,..."]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
from tests import e2e_util as E


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    variant = sys.argv[2] if len(sys.argv) > 2 else "amd"
    windows = [tuple(int(x) for x in w.split(":")) for w in (sys.argv[3] if len(sys.argv) > 3 else "2048:4096,16384:32768").split(",")]
    d = E.wgs_dataset(length)
    region = "chrW:1-%d" % length
    evs_args = ()
    if os.environ.get("SK_E2E_EVS"):  # the workflow's default: EVS on (stand-in models, tools/make_dummy_germline_models.py)
        md = tempfile.mkdtemp(prefix="sk_models_")
        subprocess.run([sys.executable, "tools/make_dummy_germline_models.py", md], check=True)
        evs_args = ("--snv-scoring-model-file", md + "/germlineSNVScoringModels.json", "--indel-scoring-model-file", md + "/germlineIndelScoringModels.json")

    def run(binary, env=None):
        with tempfile.TemporaryDirectory() as o:
            t0 = time.perf_counter()
            p = E.run(E.germline_wgs_argv(binary, o + "/", [os.path.join(d, "wgs.bam")], [region], os.path.join(d, "wgs.fa"),
                                          os.path.join(d, "chrom_depth.txt"), extra=evs_args), env=env, timeout=3600)
            dt = time.perf_counter() - t0
            body = {f: E.vcf_body(os.path.join(o, f), keep_header=True) for f in ("variants.vcf", "genome.S1.vcf")}
            return dt, body, [l for l in p.stderr.decode().splitlines() if "strelka_amd adapter" in l]

    t_ref, want, _ = min((run("starling2_ref") for _ in range(2)), key=lambda x: x[0])
    print("reference: %.2f s (%d variant records, %d gVCF lines)" % (t_ref, sum(1 for l in want["variants.vcf"] if l[0] != "#"),
                                                                     len(want["genome.S1.vcf"])), flush=True)
    for rw, sw in windows:
        legs = [("default", {}), ("device enumeration", {"SK_ENUMERATION": "2"})]
        if sw > 0:  # (with the reference's pileup the genotypes are batched per site window: a window of 0 is one ABI call per position)
            legs.append(("reference pileup", {"STRELKA_AMD_PILEUP": "0"}))
        for label, extra in legs:
            env = {"STRELKA_AMD_VERBOSE": "1", "STRELKA_AMD_READ_WINDOW": str(rw), "STRELKA_AMD_SITE_WINDOW": str(sw)}
            env.update(extra)
            best = min((run("starling2_" + variant, env) for _ in range(2)), key=lambda x: x[0])
            same = best[1] == want
            print("adapter %-20s windows %6d/%6d: %.2f s (%.2fx) identical=%s" % (label, rw, sw, best[0], t_ref / best[0], same), flush=True)
            for l in best[2][-2:]:
                print("    " + l.replace("strelka_amd adapter ", ""), flush=True)


if __name__ == "__main__":
    main()
