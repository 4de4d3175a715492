"""wall time of the reference's somatic caller, unmodified and through the adapter, on a WGS-like synthetic tumour/normal
pair (tools/make_wgs_bam.py --role normal/tumor: 40x / 110x, 150 bp, shared germline variants, clonal somatic SNVs and indels)
with the command line the somatic workflow builds for a genome segment (strelka_amd/farm.somatic_segment_argv: EVS scoring
models, callable regions).  One process each; prints the adapter's hook timers.

usage: python tools/diag/e2e_wgs_somatic.py [LENGTH=400000] [variant=amd|dbl] [label=ENV=V,ENV=V;label=...]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
from strelka_amd import farm
from tests import e2e_util as E

OUTPUTS = ("somatic.snvs.vcf", "somatic.indels.vcf", "somatic.callable.regions.bed")


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    variant = sys.argv[2] if len(sys.argv) > 2 else "amd"
    legs = [("default", {}), ("reference pileup", {"STRELKA_AMD_PILEUP": "0"})]
    if len(sys.argv) > 3:
        legs = []
        for leg in sys.argv[3].split(";"):
            label, _, envs = leg.partition("=")
            legs.append((label, dict(e.split("=") for e in envs.split(",") if e)))
    d = farm.wgs_somatic_dataset(length)
    region = "chrW:1-%d" % length

    def run(binary, env=None):
        with tempfile.TemporaryDirectory() as o:
            argv = farm.somatic_segment_argv(binary, o + "/", os.path.join(d, "normal.bam"), os.path.join(d, "tumor.bam"), [region],
                                             os.path.join(d, "normal.fa"), chrom_depth=os.path.join(d, "chrom_depth.txt"),
                                             callable_regions=True)
            t0 = time.perf_counter()
            p = E.run(argv, env=env, timeout=7200)
            dt = time.perf_counter() - t0
            body = {}
            for f in os.listdir(o):
                if f.endswith(".vcf") or f.endswith(".bed"):
                    body[f] = E.vcf_body(os.path.join(o, f), keep_header=True)
            return dt, body, [l for l in p.stderr.decode().splitlines() if "strelka_amd adapter" in l]

    t_ref, want, _ = min((run("strelka2_ref") for _ in range(1)), key=lambda x: x[0])
    print("reference: %.2f s (%s)" % (t_ref, ", ".join("%s %d lines" % (k, len(v)) for k, v in sorted(want.items()))), flush=True)
    for label, extra in legs:
        env = {"STRELKA_AMD_VERBOSE": "1"}
        env.update(extra)
        best = min((run("strelka2_" + variant, env) for _ in range(1 if variant == "dbl" else 2)), key=lambda x: x[0])
        print("adapter %-24s: %.2f s (%.2fx) identical=%s" % (label, best[0], t_ref / best[0], best[1] == want), flush=True)
        for l in best[2][-3:]:
            print("    " + l.replace("strelka_amd adapter ", ""), flush=True)


if __name__ == "__main__":
    main()
