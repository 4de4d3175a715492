"""what a caller process pays for having a GPU context at all: the drop-in on a 2 kb region (start, sk_init, two windows, exit), 1 and 16
processes at a time, against the reference on the same region"""
import os
import shutil
import sys
import tempfile

sys.path.insert(0, ".")
from strelka_amd import farm

d = farm.wgs_dataset(int(sys.argv[1]) if len(sys.argv) > 1 else 16000000)
groups = [[(0, "chrW", 1 + 50000 * i, 2000 + 50000 * i, 0)] for i in range(32)]


def argv_fn(binary):
    def fn(index, regions, prefix, skip_header):
        return farm.germline_segment_argv(binary, prefix, [os.path.join(d, "wgs.bam")], regions, os.path.join(d, "wgs.fa"),
                                          chrom_depth=os.path.join(d, "chrom_depth.txt"), skip_header=skip_header)
    return fn


for binary in ("starling2_ref", "starling2_amd"):
    for jobs in (1, 16):
        root = tempfile.mkdtemp(prefix="sk_init_")
        r = farm.run_farm(groups, argv_fn(binary), root, ("variants.vcf",), jobs=jobs, env={"STRELKA_AMD_VERBOSE": "1"})
        shutil.rmtree(root, ignore_errors=True)
        n = len(groups)
        print("%s jobs %2d: wall %.2f s; per process: wall %.3f s, user %.3f s, sys %.3f s" %
              (binary, jobs, r.wall_s, sum(r.process_s) / n, sum(r.user_s) / n, sum(r.sys_s) / n), flush=True)
