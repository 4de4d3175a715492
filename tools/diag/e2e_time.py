"""wall time of the reference's own programs, unmodified and with the adapter (GPU), on the synthetic data sets of the end-to-end
tests -- what a user of the drop-in sees today (small inputs: process start, sk_init and per-window launches are all in there)"""
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
from tests import e2e_util as E

SYNTH = os.path.join(E.REF_DIR, "synth")


def run(binary, argv_fn, env=None):
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        p = E.run(argv_fn(binary, d + "/"), env=env)
        dt = time.perf_counter() - t0
        tail = [l for l in p.stderr.decode().splitlines() if "strelka_amd adapter" in l]
        return dt, " || ".join(t.replace("strelka_amd adapter", "") for t in tail)


def main():
    big = None
    if len(sys.argv) > 1:  # a larger data set made on the spot: python tools/diag/e2e_time.py LENGTH
        import subprocess
        big = tempfile.mkdtemp(prefix="synth_big_")
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "tools/make_synth_bam.py", big, os.path.join(E.REF_DIR, "bin", "samtools"), "--seed", "11",
                        "--length", sys.argv[1]], check=True, stdout=subprocess.DEVNULL)
        print("made %s bp data set in %.0f s" % (sys.argv[1], time.perf_counter() - t0), flush=True)
    sets = (("", 60000), ("long_reads", 36000)) if big is None else ((big, int(sys.argv[1])),)
    for which, length in sets:
        d = which if big else os.path.join(SYNTH, which)
        region, fa = "chrS:1-%d" % length, os.path.join(d, "synth.fa")
        germ = lambda b, o: E.germline_argv(b, o, [os.path.join(d, "germline_S1.bam"), os.path.join(d, "germline_S2.bam")], region=region, ref=fa)
        som = lambda b, o: E.somatic_argv(b, o, os.path.join(d, "somatic_normal.bam"), os.path.join(d, "somatic_tumor.bam"), region=region, ref=fa)
        for name, fn, ref_bin, amd_bin in (("germline", germ, "starling2_ref", "starling2_amd"), ("somatic", som, "strelka2_ref", "strelka2_amd")):
            t_ref = min(run(ref_bin, fn)[0] for _ in range(2))
            for windows in ((256, 512), (2000, 4000)):
                res = {}
                for mode in ("0", "2"):
                    best = None
                    for _ in range(2):
                        t, line = run(amd_bin, fn, env={"STRELKA_AMD_VERBOSE": "1", "SK_ENUMERATION": mode,
                                                        "STRELKA_AMD_READ_WINDOW": str(windows[0]), "STRELKA_AMD_SITE_WINDOW": str(windows[1])})
                        best = (t, line) if best is None or t < best[0] else best
                    res[mode] = best
                print("%s %s windows %s: reference %.2f s; adapter host-enum %.2f s, device-enum %.2f s | %s" %
                      (os.path.basename(which) or "short_reads", name, windows, t_ref, res["0"][0], res["2"][0], res["0"][1][-110:] + ' ## ' + res["2"][1][-110:]), flush=True)


if __name__ == "__main__":
    main()
