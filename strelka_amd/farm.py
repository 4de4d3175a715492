"""The segment farm: genome segments as independent caller processes over the GPUs of one node.

The reference's only parallelism is one `starling2` / `strelka2` process per genome segment, scheduled by its pyflow workflow
(PY/ = /root/reference/src/python/lib/): segments from PY/workflowUtil.py:182-218 (getChromIntervals: every chromosome cut into
equal pieces no longer than scanSizeMb = 12 Mb, PY/strelkaSharedOptions.py:161), small ones grouped into one process
(getGenomeSegmentGroups :340-371), the per-segment outputs concatenated in segment order (PY/strelkaSharedWorkflow.py:102-147;
every segment but the first runs with --gvcf-skip-header, PY/strelkaGermlineWorkflow.py:120-121).  pyflow is Python 2 and is not
what is being replaced; this module is the small part of it the drop-in needs to be run and measured on N GPUs: the same
segment list, `jobs` processes at a time, process i on device i mod N ($STRELKA_AMD_DEVICE, read by adapter/sk_adapter_common.cpp),
outputs joined in segment order.  No collective, nothing shared between the processes but their read-only inputs.
"""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the caller programs: the reference's own main()s and translation units linked with adapter/ + libstrelka_amd.so (`*_amd`, built by
# adapter/Makefile where /root/reference is present; they travel with oracle/_ref/ because they contain the reference's object
# code, which never enters the repository), and the unmodified reference (`*_ref`, oracle/Makefile) as the CPU baseline
BIN_DIR = os.path.join(REPO, "oracle", "_ref", "bin")
MODEL_DIR = os.path.join(REPO, "oracle", "_ref", "demo")


def usable_cores():
    """cores this process may use: the affinity mask, cut down to the cgroup CPU quota where one is set"""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        cores = cores[:max(1, int(quota + 0.5))]
    return cores


def germline_segment_argv(binary, out_prefix, bams, regions, ref, chrom_depth=None, ploidy_vcf=None, nocompress_bed=None,
                          skip_header=False, extra=(), evs_models=None, report_evs_features=False):
    """The command line of one germline segment process as the workflow builds it: PY/strelkaGermlineWorkflow.py:81-147 +
    appendCommonGenomeSegmentCommandOptions (PY/strelkaSharedWorkflow.py:164-200): several --region per process (gsegGroup),
    --chrom-depth-file when the high-depth filter is on (WGS; :125-126), --ploidy-region-vcf (:131-132), --nocompress-bed
    (:128-129), --gvcf-skip-header for every segment but the first (:120-121).  evs_models = (SNV model, indel model): the
    workflow's default passes --snv-scoring-model-file / --indel-scoring-model-file (:111-115); the reference tree does not carry
    its germline models (they ship with the release packages), tools/make_dummy_germline_models.py writes small stand-ins."""
    cmd = [os.path.join(BIN_DIR, binary)]
    for r in regions:
        cmd += ["--region", r]
    cmd += ["--ref", ref, "--max-indel-size", "49", "--min-mapping-quality", "20",
            "--gvcf-output-prefix", out_prefix, "--gvcf-min-gqx", "15", "--gvcf-min-homref-gqx", "15",
            "--gvcf-max-snv-strand-bias", "10", "--enable-read-backed-phasing",
            "--stats-file", out_prefix + "runStats.xml"]
    for b in bams:
        cmd += ["--align-file", b]
    if skip_header:
        cmd.append("--gvcf-skip-header")
    if chrom_depth:
        cmd += ["--chrom-depth-file", chrom_depth]
    if nocompress_bed:
        cmd += ["--nocompress-bed", nocompress_bed]
    if ploidy_vcf:
        cmd += ["--ploidy-region-vcf", ploidy_vcf]
    cmd += ["--indel-error-models-file", os.path.join(MODEL_DIR, "indelErrorModel.json"), "--theta-file", os.path.join(MODEL_DIR, "theta.json")]
    if evs_models:
        cmd += ["--snv-scoring-model-file", evs_models[0], "--indel-scoring-model-file", evs_models[1]]
    if report_evs_features:
        cmd.append("--report-evs-features")
    return cmd + list(extra)


def somatic_segment_argv(binary, out_prefix, normal_bam, tumor_bam, regions, ref, chrom_depth=None, callable_regions=False, skip_header=False,
                         extra=()):
    """One somatic segment process as the workflow builds it: PY/strelkaSomaticWorkflow.py:74-146 with the defaults of
    src/python/bin/configureStrelkaSomaticWorkflow.py.ini, EVS models on (they ship with the reference), --strelka-chrom-depth-file /
    --strelka-max-depth-factor when the high-depth filter is on (WGS; :140-142), --strelka-skip-header for every segment but the first."""
    cmd = [os.path.join(BIN_DIR, binary)]
    for r in regions:
        cmd += ["--region", r]
    cmd += ["--ref", ref, "--max-indel-size", "49", "--min-mapping-quality", "20",
            "--somatic-snv-rate", "0.0001", "--shared-site-error-rate", "0.0000000005",
            "--shared-site-error-strand-bias-fraction", "0.0", "--somatic-indel-rate", "0.000001",
            "--shared-indel-error-factor", "2.2", "--tier2-min-mapping-quality", "0",
            "--strelka-snv-max-filtered-basecall-frac", "0.4", "--strelka-snv-max-spanning-deletion-frac", "0.75",
            "--strelka-snv-min-qss-ref", "15", "--strelka-indel-max-window-filtered-basecall-frac", "0.3",
            "--strelka-indel-min-qsi-ref", "40", "--ssnv-contam-tolerance", "0.15", "--indel-contam-tolerance", "0.15",
            "--somatic-snv-scoring-model-file", os.path.join(MODEL_DIR, "somaticSNVScoringModels.json"),
            "--somatic-indel-scoring-model-file", os.path.join(MODEL_DIR, "somaticIndelScoringModels.json"),
            "--normal-align-file", normal_bam, "--tumor-align-file", tumor_bam,
            "--somatic-snv-file", out_prefix + "somatic.snvs.vcf", "--somatic-indel-file", out_prefix + "somatic.indels.vcf"]
    if callable_regions:
        cmd += ["--somatic-callable-regions-file", out_prefix + "somatic.callable.regions.bed"]
    cmd += ["--stats-file", out_prefix + "runStats.xml"]
    if skip_header:
        cmd.append("--strelka-skip-header")
    if chrom_depth:
        cmd += ["--strelka-chrom-depth-file", chrom_depth, "--strelka-max-depth-factor", "3.0"]
    return cmd + list(extra)


def wgs_somatic_dataset(length=1000000, normal_depth=40.0, tumor_depth=110.0, seed=20260926, procs=16):
    """A WGS-like tumour / normal pair (BASELINE.json configs[2]: 110x / 40x): the same reference and germline variants, the tumour with
    two clone haplotypes carrying somatic SNVs and indels.  -> directory with normal.bam, tumor.bam, normal.fa (the pair's reference),
    chrom_depth.txt"""
    d = os.path.join(REPO, "oracle", "_ref", "synth", "wgs_somatic_%d_%g_%g_%d_p%d" % (length, normal_depth, tumor_depth, seed, procs))
    if not os.path.exists(os.path.join(d, "chrom_depth.txt")):
        os.makedirs(d, exist_ok=True)
        for role, depth, name in (("normal", normal_depth, "normal"), ("tumor", tumor_depth, "tumor")):
            subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_wgs_bam.py"), d, os.path.join(BIN_DIR, "samtools"),
                            "--length", str(length), "--depth", str(depth), "--seed", str(seed), "--procs", str(procs), "--role", role,
                            "--name", name, "--sample", name.upper()], check=True, stdout=subprocess.DEVNULL)
        with open(os.path.join(d, "chrom_depth.txt"), "w") as f:
            f.write("chrW\t%.3f\n" % normal_depth)
    return d


def wgs_dataset(length=1000000, depth=40.0, seed=20260926, procs=16):
    """A WGS-like synthetic sample (tools/make_wgs_bam.py: 150 bp reads, human variant density), made on the spot under
    oracle/_ref/synth/ (git-ignored) and kept.  -> directory with wgs.bam(.bai), wgs.fa(.fai), chrom_depth.txt"""
    d = os.path.join(REPO, "oracle", "_ref", "synth", "wgs_%d_%g_%d_p%d" % (length, depth, seed, procs))
    if not os.path.exists(os.path.join(d, "chrom_depth.txt")):
        os.makedirs(d, exist_ok=True)
        subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_wgs_bam.py"), d, os.path.join(BIN_DIR, "samtools"),
                        "--length", str(length), "--depth", str(depth), "--seed", str(seed), "--procs", str(procs)],
                       check=True, stdout=subprocess.DEVNULL)
        with open(os.path.join(d, "chrom_depth.txt"), "w") as f:  # GetChromDepth's output: chrom <tab> mean depth
            f.write("chrW\t%.3f\n" % depth)
    return d


def chrom_intervals(chrom_order, chrom_sizes, segment_size, region=None):
    """PY/workflowUtil.py:182-218 -> (chrom index, chrom, start, end, bin), 1-based closed intervals"""
    for ci, chrom in enumerate(chrom_order):
        start, end = 1, chrom_sizes[chrom]
        if region is not None:
            if region[0] != chrom:
                continue
            start = region[1] if region[1] is not None else start
            end = region[2] if region[2] is not None else end
        size = end - start + 1
        n = 1 + (size - 1) // segment_size
        base, plus = size // n, size % n
        s = start
        for i in range(n):
            seg = base + (1 if i < plus else 0)
            e = min(s + seg - 1, start + size)
            yield (ci, chrom, s, e, i)
            s = e + 1


def segment_groups(segments, min_group_size=200000):
    """PY/workflowUtil.py:340-371: consecutive small segments share a process"""
    group, head = [], 0
    for seg in segments:
        size = seg[3] - seg[2] + 1
        if group and head + size <= min_group_size:
            group.append(seg)
            head += size
        else:
            if group:
                yield group
            group, head = [seg], size
    if group:
        yield group


def region_arg(seg):
    return "%s:%d-%d" % (seg[1], seg[2], seg[3])


class FarmResult:
    def __init__(self):
        self.wall_s = 0.0
        self.process_s = []     # per process wall time
        self.user_s = []        # ... user CPU seconds (all its threads)
        self.sys_s = []         # ... system CPU seconds
        self.outputs = {}       # file name -> joined output path
        self.stderr_tails = []


def run_farm(groups, argv_fn, out_dir, output_names, n_gpus=1, jobs=None, device_offset=0, env=None, pin_cores=None, join=True):
    """Run one process per segment group, `jobs` at a time.

    groups: [[segment, ...], ...] in genome order; argv_fn(group_index, regions, out_prefix, skip_header) -> argv;
    output_names: the files each process writes under its prefix (e.g. "variants.vcf", "genome.S1.vcf"), joined in group order
    into out_dir/<name>; process i gets STRELKA_AMD_DEVICE = device_offset + i mod n_gpus; pin_cores: optional list of CPU ids, process
    slots are pinned round-robin (taskset semantics through os.sched_setaffinity in the child)."""
    os.makedirs(out_dir, exist_ok=True)
    jobs = jobs or os.cpu_count() or 1
    base_env = dict(os.environ)
    if env:
        base_env.update(env)
    res = FarmResult()
    pending = list(enumerate(groups))
    running = {}  # slot -> (index, Popen, t0, err path)
    free_slots = list(range(jobs))
    t_start = time.perf_counter()
    done = {}
    usage = {}

    def launch(slot, index, group):
        prefix = os.path.join(out_dir, "seg%04d." % index)
        argv = argv_fn(index, [region_arg(s) for s in group], prefix, index != 0)
        e = dict(base_env)
        e["STRELKA_AMD_DEVICE"] = str(device_offset + index % max(1, n_gpus))
        err = open(prefix + "stderr.txt", "wb")
        pre = None
        if pin_cores:
            core = pin_cores[slot % len(pin_cores)]
            pre = lambda: os.sched_setaffinity(0, {core})
        p = subprocess.Popen(argv, stdout=subprocess.DEVNULL, stderr=err, env=e, preexec_fn=pre)
        running[slot] = (index, p, time.perf_counter(), err, prefix)

    while pending or running:
        free_slots.sort()
        while pending and free_slots:
            slot = free_slots.pop(0)
            index, group = pending.pop(0)
            launch(slot, index, group)
        finished = []
        for slot_, (_, p_, _, _, _) in list(running.items()):
            if p_.returncode is not None:
                finished.append(slot_)
                continue
            try:
                pid, status, ru = os.wait4(p_.pid, os.WNOHANG)
            except ChildProcessError:
                pid, status, ru = p_.pid, 0, None
            if pid == p_.pid:
                p_.returncode = os.waitstatus_to_exitcode(status) if ru is not None else (p_.poll() or 0)
                usage[p_.pid] = ru
                finished.append(slot_)
        if not finished:
            time.sleep(0.002)
            continue
        for slot in finished:
            index, p, t0, err, prefix = running.pop(slot)
            err.close()
            if p.returncode != 0:
                for _, q, _, e2, _ in running.values():
                    q.kill()
                    e2.close()
                with open(prefix + "stderr.txt", "rb") as f:
                    raise RuntimeError("segment process %d failed (%d):\n%s" % (index, p.returncode, f.read().decode(errors="replace")[-3000:]))
            ru = usage.get(p.pid)
            done[index] = (time.perf_counter() - t0, prefix, ru.ru_utime if ru else 0.0, ru.ru_stime if ru else 0.0)
            free_slots.append(slot)
    res.wall_s = time.perf_counter() - t_start
    res.process_s = [done[i][0] for i in sorted(done)]
    res.user_s = [done[i][2] for i in sorted(done)]
    res.sys_s = [done[i][3] for i in sorted(done)]
    for i in sorted(done):
        with open(done[i][1] + "stderr.txt", "rb") as f:
            res.stderr_tails.append(f.read().decode(errors="replace")[-6000:])
    if join:
        # the workflow's concatenation (bgzip'd pieces joined with bgzf_cat there; the raw text here)
        for name in output_names:
            path = os.path.join(out_dir, name)
            with open(path, "wb") as out:
                for i in sorted(done):
                    with open(done[i][1] + name, "rb") as f:
                        while True:
                            chunk = f.read(1 << 22)
                            if not chunk:
                                break
                            out.write(chunk)
            res.outputs[name] = path
    return res
