#!/bin/bash
# counter passes over the headline kernel alone (tools/diag/a1_only.py); prints per-kernel sums
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/a1pmc
rm -rf $OUT; mkdir -p $OUT
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o p -- python tools/diag/a1_only.py 2048 128 ${MODE:-cols} > $OUT/$n.log 2>&1
}
run p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run p2 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY
run p3 SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LEVEL_WAVES SQ_CYCLES
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "gpurun_out/a1pmc")
for p in sorted(glob.glob("gpurun_out/a1pmc/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
    for k, d in acc.items():
        if "score" not in k:
            continue
        print(k, {c: "%.4g" % (v / max(1, cnt[(k, c)])) for c, v in d.items()})
PY
