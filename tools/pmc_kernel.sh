#!/bin/bash
# PMC counters for one profile driver run: tools/pmc_kernel.sh <a|b> <tag>
export TMPDIR=/tmp
W=${1:-b}; TAG=${2:-pmc}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -o p -- python tools/prof_a.py $W > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p2 -o p -- python tools/prof_a.py $W > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            if "at::" in k or "rocclr" in k or "elementwise" in k: continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
