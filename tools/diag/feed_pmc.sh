export TMPDIR=/tmp
OUT=${OUT:-$PWD/gpurun_out/feed_pmc}
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --only feed --steps 3 --warmup 1 > $OUT/pmc_$c.log 2>&1
done
python - <<'P'
import csv, glob, os
out = os.environ.get("OUT", "gpurun_out/feed_pmc")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = {}
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                vals.setdefault(row["Kernel_Name"].split("(")[0][-40:], []).append(float(row["Counter_Value"]))
    for k, v in vals.items():
        if "bgzf" in k:
            print(c, k, "launches", len(v), "mean KiB", sum(v) / len(v))
P
