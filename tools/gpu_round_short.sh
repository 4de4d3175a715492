TAG=r03_v24
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 420 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 200 $OUT/bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-bp 0 --e2e-somatic-bp 0 > $OUT/pmc_$c.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_a5_$c -o pmc -- python bench.py --only a5 --steps 10 --warmup 2 > $OUT/pmc_a5_$c.log 2>&1
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json 2>$OUT/pmc_traffic.err
