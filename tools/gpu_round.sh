#!/bin/bash
# one GPU-box visit: parity tests, bench, kernel-trace stats, PMC traffic passes.  Outputs under gpurun_out/$TAG.
TAG=${1:-r01_v3}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.json
# the launch line the driver uses for N > 1, with one rank (RCCL init + barrier + max-over-ranks path)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --e2e-bp 4000000 --e2e-segment-bp 1000000 --e2e-somatic-bp 800000 --e2e-somatic-segment-bp 200000 > $OUT/bench_torchrun.json 2> $OUT/bench_torchrun.err; tail -c 300 $OUT/bench_torchrun.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --e2e-bp 0 --e2e-somatic-bp 0 > $OUT/ktrace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-bp 0 --e2e-somatic-bp 0 > $OUT/pmc_$c.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_a5_$c -o pmc -- python bench.py --only a5 --steps 10 --warmup 2 > $OUT/pmc_a5_$c.log 2>&1
done
find $OUT -name "*.csv" | head -20
python tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json 2>$OUT/pmc_traffic.err; cat $OUT/pmc_traffic.json
# one caller process, reference vs drop-in, with the adapter's hook timers: germline 1 Mb with the EVS models on, the tumour-normal pair
SK_E2E_EVS=1 timeout 600 python tools/diag/e2e_wgs.py 1000000 amd 8192:0 > $OUT/e2e_wgs_evs.txt 2>&1; tail -4 $OUT/e2e_wgs_evs.txt
timeout 600 python tools/diag/e2e_wgs_somatic.py 400000 amd "default=" > $OUT/e2e_wgs_somatic.txt 2>&1; tail -4 $OUT/e2e_wgs_somatic.txt
