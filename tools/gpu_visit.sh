#!/bin/bash
# one GPU-box visit: the steps named on the command line, outputs under gpurun_out/$TAG.
#   tools/gpurun.sh --timeout S -- tools/gpu_visit.sh TAG step [step ...]
# steps: e2e (bench's germline leg; on a mismatch the parity hunt), e2e_somatic, tests (pytest -m gpu without the two bench-configuration
#        tests), tests_all, bench, smoke, ktrace, pmc_traffic, pmc_g3 (SQ counters of the germline site kernel), pmc_a5
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
[ -f gpurun_commit.txt ] && cp gpurun_commit.txt $OUT/commit.txt
for step in "$@"; do
  t0=$(date +%s)
  case $step in
    e2e)
      SK_E2E_KEEP_DIR=$OUT/kept timeout 900 python bench.py --only e2e_germline > $OUT/e2e_germline.json 2> $OUT/e2e_germline.err
      rc=$?; echo "e2e_germline rc=$rc"; tail -c 600 $OUT/e2e_germline.err
      if [ $rc -ne 0 ]; then
        timeout 1500 python tools/diag/e2e_parity_hunt.py $OUT/hunt > $OUT/hunt.log 2>&1; tail -12 $OUT/hunt.log
        rm -rf $OUT/hunt/ref $OUT/hunt/*_tmp; find $OUT/hunt -name "genome.S1.vcf" -size +2M -delete
      fi ;;
    e2e_somatic)
      SK_E2E_KEEP_DIR=$OUT/kept timeout 900 python bench.py --only e2e_somatic > $OUT/e2e_somatic.json 2> $OUT/e2e_somatic.err
      echo "e2e_somatic rc=$?"; tail -c 400 $OUT/e2e_somatic.err ;;
    tests_kernels)
      timeout 900 python -m pytest tests/test_device_enumeration.py tests/test_gpu_parity.py tests/test_read_realign.py tests/test_pipeline.py tests/test_bam_feed.py tests/test_limits.py tests/test_pileup_stream.py -m gpu -x -q > $OUT/pytest_gpu_kernels.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_kernels.log
      tail -4 $OUT/pytest_gpu_kernels.log ;;
    tests)
      timeout 1200 python -m pytest tests -m gpu -x -q -k "not at_bench_configuration" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
      tail -5 $OUT/pytest_gpu.log ;;
    tests_all)
      timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
      tail -5 $OUT/pytest_gpu.log ;;
    bench)
      timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench.json ;;
    bench_kernels)
      timeout 600 python bench.py --no-cpu-baseline --e2e-bp 0 --e2e-somatic-bp 0 --realign-processes 0 > $OUT/bench_kernels.json 2> $OUT/bench_kernels.err; echo "bench_kernels rc=$?" ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log ;;
    ktrace)
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --e2e-bp 0 --e2e-somatic-bp 0 --realign-processes 0 > $OUT/ktrace.log 2>&1
      find $OUT/ktrace -name "*kernel_stats.csv" | head -2 ;;
    ktrace_a5)  # the headline leg alone: every flatten_score_kernel launch of this trace is the full-size one (the whole bench also
                # launches it from the realignment legs, at their jobs' sizes -- its average there is over all of them)
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace_a5 -o kt -- python bench.py --only a5 --steps 20 --warmup 3 > $OUT/ktrace_a5.log 2>&1
      find $OUT/ktrace_a5 -name "*kernel_stats.csv" | head -1 | xargs grep -i flatten_score | cut -c1-200 ;;
    pmc_traffic)
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-bp 0 --e2e-somatic-bp 0 --realign-processes 0 > $OUT/pmc_$c.log 2>&1
        timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_a5_$c -o pmc -- python bench.py --only a5 --steps 10 --warmup 2 > $OUT/pmc_a5_$c.log 2>&1
      done
      python tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json 2>$OUT/pmc_traffic.err; head -c 600 $OUT/pmc_traffic.json
      # (a bench step later in this visit quotes these passes: bench.py reads the newest profiles/*_pmc_traffic.json)
      [ -s $OUT/pmc_traffic.json ] && cp $OUT/pmc_traffic.json profiles/${TAG}_pmc_traffic.json ;;
    pmc_bound)  # SQ counters of every bench kernel (which unit binds it: tools/pmc_traffic.py `bound`); run BEFORE pmc_traffic in a visit
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
                 "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_sq$i -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --e2e-bp 0 --e2e-somatic-bp 0 --realign-processes 0 > $OUT/pmc_sq$i.log 2>&1
        timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_a5_sq$i -o pmc -- python bench.py --only a5 --steps 10 --warmup 2 > $OUT/pmc_a5_sq$i.log 2>&1
        tail -c 200 $OUT/pmc_sq$i.log
      done ;;
    pmc_g3)
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
                 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
        i=$((i+1))
        timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_g3_$i -o pmc -- python bench.py --only loci --loci 16777216 --steps 3 --warmup 1 > $OUT/pmc_g3_$i.log 2>&1
        tail -1 $OUT/pmc_g3_$i.log | head -c 300
      done
      python tools/diag/pmc_kernel_sums.py $OUT/pmc_g3_* > $OUT/pmc_g3.txt 2>&1; cat $OUT/pmc_g3.txt ;;
    realign)  # the whole-read legs of one process (sparse / dense scenarios): reads per second
      timeout 600 python bench.py --only realign --steps 5 > $OUT/realign.json 2>$OUT/realign.err
      python -c "import json;d=json.load(open('$OUT/realign.json'))['legs'];print({k:'%.3e reads/s' % (v['reads']/(v['t1']-v['t0'])) for k,v in d.items()})" ;;
    seeds)  # the drop-in (this box's GPU) against the reference on fresh seeded samples: $SEEDS_ARGS = "n first variant workers"
      timeout 900 python tools/fuzz/e2e_seeds.py ${SEEDS_ARGS:-8 201 amd 8} > $OUT/fuzz_e2e_seeds.txt 2>&1; tail -3 $OUT/fuzz_e2e_seeds.txt | cut -c1-300 ;;
    a5_waves)  # F5 with 1, 2, 4 waves (reads) to a block ($SK_F5_WAVES)
      for w in ${A5_WAVES:-1 2 4}; do
        SK_F5_WAVES=$w timeout 600 python -m pytest tests/test_device_enumeration.py -m gpu -x -q > $OUT/pytest_a5_w$w.log 2>&1; echo "waves=$w pytest rc=$? $(tail -1 $OUT/pytest_a5_w$w.log)"
        SK_F5_WAVES=$w timeout 300 python bench.py --only a5 --steps 10 --warmup 2 > $OUT/a5_w$w.json 2>$OUT/a5_w$w.err; echo "waves=$w: $(grep -o '"kernel_ms": [0-9.]*' $OUT/a5_w$w.json)"
        SK_F5_WAVES=$w SK_F5_TIMING=1 timeout 300 python bench.py --only a5 --a5-reads 4096 --steps 3 --warmup 1 > $OUT/a5_4096_w$w.json 2>$OUT/a5_4096_w$w.err; grep "f5-timing" $OUT/a5_4096_w$w.err | tail -4 | cut -c1-260
      done ;;
    a5_grid)  # F5's time against the blocks of its grid ($SK_F5_GRID; 0 = a block per read)
      timeout 600 python -m pytest tests/test_device_enumeration.py -m gpu -x -q > $OUT/pytest_a5.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_a5.log; tail -3 $OUT/pytest_a5.log
      for g in ${A5_GRIDS:-0 4096 3072 2048}; do
        SK_F5_GRID=$g timeout 300 python bench.py --only a5 --steps 10 --warmup 2 > $OUT/a5_grid$g.json 2>$OUT/a5_grid$g.err; echo "grid=$g: $(grep -o '"kernel_ms": [0-9.]*' $OUT/a5_grid$g.json)"
      done ;;
    pmc_a5)  # SQ counters of F5 (flatten_score_kernel) over bench.py --only a5
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
                 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"; do
        i=$((i+1))
        timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_a5_$i -o pmc -- python bench.py --only a5 --steps 3 --warmup 1 > $OUT/pmc_a5_$i.log 2>&1
        tail -1 $OUT/pmc_a5_$i.log | head -c 300
      done
      python tools/diag/pmc_kernel_sums.py $OUT/pmc_a5_* > $OUT/pmc_a5.txt 2>&1; awk '/^==/{print} /launches=/{p=0} /flatten_score/{p=1} p' $OUT/pmc_a5.txt ;;
    pmc_pileup)
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
        i=$((i+1))
        timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_pileup_$i -o pmc -- python bench.py --only pileup --steps 3 --warmup 1 > $OUT/pmc_pileup_$i.log 2>&1
        tail -1 $OUT/pmc_pileup_$i.log | head -c 300
      done
      python tools/diag/pmc_kernel_sums.py $OUT/pmc_pileup_* 2>&1 | grep -A10 "^pileup_\|^void pileup" > $OUT/pmc_pileup.txt; cat $OUT/pmc_pileup.txt ;;
    pmc_som)
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
                 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
                 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
        i=$((i+1))
        timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_som_$i -o pmc -- python bench.py --only somatic --steps 3 --warmup 1 > $OUT/pmc_som_$i.log 2>&1
        tail -1 $OUT/pmc_som_$i.log | head -c 300
      done
      python tools/diag/pmc_kernel_sums.py $OUT/pmc_som_* > $OUT/pmc_som.txt 2>&1; cat $OUT/pmc_som.txt ;;
    som)
      timeout 300 python bench.py --only somatic --steps 5 --warmup 2 > $OUT/som.json 2>$OUT/som.err; cat $OUT/som.json; tail -2 $OUT/som.err
      timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_somatic_tiers.py -m gpu -x -q -k "somatic or tier" 2>&1 | tail -2 ;;
    g3_variants)
      for v in ${G3_VARIANTS:-0 1}; do
        SK_G3_VARIANT=$v timeout 300 python bench.py --only loci --steps 5 --warmup 2 > $OUT/loci_v$v.json 2>$OUT/loci_v$v.err; echo "variant $v: $(cat $OUT/loci_v$v.json | head -c 400)"
      done ;;
    a5_ab)
      for f in 1 0; do
        SK_A5_FUSED=$f timeout 300 python bench.py --only a5 --steps 20 --warmup 3 > $OUT/a5_fused$f.json 2>$OUT/a5_fused$f.err; echo "fused=$f: $(cat $OUT/a5_fused$f.json | head -c 600)"
        SK_A5_FUSED=$f timeout 300 python bench.py --only a5 --a5-reads 65536 --steps 10 --warmup 2 > $OUT/a5_65536_fused$f.json 2>$OUT/a5_65536_fused$f.err; echo "fused=$f 2^16 reads: $(cat $OUT/a5_65536_fused$f.json | head -c 600)"
      done ;;
    a5)
      timeout 600 python -m pytest tests/test_device_enumeration.py -m gpu -x -q > $OUT/pytest_a5.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_a5.log; tail -3 $OUT/pytest_a5.log
      timeout 300 python bench.py --only a5 --steps 10 --warmup 2 > $OUT/a5.json 2>$OUT/a5.err; echo "a5: $(cat $OUT/a5.json | head -c 500)"
      SK_F5_TIMING=1 timeout 300 python bench.py --only a5 --a5-reads 4096 --steps 10 --warmup 2 > $OUT/a5_4096.json 2>$OUT/a5_4096.err; echo "a5 4096: $(cat $OUT/a5_4096.json | head -c 300)"; grep "f5-timing" $OUT/a5_4096.err | tail -9 ;;
    feed)
      timeout 600 python -m pytest tests/test_bam_feed.py -m gpu -x -q > $OUT/pytest_feed.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_feed.log; tail -4 $OUT/pytest_feed.log
      timeout 300 python bench.py --only feed_slice --steps 8 --warmup 2 > $OUT/feed_slice.json 2>$OUT/feed_slice.err; echo "feed_slice rc=$?"; cat $OUT/feed_slice.json; tail -3 $OUT/feed_slice.err ;;
    sweep)  # $SK_SWEEP_ONLY = names of tools/diag/e2e_sweep.py's settings, $SWEEP_ARGS = "bp segment_bp procs"
      timeout 1500 python tools/diag/e2e_sweep.py $OUT/e2e_sweep.json ${SWEEP_ARGS:-32000000 4000000 8} > $OUT/e2e_sweep.log 2>&1; echo "sweep rc=$?"; tail -c 3000 $OUT/e2e_sweep.log ;;
    tests_gvcf)
      timeout 900 python -m pytest tests/test_pileup_stream.py tests/test_gpu_parity.py tests/test_e2e_adapter.py tests/test_gvcf_block.py -m gpu -x -q -k "stream or gvcf or single_sample or germline_demo or plain_runs or kernel" > $OUT/pytest_gvcf.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gvcf.log
      tail -4 $OUT/pytest_gvcf.log ;;
    sharing)  # caller processes per GPU: $SK_SHARING_JOBS (default 8,12,16), ${SHARING_ARGS:-32000000 2000000}
      SK_SHARING_REF_JOBS=${SK_SHARING_REF_JOBS:-8,16} SK_SHARING_JOBS=${SK_SHARING_JOBS:-8,12,16} timeout 1200 python tools/diag/e2e_sharing.py ${SHARING_ARGS:-32000000 2000000} > $OUT/sharing.txt 2>&1; echo "sharing rc=$?"; cat $OUT/sharing.txt | tail -12 ;;
    enum_profile)
      timeout 900 python tools/diag/enum_job_profile.py $OUT/enum_profile > $OUT/enum_profile.log 2>&1; echo "enum_profile rc=$?"; tail -c 6000 $OUT/enum_profile.log ;;
    loci)
      timeout 300 python bench.py --only loci --steps 5 --warmup 2 > $OUT/loci.json 2>$OUT/loci.err; cat $OUT/loci.json ;;
    *) echo "unknown step $step" ;;
  esac
  echo "[$step: $(( $(date +%s) - t0 )) s]"
done
# keep the merge-back small: counter CSVs are large, the summaries are what is read
find $OUT -name "*counter_collection.csv" -size +20M -delete
du -sh $OUT
