#!/bin/bash
# P2 with four waves to a block of 64 loci: the pileup tests, the stream's, the end-to-end adapter tests; the window's kernels; the legs
O=gpurun_out/r06_v51; mkdir -p $O
timeout 1500 python -m pytest tests/test_pileup.py tests/test_pileup_stream.py tests/test_gpu_parity.py tests/test_e2e_adapter.py tests/test_somatic_tiers.py tests/test_full_size.py -m gpu -x -q > $O/pytest_pileup.log 2>&1; tail -3 $O/pytest_pileup.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python tools/diag/stream_window_trace.py 2>&1 | grep rep
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp $f $O/stream_window_kernel_stats.csv
grep -E 'pileup_column|germline_site_fused' $O/stream_window_kernel_stats.csv | cut -c1-60,100-200
python bench.py --only pileup --steps 10 --warmup 2 2>/dev/null | tail -1
