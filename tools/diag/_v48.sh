#!/bin/bash
# 176 seeded end-to-end runs on the GPU at the head of the round (as r06_v29's, fresh seeds)
O=gpurun_out/r06_v48; mkdir -p $O
timeout 900 python tools/fuzz/e2e_seeds.py 48 2001 amd 8 adversarial > $O/fuzz_adversarial.txt 2>&1; tail -1 $O/fuzz_adversarial.txt
timeout 900 python tools/fuzz/e2e_seeds.py 48 2101 amd 8 > $O/fuzz_germline.txt 2>&1; tail -1 $O/fuzz_germline.txt
timeout 900 python tools/fuzz/e2e_seeds.py 32 2201 amd 8 somatic > $O/fuzz_somatic.txt 2>&1; tail -1 $O/fuzz_somatic.txt
timeout 900 python tools/fuzz/e2e_seeds.py 24 2301 amd 8 multi > $O/fuzz_multi.txt 2>&1; tail -1 $O/fuzz_multi.txt
SK_FUZZ_HARD=1 timeout 900 python tools/fuzz/e2e_seeds.py 24 2401 amd 8 > $O/fuzz_hard.txt 2>&1; tail -1 $O/fuzz_hard.txt
cat $O/fuzz_adversarial.txt $O/fuzz_germline.txt $O/fuzz_somatic.txt $O/fuzz_multi.txt $O/fuzz_hard.txt > $O/fuzz_e2e_seeds_gpu_176_runs.txt
