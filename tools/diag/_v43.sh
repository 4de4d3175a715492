#!/bin/bash
# seeded end-to-end runs on the GPU at the head (the push in two halves): germline, adversarial, tumour / normal, two-sample, hard
O=gpurun_out/r06_v43; mkdir -p $O
{
python tools/fuzz/e2e_seeds.py 24 701 amd 8 germline
python tools/fuzz/e2e_seeds.py 24 801 amd 8 adversarial
python tools/fuzz/e2e_seeds.py 16 901 amd 8 somatic
python tools/fuzz/e2e_seeds.py 12 1001 amd 6 multi
SK_FUZZ_HARD=1 python tools/fuzz/e2e_seeds.py 12 1101 amd 6
STRELKA_AMD_BROKER=0 python tools/fuzz/e2e_seeds.py 8 1201 amd 8 germline
STRELKA_AMD_BROKER=0 python tools/fuzz/e2e_seeds.py 8 1301 amd 8 somatic
} > $O/fuzz_e2e_seeds_gpu.txt 2>&1
grep -c identical $O/fuzz_e2e_seeds_gpu.txt; grep 'seeds identical' $O/fuzz_e2e_seeds_gpu.txt; grep -i 'differs\|error\|Traceback' $O/fuzz_e2e_seeds_gpu.txt | head
