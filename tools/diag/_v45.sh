#!/bin/bash
# after SK_MAX_SAMPLES 4 -> 8 (frames and table entries of the device search grew) and the xwide allele groups: the GPU suite, the kernel
# legs that touch them, seeded end-to-end runs
O=gpurun_out/r06_v45; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --only a5 --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-300 | tee $O/a5.json
python bench.py --only realign --steps 5 > $O/realign.json 2>$O/realign.err
python -c "import json;d=json.load(open('$O/realign.json'))['legs'];print({k:'%.3e reads/s' % (v['reads']/(v['t1']-v['t0'])) for k,v in d.items()})" | tee $O/realign.txt
{
python tools/fuzz/e2e_seeds.py 12 1401 amd 8 germline
python tools/fuzz/e2e_seeds.py 12 1501 amd 8 adversarial
python tools/fuzz/e2e_seeds.py 8 1601 amd 8 somatic
python tools/fuzz/e2e_seeds.py 8 1701 amd 8 multi
} > $O/fuzz_e2e_seeds_gpu.txt 2>&1
grep 'seeds identical' $O/fuzz_e2e_seeds_gpu.txt
