O=gpurun_out/r06_v29; mkdir -p $O
timeout 900 python tools/fuzz/e2e_seeds.py 48 1001 amd 8 adversarial > $O/fuzz_adversarial.txt 2>&1; tail -1 $O/fuzz_adversarial.txt
timeout 900 python tools/fuzz/e2e_seeds.py 48 1101 amd 8 > $O/fuzz_germline.txt 2>&1; tail -1 $O/fuzz_germline.txt
timeout 900 python tools/fuzz/e2e_seeds.py 32 1201 amd 8 somatic > $O/fuzz_somatic.txt 2>&1; tail -1 $O/fuzz_somatic.txt
timeout 900 python tools/fuzz/e2e_seeds.py 24 1301 amd 8 multi > $O/fuzz_multi.txt 2>&1; tail -1 $O/fuzz_multi.txt
SK_FUZZ_HARD=1 timeout 900 python tools/fuzz/e2e_seeds.py 24 1401 amd 8 > $O/fuzz_hard.txt 2>&1; tail -1 $O/fuzz_hard.txt
for k in 1 2; do SK_TEST_SEED_OFFSET=$((k*1000)) timeout 900 python -m pytest tests -m gpu -q -x -k "not at_bench_configuration and not golden and not reference" > $O/pytest_shifted_$k.txt 2>&1; tail -1 $O/pytest_shifted_$k.txt; done
