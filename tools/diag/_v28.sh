export SEEDS_ARGS="16 501 amd 8 adversarial"
bash tools/gpu_visit.sh r06_v28 tests_all smoke seeds ktrace pmc_bound pmc_traffic bench
O=gpurun_out/r06_v28
timeout 600 python tools/fuzz/e2e_seeds.py 8 601 amd 8 > $O/fuzz_e2e_seeds_germline.txt 2>&1; tail -1 $O/fuzz_e2e_seeds_germline.txt
timeout 600 python tools/fuzz/e2e_seeds.py 8 701 amd 8 somatic > $O/fuzz_e2e_seeds_somatic.txt 2>&1; tail -1 $O/fuzz_e2e_seeds_somatic.txt
timeout 600 python tools/fuzz/e2e_seeds.py 8 801 amd 8 multi > $O/fuzz_e2e_seeds_multi.txt 2>&1; tail -1 $O/fuzz_e2e_seeds_multi.txt
STRELKA_AMD_BROKER=1 timeout 1200 python -m pytest tests -m gpu -q -k "not at_bench_configuration" > $O/pytest_gpu_as_broker_client.txt 2>&1; tail -2 $O/pytest_gpu_as_broker_client.txt
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
