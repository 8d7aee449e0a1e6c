mkdir -p gpurun_out/r5/mvn
timeout 1500 python -m pytest tests/test_gpu_mvn.py -q -x 2>&1 | tail -5 > gpurun_out/r5/mvn/mvn_tests.txt
timeout 600 python tools/r5/mvn_spill_time.py > gpurun_out/r5/mvn/mvn_spill_time.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -k "gp_marginal or very_long" --durations=5 2>&1 | tail -14 > gpurun_out/r5/mvn/marginal_tests.txt
