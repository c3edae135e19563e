set -e
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py tests/test_gpu_lsi.py -q -m gpu -k "regulariz or reference_lexlse_suite" -x > gpurun_out/reg_tests.log 2>&1
for w in ${REG_WAVES:-8 4}; do echo "== LEXLS_REG_LDS_WAVES=$w" >> gpurun_out/reg_ab.log; LEXLS_REG_LDS_WAVES=$w timeout -k 10 120 python scripts/time_reg.py 4096 | grep -v "type 0" >> gpurun_out/reg_ab.log 2>&1; done
