mkdir -p gpurun_out/r05/final
MVS_TEST_HOOKS=1 MVS_ONECALL_BANDS=3 timeout 1200 python tests/perf/fuzz_api.py 8000 10 60 > gpurun_out/r05/final/fuzz_api_banded.log 2>&1
timeout 1500 python tests/perf/stress_fx.py 700000 3000 > gpurun_out/r05/final/stress_fx_sep.log 2>&1
tail -n 1 gpurun_out/r05/final/fuzz_api_banded.log gpurun_out/r05/final/stress_fx_sep.log
