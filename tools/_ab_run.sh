python -m pytest tests/test_gpu_rollout.py tests/test_gpu_harness.py tests/test_gpu_planning.py tests/test_gpu_random_configs.py tests/test_gpu_env_streams.py -x -q > gpurun_out/r02x_tests.log 2>&1; tail -2 gpurun_out/r02x_tests.log
python tools/kbench.py --work c1,c2,pend,acro,c3 --spec --rollout 64 > gpurun_out/r02x_rollout.log 2>&1; cat gpurun_out/r02x_rollout.log | cut -c1-150
python tools/kbench.py --work c1,c2,pend,acro,c3,mcar --spec > gpurun_out/r02x_step.log 2>&1; cat gpurun_out/r02x_step.log | cut -c1-150
