python -m pytest tests/test_gpu_env_streams.py tests/test_gpu_parity.py tests/test_gpu_rollout.py tests/test_gpu_specialized.py tests/test_gpu_harness.py tests/test_gpu_planning.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r02z_tests.log 2>&1; tail -2 gpurun_out/r02z_tests.log
python tools/kbench.py --work c1,c2,pend,acro,c3,mcar --spec > gpurun_out/r02z_step.log 2>&1; cat gpurun_out/r02z_step.log | cut -c1-150
python tools/kbench.py --work pend,acro,mcar --spec --rollout 64 2>&1 | cut -c1-120
python tools/kbench.py --work pend,acro --n 262144 --spec 2>&1 | cut -c1-150
