python -m pytest tests/test_gpu_env_streams.py tests/test_gpu_parity.py tests/test_gpu_rollout.py tests/test_gpu_reference_update_fns.py -x -q > gpurun_out/r02s_tests.log 2>&1; tail -3 gpurun_out/r02s_tests.log
rm -f gpurun_out/r02s_ab.log
for n in 1048576 4194304 16777216; do
  it=300; [ $n -gt 5000000 ] && it=60
  python tools/ab.py "lib:spec,lib:spec:-DNSG_MIN_WAVES=8" c1 3 $n $it >> gpurun_out/r02s_ab.log 2>&1
done
cat gpurun_out/r02s_ab.log
