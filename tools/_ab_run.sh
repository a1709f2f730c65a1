python -m pytest tests/test_gpu_env_streams.py tests/test_gpu_parity.py tests/test_gpu_rollout.py tests/test_gpu_specialized.py -x -q 2>&1 | tail -1
NSG_HELPER_WAVE=1 NSG_SPEC_FLAGS="-DNSG_HELPER_LANES=64 -DNSG_MIN_WAVES=8" python -m pytest tests/test_gpu_env_streams.py tests/test_gpu_specialized.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -1
rm -f gpurun_out/r02ab_ab.log
for n in 262144 524288 1048576 4194304 16777216; do
  it=300; [ $n -gt 5000000 ] && it=60
  python tools/ab.py "lib:spec,lib:spec:-DNSG_MIN_WAVES=8" c1 2 $n $it >> gpurun_out/r02ab_ab.log 2>&1
  NSG_HELPER_WAVE=1 python tools/ab.py "lib:spec:-DNSG_HELPER_LANES=64 -DNSG_MIN_WAVES=8" c1 2 $n $it >> gpurun_out/r02ab_ab.log 2>&1
done
cat gpurun_out/r02ab_ab.log
