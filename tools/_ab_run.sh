NSG_SPEC_FLAGS="-DNSG_BALANCED=1" python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
rm -f gpurun_out/r02am_ab.log
for n in 524288 786432 1048576; do
  python tools/ab.py "lib:spec,lib:spec:-DNSG_BALANCED=1" c1 3 $n 400 >> gpurun_out/r02am_ab.log 2>&1
done
for w in c2 c3 pend mcar; do
  python tools/ab.py "lib:spec,lib:spec:-DNSG_BALANCED=1" $w 2 1048576 300 >> gpurun_out/r02am_ab.log 2>&1
done
cat gpurun_out/r02am_ab.log
