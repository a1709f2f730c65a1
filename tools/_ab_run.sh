rm -f gpurun_out/r02n_ab.log
for n in 1048576 4194304 16777216; do
  it=300; [ $n -gt 5000000 ] && it=60
  python tools/ab.py "lib:spec,lib:spec:-DNSG_X_NO_RECORD_IO,lib:spec:-DNSG_X_NO_RECORD_IO -DNSG_X_RESET_INLANE,lib:spec:-DNSG_X_INLINE_RESET" c1 2 $n $it >> gpurun_out/r02n_ab.log 2>&1
done
python tools/ab.py "lib:spec,lib:spec:-DNSG_X_NO_RECORD_IO" c1 2 1048576 300 >> gpurun_out/r02n_ab.log 2>&1
cat gpurun_out/r02n_ab.log
