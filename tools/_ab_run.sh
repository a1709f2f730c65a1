rm -f gpurun_out/r02ad_ab.log
for w in c1 c2; do for n in 16384 65536 131072 262144 524288; do
  python tools/ab.py "lib:spec,lib:spec:-DNSG_CARTPOLE_INLANE=1" $w 2 $n 500 >> gpurun_out/r02ad_ab.log 2>&1
done; done
cat gpurun_out/r02ad_ab.log
