rm -f gpurun_out/r02al_ab.log
for w in acro pend; do
  python tools/ab.py "lib:spec,lib:spec:-DNSG_SINCOS_STAGES=2" $w 3 1048576 300 >> gpurun_out/r02al_ab.log 2>&1
done
python tools/ab.py "lib:spec,lib:spec:-DNSG_SINCOS_STAGES=2" acro 3 262144 300 >> gpurun_out/r02al_ab.log 2>&1
python tools/ab.py "lib:spec,lib:spec:-DNSG_SINCOS_STAGES=2" acro 2 1048576 640 64 >> gpurun_out/r02al_ab.log 2>&1
python tools/ab.py "lib:spec,lib:spec:-DNSG_SINCOS_STAGES=2" pend 2 1048576 640 64 >> gpurun_out/r02al_ab.log 2>&1
cat gpurun_out/r02al_ab.log
