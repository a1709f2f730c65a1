rm -f gpurun_out/r02ak_ab.log
for w in acro pend c1 c2 mcar; do
  python tools/ab.py "lib:spec,lib:spec:-DNSG_SINCOS_FMA=1" $w 3 1048576 300 >> gpurun_out/r02ak_ab.log 2>&1
done
python tools/ab.py "lib:spec,lib:spec:-DNSG_SINCOS_FMA=1" acro 2 262144 300 >> gpurun_out/r02ak_ab.log 2>&1
python tools/ab.py "lib:spec,lib:spec:-DNSG_SINCOS_FMA=1" acro 2 1048576 640 64 >> gpurun_out/r02ak_ab.log 2>&1
python tools/ab.py "lib:spec,lib:spec:-DNSG_SINCOS_FMA=1" c1 2 1048576 640 64 >> gpurun_out/r02ak_ab.log 2>&1
python tools/ab.py "lib:spec,lib:spec:-DNSG_SINCOS_FMA=1" pend 2 1048576 640 64 >> gpurun_out/r02ak_ab.log 2>&1
cat gpurun_out/r02ak_ab.log
