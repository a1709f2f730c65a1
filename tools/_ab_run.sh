python -m pytest tests -m gpu -x -q 2>&1 | tail -3
rm -f gpurun_out/r02aj_ab.log
for w in c1 c2 acro pend c3 mcar; do
  python tools/ab.py "prev:spec,lib:spec" $w 2 1048576 300 >> gpurun_out/r02aj_ab.log 2>&1
done
NSG_GRID_CAP=1536 python tools/ab.py "lib:spec" acro 2 1048576 300 2>&1 | sed "s/^/grid1536 /" >> gpurun_out/r02aj_ab.log
python tools/ab.py "prev:spec,lib:spec" c2 2 524288 300 >> gpurun_out/r02aj_ab.log 2>&1
python tools/ab.py "prev:spec,lib:spec" c2 2 4194304 100 >> gpurun_out/r02aj_ab.log 2>&1
cat gpurun_out/r02aj_ab.log
