rm -f gpurun_out/r02k_ab.log
for cap in 1024 1536 2048 3072 4096; do echo "cap $cap" >> gpurun_out/r02k_ab.log; NSG_GRID_CAP=$cap python tools/ab.py "lib:spec,lib:spec:-DNSG_X_INLINE_RESET" c1 2 1048576 300 >> gpurun_out/r02k_ab.log 2>&1; done
for cap in 2048 4096 16384; do echo "cap $cap" >> gpurun_out/r02k_ab.log; NSG_GRID_CAP=$cap python tools/ab.py "lib:spec" c1 2 4194304 100 >> gpurun_out/r02k_ab.log 2>&1; done
for cap in 2048 4096; do echo "cap $cap" >> gpurun_out/r02k_ab.log; NSG_GRID_CAP=$cap python tools/ab.py "lib:spec" c3 2 1048576 300 >> gpurun_out/r02k_ab.log 2>&1; done
cat gpurun_out/r02k_ab.log
