#!/bin/bash
# Round-4 profile collection (GPU box; run from the repo root through gpurun).  Every profiler invocation is bounded by `timeout`;
# counters are collected in their own passes (--pmc with --kernel-trace only).  Summaries are copied into profiles/ afterwards by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04prof; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>/dev/null
NSG_NO_HIPRTC=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_no_hiprtc.json 2>/dev/null
NSG_PREBUILT_DIR=off NSG_SPEC_CACHE=off python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_hiprtc_units.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-all-configs > $O/bench_under_rocprof.json 2> $O/stats.log
echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hbm-resident --no-all-configs > $O/pmc_$c.log 2>&1; echo "$c rc=$?"
done
python tools/measure_all.py > $O/kbench_all_configs.json 2> $O/kbench_all.err
timeout -k 10 200 python tools/resident_probe.py c2 65536 4000 > $O/resident_probe_c2.json 2> $O/resident_probe.err
timeout -k 10 200 python tools/resident_probe.py c1 65536 4000 > $O/resident_probe_c1.json 2>> $O/resident_probe.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_resident -o s -- python3 tools/kbench.py --work c2 --n 65536 --resident 4000 --spec > $O/resident_under_rocprof.log 2> $O/stats_resident.log
echo "resident stats rc=$?"
ls $O
