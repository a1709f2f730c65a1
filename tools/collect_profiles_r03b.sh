#!/bin/bash
# Round-3 follow-up collection after the grid envs' table hint (status byte bits 1-7): C3's counters and times, every config's times,
# the bench line.  Same rules as collect_profiles_r03.sh.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bprof; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_c3_$c -o p -- python3 tools/kbench.py --work c3 --iters 60 --spec > $O/pmc_c3_$c.log 2>&1; echo "c3 $c rc=$?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o s -- python3 tools/kbench.py --work c3 --iters 300 --spec > $O/kbench_c3_under_rocprof.log 2> $O/stats_c3.log
echo "stats c3 rc=$?"
python tools/measure_all.py > $O/kbench_all_configs.json 2> $O/kbench_all.err
python bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>/dev/null
python tools/pmc_summarize.py $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE nsg_spec_step r03b_step_kernel_c3_frozenlake_specialised_table_hint 1048576 96 > $O/pmc_c3_summary.json
cp profiles/pmc_traffic.json $O/pmc_traffic.json
ls $O
