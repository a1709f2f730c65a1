#!/bin/bash
# Round-3 profile collection (GPU box; run from the repo root through gpurun).  Every profiler invocation is bounded by `timeout`;
# counters are collected in their own passes (--pmc with --kernel-trace only).  Summaries are copied into profiles/ afterwards by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03prof; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-all-configs > $O/bench_under_rocprof.json 2> $O/stats.log
echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hbm-resident --no-all-configs > $O/pmc_$c.log 2>&1; echo "$c rc=$?"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc24_$c -o p -- python3 tools/kbench.py --work c1 --n 16777216 --iters 20 --spec > $O/pmc24_$c.log 2>&1; echo "$c 2^24 rc=$?"
done
for w in c2 c3 pend acro; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${w}_$c -o p -- python3 tools/kbench.py --work $w --iters 60 --spec > $O/pmc_${w}_$c.log 2>&1; echo "$w $c rc=$?"
  done
done
for c in FETCH_SIZE WRITE_SIZE; do   # Pendulum without episode accounting (BASELINE's C4 does not ask for returns)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_pendnt_$c -o p -- python3 tools/kbench.py --work pend --iters 60 --spec --no-track > $O/pmc_pendnt_$c.log 2>&1; echo "pend no-track $c rc=$?"
done
for w in c1 acro; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq1_$w -o p -- python3 tools/kbench.py --work $w --iters 60 --spec > $O/sq1_$w.log 2>&1; echo "sq1 $w rc=$?"
  timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2_$w -o p -- python3 tools/kbench.py --work $w --iters 60 --spec > $O/sq2_$w.log 2>&1; echo "sq2 $w rc=$?"
done
python tools/measure_all.py > $O/kbench_all_configs.json 2> $O/kbench_all.err
python tools/kbench.py --work c1,c2,c3,pend,acro,mcar --n 4194304 --iters 100 --spec > $O/kbench_2p22.log 2>&1
# 2^24: the kernel-trace average and kbench / bench figures of the SAME box, three repetitions each side of the profiled run
python tools/kbench.py --work c1 --n 16777216 --iters 100 --spec > $O/kbench24_before.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats24 -o s -- python3 tools/kbench.py --work c1 --n 16777216 --iters 100 --spec > $O/kbench24_under_rocprof.log 2> $O/stats24.log
echo "stats24 rc=$?"
python tools/kbench.py --work c1 --n 16777216 --iters 100 --spec > $O/kbench24_after.log 2>&1
python - > $O/size_sweep.json <<'PY'
import json, subprocess, sys
rows = []
for work, sizes in (("c1", (12, 14, 16, 18, 19, 20, 21, 22, 23, 24)), ("c2", (14, 15, 16, 17, 18))):
    for lg in sizes:
        for extra, tag in (([], "nsg_step"), (["--rollout", "64"], "nsg_rollout_k64")):
            if tag != "nsg_step" and lg in (15, 17, 19, 21, 23):
                continue
            out = subprocess.run([sys.executable, "tools/kbench.py", "--work", work, "--n", str(1 << lg), "--spec", "--iters", "300" if lg <= 22 else "60"] + extra,
                                 capture_output=True, text=True).stdout
            for line in out.splitlines():
                if line.startswith(work + " "):
                    rows.append({"workload": work, "envs_log2": lg, "api": tag, **json.loads(line.split(" ", 1)[1])})
print(json.dumps({"workloads": "C1 (CartPole masspole IncrementUpdate / ContinuousScheduler) and C2 (gravity RandomWalk / PeriodicScheduler(3)), config-specialised kernels, one GPU box, one call",
                  "infinity_cache_MiB": 256, "rows": rows}, indent=1))
PY
ls $O
