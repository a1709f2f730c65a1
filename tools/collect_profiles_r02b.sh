cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02prof7; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats24 -o s -- python3 tools/kbench.py --work c1 --n 16777216 --iters 100 --spec > $O/kbench24_under_rocprof.log 2> $O/stats24.log
echo "stats24 rc=$?"
python tools/kbench.py --work c1,c2,c3,pend,acro,mcar --n 4194304 --iters 100 --spec > $O/kbench_2p22.log 2>&1
python - > $O/size_sweep.json <<'PY'
import json, subprocess, sys
rows = []
for lg in (12, 14, 16, 18, 19, 20, 21, 22, 23, 24):
    for extra, tag in (([], "nsg_step"), (["--rollout", "64"], "nsg_rollout_k64")):
        if tag != "nsg_step" and lg in (19, 21, 23):
            continue
        out = subprocess.run([sys.executable, "tools/kbench.py", "--work", "c1", "--n", str(1 << lg), "--spec", "--iters", "300" if lg <= 22 else "60"] + extra,
                             capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith("c1 "):
                rows.append({"envs_log2": lg, "api": tag, **json.loads(line.split(" ", 1)[1])})
print(json.dumps({"workload": "C1 (CartPole masspole IncrementUpdate / ContinuousScheduler), config-specialised kernels, one GPU box, one call",
                  "rows_bytes_per_env": 124, "infinity_cache_MiB": 256, "rows": rows}, indent=1))
PY
ls $O
