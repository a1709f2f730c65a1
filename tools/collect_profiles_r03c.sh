#!/bin/bash
# Round-3 follow-up: HBM-side counters of C4's one-launch group step (nsg_spec_group, Pendulum 2^18 + Acrobot 2^18).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03dprof; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_c4_$c -o p -- python3 tools/kbench.py --work pend,acro --n 262144 --iters 60 --spec > $O/pmc_c4_$c.log 2>&1; echo "c4 $c rc=$?"
done
python tools/pmc_summarize.py $O/pmc_c4_FETCH_SIZE $O/pmc_c4_WRITE_SIZE nsg_spec_group r03_step_group_c4_pendulum_acrobot_specialised 262144 210 > $O/pmc_c4_summary.json
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cat $O/pmc_c4_summary.json
