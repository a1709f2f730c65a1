#!/bin/bash
# Round-4 second collection (GPU box; through gpurun from the repo root): the fused policy rollouts (nsg_rollout_policy).
# Profiler runs are bounded by `timeout`; summaries are copied into profiles/ afterwards by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04bprof; mkdir -p $O
# kernel level + closed loops on the PREBUILT units (no runtime compiler); the planner's copies are planning-copy configurations
# (NSG_F_SIM_ENV), which nobody prebuilds: their units come from hiprtc
NSG_NO_HIPRTC=1 python3 tools/policy_probe.py kernel closed > $O/policy_probe_prebuilt_units.json 2> $O/policy_probe.err
echo "probe (prebuilt) rc=$?"
python3 tools/policy_probe.py > $O/policy_probe.json 2>> $O/policy_probe.err
echo "probe rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 tools/policy_probe.py kernel closed > $O/policy_probe_under_rocprof.json 2> $O/stats.log
echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_planner -o s -- python3 tools/policy_probe.py planner > $O/policy_probe_planner_under_rocprof.json 2> $O/stats_planner.log
echo "planner stats rc=$?"
ls $O
