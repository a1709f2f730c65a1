#!/bin/bash
# SQ counters of the fused policy rollout (nsg_spec_rollout_policy), one config per pass so that the shared kernel name is unambiguous.
# Counters in their own passes, --kernel-trace only (no other trace domain).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04cprof; mkdir -p $O
for w in c1 c3 acro; do
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq1_$w -o p -- python3 tools/policy_probe.py kernel:$w > $O/sq1_$w.log 2>&1; echo "sq1 $w rc=$?"
  timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2_$w -o p -- python3 tools/policy_probe.py kernel:$w > $O/sq2_$w.log 2>&1; echo "sq2 $w rc=$?"
  python3 tools/pmc_sq.py $O/sq1_$w > $O/sq_counters_policy_$w.txt; python3 tools/pmc_sq.py $O/sq2_$w >> $O/sq_counters_policy_$w.txt
  rm -rf $O/sq1_$w $O/sq2_$w
done
ls $O
