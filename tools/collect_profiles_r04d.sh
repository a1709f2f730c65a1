#!/bin/bash
# SQ counters of the NSG_F_LIBM_EXACT step kernels (Acrobot, Pendulum, C1) next to the default ones: what bounds an exact unit.
# Counters in their own passes, --kernel-trace only (no other trace domain).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04dprof; mkdir -p $O
for w in acro pend c1; do
  n=262144; [ $w = c1 ] && n=1048576
  for mode in default exact; do
    x=""; [ $mode = exact ] && x="--exact"
    timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq1_${w}_$mode -o p -- python3 tools/kbench.py --work $w --n $n --spec --iters 100 $x > $O/sq1_${w}_$mode.log 2>&1; echo "sq1 $w $mode rc=$?"
    timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2_${w}_$mode -o p -- python3 tools/kbench.py --work $w --n $n --spec --iters 100 $x > $O/sq2_${w}_$mode.log 2>&1; echo "sq2 $w $mode rc=$?"
    (echo "== $w $mode (tools/kbench.py --work $w --n $n --spec --iters 100 $x)"; python3 tools/pmc_sq.py $O/sq1_${w}_$mode | grep -A12 nsg_spec_step; python3 tools/pmc_sq.py $O/sq2_${w}_$mode | grep -A12 nsg_spec_step; grep "^$w " $O/sq1_${w}_$mode.log) >> $O/sq_counters_exact.txt
    rm -rf $O/sq1_${w}_$mode $O/sq2_${w}_$mode
  done
done
cat $O/sq_counters_exact.txt
