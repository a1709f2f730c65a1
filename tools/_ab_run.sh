cd $GRAFT_REPO_ROOT
run() { # tag flags counters
  NSG_SPEC_FLAGS="$2" timeout -k 5 150 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d gpurun_out/sq_$1 -o p -- python3 tools/kbench.py --work c1 --n 1048576 --iters 100 --spec > gpurun_out/sq_$1.log 2>&1
  echo "$1 rc=$?"
}
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
run base1 "" "$P1" && run base2 "" "$P2" && run inl1 "-DNSG_X_INLINE_RESET" "$P1" && run inl2 "-DNSG_X_INLINE_RESET" "$P2" && run fake1 "-DNSG_X_FAKE_STEP" "$P1" && run fake2 "-DNSG_X_FAKE_STEP" "$P2"
ls gpurun_out/sq_base1
