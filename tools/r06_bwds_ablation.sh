cd $GRAFT_REPO_ROOT
for v in hip abl1 abl2 abl4 abl6; do
  export TTRAP_LIB=libttrap_$v.so
  echo "== $v"; KB_C=16 KB_D=1,2,3 KB_WHAT=bwd KB_N=10 python tools/kb_level.py 2>&1 | grep bwd
  KB_C=16 KB_D=1 KB_WHAT=bwd KB_N=3 bash tools/pmc_level.sh r06_$v > /dev/null 2>&1
  grep -A17 "k_wrb_bwds" gpurun_out/pmc_r06_$v/summary.txt | grep -E "k_wrb_bwds|FETCH|BANK|IDX_ACTIVE|BUSY_CYCLES|INSTS_VALU|INSTS_LDS|WAIT_ANY|WAVE_CYCLES|ACTIVE_INST_LDS|WAIT_INST_LDS"
done
