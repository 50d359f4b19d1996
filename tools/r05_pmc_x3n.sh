#!/bin/bash
# rocprofv3 --pmc passes of the narrow split-operand blocks at the inference shape (96 chunks)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
KB_N=3 PMC_PY=tools/kb_x3n.py bash tools/pmc_level.sh r05_x3n > /dev/null 2>&1
cat gpurun_out/pmc_r05_x3n/summary.txt | grep -E "^k_|FETCH|WAIT_ANY|WAVE_CYCLES|ACTIVE_INST_VALU|MFMA|INSTS_VALU|SQ_WAVES|BUSY_CYCLES|LDS_IDX|BANK"
python tools/kb_x3n.py
