#!/bin/bash
# SQ counter passes over one batch of Farneback launches (+ one VR call): tools/pmc_kernels.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; mkdir -p "$out"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -- python3 tools/pmc_narrow.py --B 4 --levels 5 --vr 1 > "$out/pass$i.out" 2> "$out/pass$i.err" || exit 1
done
python tools/pmc_summarise.py "$out" > "$out/summary.txt"
