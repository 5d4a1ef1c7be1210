#!/bin/bash
# SQ counter passes over create_flow of 4 full-disk frames: what holds the sampled row blur back?  tools/pmc_blur_r6.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; mkdir -p "$out"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -- python3 tools/fb_only.py 4 > "$out/pass$i.out" 2> "$out/pass$i.err" || { echo "pass $i failed"; tail -3 "$out/pass$i.err"; }
done
python tools/pmc_summarise.py "$out" k_fb_blur > "$out/summary.txt"
python tools/pmc_summarise.py "$out" k_fb_polyexp >> "$out/summary.txt"
rm -rf "$out"/pass*/
cat "$out/summary.txt"
