#!/bin/bash
# PMC passes over the wide log-pdf kernel (and the tile kernel at d = 128, 176 beside it): where a wave's cycles go.
# Counters in their own runs, --kernel-trace only beside them.   bash scripts/wide_pmc.sh <tag> -> gpurun_out/wide_pmc_<tag>/summary.md
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/wide_pmc_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "MfmaUtil" "TA_BUSY_avr TD_BUSY_avr TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$R/scripts/wide_pmc_probe.py" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$OUT/failed.txt"
done
python3 "$R/scripts/summarize_mh_pmc.py" "$OUT" logpdf_mfma > "$OUT/summary.md"
cat "$OUT/summary.md"
