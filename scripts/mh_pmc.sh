#!/bin/bash
# PMC passes over the Metropolis resampler: counters in their own runs, --kernel-trace only beside them.
#   bash scripts/mh_pmc.sh <tag>   -> gpurun_out/mh_pmc_<tag>/summary.md
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/mh_pmc_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "VALUBusy" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$R/scripts/mh_pmc_probe.py" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$OUT/failed.txt"
done
python3 "$R/scripts/summarize_mh_pmc.py" "$OUT" > "$OUT/summary.md"
cat "$OUT/summary.md"
