#!/bin/bash
# Device-side time per step of the sharded filter loop (rocprofv3 kernel trace: first step kernel's start to the last
# one's end), one device against 2 / 4 / 8 shards rehearsed on device 0.   bash scripts/multi_filter_profile.sh > out.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1000000 2 100" "1000000 8 50" "200000 64 20" "1000 2 1000"; do
  for nd in 0 2 4 8; do
    rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 scripts/multi_filter_trace.py $cfg $nd > /tmp/out.txt 2>&1
    echo "== N d T = $cfg, shards = $nd (0: cusmc_pf_run_host on one device)"; grep -E "^[0-9]" /tmp/out.txt | tail -1
    python3 scripts/kernel_gaps.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) 1 | grep -E "pf_step_kernel|metropolis|loop span|gather|propagate_mfma|logpdf" | grep -v "^q"
  done
done
