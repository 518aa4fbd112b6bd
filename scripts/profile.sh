#!/bin/bash
# Collects the rocprofv3 evidence for the headline kernel on the GPU box and writes the
# summaries under gpurun_out/prof_<tag>/ (copy what should be judged into profiles/).
#   usage (through gpurun):  bash scripts/profile.sh r01
# Passes: (1) --kernel-trace --stats  (2) --pmc FETCH_SIZE  (3) --pmc WRITE_SIZE  (4) --kernel-trace
# --stats of scripts/bench_configs.py -- counters in
# their own runs, as MI355X_MICROARCH.md prescribes (TCC has 4 slots: FETCH_SIZE 3 + WRITE_SIZE 2).
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu"   # the driver's command (BENCH_rNN.json: `python3 bench.py --gpus 1 --steps 20 --warmup 5`)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" $ARGS > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" --gpus 1 --no-pmc --no-cpu --steps 5 --warmup 2 > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$R/bench.py" --gpus 1 --no-pmc --no-cpu --steps 5 --warmup 2 > "$OUT/write.log" 2>&1
# (4) the other BASELINE configs' kernels (resampler, wide kernel, fused filter step, proposal draws)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/configs" -- python3 "$R/scripts/bench_configs.py" > "$OUT/configs.log" 2>&1
# (5), (6) matrix-core utilisation and LDS bank conflicts of every MFMA kernel
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d "$OUT/mfma" -- python3 "$R/scripts/mfma_util_probe.py" > "$OUT/mfma.log" 2>&1
rocprofv3 --pmc LdsBankConflict --kernel-trace --output-format csv -d "$OUT/lds" -- python3 "$R/scripts/mfma_util_probe.py" > "$OUT/lds.log" 2>&1
python3 "$R/scripts/summarize_pmc.py" "$OUT/mfma" MfmaUtil > "$OUT/pmc_mfma_util.md"
python3 "$R/scripts/summarize_pmc.py" "$OUT/lds" LdsBankConflict > "$OUT/pmc_lds_conflicts.md"
python3 "$R/scripts/summarize_profile.py" "$OUT" > "$OUT/summary.md"
cat "$OUT/summary.md"
