#!/bin/bash
# Collects the per-workload rocprofv3 evidence the profiles/ directory holds, on a GPU box:
#   scripts/profile_round.sh <out dir under gpurun_out/> [workload ...]
# per workload: one --kernel-trace --stats run of bench.py (kernel_stats.csv + bench.py's own line under the
# trace), two separate counter runs (FETCH_SIZE, WRITE_SIZE), reduced by scripts/traffic_from_pmc.py, and one run
# with the SQ issue counters, reduced by scripts/issue_from_pmc.py.
# Every GPU step runs under its own time limit and the script stops at the first failure.
set -e -o pipefail
out=${1:?out dir}; shift
wls=${@:-llama8b llama70b_shard c2}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"
out=$(cd "$out" && pwd)
export TMPDIR=/tmp
cd "$root"
for w in $wls; do
  flags="--workload $w --no-cpu-baseline --no-extra"
  echo "== $w: kernel trace" 
  timeout -k 10 280 rocprofv3 --kernel-trace --stats -f csv -d "$out/kt_$w" -o kt -- \
      python3 bench.py $flags --steps 5 --warmup 2 > "$out/${w}_bench_under_trace.json" 2> "$out/kt_$w.log"
  cp "$(find "$out/kt_$w" -name '*kernel_stats.csv' | head -1)" "$out/${w}_kernel_stats.csv"
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== $w: $c"
    timeout -s KILL 280 rocprofv3 --pmc $c -f csv -d "$out/pmc_${c}_$w" -o pmc -- \
        python3 bench.py $flags --steps 1 --warmup 1 > "$out/pmc_${c}_$w.json" 2> "$out/pmc_${c}_$w.log"
  done
  echo "== $w: SQ counters"
  timeout -s KILL 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      -f csv -d "$out/pmc_SQ_$w" -o pmc -- \
      python3 bench.py $flags --steps 1 --warmup 1 > "$out/pmc_SQ_$w.json" 2> "$out/pmc_SQ_$w.log"
  python3 scripts/issue_from_pmc.py "$(find "$out/pmc_SQ_$w" -name '*counter_collection.csv' | head -1)" "$w" > "$out/${w}_sq_issue.txt"
  rm -rf "$out/pmc_SQ_$w"
  python3 scripts/traffic_from_pmc.py "$(find "$out/pmc_FETCH_SIZE_$w" -name '*counter_collection.csv' | head -1)" \
      "$(find "$out/pmc_WRITE_SIZE_$w" -name '*counter_collection.csv' | head -1)" "$w" > "$out/${w}_pmc_traffic.txt"
  # the raw traces are large: only the summaries travel back
  rm -rf "$out/kt_$w" "$out/pmc_FETCH_SIZE_$w" "$out/pmc_WRITE_SIZE_$w"
done
cp profiles/traffic.json "$out/traffic.json"
cp profiles/issue.json "$out/issue.json"
echo done
