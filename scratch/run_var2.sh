#!/bin/bash
# usage: run_var2.sh <label> [env...]
export TMPDIR=/tmp
lbl=$1; shift
env "$@" true
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/var2_$lbl -- python3 scratch/var2.py run > gpurun_out/var2_$lbl.log 2>&1
echo "== $lbl $@"
python3 scratch/var2.py report gpurun_out/var2_$lbl | grep -E "f16  (all|none|stats)|bf16 stats"
