#!/bin/bash
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/gcl_kt"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export GCR_BENCH_FORCE_DIST=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*.db" -delete 2>/dev/null || true
tail -c 600 "$OUT/bench.json"
