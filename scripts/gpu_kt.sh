#!/bin/bash
# rocprofv3 --kernel-trace --stats of one probe program on the GPU box, output under gpurun_out/<dir>.
#   usage (from the repo root on the box):  bash scripts/gpu_kt.sh <dir> <program.py> [args...]
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/$1"
shift
PROG="$1"
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/$PROG" "$@" > "$OUT/kt.log" 2>&1
find "$OUT" -name "*.db" -delete 2>/dev/null || true
tail -2 "$OUT/kt.log"
