#!/bin/bash
# Builds libgcr from the sources of a git revision into build/<name>.so (git-ignored, travels to the GPU box with the
# snapshot) for in-process A/B runs: scripts/perf_infonce.py --lib build/<name>.so.  Never writes into recommendation_amd/.
#   usage: scripts/build_ab.sh <git-rev> <name>
set -e -o pipefail
REV="${1:?git revision}"; NAME="${2:?output name}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP="$(mktemp -d /tmp/gcr_ab_XXXX)"
git -C "$ROOT" archive "$REV" recommendation_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$ROOT/build"
OBJS=()
for src in "$TMP"/recommendation_amd/csrc/*.hip "$TMP"/recommendation_amd/csrc/*.cpp; do
  obj="$TMP/$(basename "$src").o"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I "$TMP/include" \
    -I "$TMP/recommendation_amd/csrc" -c "$src" -o "$obj" &
  OBJS+=("$obj")
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "${OBJS[@]}" -o "$ROOT/build/$NAME.so"
rm -rf "$TMP"
echo "$ROOT/build/$NAME.so"
