#!/bin/bash
# One GPU-box pass that produces every rocprofv3 tree profiles/summarize.py reads.
#   usage (from the repo root on the box):  bash scripts/gpu_profile_round.sh prof_r02 [hbm|nce|all]
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md §HBM; gpurun refuses --pmc
# together with the hip/hsa trace domains); the program goes directly after `--`.
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/${1:-prof}"
WHAT="${2:-all}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = "hbm" ] || [ "$WHAT" = "all" ]; then
  for wl in cfg2 cfg4; do
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_$wl" -- python3 "$R/profiles/pmc_probe.py" --workload $wl > "$OUT/pmc_fetch_$wl.log" 2>&1
    echo "pmc fetch $wl done"
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_$wl" -- python3 "$R/profiles/pmc_probe.py" --workload $wl > "$OUT/pmc_write_$wl.log" 2>&1
    echo "pmc write $wl done"
  done
  for wl in cfg2c cfg4c; do
    for ro in before after; do
      flag=""; [ "$ro" = "after" ] && flag="--reorder"
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_${wl}_$ro" -- python3 "$R/profiles/pmc_probe.py" --workload $wl $flag > "$OUT/pmc_fetch_${wl}_$ro.log" 2>&1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_${wl}_$ro" -- python3 "$R/profiles/pmc_probe.py" --workload $wl $flag > "$OUT/pmc_write_${wl}_$ro.log" 2>&1
      echo "pmc $wl $ro done"
    done
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/bench_kt.json" 2> "$OUT/bench_kt.err"
  echo "kernel trace bench done"
fi
if [ "$WHAT" = "nce" ] || [ "$WHAT" = "all" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_nce" -- python3 "$R/profiles/infonce_probe.py" > "$OUT/kt_nce.log" 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_nce" -- python3 "$R/profiles/infonce_probe.py" > "$OUT/pmc_nce.log" 2>&1
  echo "infonce passes done"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_ncl" -- python3 "$R/profiles/ncl_step_probe.py" > "$OUT/kt_ncl.log" 2>&1
  echo "ncl step pass done"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_rank" -- python3 "$R/profiles/rank_probe.py" > "$OUT/kt_rank.log" 2>&1
  echo "rank pass done"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_lg" -- python3 "$R/profiles/lightgcn_step_probe.py" > "$OUT/kt_lg.log" 2>&1
  echo "lightgcn step pass done"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_bce" -- python3 "$R/profiles/bce_probe.py" > "$OUT/kt_bce.log" 2>&1
  echo "bce pass done"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_kme" -- python3 "$R/profiles/kmeans_estep_probe.py" > "$OUT/kt_kme.log" 2>&1
  echo "k-means e_step pass done"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_cfg5" -- python3 "$R/bench.py" --workload cfg5 --steps 5 --warmup 2 > "$OUT/bench_cfg5_kt.json" 2> "$OUT/bench_cfg5_kt.err"
  echo "cfg5 pass done"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_gcl" -- python3 "$R/profiles/gcl_step_probe.py" > "$OUT/kt_gcl.log" 2>&1
  echo "gcl step pass done"
fi
# keep the merge-back small: only the CSV summaries are read afterwards
find "$OUT" -name "*.db" -delete 2>/dev/null || true
du -sh "$OUT"
