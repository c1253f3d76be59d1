#!/bin/bash
# Rehearsal of the N > 1 bench paths on a ONE-GPU box: N = 1 through the sharded code, then two ranks sharing
# the GPU over gloo (RCCL refuses two ranks on one device) — weak, strong and config 5.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/dist"
mkdir -p "$O"
cd "$R"
run() { name=$1; shift; echo "== $name"; "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "rc=$?"; tail -c 400 "$O/$name.err"; python3 - "$O/$name.json" <<'PY'
import json, sys
try:
    b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: b.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling", "dist_backend", "dist_world", "infonce_pairs_per_s")})
    print("  workload:", b["config"]["workload"][:160])
    if b.get("gcl_step"): print("  gcl:", {k: v for k, v in b["gcl_step"].items() if k != "note"})
    if b.get("extra") and "fwd_bwd_ms" in b["extra"]: print("  fwd_bwd_ms", b["extra"]["fwd_bwd_ms"])
except Exception as e:
    print("no json line:", e)
PY
}
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
GCR_BENCH_FORCE_DIST=1 GCR_BENCH_FORCE_COLLECTIVES=1 run n1_rccl python3 bench.py --steps 5 --warmup 2
GCR_BENCH_FORCE_COLLECTIVES=1 run cfg5_n1_rccl python3 bench.py --workload cfg5 --steps 5 --warmup 2
GCR_BENCH_REHEARSE_ONE_GPU=1 run n2_weak $TR --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1
GCR_BENCH_REHEARSE_ONE_GPU=1 run n2_strong $TR --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --scaling strong --workload cfg2 --no-extra
run cfg5_n1 python3 bench.py --workload cfg5 --steps 5 --warmup 2
GCR_BENCH_REHEARSE_ONE_GPU=1 run cfg5_n2 $TR --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --workload cfg5
