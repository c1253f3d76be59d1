#!/usr/bin/env python3
"""ISA check behind DESIGN 4.2b's ring analysis: in every instantiation of infonce_pipe_kernel, no LDS instruction may
stand between the last `s_waitcnt ... lgkmcnt(0)` and an `s_barrier` (every read has returned and every write has landed
when a barrier separates it from its counterpart).  Textual scan of the gfx950 assembly:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I recommendation_amd/csrc -c recommendation_amd/csrc/gcr_infonce.hip \\
          -o /tmp/isa/infonce.o -save-temps=obj
    python scripts/exp/check_barrier_waits.py /tmp/isa/gcr_infonce-hip-amdgcn-amd-amdhsa-gfx950.s"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
cur, agg, bad = None, collections.Counter(), []
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
        if "infonce_pipe_kernel" in cur:
            agg["kernels"] += 1
    if cur and "infonce_pipe_kernel" in cur and re.match(r"\s*s_barrier", l):
        agg["barriers"] += 1
        j = i - 1
        while j > 0:
            t = lines[j].strip()
            if re.match(r"^_Z\w+:", t) or t.startswith("s_barrier") or (t.startswith("s_waitcnt") and "lgkmcnt(0)" in t):
                break
            if t.startswith("ds_"):
                agg["ds_between_wait_and_barrier"] += 1
                bad.append((cur[-50:], t.split()[0], i - j))
                break
            j -= 1
print(dict(agg))
for b in bad[:20]:
    print("UNWAITED", b)
sys.exit(1 if bad else 0)
