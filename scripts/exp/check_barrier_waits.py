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
# Second check: in the 512-thread form (template argument NW = 8) the two wave groups take some barriers
# inside group-dependent branches (waves 4..7 one more in front of the loop, waves 0..3 one more behind it), which is only
# sound while the branch is UNIFORM — taken by whole waves — so that a barrier is never executed under a partial EXEC mask.
# LLVM lowers a branch it has proven uniform to s_cbranch_scc0 / scc1 (scalar compare) or s_cbranch_vccz / vccnz (a uniform
# value compared in the vector ALU, e.g. a 64-bit tile counter), and a possibly DIVERGENT one to an EXEC-masked region
# (s_and_saveexec + s_cbranch_execz): no forward branch that skips an s_barrier in those kernels may be of the EXEC kind.
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
vector_guarded = []
cur = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
    if not (cur and "infonce_pipe_kernel" in cur and cur.rstrip("_").find("ELi8EE") >= 0):
        continue
    m = re.match(r"\s*(s_cbranch_\w+)\s+(\.LBB\d+_\d+)", l)
    if m and labels.get(m.group(2), -1) > i:                          # forward branch: does it skip a barrier?
        skipped = any(re.match(r"\s*s_barrier", x) for x in lines[i + 1:labels[m.group(2)]])
        if skipped:
            agg["barriers_in_8wave_branches"] += 1
            if m.group(1) in ("s_cbranch_execz", "s_cbranch_execnz"):
                vector_guarded.append((cur[-60:], m.group(1), i + 1))
print(dict(agg))
for b in bad[:20]:
    print("UNWAITED", b)
for b in vector_guarded[:20]:
    print("BARRIER INSIDE AN EXEC-MASKED REGION", b)
sys.exit(1 if (bad or vector_guarded) else 0)
