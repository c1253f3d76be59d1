#!/bin/bash
# sweep the InfoNCE block-count knob (one process per setting: the knob is read once)
for b in 256 512 768 1024 1536 2048; do
  echo "== GCR_INFONCE_BLOCKS=$b"
  GCR_INFONCE_BLOCKS=$b python scripts/perf_infonce.py 2>&1 | grep -E "M=2048 N=1000000 d=64|M=8192|M=100000"
done
