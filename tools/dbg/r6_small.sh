#!/bin/bash
# round 6: the short-input MSM -- parity first, then the latency ladder
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_small; rm -rf "$O"; mkdir -p "$O"
timeout -s KILL 900 python3 -m pytest tests/test_gpu_small.py -x -q > "$O/tests.txt" 2>&1; tail -15 "$O/tests.txt"
timeout -s KILL 300 python3 bench.py --small-only > "$O/small.json" 2> "$O/small.err"; cat "$O/small.json"; tail -3 "$O/small.err"
