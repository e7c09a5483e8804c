#!/bin/bash
# kernel timelines of one blocking kg_msm at a few lengths (rocprofv3 --kernel-trace; tools/dbg/call_timeline.py prints the last call but one)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for lg in "$@"; do
  O=gpurun_out/tl_$lg; rm -rf "$O"; mkdir -p "$O"
  timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 tools/dbg/blocking_trace.py $lg 12 > "$O/run.txt" 2>&1
  echo "== 2^$lg"; grep blocking "$O/run.txt"; python3 tools/dbg/call_timeline.py "$O" ${KEY:-k_prep_scalars}
done
