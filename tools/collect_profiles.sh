#!/bin/bash
# Collects the artefacts profiles/ holds for a round, on the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
# then, back in the container:  python tools/install_profiles.py r02
# Every leg is bounded by its own timeout; PMC passes run alone (never with a trace domain).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/collect
rm -rf "$O"; mkdir -p "$O"
# 1. the bench line itself (default flags: what the driver runs)
timeout -s KILL 400 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
# 2. the same command under the kernel trace (kernel averages must agree with the line's HIP-event figures)
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_bench" -- python3 bench.py --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/kt_bench.err"
# 3. MSM legs only (the kernels of the headline metric without NTT / Groth16 launches in the averages)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_msm" -- python3 bench.py --no-cpu-baseline --no-ntt --no-groth16 > "$O/msm_under_rocprof.json" 2> "$O/kt_msm.err"
# 4. HBM traffic: one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d "$O/pmc_$c" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-groth16 > "$O/pmc_$c.json" 2> "$O/pmc_$c.err"
done
ls -R "$O" | head -60
tail -c 600 "$O/bench.json"
