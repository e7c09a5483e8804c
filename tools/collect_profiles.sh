#!/bin/bash
# Collects the artefacts profiles/ holds for a round, on the GPU box:
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh'
# then, back in the container:  python tools/install_profiles.py r04
# Every leg is bounded by its own timeout; PMC passes run alone (never with a trace domain).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/collect
rm -rf "$O"; mkdir -p "$O"
# 1. the bench line itself (default flags: what the driver runs)
timeout -s KILL 700 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
# 2. the same command under the kernel trace (kernel averages must agree with the line's HIP-event figures)
timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_bench" -- python3 bench.py --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/kt_bench.err"
# 3. MSM legs only (the kernels of the headline metric without NTT / Groth16 launches in the averages)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_msm" -- python3 bench.py --no-cpu-baseline --no-ntt --no-groth16 --no-nova --no-skew > "$O/msm_under_rocprof.json" 2> "$O/kt_msm.err"
# 4. HBM traffic: one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d "$O/pmc_$c" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-groth16 --no-nova --no-skew > "$O/pmc_$c.json" 2> "$O/pmc_$c.err"
done
# 5. the transform alone (2^22 forward NTT: the two k_ntt_tile kernels without the prover's 2^18 launches in the averages)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_ntt" -- python3 bench.py --ntt-only --steps 300 > "$O/ntt_under_rocprof.json" 2> "$O/kt_ntt.err"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d "$O/pmcntt_$c" -- python3 bench.py --ntt-only --steps 10 > "$O/pmcntt_$c.json" 2> "$O/pmcntt_$c.err"
done
# 6. SQ issue / wait counters of the same leg
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmcntt_SQ" -- python3 bench.py --ntt-only --steps 10 > "$O/pmcntt_SQ.json" 2> "$O/pmcntt_SQ.err"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_SQ" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ntt --no-groth16 --no-nova --no-skew > "$O/pmc_SQ.json" 2> "$O/pmc_SQ.err"
# 7. the multiplier and the transform's register pass alone (tools/ubench; built in-tree before the call)
if [ -x tools/ubench/mul_rate ]; then
  ( cd tools/ubench; echo "== fp29.h as shipped (one multiply-accumulate chain per column)"; timeout -s KILL 120 ./mul_rate | grep -E "SIMD=(1|2|4|8) "
    if [ -x ./mul_rate_c ]; then echo "== -DKG_NO_ASM_MAC (the columns left to the compiler)"; timeout -s KILL 120 ./mul_rate_c | grep -E "SIMD=(1|2|4|8) "; fi ) > "$O/mul_rate.txt" 2>&1
fi
if [ -x tools/ubench/mfma_const_mul ]; then ( cd tools/ubench; timeout -s KILL 120 ./mfma_const_mul ) > "$O/mfma_const_mul.txt" 2>&1; fi
if [ -x tools/ubench/group_scatter ]; then ( cd tools/ubench; timeout -s KILL 120 ./group_scatter 24 20; timeout -s KILL 120 ./group_scatter 20 16 ) > "$O/group_scatter.txt" 2>&1; fi
if [ -x tools/ubench/ntt_pass_rate ]; then ( cd tools/ubench; timeout -s KILL 120 ./ntt_pass_rate ) > "$O/ntt_pass_rate.txt" 2>&1; fi
ls -R "$O" | head -80
tail -c 600 "$O/bench.json"
