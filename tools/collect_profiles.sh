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
# 3. THE HEADLINE ALONE (bench.py --headline-only: pre-warm, warm-up, five timed rounds of the four-deep loop and nothing else): every
#    k_acc_tasks<Fq> launch of the trace is a full 2^20-pair launch of the pipelined loop, so the stats file's average for that kernel IS
#    roofline.kernel_ms of the line printed under the profiler (the pre-warm's cold-clock launches are a sixth of the launches)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_msm" -- python3 bench.py --headline-only --steps 200 --warmup 10 > "$O/msm_under_rocprof.json" 2> "$O/kt_msm.err"
python3 tools/kernel_table.py "$O/kt_msm" --per 1210 --last-of k_acc_tasks 1000 --title "bench.py --headline-only --steps 200 --warmup 10 under rocprofv3 --kernel-trace (1210 MSMs of 2^20 pairs, the last 1000 timed); us per unit = per MSM" > "$O/msm_kernel_table.txt" 2>&1
# 3b. the prover alone (bench.py --groth16-only: setup, 17 warm + 3 x 16 blocking proofs, 2 + 3 x 32 proofs two in flight): the phase table behind DESIGN.md section 10
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_g16" -- python3 bench.py --groth16-only --steps 16 > "$O/g16_under_rocprof.json" 2> "$O/kt_g16.err"
python3 tools/kernel_table.py "$O/kt_g16" --per 163 --split-grid k_acc_tasks --title "bench.py --groth16-only --steps 16 under rocprofv3 --kernel-trace: 2 setups + 163 proofs of 2^18 constraints (1 + 16 warm, 3 x 16 blocking, 2 + 3 x 32 two in flight: medians of three rounds); us per unit = per proof (setup kernels included in the list, not in a proof)" > "$O/g16_phase_table.txt" 2>&1
# 3c. the G2 MSM alone (2^18 pairs: 24 + 3 x 20 four in flight, 1 + 5 alone, 2 + 20 blocking in two window groups)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_g2" -- python3 bench.py --msm-g2-only --steps 20 --rounds 3 --no-cpu-baseline > "$O/g2_under_rocprof.json" 2> "$O/kt_g2.err"
python3 tools/kernel_table.py "$O/kt_g2" --split-grid k_acc_tasks --title "bench.py --msm-g2-only --steps 20 --rounds 3 under rocprofv3 --kernel-trace" > "$O/g2_kernel_table.txt" 2>&1
# 4. HBM traffic: one counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d "$O/pmc_$c" -- python3 bench.py --headline-only --steps 3 --warmup 1 --rounds 1 --prewarm 4 > "$O/pmc_$c.json" 2> "$O/pmc_$c.err"
done
# 5. the transform alone (2^22 forward NTT: the two k_ntt_tile kernels without the prover's 2^18 launches in the averages)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_ntt" -- python3 bench.py --ntt-only --ntt-variant dft --steps 300 > "$O/ntt_under_rocprof.json" 2> "$O/kt_ntt.err"
# 5b. the other three transforms of fft.rs:100-127, one run each
for v in idft coset_dft coset_idft; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_ntt_$v" -- python3 bench.py --ntt-only --ntt-variant $v --steps 300 > "$O/ntt_${v}_under_rocprof.json" 2> "$O/kt_ntt_$v.err"
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d "$O/pmcntt_$c" -- python3 bench.py --ntt-only --ntt-variant dft --steps 10 > "$O/pmcntt_$c.json" 2> "$O/pmcntt_$c.err"
done
# 6. SQ issue / wait counters of the same leg
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmcntt_SQ" -- python3 bench.py --ntt-only --ntt-variant dft --steps 10 > "$O/pmcntt_SQ.json" 2> "$O/pmcntt_SQ.err"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_SQ" -- python3 bench.py --headline-only --steps 3 --warmup 1 --rounds 1 --prewarm 4 > "$O/pmc_SQ.json" 2> "$O/pmc_SQ.err"
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
