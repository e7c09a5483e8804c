#!/bin/bash
# the whole -m gpu suite under non-default settings of the round's knobs (every setting must give the same bits)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/knobs6
run() { name=$1; shift; env "$@" timeout -s KILL 1500 python3 -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_multirank.py > gpurun_out/knobs6/$name.txt 2>&1; echo "$name ($*): $(grep -E 'passed|failed|error' gpurun_out/knobs6/$name.txt | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/knobs6/$name.txt | head -5; }
if [ "${ONLY:-A}" = A ] || [ -z "${ONLY+x}" ]; then run A KG_SMALL_MAX=0 KG_COOP_TAIL=0 KG_BLOCKING_REDUCE_INLINE=0; fi
if [ "${ONLY:-B}" = B ]; then run B KG_SMALL_KT_FROM=0 KG_SMALL_MAX_FLIGHT=32768 KG_POOL_MB=64; fi
if [ "${ONLY:-C}" = C ]; then run C KG_SMALL_KT_FROM=8192 KG_SMALL_MAX_FLIGHT=0 KG_QUEUE_PLACEMENT=0 KG_G16_H_EARLY=0; fi
if [ "${ONLY:-D}" = D ]; then run D KG_SMALL_GLV=0 KG_G16_BLIND_EARLY=0; fi
if [ "${ONLY:-E}" = E ]; then run E KG_SMALL_GLV=2 KG_SMALL_KT_FROM=0 KG_G16_G2_GLV_MAX=20000; fi
