#!/bin/bash
# the whole -m gpu suite under three non-default knob sets (every setting must give the same bits)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/knobs
run() { name=$1; shift; env "$@" timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_bench_multirank.py > gpurun_out/knobs/$name.txt 2>&1; echo "$name: $(grep -E 'passed|failed|error' gpurun_out/knobs/$name.txt | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/knobs/$name.txt | head -5; }
run A KG_QUEUE_PLACEMENT=0 KG_POOL_MB=0 KG_MSM_GROUPS=3 KG_WIDE_WINDOW=0
run B KG_HOST_SLICES=5 KG_HOST_FIRST_DIV=3 KG_BLOCKING_TABLES_LOG=20 KG_HOST_SLICE_TABLES=0 KG_GROUP_ACCQ=1 KG_HOST_ACCQ=1
run C KG_SORT_ALONE=0 KG_GATHER_FUSE=0 KG_HOT_SUM=0 KG_G16_H_EARLY=0 KG_NTT_DIRECT_MAX_LOG=0 KG_TABLE64=0 KG_FMT64_MIN_LOG=30
