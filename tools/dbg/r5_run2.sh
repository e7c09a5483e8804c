cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 1500 python -m pytest tests/test_gpu_holes.py tests/test_gpu_host_scalars.py -x -q -m gpu > gpurun_out/r5c/pytest_holes.txt 2>&1
tail -30 gpurun_out/r5c/pytest_holes.txt
timeout 900 python bench.py > gpurun_out/r5c/bench.json 2> gpurun_out/r5c/bench.err
tail -c 600 gpurun_out/r5c/bench.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r5c/bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: l[k] for k in ("value", "ms_per_step", "blocking_ms", "msm_host_scalars")}, indent=1))
nc = l["nova_commit"]
print(json.dumps({k: {kk: nc[k].get(kk) for kk in ("ms_per_commit", "from_host")} for k in ("g1_fr", "grumpkin_fq")}, indent=1))
print(json.dumps(nc.get("rank_unit"), indent=1))
PY
