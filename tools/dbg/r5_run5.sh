cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
timeout 1500 python -m pytest tests/test_gpu_large.py -x -q -m gpu -k "groth16_2_18 or commit_2_24_matches" 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/r5n/bench.json 2> gpurun_out/r5n/bench.err
tail -c 300 gpurun_out/r5n/bench.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r5n/bench.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_step", "blocking_ms")}, l["msm_host_scalars"]["ms_per_msm"], l["roofline"]["traffic_detail"]["source"])
g = l["groth16"]; print(g["ms_per_proof"], g["ms_per_proof_blocking"], g["setup_ms"], g["window_tables"])
PY
