cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tables.py tests/test_gpu_host_scalars.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r5j/pytest.txt 2>&1
tail -8 gpurun_out/r5j/pytest.txt
timeout 900 python bench.py > gpurun_out/r5j/bench.json 2> gpurun_out/r5j/bench.err
tail -c 400 gpurun_out/r5j/bench.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r5j/bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: l[k] for k in ("value", "ms_per_step", "blocking_ms")}, indent=1))
print(json.dumps(l["msm_host_scalars"], indent=1)); print(json.dumps(l["msm_strong"], indent=1))
print("ntt", l["ntt"]["ms"], "g16", l["groth16"]["ms_per_proof"], l["groth16"]["ms_per_proof_blocking"], l["groth16"].get("setup_ms"), l["groth16"].get("setup_first_ms"), l["groth16"]["window_tables"]["ms_per_proof"])
nc = l["nova_commit"]
print({k: (nc[k]["ms_per_commit"], nc[k]["from_host"]["ms_per_commit"], nc[k]["from_host"]["matches_resident"]) for k in ("g1_fr", "grumpkin_fq")}, nc["rank_unit"])
PY
