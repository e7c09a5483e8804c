cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5z
g++ -O2 -std=c++17 -o /tmp/host_cost tools/host/host_cost.cpp -Lkogarashi_amd -lkogarashi_amd -Wl,-rpath,$PWD/kogarashi_amd || exit 1
/tmp/host_cost 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5z/host_cost.txt
/tmp/host_cost 22 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5z/host_cost.txt
timeout 900 python bench.py > gpurun_out/r5z/bench.json 2> gpurun_out/r5z/bench.err
tail -c 300 gpurun_out/r5z/bench.err
python tools/dbg/bench_line.py gpurun_out/r5z/bench.json
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r5z/bench.json").read().strip().splitlines()[-1])
g = l["groth16"]; print("setup", g["setup_ms"], g["setup_first_ms"], "host witness", g["window_tables"].get("ms_per_proof_blocking_host_witness"), "host scalars", l["msm_host_scalars"]["ms_per_msm"], l["blocking_ms"])
PY
