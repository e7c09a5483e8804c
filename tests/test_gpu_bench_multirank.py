"""The N > 1 path of bench.py as the driver starts it (`python bench.py --gpus N`: launch_ranks spawns the ranks before anything
touches the GPU), on the one-GPU box: KG_BENCH_SELFTEST=1 puts every rank on cuda:0 and exchanges over gloo -- the control flow of
the sharded MSM, the strong-scaled Nova commitment and the sharded Groth16 proof, not a measurement.  Checked against the
one-rank run of the same command: the combined commitments must be the same points (nova/src/pedersen.rs:15-20)."""
import functools
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@functools.lru_cache(maxsize=None)
def _bench(gpus: int) -> dict:
    env = dict(os.environ, KG_BENCH_SELFTEST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--prewarm", "2", "--log-n", "16",
           "--nova-log-n", "16", "--groth16-log-m", "12", "--g2-log-n", "12", "--rounds", "2", "--no-cpu-baseline", "--no-ntt", "--no-skew", "--no-small"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_ranks_run_the_whole_line_and_combine_to_the_one_rank_points():
    one = _bench(1)
    two = _bench(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["sharding"] == "index range" and two["scaling"] == "weak"
    nc1, nc2 = one["nova_commit"], two["nova_commit"]
    assert nc1["ranks"] == 1 and nc2["ranks"] == 2 and nc2["pairs_per_rank"] * 2 == nc2["pairs_total"] == nc1["pairs_total"]
    for leg in ("g1_fr", "grumpkin_fq"):             # one commitment cut over the ranks = the same point
        assert nc1[leg]["point"] == nc2[leg]["point"] != "identity", leg
    g = two["groth16"]
    assert g["replicas"] == 2 and g["pipelined_matches_blocking"]
    assert g["sharded"]["contexts"] == 2 and g["sharded"]["matches_single_context"]
    assert "sharded" not in one["groth16"]
    # the strong-scaled MSM: ONE 2^16-pair MSM cut over the ranks is the point the one-rank run computes on the whole range
    s1, s2 = one["msm_strong"], two["msm_strong"]
    assert s1["ranks"] == 1 and s2["ranks"] == 2 and s2["scaling"] == "strong" and s2["pairs_per_rank"] * 2 == s2["pairs_total"] == s1["pairs_total"]
    assert s1["point"] == s2["point"] != "identity" and s2["ms_per_msm"] > 0
    # the line says how the ranks met: the backend and the world size the process group itself reports (a future SCALE record proves RCCL saw N ranks)
    assert one["rccl"] == {"backend": None, "ranks": 1, "hosts": 1, "selftest": False}
    assert two["rccl"] == {"backend": "gloo", "ranks": 2, "hosts": 1, "selftest": True}
    assert two["rounds"] == 2 and len(two["rounds_ms"]) == 2 and next(iter(two)) == "summary"
    assert two["msm_g2"]["pipelined_matches_blocking"] and not two["summary"]["checks"]["failed"]
    # the waiting ranks of the sharded proof park on a host-side (gloo) barrier, never on an RCCL kernel of the GPUs being timed
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src.split("if world > 1 and circuit == \"chain\":")[1].split("return out")[0]
    assert body.count('env["host_barrier"]()') == 2 and "sync()" not in body
    assert 'dist.new_group(backend="gloo")' in src


def test_three_ranks_an_odd_split_with_a_ragged_slice():
    """2^16 pairs over three ranks: kg_shard_range hands out 21846 + 21845 + 21845 pairs (no rank's slice is a power of two, rank 0 holds
    the extra pair) -- the combined MSM and commitments are still the one-rank points, and the sharded proof runs on three contexts."""
    one = _bench(1)
    three = _bench(3)
    assert three["n_gpus"] == 3 and three["rccl"]["ranks"] == 3
    s1, s3 = one["msm_strong"], three["msm_strong"]
    assert s3["pairs_per_rank"] == 21846 and s3["pairs_total"] == s1["pairs_total"] == 1 << 16
    assert s1["point"] == s3["point"] != "identity"
    for leg in ("g1_fr", "grumpkin_fq"):
        assert one["nova_commit"][leg]["point"] == three["nova_commit"][leg]["point"] != "identity", leg
    assert three["nova_commit"]["pairs_per_rank"] == 21846
    g = three["groth16"]
    assert g["replicas"] == 3 and g["sharded"]["contexts"] == 3 and g["sharded"]["matches_single_context"]
    assert not three["summary"]["checks"]["failed"]
