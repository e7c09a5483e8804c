#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM throughput at 2^20 pairs per GPU (BASELINE.json configs[1]) on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step is one full MSM (kg_msm: scalars + bases resident in HBM -> one projective point on the host).  With N > 1
the index range of an N * 2^20 commitment is sharded: every rank runs the same pipeline on its own 2^20 slice and
the per-rank affine partial sums (17 words) are all-gathered over RCCL and added (weak scaling).
Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the roofline / cpu_baseline fields."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4B6F676172617368
LOG_N = 20
G1_BYTES_PER_PAIR = 96          # 32 B scalar + 64 B affine base
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--window", type=int, default=0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import kogarashi_amd as K

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    n = 1 << args.log_n

    ctx = K.Context(local_rank)
    # every kernel of the library goes to ONE explicit (non-default) torch stream, so torch.cuda.Event and the
    # library's own HIP events bracket the same queue
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    if args.window:
        ctx.set_msm_window(args.window)

    # synthetic inputs, generated on the device; rank r owns slice [r*n, (r+1)*n) of the global index range
    bases = torch.empty(n * 8, dtype=torch.int64, device=dev)
    scalars = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_bases(K.KG_G1, SEED + 1, rank * n, n, bases.data_ptr())
    ctx.gen_scalars(K.KG_FR, SEED + 2, rank * n, n, scalars.data_ptr())
    torch.cuda.synchronize()

    from kogarashi_amd import dist as kdist

    def step():
        out = ctx.msm(K.KG_G1, bases.data_ptr(), 0, scalars.data_ptr(), n)
        xy, inf = out[:8], int(not out[8:].any())
        if world > 1:
            # exchange step: one all_gather of 9 words per rank over RCCL, every rank adds the partial sums
            xy, inf = kdist.combine_partials(ctx, K.KG_G1, xy, inf, device=dev)
        return xy, inf

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    acc_ms = []
    phase_sum = {}
    for _ in range(args.steps):
        res = step()
        ph = ctx.profile_last()
        acc_ms.append(ph.get("accumulate", float("nan")))
        for k_, v_ in ph.items():
            phase_sum[k_] = phase_sum.get(k_, 0.0) + v_
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    value = world * n * args.steps / elapsed
    acc_avg_ms = float(np.mean(acc_ms))
    achieved = G1_BYTES_PER_PAIR * n / (acc_avg_ms * 1e-3) / 1e9
    line = {
        "metric": "bn254_g1_msm_pairs_per_sec", "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, 64-bit accumulate)", "data": "synthetic",
        "config": {"workload": f"bn254 G1 MSM, 2^{args.log_n} uniform Fr scalars x uniform G1 bases per GPU, inputs resident in HBM",
                   "pairs_per_gpu": n, "sharding": "index range" if world > 1 else "none"},
        "roofline": {"bound": "hbm", "kernel": "k_accumulate (bucket accumulation, one launch per MSM)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel_ms": acc_avg_ms, "algorithmic_bytes_per_launch": G1_BYTES_PER_PAIR * n},
        "phases_ms_per_step": {k_: v_ / args.steps for k_, v_ in phase_sum.items()},
    }

    if rank == 0 and world == 1:
        if not args.no_ntt:
            line["ntt"] = bench_ntt(ctx, torch, dev, K)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ctx, K, bases, scalars, n, res)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def bench_ntt(ctx, torch, dev, K, log_n=22, steps=10):
    """secondary line: forward Fr NTT at 2^22 (BASELINE.json configs[2]), 64 algorithmic bytes per element."""
    n = 1 << log_n
    data = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_scalars(K.KG_FR, SEED + 3, 0, n, data.data_ptr())
    for _ in range(2):
        ctx.ntt(data.data_ptr(), log_n, False, False)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(steps):
        ctx.ntt(data.data_ptr(), log_n, False, False)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / steps
    gbs = 64.0 * n / (ms * 1e-3) / 1e9
    return {"metric": "bn254_fr_ntt_elements_per_sec", "log_n": log_n, "value": n / (ms * 1e-3), "ms": ms,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes": 64 * n}}


def cpu_baseline(ctx, K, bases, scalars, n, gpu_result):
    """The oracle's restatement of groth16::msm_curve_addition (one thread per window, like rayon's par_iter_mut)
    timed on this box's host cores on the SAME inputs; also cross-checks the GPU result at full size."""
    import numpy as np
    from oracle import oracle as O
    hb = bases.cpu().numpy().view(np.uint64).reshape(n, 8)
    hs = scalars.cpu().numpy().view(np.uint64).reshape(n, 4)
    sample = min(n, 1 << 20)
    lg = sample.bit_length()
    nwin = 256 // ((lg * 69 // 100) + 2) + 1
    threads = max(1, min(nwin, os.cpu_count() or 1))
    t0 = time.perf_counter()
    r = O.msm("g1", hb[:sample], hs[:sample], None, threads=threads)
    dt = time.perf_counter() - t0
    out = {"value": sample / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
           "sample": f"first {sample} of the {n} pairs, reference window rule (c = {(lg * 69 // 100) + 2}), {dt:.2f} s"}
    if sample == n:
        xy, inf = O.to_affine("g1", r)
        gx, ginf = gpu_result
        out["gpu_matches_cpu_at_full_size"] = bool(inf == ginf and (inf or (xy == gx).all()))
    return out


if __name__ == "__main__":
    main()
