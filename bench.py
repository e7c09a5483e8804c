#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM throughput at 2^20 pairs per GPU (BASELINE.json configs[1]) on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process IS a rank; started bare, it spawns
the N ranks itself as child processes BEFORE anything touches the GPU (launch_ranks) and exits with their status.  Either
way the process group's world size must equal --gpus.

A step is one full MSM (kg_msm: scalars + bases resident in HBM -> one projective point on the host).  The headline is the
MEDIAN of --rounds (5) timed rounds of K steps each, every round bracketed by a barrier + device synchronisation, max over
ranks (`rounds_ms` lists them).  With N > 1 the index range of an N * 2^20 commitment is sharded: every rank runs the same
pipeline on its own 2^20 slice and the per-rank affine partial sums (9 words) are all-gathered over RCCL and added (weak
scaling).

Rank 0 prints ONE JSON line.  Its first key is `summary`: every leg's headline figure, flat.  The legs: `roofline` /
`valu_roofline` / `cpu_baseline` of the headline, `blocking_ms`, `msm_host_scalars`, `msm_strong`, `msm_skewed`, `msm_g2`
(2^18 G2 pairs, 160 B per pair), `small` (short blocking calls: the reference's own test sizes), `ntt` (2^22 forward, and
`variants`: idft / coset_dft / coset_idft), `groth16` (2^18 constraints), `nova_commit` (2^24 pairs, both curves).  Prose
notes are left out unless --notes is given (the driver keeps the last 8 KB of stdout: the line must fit); what every field
means is in DESIGN.md section 8.

Profiling aids (tools/collect_profiles.sh): --headline-only (pre-warm, warm-up and the timed rounds, nothing else: the
k_acc_tasks average of a kernel trace of this command is roofline.kernel_ms), --ntt-only, --msm-g2-only, --groth16-only."""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The library's queues (main, scalar, two reduction queues) want a hardware queue each; the runtime reads this when HIP
# initialises, which torch does before the library is loaded (csrc/capi.cpp sets the same default for hosts that load it first).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

SEED = 0x4B6F676172617368
LOG_N = 20
G1_BYTES_PER_PAIR = 96          # 32 B scalar + 64 B affine base
G2_BYTES_PER_PAIR = 160         # 32 B scalar + 128 B affine G2 base
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
MADD_PEAK_G = 17.3              # measured: the bucket kernel's addition routine, operands in registers, 4 waves/SIMD (profiles/r03_mul_rate.txt; 16.5 before round 3's column chains)
MAD_PEAK_T = 33.0               # measured chip-wide v_mad_u64_u32 issue rate, T instructions/s (profiles/r01_valu_rates.txt)
MADS_PER_ADDITION = 1467        # add_mixed_signed (curve.h): 6 products x 162 + 2 squares x 126 + one double product x 243 multiply-accumulates (fp29.h)
MADS_PER_G2_ADDITION = 4536     # add_mixed over Fq2 (curve.h, fp29.h Fp2): 6 products x 486 + 2 squares x 324 + one double product x 972
WARM_PROOFS = 16                # untimed proofs in front of a timed Groth16 section (clock ramp after an idle period)
MUL_PEAK_G = 175.0              # measured Montgomery products/s, 4 waves/SIMD (profiles/r03_mul_rate.txt)
NOTES = False                   # --notes


def note(d, text):
    """an explanatory note on a leg -- only with --notes (the default line must fit the 8 KB of stdout the driver keeps)"""
    if NOTES:
        d["note"] = text
    return d


def point_digest(xy, inf):
    """fingerprint of an affine point (equal points <=> equal digests): 16 hex digits of sha256 over its words"""
    import hashlib
    import numpy as np
    return "identity" if inf else hashlib.sha256(np.ascontiguousarray(xy, dtype=np.uint64).tobytes()).hexdigest()[:16]


def r3(x):
    return None if x is None else round(float(x), 4)


def compact(x, digits=5):
    """floats to `digits` significant digits, recursively (the line must fit the 8 KB of stdout the driver keeps)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k_: compact(v_, digits) for k_, v_ in x.items()}
    if isinstance(x, (list, tuple)):
        return [compact(v_, digits) for v_ in x]
    return x


def ntt_cost(K, log_n):
    """Multiply-accumulates per element and a description of the kernels of a 2^log_n transform, from the library's own plan
    (kg_ntt_plan).  A step of 2^m points has m radix-2 stages; stage 1 has no twiddle products, stage 2 on half of its butterflies,
    every later stage half a product per element: m/2 - 0.75 in-tile products per element in Shoup form (fp29.h mulc: 143
    multiply-accumulates), plus one Montgomery product (162) per element between two steps; above 2^22 (ntt.hip direct_a_max_log,
    KG_NTT_DIRECT_MAX_LOG) the first boundary composes its twiddle from two table entries (one more product) instead of reading it
    from a table of n entries (36 B of HBM traffic per element)."""
    plan = K.lib.ntt_plan(log_n)
    direct_a = len(plan) >= 2 and log_n <= min(22, int(os.environ.get("KG_NTT_DIRECT_MAX_LOG", "22")))
    composed = len(plan) >= 2 and not direct_a
    mads = sum((m / 2 - 0.75) * 143 for m, _ in plan) + (len(plan) - 1 + (1 if composed else 0)) * 162
    kern = "k_ntt_tile steps " + " x ".join(f"2^{m}" for m, _ in plan) + f", {1 << plan[0][1]}-element tiles"
    return mads, kern, plan


def single_rank_env(torch, dev):
    """the rank environment of the legs for one process without a process group (the --*-only modes, tools/dbg)"""
    return {"world": 1, "rank": 0, "barrier": torch.cuda.synchronize, "host_barrier": torch.cuda.synchronize, "max_over_ranks": lambda x: x, "xdev": dev,
            "kdist": None}


def timed_rounds(env, fn, steps, rounds):
    """`rounds` timed rounds of fn(steps), each bracketed by the rank barrier + device synchronisation, max over ranks per round;
    returns (median seconds per round, [ms per step of every round], the last result)"""
    out, res = [], None
    gc.collect()
    for _ in range(rounds):
        env["barrier"]()
        t0 = time.perf_counter()
        res = fn(steps)
        env["barrier"]()
        out.append(env["max_over_ranks"](time.perf_counter() - t0))
    med = sorted(out)[len(out) // 2]
    return med, [round(x / steps * 1e3, 4) for x in out], res


def bench_msm_strong(ctx, torch, dev, K, env, log_n, steps):
    """ONE G1 MSM of 2^log_n pairs (the headline's pairs: same seeds, global index range) cut over the N ranks by kg_shard_range: every
    rank runs a BLOCKING kg_msm on its contiguous slice (2^log_n / N pairs: latency-bound below ~2^18) and the N affine partial sums
    meet in one all_gather of 9 words per rank (kogarashi_amd/dist.py) -- strong scaling of a fixed job, beside the weak-scaled headline.
    Unmeasured on multi-GPU hardware until the driver has an 8-GPU node; the point is the same for every N (tests/test_gpu_bench_multirank.py)."""
    gc.collect()
    from kogarashi_amd.lib import shard_range
    world, rank, kdist, xdev = env["world"], env["rank"], env["kdist"], env["xdev"]
    total = 1 << log_n
    lo, hi = shard_range(total, rank, world)
    nl = hi - lo
    b = torch.empty(max(nl, 1) * 8, dtype=torch.int64, device=dev)
    sc = torch.empty(max(nl, 1) * 4, dtype=torch.int64, device=dev)
    if nl:
        ctx.gen_bases(K.KG_G1, SEED + 1, lo, nl, b.data_ptr())
        ctx.gen_scalars(K.KG_FR, SEED + 2, lo, nl, sc.data_ptr())
    ctx.sync()

    def one():
        out = ctx.msm(K.KG_G1, b.data_ptr(), 0, sc.data_ptr(), nl)
        xy, inf = out[:8], int(not out[8:].any())
        if world > 1:
            xy, inf = kdist.combine_partials(ctx, K.KG_G1, xy, inf, device=xdev)
        return xy, inf
    for _ in range(3):
        got = one()
    env["barrier"]()
    t0 = time.perf_counter()
    for _ in range(steps):
        got = one()
    env["barrier"]()
    dt = env["max_over_ranks"](time.perf_counter() - t0) / steps
    return {"metric": "bn254_g1_msm_pairs_per_sec (one MSM over all ranks)", "log_n": log_n, "pairs_total": total, "pairs_per_rank": nl, "ranks": world,
            "scaling": "strong", "ms_per_msm": r3(dt * 1e3), "value": total / dt, "unit": "pairs/s",
            "exchange": "all_gather of 9 words per rank (RCCL)" if world > 1 else "none",
            "point": point_digest(got[0], got[1])}


def window_adds(n):
    """bucket additions of one MSM: one per (window, scalar) pair with a non-zero digit ~ W * n; c from the library's own rule"""
    from kogarashi_amd.lib import msm_pick_window
    c = msm_pick_window(n)                  # kg_msm_pick_window (no device needed)
    return ((255 + c - 1) // c) * n


def launch_ranks(args):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE): start the N ranks as CHILD processes of this one, before
    anything here has touched the GPU (no torch, no HIP library loaded yet -- a process that has initialised the GPU must
    never be replaced), wait for all of them and exit non-zero if any failed.  Rank 0 inherits stdout and prints the line."""
    import signal
    import socket
    import subprocess
    rc = 0
    for attempt in range(3):
        with socket.socket() as sk:                 # a free rendezvous port on the loop-back interface
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        # (the port is free when it is picked, not reserved: if another process takes it before rank 0 binds it, the ranks fail
        # within seconds and the launch is repeated with a new one)
        t_start = time.time()
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = 0
        try:
            pending = list(procs)
            while pending:
                for pr in list(pending):
                    code = pr.poll()
                    if code is None:
                        continue
                    pending.remove(pr)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 1
                        for other in pending:        # one rank failed: the others would wait in a collective for ever
                            other.send_signal(signal.SIGTERM)
                time.sleep(0.05)
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        if rc == 0 or time.time() - t_start > 30.0:    # a late failure is not a rendezvous problem: report it
            break
    return rc


def main():
    global NOTES
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5, help="timed rounds of --steps steps each; the headline is the median round")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-groth16", action="store_true")
    ap.add_argument("--no-nova", action="store_true")
    ap.add_argument("--no-g2", action="store_true")
    ap.add_argument("--no-small", action="store_true")
    ap.add_argument("--no-skew", action="store_true", help="skip the skewed-scalar legs (msm_skewed, groth16.skewed_witness)")
    ap.add_argument("--headline-only", action="store_true", help="profiling aid: pre-warm, warm-up and the timed rounds of the headline, nothing else")
    ap.add_argument("--ntt-only", action="store_true", help="profiling aid: only the NTT leg; prints {\"ntt\": ...}")
    ap.add_argument("--ntt-variant", default="all", help="with --ntt-only: dft | idft | coset_dft | coset_idft | all")
    ap.add_argument("--msm-g2-only", action="store_true", help="profiling aid: only the G2 MSM leg; prints {\"msm_g2\": ...}")
    ap.add_argument("--groth16-only", action="store_true", help="profiling aid: only the prover (no tables, no CPU leg); prints {\"groth16\": ...}")
    ap.add_argument("--small-only", action="store_true", help="only the short blocking calls; prints {\"small\": ...}")
    ap.add_argument("--notes", action="store_true", help="keep the prose notes in the line")
    ap.add_argument("--nova-log-n", type=int, default=24, help="pairs of the Nova commitment (whole job, cut over the ranks)")
    ap.add_argument("--groth16-log-m", type=int, default=18)
    ap.add_argument("--g2-log-n", type=int, default=18)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--depth", type=int, default=4, help="MSM steps in flight (1..4)")
    ap.add_argument("--prewarm", type=int, default=200, help="untimed steps before the W warm-up steps (first touch, clock ramp)")
    ap.add_argument("--stream-ordered-inputs", action="store_true", help="do not declare the (static, synchronised) inputs complete")
    args = ap.parse_args()
    # like timeit: no cyclic garbage collection inside timed loops (a generation-2 pass is a 30-40 ms pause that lands in whichever call
    # happens to allocate the object that triggers it -- tools/dbg/anom_1024.py); every leg collects before it starts its clock
    gc.disable()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.rounds < 1:
        ap.error("--rounds must be >= 1")
    NOTES = args.notes

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")

    import numpy as np
    import torch
    import kogarashi_amd as K

    # KG_BENCH_SELFTEST=1: control-flow check of the N > 1 path on a ONE-GPU box -- every rank uses cuda:0 and the
    # exchange goes over gloo with host tensors (RCCL refuses two ranks on one device).  Never a measurement.
    selftest = world > 1 and os.environ.get("KG_BENCH_SELFTEST") == "1"
    dist = None
    host_group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            local_rank = 0
        elif torch.cuda.device_count() <= local_rank:       # counting devices does not initialise the GPU
            sys.exit(f"bench.py: rank {rank} needs cuda:{local_rank} but this box has {torch.cuda.device_count()} GPU(s) "
                     "(KG_BENCH_SELFTEST=1 runs the N > 1 control flow with every rank on cuda:0 -- not a measurement)")
        torch.cuda.set_device(local_rank)
        if selftest:
            dist.init_process_group("gloo")
            host_group = None                                # the default group is gloo already
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # a second, host-only group: ranks that merely WAIT for rank 0 (the sharded proof drives GPUs 0..2 from one process) park on
            # a gloo barrier -- a barrier of the nccl group is an RCCL kernel spinning on the very GPUs rank 0 is timing
            host_group = dist.new_group(backend="gloo")
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    xdev = None if selftest else dev     # where exchanged tensors live
    n = 1 << args.log_n

    ctx = K.Context(local_rank)
    if args.ntt_only or args.msm_g2_only or args.groth16_only or args.small_only:
        torch.cuda.synchronize()
        env = single_rank_env(torch, dev)
        ctx.set_inputs_complete(True)
        if args.ntt_only:
            out = {"ntt": bench_ntt(ctx, torch, dev, K, env, steps=max(args.steps, 10), variants=args.ntt_variant)}
        elif args.msm_g2_only:
            out = {"msm_g2": bench_msm_g2(ctx, torch, dev, K, env, args.g2_log_n, steps=max(args.steps, 10), rounds=args.rounds, cpu=not args.no_cpu_baseline)}
        elif args.small_only:
            out = {"small": bench_small(ctx, torch, dev, K)}
        else:
            out = {"groth16": bench_groth16(ctx, torch, dev, K, env, args.groth16_log_m, steps=max(args.steps, 8), cpu=False, tables=False, from_witness=False)}
        print(json.dumps(compact(out)), flush=True)
        return
    # The library launches on its own queues (main queue: accumulations; scalar-side queue: digit extraction, sort, base
    # conversion; two reduction queues) and brackets its phases with HIP events recorded on those queues; the timed region
    # is bracketed by device-wide synchronisation.
    if not args.stream_ordered_inputs:
        ctx.set_inputs_complete(True)          # inputs are generated once and synchronised before the timed region
    if args.window:
        ctx.set_msm_window(args.window)

    # synthetic inputs, generated on the device; rank r owns slice [r*n, (r+1)*n) of the global index range
    bases = torch.empty(n * 8, dtype=torch.int64, device=dev)
    scalars = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_bases(K.KG_G1, SEED + 1, rank * n, n, bases.data_ptr())
    ctx.gen_scalars(K.KG_FR, SEED + 2, rank * n, n, scalars.data_ptr())
    ctx.sync()
    torch.cuda.synchronize()

    from kogarashi_amd import dist as kdist

    def finish(ticket):
        out = ctx.msm_end(K.KG_G1, ticket)
        xy, inf = out[:8], int(not out[8:].any())
        if world > 1:
            # exchange step: one all_gather of 9 words per rank over RCCL, every rank adds the partial sums
            xy, inf = kdist.combine_partials(ctx, K.KG_G1, xy, inf, device=xdev)
        return xy, inf

    depth = max(1, min(args.depth, 4))

    def run(k):
        """k MSM steps, software-pipelined `depth` deep through kg_msm_begin / kg_msm_end (tickets 0..3): while step i
        accumulates, step i+1 is sorted on the scalar queue and steps i-1, i-2 finish their bucket reductions (reduction
        queues) and host tails.  Every step's result is produced inside the loop."""
        res = None
        for i in range(k):
            ctx.msm_begin(K.KG_G1, bases.data_ptr(), 0, scalars.data_ptr(), n, i % 4)
            if i >= depth - 1:
                res = finish((i - depth + 1) % 4)
        for i in range(max(k - depth + 1, 0), k):
            res = finish(i % 4)
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def host_barrier():
        """all ranks meet WITHOUT touching a GPU (gloo): for waits during which another rank is timing work on this rank's device"""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=host_group)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=xdev if xdev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    env = {"world": world, "rank": rank, "barrier": barrier, "host_barrier": host_barrier, "max_over_ranks": max_over_ranks, "xdev": xdev, "kdist": kdist}
    run(args.prewarm)                         # untimed: first-touch allocations and the clock ramp of a cold GPU (~0.3 s)
    if args.warmup:
        run(args.warmup)
    ctx.profile_enable(True)
    # the timed region: --rounds rounds of EXACTLY --steps steps, each bracketed by barrier + synchronisation; the median round is the
    # headline (a 27 ms window moves 2-4 % between runs on a box whose host cores are shared)
    elapsed, rounds_ms, res = timed_rounds(env, run, args.steps, args.rounds)
    summary = ctx.profile_summary()           # HIP events recorded on the launch streams during ALL timed rounds
    ctx.profile_enable(False)
    acc_avg_ms = summary["accumulate"][0] / summary["accumulate"][1]
    phase_avg = {k_: r3(v_[0] / v_[1]) for k_, v_ in summary.items()}

    value = world * n * args.steps / elapsed
    achieved = G1_BYTES_PER_PAIR * n / (acc_avg_ms * 1e-3) / 1e9
    traffic = pmc_traffic(args.log_n)
    rccl = {"backend": dist.get_backend() if world > 1 else None, "ranks": dist.get_world_size() if world > 1 else 1,
            "hosts": 1, "selftest": bool(selftest)}
    line = {
        "metric": "bn254_g1_msm_pairs_per_sec", "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32x9 (29-bit limbs)", "data": "synthetic",
        "config": {"workload": f"bn254 G1 MSM, 2^{args.log_n} uniform Fr scalars x uniform G1 bases per GPU, resident in HBM",
                   "pairs_per_gpu": n, "sharding": "index range" if world > 1 else "none"},
        "rounds": args.rounds, "rounds_ms": rounds_ms,
        "rccl": rccl,
        "roofline": note({"bound": "hbm", "kernel": "k_acc_tasks<Fq>",
                          "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                          "traffic": traffic["corrected"] if traffic else None, "kernel_ms": acc_avg_ms, "launches": summary["accumulate"][1],
                          "algorithmic_bytes_per_launch": G1_BYTES_PER_PAIR * n,
                          "traffic_uncorrected": traffic["uncorrected"] if traffic else None, "traffic_source": traffic["source"] if traffic else None},
                         "VALU-bound kernel (16 n point additions, see valu_roofline); in the timed region its launches overlap the next step's sort "
                         "and the previous steps' reductions (service kernels run at wave priority 3 beside it), so kernel_ms there is longer than "
                         "isolated.kernel_ms while ms_per_step is shorter than their sum; kernel_ms = HIP events over every launch of all timed rounds"),
        "pipelining": f"{depth} in flight",
    }
    if NOTES:
        line["phases_ms_per_step"] = phase_avg
    if args.headline_only:
        if rank == 0:
            print(json.dumps(compact({"summary": {"msm_ms_per_step": r3(line["ms_per_step"]), "msm_pairs_per_s": value, "acc_kernel_ms": r3(acc_avg_ms)}, **line})), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # The same kernel with nothing beside it (a few blocking MSMs after the timed region): in the timed region the
    # accumulation shares the chip with the next step's sort and the previous steps' reductions, which is what makes the
    # step shorter and the kernel's own launch longer.
    barrier()
    def groups(g):            # (an older build of the library, in an A/B run through KG_LIB_PATH, has no such knob and no window groups)
        try:
            ctx.set_msm_groups(g)
        except AttributeError:
            pass
    groups(1)                 # the kernel alone: ONE accumulation launch per MSM (a blocking call otherwise runs in two window groups)
    ctx.msm(K.KG_G1, bases.data_ptr(), 0, scalars.data_ptr(), n)
    ctx.profile_enable(True)
    for _ in range(5):
        ctx.msm(K.KG_G1, bases.data_ptr(), 0, scalars.data_ptr(), n)
    iso = ctx.profile_summary()
    ctx.profile_enable(False)
    groups(0)
    for _ in range(2):        # untimed: the blocking call's own queues, work spaces and result slots (window groups) are set up on first use
        ctx.msm(K.KG_G1, bases.data_ptr(), 0, scalars.data_ptr(), n)
    blk_rounds = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(4):
            ctx.msm(K.KG_G1, bases.data_ptr(), 0, scalars.data_ptr(), n)
        blk_rounds.append((time.perf_counter() - t0) / 4 * 1e3)
    blocking_ms = sorted(blk_rounds)[2]
    iso_ms = iso["accumulate"][0] / iso["accumulate"][1]
    iso_achieved = G1_BYTES_PER_PAIR * n / (iso_ms * 1e-3) / 1e9
    adds = window_adds(n)
    line["blocking_ms"] = blocking_ms          # wall time of one isolated kg_msm call (nothing in flight), host finish included; median of five rounds of four
    line["queues"] = {"placement": ctx.queue_placement() if hasattr(ctx, "queue_placement") else None}
    line["roofline"]["isolated"] = {"kernel_ms": iso_ms, "achieved": iso_achieved, "frac": iso_achieved / HBM_PEAK_GBS}
    # the bound that actually limits the kernel, against the MACHINE: multiply-accumulate instructions per second vs the
    # chip's measured v_mad_u64_u32 issue rate; and against the same addition routine with operands in registers
    line["valu_roofline"] = note({"bound": "valu", "unit": "T v_mad_u64_u32/s", "mads_per_addition": MADS_PER_ADDITION, "additions_per_launch": adds,
                                  "achieved": adds * MADS_PER_ADDITION / (iso_ms * 1e-3) / 1e12, "peak": MAD_PEAK_T,
                                  "frac": adds * MADS_PER_ADDITION / (iso_ms * 1e-3) / 1e12 / MAD_PEAK_T,
                                  "routine_frac": adds / (iso_ms * 1e-3) / 1e9 / MADD_PEAK_G},      # against the addition routine with operands in registers (17.3 G/s)
                                 "isolated launches; the remaining ~30 % of issue slots go to the shifts / masks / carries of the 29-bit limbs, "
                                 "the lazy-reduction bookkeeping and the gathers; rocprofv3 SQ counters (profiles/r05_msm_sq_counters.json): VALU issue "
                                 "busy 88 % of the SIMD cycles of a launch")

    if rank == 0 and world == 1:
        # informational: the same steps with the bases registered (kg_bases_register: converted to the internal
        # form once, as a resident CRS / commitment key would be); never the headline value
        ctx.bases_register(K.KG_G1, bases.data_ptr(), 0, n)
        run(2)
        barrier()
        t0 = time.perf_counter()
        res_reg = run(args.steps)
        barrier()
        line["registered_bases"] = {"ms_per_step": r3((time.perf_counter() - t0) / args.steps * 1e3),
                                    "matches_unregistered": bool((res_reg[0] == res[0]).all() and res_reg[1] == res[1])}
        if (1 << 16) <= n <= (1 << 20):
            # and with window tables on top (kg_bases_precompute: 2^(17 w) * P for the 15 windows, one bucket set for all of them):
            # 6 % fewer additions and a 16x smaller reduction, but the gathers leave the Infinity Cache (1.1 GB table at 2^20) --
            # about even at this size, a gain at the prover's 2^18 (groth16.window_tables)
            t0 = time.perf_counter()
            ctx.bases_precompute(bases.data_ptr())
            ctx.sync()
            build_ms = (time.perf_counter() - t0) * 1e3
            run(4)
            barrier()
            t0 = time.perf_counter()
            res_tab = run(args.steps)
            barrier()
            line["registered_bases"]["window_tables"] = {"ms_per_step": r3((time.perf_counter() - t0) / args.steps * 1e3), "build_ms": r3(build_ms),
                                                         "matches_unregistered": bool((res_tab[0] == res[0]).all() and res_tab[1] == res[1])}
        # The call the reference's call sites make (groth16/src/msm.rs:6: fixed bases, a fresh `coeffs` slice per call): bases registered
        # once, the 2^log_n scalars in pageable HOST memory, uploaded inside the call in index slices under the accumulations
        # (kg_msm_host_scalars).  PCIe-inclusive: reported beside blocking_ms, never as `value`.
        hs = scalars.cpu().numpy().view(np.uint64).reshape(n, 4)
        # host-side timing of a call that spends a third of its time in host threads (upload, enqueue): five alternating rounds of six calls,
        # the median round reported and all rounds listed (the boxes' host cores are shared with other tenants)
        for _ in range(3):
            hres = ctx.msm_host_scalars(K.KG_G1, bases.data_ptr(), 0, hs, n)
            rres = ctx.msm(K.KG_G1, bases.data_ptr(), 0, scalars.data_ptr(), n)
        host_rounds, res_rounds = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(6):
                hres = ctx.msm_host_scalars(K.KG_G1, bases.data_ptr(), 0, hs, n)
            host_rounds.append((time.perf_counter() - t0) / 6 * 1e3)
            t0 = time.perf_counter()
            for _ in range(6):
                rres = ctx.msm(K.KG_G1, bases.data_ptr(), 0, scalars.data_ptr(), n)
            res_rounds.append((time.perf_counter() - t0) / 6 * 1e3)
        host_ms, res_ms = sorted(host_rounds)[2], sorted(res_rounds)[2]
        stage = torch.empty(n * 4, dtype=torch.int64, device=dev)
        ctx.write(stage.data_ptr(), hs)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.write(stage.data_ptr(), hs)                # one synchronous kg_memcpy_h2d of the whole slice: what the call replaced
        up_ms = (time.perf_counter() - t0) / 5 * 1e3
        del stage
        line["msm_host_scalars"] = note({"ms_per_msm": r3(host_ms), "resident_blocking_ms": r3(res_ms), "over_resident_ms": r3(host_ms - res_ms),
                                         "rounds_ms": [round(x, 2) for x in host_rounds],
                                         "plain_upload_ms": r3(up_ms), "upload_gb_per_s": r3(32 * n / (up_ms * 1e-3) / 1e9),
                                         "matches_resident": bool((hres == rres).all())},
                                        "kg_msm_host_scalars: registered bases, pageable host scalars (32 B per pair over PCIe inside the call), "
                                        "blocking, host finish included; resident_blocking_ms = kg_msm on the same registered bases, same loop")
        del hs
        ctx.bases_unregister(bases.data_ptr())
    cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if rank == 0 and world == 1 and not args.no_skew:
        line["msm_skewed"] = bench_msm_skewed(ctx, torch, dev, K, bases, n, run, barrier, args.steps, elapsed / args.steps * 1e3, blocking_ms, cpu)
    if cpu:
        line["cpu_baseline"] = cpu_baseline(ctx, K, bases, scalars, n, res)
        line["cpu_plumbing_2_10"] = cpu_plumbing(ctx, K)
    del bases, scalars
    # the strong-scaled counterpart of the headline: ONE 2^log_n MSM cut over the ranks (north_star: "splitting the scalar/base array
    # across the 8 GPUs"); at N = 1 it is the blocking kg_msm of the whole range
    line["msm_strong"] = bench_msm_strong(ctx, torch, dev, K, env, args.log_n, args.steps)
    if not args.no_g2:
        line["msm_g2"] = bench_msm_g2(ctx, torch, dev, K, env, args.g2_log_n, steps=10, rounds=3, cpu=cpu)
    if rank == 0 and world == 1 and not args.no_small:
        line["small"] = bench_small(ctx, torch, dev, K)
    if not args.no_ntt:
        line["ntt"] = bench_ntt(ctx, torch, dev, K, env)
    if not args.no_nova:
        line["nova_commit"] = bench_nova_commit(ctx, torch, dev, K, env, args.nova_log_n, cpu=cpu, skew=not args.no_skew)
    if not args.no_groth16:
        line["groth16"] = bench_groth16(ctx, torch, dev, K, env, args.groth16_log_m, cpu=cpu)
        if not args.no_skew:
            # the same proof size on a circuit whose witness is 0/1-heavy (prover.rs:53-65 meets such aux vectors)
            sk = bench_groth16(ctx, torch, dev, K, env, args.groth16_log_m, steps=4, cpu=cpu, circuit="boolean", from_witness=False)
            line["groth16"]["skewed_witness"] = {
                "ms_per_proof": sk["ms_per_proof"], "ms_per_proof_blocking": sk["ms_per_proof_blocking"],
                "ratio_to_uniform": r3(sk["ms_per_proof"] / line["groth16"]["ms_per_proof"]),
                "pipelined_matches_blocking": sk["pipelined_matches_blocking"],
                "window_tables": {k_: sk["window_tables"][k_] for k_ in ("ms_per_proof", "ms_per_proof_blocking", "proofs_match")} if "window_tables" in sk else None,
                "cpu_baseline": sk.get("cpu_baseline")}
    if rank == 0:
        out = {"summary": make_summary(line), **line}
        text = json.dumps(compact(out), separators=(",", ":"))
        print(text, flush=True)
        print(f"bench.py: the line is {len(text)} bytes", file=sys.stderr)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def make_summary(line):
    """every leg's headline figure, flat, FIRST in the line (ms unless the key says otherwise); `checks` = every parity flag the legs carry"""
    def g(d, *ks):
        for k_ in ks:
            if not isinstance(d, dict) or k_ not in d:
                return None
            d = d[k_]
        return d
    s = {"msm_ms_per_step": r3(line["ms_per_step"]), "msm_pairs_per_s": line["value"], "acc_kernel_ms": r3(g(line, "roofline", "kernel_ms")),
         "acc_kernel_isolated_ms": r3(g(line, "roofline", "isolated", "kernel_ms")), "hbm_frac": r3(g(line, "roofline", "frac")),
         "blocking_ms": r3(line.get("blocking_ms")), "host_scalars_ms": g(line, "msm_host_scalars", "ms_per_msm"),
         "host_scalars_over_resident_ms": g(line, "msm_host_scalars", "over_resident_ms"), "msm_strong_ms": g(line, "msm_strong", "ms_per_msm"),
         "msm_skewed_ms": g(line, "msm_skewed", "ms_per_step"),
         "g2_ms_per_step": g(line, "msm_g2", "ms_per_step"), "g2_blocking_ms": g(line, "msm_g2", "blocking_ms"), "g2_pairs_per_s": g(line, "msm_g2", "value"),
         "msm_2p10_blocking_ms": g(line, "small", "msm_blocking_ms", "1024"), "msm_32_blocking_ms": g(line, "small", "msm_blocking_ms", "32"),
         "proof_2p10_blocking_ms": g(line, "small", "proof_2p10_blocking_ms"),
         "ntt_ms": g(line, "ntt", "ms"), "ntt_idft_ms": g(line, "ntt", "variants", "idft", "ms"), "ntt_coset_dft_ms": g(line, "ntt", "variants", "coset_dft", "ms"),
         "ntt_coset_idft_ms": g(line, "ntt", "variants", "coset_idft", "ms"),
         "proof_ms": g(line, "groth16", "ms_per_proof"), "proof_blocking_ms": g(line, "groth16", "ms_per_proof_blocking"),
         "proof_tables_ms": g(line, "groth16", "window_tables", "ms_per_proof"), "proof_tables_blocking_ms": g(line, "groth16", "window_tables", "ms_per_proof_blocking"),
         "setup_ms": g(line, "groth16", "setup_ms"),
         "commit_g1_ms": g(line, "nova_commit", "g1_fr", "ms_per_commit"), "commit_grumpkin_ms": g(line, "nova_commit", "grumpkin_fq", "ms_per_commit"),
         "commit_g1_from_host_ms": g(line, "nova_commit", "g1_fr", "from_host", "ms_per_commit"),
         "commit_grumpkin_from_host_ms": g(line, "nova_commit", "grumpkin_fq", "from_host", "ms_per_commit"),
         "cpu_msm_pairs_per_s": g(line, "cpu_baseline", "value"), "cpu_proofs_per_s": g(line, "groth16", "cpu_baseline", "value")}
    flags = []

    def walk(d, path):
        for k_, v_ in d.items():
            if isinstance(v_, dict):
                walk(v_, path + [k_])
            elif isinstance(v_, bool) and ("match" in k_ or "add_up" in k_):
                flags.append((".".join(path + [k_]), v_))
    walk(line, [])
    s["checks"] = {"passed": sum(1 for _, v_ in flags if v_), "failed": [k_ for k_, v_ in flags if not v_]}
    return {k_: v_ for k_, v_ in s.items() if v_ is not None}


def pmc_traffic(log_n):
    """KG_BENCH_PMC=<file>: a fresh pmc_summary.py JSON (same-day FETCH_SIZE / WRITE_SIZE passes) instead of the committed one.
    Memory-side bytes per k_acc_tasks launch from the committed rocprofv3 --pmc passes (profiles/): FETCH_SIZE and
    WRITE_SIZE are collected in separate runs of this same command.  MI355X_MICROARCH.md prescribes doubling FETCH_SIZE on
    gfx950 for wide coalesced streams; this kernel's reads are scattered 8-byte-per-lane gathers, for which the correction is
    uncalibrated (an upper estimate) -- both figures are reported.  Only valid for the configuration it was measured on (2^20 pairs); null otherwise."""
    for name in ("r06_pmc_hbm.json", "r05_pmc_hbm.json", "r04_pmc_hbm.json", "r03_pmc_hbm.json", "r02_pmc_hbm.json", "r01_m_pmc_hbm.json"):
        path = os.environ.get("KG_BENCH_PMC") or os.path.join(ROOT, "profiles", name)
        if log_n == LOG_N and os.path.exists(path):
            with open(path) as f:
                t = json.load(f)["k_acc_tasks_traffic_bytes_per_launch"]
            return {"uncorrected": t["fetch_reported"] + t["write"], "corrected": t["total_corrected"], "fetch_reported": t["fetch_reported"],
                    "write": t["write"], "source": os.path.relpath(path, ROOT)}
    return None


def bench_msm_g2(ctx, torch, dev, K, env, log_n=18, steps=10, rounds=3, cpu=False):
    """BN254 G2 MSM (north_star: "Pippenger bucket MSM over G1/G2"; the prover's b_g2 query, groth16/src/prover.rs:64-65, bn254/src/g2.rs:16-20):
    2^log_n uniform Fr scalars against G2 bases k_i * G2 (cofactor != 1: generator multiples, made on the device by kg_fixed_base_mul),
    four in flight and blocking.  160 algorithmic bytes per pair; the dominant kernel is k_acc_tasks<Fq2> (250 VGPRs, two
    waves per SIMD), timed alone (one launch per MSM) for `roofline.isolated` and `valu_roofline`.  CPU leg: the oracle's Pippenger
    restatement on the same pairs, which also checks the GPU point."""
    gc.collect()
    import numpy as np
    world, rank = env["world"], env["rank"]
    n = 1 << log_n
    k = torch.empty(n * 4, dtype=torch.int64, device=dev)
    xy = torch.empty(n * 16, dtype=torch.int64, device=dev)
    inf = torch.zeros(n, dtype=torch.uint8, device=dev)
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_scalars(K.KG_FR, SEED + 60, rank * n, n, k.data_ptr())
    ctx.fixed_base_mul(K.KG_G2, k.data_ptr(), n, xy.data_ptr(), inf.data_ptr())
    ctx.gen_scalars(K.KG_FR, SEED + 61, rank * n, n, sc.data_ptr())
    ctx.sync()
    args_ = (K.KG_G2, xy.data_ptr(), inf.data_ptr(), sc.data_ptr(), n)

    def run(kk, depth=4):
        res = None
        for i in range(kk):
            ctx.msm_begin(*args_, i % 4)
            if i >= depth - 1:
                res = ctx.msm_end(K.KG_G2, (i - depth + 1) % 4)
        for i in range(max(kk - depth + 1, 0), kk):
            res = ctx.msm_end(K.KG_G2, i % 4)
        return res
    run(24)
    ctx.profile_enable(True)
    elapsed, rounds_ms, res = timed_rounds(env, run, steps, rounds)
    summ = ctx.profile_summary()
    ctx.profile_enable(False)
    kphase = "accumulate" if "accumulate" in summ else "small_msm"       # (lengths within the short-input kernel's reach: one launch per MSM)
    pipe_kernel_ms = summ[kphase][0] / summ[kphase][1]
    ctx.set_msm_groups(1)                      # the kernel alone: ONE accumulation launch per MSM
    ctx.msm(*args_)
    ctx.profile_enable(True)
    for _ in range(5):
        ctx.msm(*args_)
    iso = ctx.profile_summary()
    ctx.profile_enable(False)
    ctx.set_msm_groups(0)
    for _ in range(2):
        blk = ctx.msm(*args_)
    br = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(4):
            blk = ctx.msm(*args_)
        br.append((time.perf_counter() - t0) / 4 * 1e3)
    iso_ms = iso[kphase][0] / iso[kphase][1]
    adds = window_adds(n)
    gbs = G2_BYTES_PER_PAIR * n / (pipe_kernel_ms * 1e-3) / 1e9
    iso_gbs = G2_BYTES_PER_PAIR * n / (iso_ms * 1e-3) / 1e9
    out = {"metric": "bn254_g2_msm_pairs_per_sec", "log_n": log_n, "value": world * n * steps / elapsed, "unit": "pairs/s", "ms_per_step": r3(elapsed / steps * 1e3),
           "rounds_ms": rounds_ms, "blocking_ms": r3(sorted(br)[2]), "pipelined_matches_blocking": bool((res == blk).all()),
           "roofline": {"bound": "hbm", "kernel": "k_acc_tasks<Fq2>", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "kernel_ms": pipe_kernel_ms, "launches": summ[kphase][1], "algorithmic_bytes_per_launch": G2_BYTES_PER_PAIR * n, "traffic": None,
                        "isolated": {"kernel_ms": iso_ms, "achieved": iso_gbs, "frac": iso_gbs / HBM_PEAK_GBS}},
           "valu_roofline": {"bound": "valu", "unit": "T v_mad_u64_u32/s", "mads_per_addition": MADS_PER_G2_ADDITION, "additions_per_launch": adds,
                             "achieved": adds * MADS_PER_G2_ADDITION / (iso_ms * 1e-3) / 1e12, "peak": MAD_PEAK_T,
                             "frac": adds * MADS_PER_G2_ADDITION / (iso_ms * 1e-3) / 1e12 / MAD_PEAK_T}}
    if cpu:
        from oracle import oracle as O
        hb = xy.cpu().numpy().view(np.uint64).reshape(n, 16)
        hi = inf.cpu().numpy()
        hs = sc.cpu().numpy().view(np.uint64).reshape(n, 4)
        lg = n.bit_length()
        c_ref = (lg * 69 // 100) + 2
        threads = max(1, min(256 // c_ref + 1, os.cpu_count() or 1))
        t0 = time.perf_counter()
        want_xy, want_inf = O.to_affine("g2", O.msm("g2", hb, hs, hi, threads=threads))
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / cdt, "unit": "pairs/s", "cores": threads, "kind": "port",
                               "sample": f"all {n} pairs, reference window rule (c = {c_ref}), {cdt:.2f} s",
                               "gpu_matches_cpu_at_full_size": bool(not want_inf and blk[16:].any() and (blk[:16] == want_xy).all())}
    del k, xy, inf, sc
    return out


def bench_small(ctx, torch, dev, K):
    """Short BLOCKING calls, the sizes the reference's own tests and BASELINE configs[0] live at (groth16/src/msm.rs:118-135: 32 pairs;
    bn254/benches: 2^10; groth16/src/lib.rs:29-77: a handful of constraints): latency of kg_msm on resident arrays for n = 16 ... 2^14 on
    the three curves, and a blocking proof of 2^10 constraints.  Median of five rounds of eight calls."""
    gc.collect()
    import numpy as np
    out = {"msm_blocking_ms": {}, "grumpkin_blocking_ms": {}, "g2_blocking_ms": {}}

    def lat(fn, reps=8):
        for _ in range(3):
            fn()
        rr = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            rr.append((time.perf_counter() - t0) / reps * 1e3)
        return r3(sorted(rr)[2])
    nmax = 1 << 14
    for name, curve, fld in (("msm_blocking_ms", K.KG_G1, K.KG_FR), ("grumpkin_blocking_ms", K.KG_GRUMPKIN, K.KG_FQ)):
        b = torch.empty(nmax * 8, dtype=torch.int64, device=dev)
        s = torch.empty(nmax * 4, dtype=torch.int64, device=dev)
        ctx.gen_bases(curve, SEED + 80, 0, nmax, b.data_ptr())
        ctx.gen_scalars(fld, SEED + 81, 0, nmax, s.data_ptr())
        ctx.sync()
        for nn in ((16, 32, 256, 1024, 4096, 16384) if curve == K.KG_G1 else (32, 1024)):
            out[name][str(nn)] = lat(lambda: ctx.msm(curve, b.data_ptr(), 0, s.data_ptr(), nn))
        del b, s
    k = torch.empty(1024 * 4, dtype=torch.int64, device=dev)
    xy = torch.empty(1024 * 16, dtype=torch.int64, device=dev)
    inf = torch.zeros(1024, dtype=torch.uint8, device=dev)
    ctx.gen_scalars(K.KG_FR, SEED + 82, 0, 1024, k.data_ptr())
    ctx.fixed_base_mul(K.KG_G2, k.data_ptr(), 1024, xy.data_ptr(), inf.data_ptr())
    ctx.sync()
    for nn in (32, 1024):
        out["g2_blocking_ms"][str(nn)] = lat(lambda: ctx.msm(K.KG_G2, xy.data_ptr(), inf.data_ptr(), k.data_ptr(), nn))
    del k, xy, inf
    # a blocking proof of 2^10 constraints (chain circuit, real CRS from the device setup)
    from kogarashi_amd import synthetic as syn
    from kogarashi_amd.api import groth16_setup
    from kogarashi_amd.lib import Groth16Crs
    m = 1 << 10
    cc = syn.ChainCircuit(m)
    P = groth16_setup(cc.a, cc.b, cc.c, m, cc.l, cc.m_l_1, syn.fixed_toxic(), syn.FrOps, ctx=ctx)
    up = lambda arr: torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).to(dev)
    names = ("h", "l", "a", "b_g1", "b_g2")
    dev_arr = {nm: up(P[nm]) for nm in names}
    dev_inf = {nm: (torch.from_numpy(P[nm + "_inf"]).to(dev) if P[nm + "_inf"].any() else None) for nm in names}
    crs = Groth16Crs()
    crs.m, crs.l, crs.m_l_1 = m, cc.l, cc.m_l_1
    for nm in names:
        setattr(crs, "d_" + nm, dev_arr[nm].data_ptr())
        if dev_inf[nm] is not None:
            setattr(crs, "d_" + nm + "_inf", dev_inf[nm].data_ptr())
        ctx.bases_register(K.KG_G2 if nm == "b_g2" else K.KG_G1, dev_arr[nm].data_ptr(), dev_inf[nm].data_ptr() if dev_inf[nm] is not None else 0,
                           dev_arr[nm].numel() // (16 if nm == "b_g2" else 8))
    for i in range(8):
        crs.alpha_g1[i], crs.beta_g1[i], crs.delta_g1[i] = int(P["vk_g1"][0, i]), int(P["vk_g1"][1, i]), int(P["vk_g1"][2, i])
    for i in range(16):
        crs.beta_g2[i], crs.delta_g2[i] = int(P["vk_g2"][0, i]), int(P["vk_g2"][1, i])
    r, s_ = syn.fixed_rs()
    d = [up(x_) for x_ in (cc.a_eval, cc.b_eval, cc.c_eval, cc.x, cc.w)]
    out["proof_2p10_blocking_ms"] = lat(lambda: ctx.groth16_prove(crs, *[t.data_ptr() for t in d], r, s_), reps=4)
    for nm in names:
        ctx.bases_unregister(dev_arr[nm].data_ptr())
    return out


def bench_ntt(ctx, torch, dev, K, env, log_n=22, steps=10, warmup=100, variants="all"):
    """secondary line: forward Fr NTT at 2^22 (BASELINE.json configs[2]), 64 algorithmic bytes per element, and -- `variants` -- the
    other three transforms of groth16/src/fft.rs:100-127 (idft, coset_dft, coset_idft: the n^-1 scale and the coset shifts are fused
    into the first load / last store of the same kernels) at the same size.  The transform does not shard (it would need an
    all-to-all transpose): with N ranks every rank transforms its own vector (replicas) and `value` is the aggregate."""
    gc.collect()
    world, rank = env["world"], env["rank"]
    n = 1 << log_n
    data = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_scalars(K.KG_FR, SEED + 3, rank * n, n, data.data_ptr())

    def timed(inverse, coset, wu):
        for _ in range(wu):       # untimed: tables, first touch, and the clock ramp after the CPU legs (the GPU sat idle behind them)
            ctx.ntt(data.data_ptr(), log_n, inverse, coset)
        ctx.sync()
        # the library brackets every transform with HIP events on the queue it launches on ("ntt" phase)
        ctx.profile_enable(True)
        env["barrier"]()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.ntt(data.data_ptr(), log_n, inverse, coset)
        ctx.sync()
        env["barrier"]()
        wall_ms = env["max_over_ranks"](time.perf_counter() - t0) / steps * 1e3
        tot, cnt = ctx.profile_summary()["ntt"]
        ctx.profile_enable(False)
        return tot / cnt, wall_ms
    mads, kernel_note, plan = ntt_cost(K, log_n)
    table = {"dft": (False, False), "idft": (True, False), "coset_dft": (False, True), "coset_idft": (True, True)}
    if variants not in ("all", "dft"):            # --ntt-only --ntt-variant X: that transform alone (its kernels' rocprof averages)
        ms, wall_ms = timed(*table[variants], warmup)
        return {"log_n": log_n, "variant": variants, "ms": r3(ms), "wall_ms": r3(wall_ms), "value": world * n / (wall_ms * 1e-3), "steps": steps}
    ms, wall_ms = timed(False, False, warmup)
    gbs = 64.0 * n / (ms * 1e-3) / 1e9
    out = {"metric": "bn254_fr_ntt_elements_per_sec", "log_n": log_n, "value": world * n / (wall_ms * 1e-3), "ms": r3(ms), "wall_ms": r3(wall_ms),
           "replicas": world, "steps": steps, "warmup": warmup,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes": 64 * n, "kernel": kernel_note, "traffic": ntt_pmc_traffic(log_n)}}
    note(out, "value = replicas x n / wall time per transform (max over ranks); ms = HIP-event duration of one transform on rank 0")
    if mads:
        out["valu_roofline"] = {"bound": "valu", "unit": "T v_mad_u64_u32/s", "mads_per_element": mads,
                                "achieved": mads * n / (ms * 1e-3) / 1e12, "peak": MAD_PEAK_T, "frac": mads * n / (ms * 1e-3) / 1e12 / MAD_PEAK_T}
    if variants == "all":
        out["variants"] = {}
        for name in ("idft", "coset_dft", "coset_idft"):
            vms, vwall = timed(*table[name], 20)
            out["variants"][name] = {"ms": r3(vms), "wall_ms": r3(vwall), "value": world * n / (vwall * 1e-3), "hbm_frac": r3(64.0 * n / (vms * 1e-3) / 1e9 / HBM_PEAK_GBS)}
    return out


def ntt_pmc_traffic(log_n):
    """HBM bytes per 2^22 transform from the committed counter passes (profiles/rNN_ntt_pmc_hbm.json: the transform's launches,
    FETCH_SIZE x2-corrected + WRITE_SIZE), or null"""
    if log_n != 22:
        return None
    for name in ("r06_ntt_pmc_hbm.json", "r05_ntt_pmc_hbm.json", "r04_ntt_pmc_hbm.json", "r03_ntt_pmc_hbm.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                with open(path) as f:
                    t = json.load(f)
                v = t.get("ntt_traffic_bytes_per_transform")
                return (v["total_corrected"] if isinstance(v, dict) else v) if v is not None else None
            except (OSError, ValueError, KeyError):
                return None
    return None


def bench_nova_commit(ctx, torch, dev, K, env, log_n=24, cpu=False, steps=5, skew=False):
    """BASELINE.json configs[4]: Nova's Pedersen commitment (nova/src/pedersen.rs:15-20) over 2^24 generators, on both
    curves of the cycle (bn254 G1 with Fr scalars, Grumpkin with Fq scalars; nova/src/driver.rs:9-42).  ONE commitment of
    2^log_n pairs is cut over the N ranks by kg_shard_range; every rank commits its slice (kg_commit) and the N affine
    partial sums are all-gathered and added (strong scaling of a fixed job: ms_per_commit is the number to watch).  The key
    g is fixed by PedersenCommitment::new, so it is registered (resident internal form) like a CRS vector.
    CPU legs (N = 1 only): the reference's naive fold timed on a prefix and extrapolated (2^24 of it would take hours), and
    the oracle's Pippenger restatement (msm_curve_addition) on all 2^24 pairs -- which also checks the GPU point."""
    gc.collect()
    import numpy as np
    from kogarashi_amd.lib import shard_range
    world, rank, kdist, xdev = env["world"], env["rank"], env["kdist"], env["xdev"]
    total = 1 << log_n
    lo, hi = shard_range(total, rank, world)
    nl = hi - lo
    out = {"metric": "nova_pedersen_commit_pairs_per_sec", "log_n": log_n, "pairs_total": total, "pairs_per_rank": nl, "ranks": world,
           "scaling": "strong", "key": "registered", "unit": "pairs/s", "algorithmic_bytes_per_commit": 96 * total}
    for name, curve, fld, cv in (("g1_fr", K.KG_G1, K.KG_FR, "g1"), ("grumpkin_fq", K.KG_GRUMPKIN, K.KG_FQ, "gk")):
        g = torch.empty(nl * 8, dtype=torch.int64, device=dev)
        m = torch.empty(nl * 4, dtype=torch.int64, device=dev)
        ctx.gen_bases(curve, SEED + 40 + curve, lo, nl, g.data_ptr())
        ctx.gen_scalars(fld, SEED + 41, lo, nl, m.data_ptr())
        ctx.sync()
        ctx.bases_register(curve, g.data_ptr(), 0, nl)

        def one():
            xy, inf = ctx.commit(curve, g.data_ptr(), 0, m.data_ptr(), nl)
            if world > 1:
                xy, inf = kdist.combine_partials(ctx, curve, xy, inf, device=xdev)
            return xy, inf
        one(); one()          # twice: consecutive calls alternate between the library's two sort spaces, and each is sized on first use
        env["barrier"]()
        t0 = time.perf_counter()
        for _ in range(steps):
            got = one()
        env["barrier"]()
        dt = env["max_over_ranks"](time.perf_counter() - t0) / steps
        leg = {"ms_per_commit": r3(dt * 1e3), "value": total / dt,
               "point": point_digest(got[0], got[1]),     # the commitment itself: equal for every N
               "hbm_frac": 96 * total / dt / 1e9 / (HBM_PEAK_GBS * world)}
        if world == 1:
            # As nova/src/pedersen.rs:15-20 is called: the key resident (kg_sharded_key over this one context), m a pageable HOST vector --
            # kg_sharded_key_commit uploads it in index slices under the accumulations.  PCIe-inclusive; beside ms_per_commit, never `value`.
            from kogarashi_amd.lib import ShardedKey
            hm = m.cpu().numpy().view(np.uint64).reshape(nl, 4)
            hg = g.cpu().numpy().view(np.uint64).reshape(nl, 8)
            key = ShardedKey([ctx], curve, hg)
            del hg
            for _ in range(2):
                hxy, hinf = key.commit(hm)
            t0 = time.perf_counter()
            for _ in range(steps):
                hxy, hinf = key.commit(hm)
            hdt = (time.perf_counter() - t0) / steps
            key.close()
            leg["from_host"] = {"ms_per_commit": r3(hdt * 1e3), "ratio_to_resident": r3(hdt / dt),
                                "matches_resident": bool(bool(hinf) == bool(got[1]) and (bool(hinf) or (np.asarray(hxy) == np.asarray(got[0])).all()))}
            del hm
        if skew and name == "g1_fr" and world == 1:
            # the same commitment of a witness-like vector (half ones, a fifth zeros: what a folded R1CS witness looks like,
            # nova/src/relaxed_r1cs/witness.rs:56-70) -- sort-bound instead of accumulation-bound; checked by committing the halves
            from kogarashi_amd import synthetic as syn
            hw = m.cpu().numpy().view(np.uint64).reshape(nl, 4)
            syn.witness_like(hw, 23)
            mw = torch.from_numpy(hw.view(np.int64).reshape(-1)).to(dev)
            del hw
            for _ in range(2):
                ctx.commit(curve, g.data_ptr(), 0, mw.data_ptr(), nl)
            t0 = time.perf_counter()
            for _ in range(3):
                wxy, winf = ctx.commit(curve, g.data_ptr(), 0, mw.data_ptr(), nl)
            wdt = (time.perf_counter() - t0) / 3
            h = nl // 2
            a = ctx.commit(curve, g.data_ptr(), 0, mw.data_ptr(), h)
            b = ctx.commit(curve, g.data_ptr() + h * 64, 0, mw.data_ptr() + h * 32, nl - h)
            sxy, sinf = ctx.points_sum_affine(curve, np.stack([a[0], b[0]]), np.array([a[1], b[1]], dtype=np.uint8))
            leg["witness_like"] = {"ms_per_commit": r3(wdt * 1e3), "ratio_to_uniform": r3(wdt / dt),
                                   "halves_add_up": bool(bool(sinf) == bool(winf) and (bool(winf) or (np.asarray(sxy) == np.asarray(wxy)).all()))}
            del mw
        if cpu:
            from oracle import oracle as O
            hg = g.cpu().numpy().view(np.uint64).reshape(nl, 8)
            hm = m.cpu().numpy().view(np.uint64).reshape(nl, 4)
            # (1) the reference's own commit: a sequential fold of naive scalar multiplications (pedersen.rs:15-20) -- on G1 (Grumpkin's is the same loop)
            if name == "g1_fr":
                k = 1 << 12
                t0 = time.perf_counter()
                O.commit_naive(cv, hg[:k], hm[:k])
                cdt = time.perf_counter() - t0
                if cdt < 1.6 and nl >= (1 << 14):
                    k = 1 << 14
                    t0 = time.perf_counter()
                    O.commit_naive(cv, hg[:k], hm[:k])
                    cdt = time.perf_counter() - t0
                leg["cpu_naive_fold"] = {"value": k / cdt, "unit": "pairs/s", "cores": 1, "kind": "port",
                                         "sample": f"first {k} pairs, {cdt:.2f} s", "extrapolated_s_per_commit": round(cdt * total / k)}
            # (2) like for like at full size: the oracle's restatement of msm_curve_addition, one thread per window
            lg = nl.bit_length()
            c_ref = (lg * 69 // 100) + 2
            threads = max(1, min(256 // c_ref + 1, os.cpu_count() or 1))
            t0 = time.perf_counter()
            want = O.to_affine(cv, O.msm(cv, hg, hm, None, threads=threads))
            cdt = time.perf_counter() - t0
            leg["cpu_pippenger"] = {"value": nl / cdt, "unit": "pairs/s", "cores": threads, "kind": "port",
                                    "sample": f"all {nl} pairs, c = {c_ref}, {cdt:.2f} s",
                                    "gpu_matches_cpu_at_full_size": bool(want[1] == got[1] and (want[1] or (want[0] == got[0]).all()))}
            del hg, hm
        ctx.bases_unregister(g.data_ptr())
        del g, m
        out[name] = leg
    out["value"] = out["g1_fr"]["value"]
    if world == 1 and log_n >= 23:
        # the per-rank unit of the 8-GPU configuration: ONE blocking commit of 2^(log_n - 3) registered pairs (what each rank of
        # BASELINE configs[4] runs before the all-gather), beside half of a 2^(log_n - 2)-pair commit for the latency overhead
        unit = {}
        for lg in (log_n - 3, log_n - 2):
            nn = 1 << lg
            g = torch.empty(nn * 8, dtype=torch.int64, device=dev)
            m = torch.empty(nn * 4, dtype=torch.int64, device=dev)
            ctx.gen_bases(K.KG_G1, SEED + 50, 0, nn, g.data_ptr())
            ctx.gen_scalars(K.KG_FR, SEED + 51, 0, nn, m.data_ptr())
            ctx.sync()
            ctx.bases_register(K.KG_G1, g.data_ptr(), 0, nn)
            for _ in range(3):
                ctx.commit(K.KG_G1, g.data_ptr(), 0, m.data_ptr(), nn)
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.commit(K.KG_G1, g.data_ptr(), 0, m.data_ptr(), nn)
            unit[lg] = (time.perf_counter() - t0) / 10 * 1e3
            if lg == log_n - 3:
                hm = m.cpu().numpy().view(np.uint64).reshape(nn, 4)
                for _ in range(3):
                    ctx.commit_host_scalars(K.KG_G1, g.data_ptr(), 0, hm, nn)
                t0 = time.perf_counter()
                for _ in range(10):
                    ctx.commit_host_scalars(K.KG_G1, g.data_ptr(), 0, hm, nn)
                unit["host"] = (time.perf_counter() - t0) / 10 * 1e3
                del hm
            ctx.bases_unregister(g.data_ptr())
            del g, m
        a, b = unit[log_n - 3], unit[log_n - 2]
        out["rank_unit"] = note({"log_n": log_n - 3, "blocking_ms_per_commit": r3(a), "half_of_twice_the_size_ms": r3(b / 2), "ratio": r3(a / (b / 2)),
                                 "from_host_ms_per_commit": r3(unit["host"])},
                                "blocking kg_commit, registered key, bn254 G1: the slice one rank of the 8-GPU configuration commits; "
                                "from_host = the same slice with its scalars in pageable host memory (kg_commit_host_scalars, PCIe-inclusive)")
    return out


def cpu_plumbing(ctx, K, log_n=10):
    """BASELINE.json configs[0]: bn254 G1 MSM over 2^10 random pairs on the CPU (the reference's bn254/benches size), with the blocking
    kg_msm_host of the same host arrays beside it (latency: see `small` for the resident calls)."""
    import numpy as np
    from oracle import oracle as O
    n = 1 << log_n
    hb = O.gen_bases(0, SEED + 1, 0, n)
    hs = O.gen_scalars(0, SEED + 2, 0, n)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        r = O.msm("g1", hb, hs, None, threads=1)
    dt = (time.perf_counter() - t0) / reps
    xy, inf = O.to_affine("g1", r)
    inf0 = np.zeros(n, dtype=np.uint8)
    got = ctx.msm_host(K.KG_G1, hb, inf0, hs, n)
    t0 = time.perf_counter()
    for _ in range(reps):
        got = ctx.msm_host(K.KG_G1, hb, inf0, hs, n)
    gdt = (time.perf_counter() - t0) / reps
    return {"value": n / dt, "unit": "pairs/s", "cores": 1, "kind": "port", "sample": f"{reps} x 2^{log_n} pairs, {dt * 1e3:.2f} ms each",
            "gpu_host_call_ms": r3(gdt * 1e3), "gpu_matches_cpu": bool(not inf and (got[:8] == xy).all())}


def bench_msm_skewed(ctx, torch, dev, K, bases, n, run_uniform, barrier, steps, uniform_ms, uniform_blocking_ms, cpu):
    """The headline MSM with scalars distributed like a Groth16 aux vector (half ones, a fifth zeros, a tenth -1, a tenth a small
    value, a tenth uniform: kogarashi_amd.synthetic.witness_like) instead of uniform ones -- SURVEY.md 7, hard part (iii): half of
    window 0's entries meet one bucket and the upper windows of the small values are empty, so the sort's segments, the task
    cutting and the partial-sum rounds carry the load balance.  Same bases, same pipeline depth; `partial_rounds_per_msm` counts
    the extra summation rounds (k_sum_tasks) an MSM needed."""
    import numpy as np
    from kogarashi_amd import synthetic as syn
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ctx.gen_scalars(K.KG_FR, SEED + 7, 0, n, sc.data_ptr())
    ctx.sync()
    h = sc.cpu().numpy().view(np.uint64).reshape(n, 4).copy()
    syn.witness_like(h, 11)
    sc.copy_(torch.from_numpy(h.view(np.int64).reshape(-1)))
    torch.cuda.synchronize()

    def run(k, depth=4):
        res = None
        for i in range(k):
            ctx.msm_begin(K.KG_G1, bases.data_ptr(), 0, sc.data_ptr(), n, i % 4)
            if i >= depth - 1:
                res = ctx.msm_end(K.KG_G1, (i - depth + 1) % 4)
        for i in range(max(k - depth + 1, 0), k):
            res = ctx.msm_end(K.KG_G1, i % 4)
        return res
    run(8)
    barrier()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    res = run(steps)
    barrier()
    ms = (time.perf_counter() - t0) / steps * 1e3
    summ = ctx.profile_summary()
    ctx.profile_enable(False)
    for _ in range(2):
        blk = ctx.msm(K.KG_G1, bases.data_ptr(), 0, sc.data_ptr(), n)
    t0 = time.perf_counter()
    for _ in range(5):
        blk = ctx.msm(K.KG_G1, bases.data_ptr(), 0, sc.data_ptr(), n)
    blocking = (time.perf_counter() - t0) / 5 * 1e3
    out = {"ms_per_step": r3(ms), "ratio_to_uniform": r3(ms / uniform_ms), "blocking_ms": r3(blocking), "blocking_ratio_to_uniform": r3(blocking / uniform_blocking_ms),
           "partial_rounds_per_msm": (summ["partial_round"][1] / steps) if "partial_round" in summ else 0.0,
           "pipelined_matches_blocking": bool((res == blk).all()),
           }
    if cpu:
        from oracle import oracle as O
        hb = bases.cpu().numpy().view(np.uint64).reshape(n, 8)
        want_xy, want_inf = O.to_affine("g1", O.msm("g1", hb, h, None, threads=17))
        out["gpu_matches_cpu_at_full_size"] = bool(not want_inf and (res[:8] == want_xy).all())
    del sc
    return out


def median_rounds(fn, sync, mx, rounds=3):
    """fn() -> (units, result) timed `rounds` times between synchronisations; (median seconds per unit, [ms per unit of every round], last result).
    One timed window of 25-50 ms catches a runtime stall of a few milliseconds every few runs (profiles/r06_stalls.txt): +10 % on the figure."""
    secs, res = [], None
    for _ in range(rounds):
        sync()
        t0 = time.perf_counter()
        units, res = fn()
        sync()
        secs.append(mx(time.perf_counter() - t0) / units)
    return sorted(secs)[len(secs) // 2], [round(x * 1e3, 4) for x in secs], res


def bench_groth16(ctx, torch, dev, K, env, log_m=18, steps=8, cpu=True, tables=True, tickets=2, circuit="chain", from_witness=True):
    """secondary line: Groth16 prove at m = 2^log_m constraints (BASELINE.json configs[3]).  Circuit: the chain
    t_{i+1} = t_i * (t_i + 1) (x = [1, t_0], w = t_1..t_m); CRS:
    a real CRS from a fixed toxic waste, generated on the device; fixed (r, s).  The CPU leg runs the oracle's create_proof on the same inputs and compares the proof."""
    gc.collect()
    import numpy as np
    from kogarashi_amd.lib import Groth16Crs
    world, sync, mx = env["world"], env["barrier"], env["max_over_ranks"]     # N ranks: one prover per rank (replicas), aggregate proofs/s
    from kogarashi_amd import synthetic as syn
    m = 1 << log_m
    cc = syn.ChainCircuit(m) if circuit == "chain" else syn.BooleanHeavyCircuit(m)
    l, m_l_1 = cc.l, cc.m_l_1
    a_ev, b_ev, c_ev, x, w = cc.a_eval, cc.b_eval, cc.c_eval, cc.x, cc.w
    up = lambda arr: torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).to(dev)
    d_a, d_b, d_c, d_x, d_w = up(a_ev), up(b_ev), up(c_ev), up(x), up(w)
    # a REAL CRS for this circuit: ZkSnark::setup composed from the device primitives (kogarashi_amd.api.groth16_setup,
    # parity-tested against the oracle in tests/test_gpu_groth16.py), toxic waste fixed
    from kogarashi_amd.api import groth16_setup
    a_csr, b_csr, c_csr, FrOps = cc.a, cc.b, cc.c, syn.FrOps
    toxic = syn.fixed_toxic()
    t0 = time.perf_counter()
    P = groth16_setup(a_csr, b_csr, c_csr, m, l, m_l_1, toxic, FrOps, ctx=ctx)
    setup_first_ms = (time.perf_counter() - t0) * 1e3          # first call on the context: builds the generator window tables too
    t0 = time.perf_counter()
    P = groth16_setup(a_csr, b_csr, c_csr, m, l, m_l_1, toxic, FrOps, ctx=ctx)
    setup_ms = (time.perf_counter() - t0) * 1e3                # kg_groth16_setup_bn254 + upload of the matrices + download of Parameters
    dev_arr = {name: up(P[name]) for name in ("h", "l", "a", "b_g1", "b_g2")}
    dev_inf = {name: (torch.from_numpy(P[name + "_inf"]).to(dev) if P[name + "_inf"].any() else None) for name in ("h", "l", "a", "b_g1", "b_g2")}
    vk_g1, vk_g2 = P["vk_g1"], P["vk_g2"]
    crs = Groth16Crs()
    crs.m, crs.l, crs.m_l_1 = m, l, m_l_1
    for name in ("h", "l", "a", "b_g1", "b_g2"):
        setattr(crs, "d_" + name, dev_arr[name].data_ptr())
        if dev_inf[name] is not None:
            setattr(crs, "d_" + name + "_inf", dev_inf[name].data_ptr())
        # Parameters is immutable: like kogarashi_amd.api.Prover, convert each CRS vector to the internal form once
        ctx.bases_register(K.KG_G2 if name == "b_g2" else K.KG_G1, dev_arr[name].data_ptr(),
                           dev_inf[name].data_ptr() if dev_inf[name] is not None else 0, dev_arr[name].numel() // (16 if name == "b_g2" else 8))
    for i in range(8):
        crs.alpha_g1[i], crs.beta_g1[i], crs.delta_g1[i] = int(vk_g1[0, i]), int(vk_g1[1, i]), int(vk_g1[2, i])
    for i in range(16):
        crs.beta_g2[i], crs.delta_g2[i] = int(vk_g2[0, i]), int(vk_g2[1, i])
    r, s_ = syn.fixed_rs()
    args_ = (crs, d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), d_x.data_ptr(), d_w.data_ptr(), r, s_)
    prove = lambda: ctx.groth16_prove(*args_)
    proof = prove()
    for _ in range(WARM_PROOFS):          # untimed: the GPU idled behind the CPU legs; ~60 ms of load bring it back to its clock
        prove()
    def blocking_round():
        pr = None
        for _ in range(steps):
            pr = prove()
        return steps, pr
    dt_blocking, blocking_rounds, proof = median_rounds(blocking_round, sync, mx)
    # throughput: proofs issued back to back, two in flight (kg_groth16_prove_begin / _end) -- proof i+1's transforms and
    # sorts run under proof i's last reduction and host assembly; every proof is produced inside the timed region
    def run(k, depth=tickets):
        last = None
        for i in range(k):
            ctx.groth16_prove_begin(*args_, i % depth)
            if i >= depth - 1:
                last = ctx.groth16_prove_end((i - depth + 1) % depth)
        for i in range(max(k - depth + 1, 0), k):
            last = ctx.groth16_prove_end(i % depth)
        return last
    run(2)
    k_pipe = max(2 * steps, 4)
    dt, flight_rounds, proof_p = median_rounds(lambda: (k_pipe, run(k_pipe)), sync, mx)
    out = {"metric": "groth16_proofs_per_sec", "log_m": log_m, "value": world / dt, "ms_per_proof": r3(dt * 1e3), "replicas": world,
           "ms_per_proof_blocking": r3(dt_blocking * 1e3), "rounds_ms": flight_rounds, "blocking_rounds_ms": blocking_rounds,
           "pipelined_matches_blocking": bool(all((proof_p[i] == proof[i]).all() for i in range(4))),
           "algorithmic_bytes_per_proof": (7 * 64 + 4 * 32 + 4 * 96 + 160) * m,      # 1120 B per constraint
           "setup_ms": r3(setup_ms), "setup_first_ms": r3(setup_first_ms)}
    note(out, "setup: ZkSnark::setup (zksnark.rs:17-127) through kg_groth16_setup_bn254: matrices uploaded, CRS computed on the device, "
              "Parameters downloaded; first = with the one-off generator window tables of the context")
    out["roofline"] = {"bound": "hbm", "achieved": out["algorithmic_bytes_per_proof"] / dt / 1e9, "frac": out["algorithmic_bytes_per_proof"] / dt / 1e9 / HBM_PEAK_GBS}
    # the same proofs with window tables on the five CRS vectors (kg_bases_precompute: 2^(c w) * P for every window, built once
    # per CRS): one bucket set for all windows of an MSM; proofs must be bit-identical
    tables = tables and (1 << 16) <= (l + m_l_1) <= (1 << 20) and (m - 1) >= (1 << 16)      # kg_bases_precompute: MSMs of 2^16 .. 2^20 scalars
    if tables:
        nz = l + m_l_1
        t0 = time.perf_counter()
        for name in ("a", "b_g1", "b_g2", "l"):
            ctx.bases_precompute(dev_arr[name].data_ptr(), nz)
        ctx.bases_precompute(dev_arr["h"].data_ptr(), m - 1)
        ctx.sync()
        build_ms = (time.perf_counter() - t0) * 1e3
        proof_t = prove()
        for _ in range(WARM_PROOFS):
            prove()
        dt_tb, _, proof_t = median_rounds(blocking_round, sync, mx)
        run(2)
        dt_t, _, proof_tp = median_rounds(lambda: (k_pipe, run(k_pipe)), sync, mx)
        out["window_tables"] = {"ms_per_proof": r3(dt_t * 1e3), "ms_per_proof_blocking": r3(dt_tb * 1e3), "value": world / dt_t, "build_ms": r3(build_ms),
                                "proofs_match": bool(all((proof_t[i] == proof[i]).all() and (proof_tp[i] == proof[i]).all() for i in range(4)))}
    if tables and from_witness:
        # and from the witness alone: the constraint matrices resident as CSR, cs.evaluate() on the device at the head of the
        # transform chains (kg_groth16_prove_r1cs_begin) -- what the patched create_proof calls
        csr_dev = [tuple(torch.from_numpy(np.ascontiguousarray(np.asarray(x_, dtype=np.uint64)).view(np.int64).reshape(-1)).to(dev) for x_ in trip)
                   for trip in (a_csr, b_csr, c_csr)]
        csr_ptr = [tuple(t.data_ptr() for t in trip) for trip in csr_dev]
        w_args = (crs, csr_ptr[0], csr_ptr[1], csr_ptr[2], d_x.data_ptr(), d_w.data_ptr(), r, s_)

        def run_w(k, depth=tickets):
            last = None
            for i in range(k):
                ctx.groth16_prove_r1cs_begin(*w_args, i % depth)
                if i >= depth - 1:
                    last = ctx.groth16_prove_end((i - depth + 1) % depth)
            for i in range(max(k - depth + 1, 0), k):
                last = ctx.groth16_prove_end(i % depth)
            return last
        run_w(2)
        sync()
        t0 = time.perf_counter()
        proof_w = run_w(k_pipe)
        sync()
        dt_w = mx(time.perf_counter() - t0) / k_pipe
        out["window_tables"].update({"ms_per_proof_from_witness": r3(dt_w * 1e3), "from_witness_matches": bool(all((proof_w[i] == proof[i]).all() for i in range(4)))})
        # and as the patched Prover::create_proof calls it (rust/kogarashi-amd/src/groth16.rs prove_cs_with): BLOCKING, x = cs.x() and w = cs.w()
        # in host memory, uploaded into kg_malloc buffers per proof (the pool hands the same blocks back) -- PCIe-inclusive, never `value`
        hx, hw = np.ascontiguousarray(x), np.ascontiguousarray(w)

        def host_proof():
            dx, dw = ctx.upload(hx), ctx.upload(hw)
            return ctx.groth16_prove_r1cs(crs, csr_ptr[0], csr_ptr[1], csr_ptr[2], dx.ptr, dw.ptr, r, s_)
        for _ in range(3):
            proof_h = host_proof()
        t0 = time.perf_counter()
        for _ in range(steps):
            proof_h = host_proof()
        dt_h = (time.perf_counter() - t0) / steps
        out["window_tables"].update({"ms_per_proof_blocking_host_witness": r3(dt_h * 1e3),
                                     "host_witness_matches": bool(all((proof_h[i] == proof[i]).all() for i in range(4)))})
    # the CPU leg LAST: a second of 32 busy host threads in front of a timed GPU leg costs it 3-5 % (clocks, host threads)
    if cpu:
        from oracle import oracle as O
        Pc = {name: P[name] for name in ("h", "l", "a", "b_g1", "b_g2")}
        Pc.update({name + "_inf": (P[name + "_inf"] if P[name + "_inf"].any() else None) for name in ("h", "l", "a", "b_g1", "b_g2")})
        Pc["vk_g1"], Pc["vk_g2"] = vk_g1, np.concatenate([vk_g2, vk_g2[:1]])
        cs = O.R1cs((np.zeros(m + 1, dtype=np.uint64),) * 3, (np.zeros(m + 1, dtype=np.uint64),) * 3, (np.zeros(m + 1, dtype=np.uint64),) * 3, x, w)
        threads = min(32, os.cpu_count() or 1)
        t0 = time.perf_counter()
        want = O.groth16_prove(cs, Pc, r, s_, threads=threads, evals=(a_ev, b_ev, c_ev))
        cdt = time.perf_counter() - t0
        same = all((g == w_).all() for g, w_ in zip(proof[:3], want[:3])) and (proof[3] == want[3]).all()
        out["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "proofs/s", "cores": threads, "kind": "port",
                               "sample": f"one proof at m = 2^{log_m}, {cdt:.2f} s", "gpu_matches_cpu_at_full_size": bool(same)}
    for name in ("h", "l", "a", "b_g1", "b_g2"):
        ctx.bases_unregister(dev_arr[name].data_ptr())
    if world > 1 and circuit == "chain":
        # ONE proof over several GPUs, task-parallel (kg_groth16_prove_sharded: the G2 query | the three G1 queries | transforms
        # and h's MSM on contexts 0, 1, 2): rank 0 drives contexts on the first min(3, N) devices while the other ranks wait.
        # (KG_BENCH_SELFTEST: every context on cuda:0 -- control flow only.)  The waiting ranks park on a HOST barrier (gloo): an RCCL
        # barrier kernel would spin on the GPUs whose proof is being timed.
        env["host_barrier"]()
        if env["rank"] == 0:
            from kogarashi_amd.api import ShardedProver
            n_ctx = min(3, world)
            selftest = os.environ.get("KG_BENCH_SELFTEST") == "1"
            ctxs = [K.Context(0 if selftest else i) for i in range(n_ctx)]
            for c_ in ctxs:
                c_.set_inputs_complete(True)
            params = {name: P[name] for name in ("h", "l", "a", "b_g1", "b_g2")}
            params.update({name + "_inf": P[name + "_inf"] for name in ("h", "l", "a", "b_g1", "b_g2")})
            params["vk_g1"], params["vk_g2"] = vk_g1, vk_g2
            sp = ShardedProver(params, m, l, m_l_1, ctxs)
            ptr, keep = sp.upload_inputs(a_ev, b_ev, c_ev, x, w)
            for _ in range(3):
                proof_s = sp.prove_resident(ptr, r, s_)
            t0 = time.perf_counter()
            for _ in range(steps):
                proof_s = sp.prove_resident(ptr, r, s_)
            dt_s = (time.perf_counter() - t0) / steps
            out["sharded"] = {"contexts": n_ctx, "devices": [0 if selftest else i for i in range(n_ctx)], "ms_per_proof": r3(dt_s * 1e3),
                              "matches_single_context": bool(all((proof_s[i] == proof[i]).all() for i in range(4)))}
            del sp, keep
            for c_ in ctxs:
                c_.close()
        env["host_barrier"]()
    return out


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
