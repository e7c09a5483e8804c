"""Round-5 test holes: MSMs longer than 2^24 (groth16/src/msm.rs:6-48 takes any length), the caller-stream
contract of kg_ctx_set_stream, and allocation failures (the library never aborts -- KG_ERR_OOM, context intact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x4B6F676172617368


@pytest.fixture(scope="module")
def ctx():
    import kogarashi_amd as K
    c = K.Context(0)
    yield c
    c.close()


# ---- MSMs beyond 2^24 pairs ------------------------------------------------------------------------------------------------------
def test_msm_2_25_matches_the_oracles_pippenger(ctx, oracle):
    """2^25 pairs: past the two-pass sort's index field, kg_msm runs four index slices of 2^23 (c = 17 each) and adds their sums.
    Against the oracle's restatement of msm_curve_addition at full size (window rule c = 19, one thread per window)."""
    import kogarashi_amd as K
    n = 1 << 25
    g, m = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, SEED + 60, 0, n, g.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 61, 0, n, m.ptr)
    got = ctx.msm(K.KG_G1, g.ptr, 0, m.ptr, n)
    want_xy, want_inf = oracle.to_affine("g1", oracle.msm("g1", g.numpy(), m.numpy(), None, threads=14))
    assert not want_inf and (got[:8] == want_xy).all() and (got[8:] == oracle.f_consts(1)["r"]).all()
    # the same pairs with the scalars in host memory (eight slices of 2^22 under the uploads)
    assert (ctx.msm_host_scalars(K.KG_G1, g.ptr, 0, m.numpy(), n) == got).all()


def test_msm_2_26_plus_ragged_equals_the_sum_of_its_parts(ctx):
    """2^26 + 12345 pairs: slices longer than 2^24 fall back to c = 16 with the ONE-pass sort (whole-window histogram in LDS, 32-bit
    entries).  Size-independent property: MSM(whole) = sum of MSM(part) over a ragged cut into parts the tested paths cover
    (<= 2^24: two-pass sort, window groups / wide windows), added with kg_points_sum_affine."""
    import kogarashi_amd as K
    n = (1 << 26) + 12345
    g, m = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, SEED + 62, 0, n, g.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 63, 0, n, m.ptr)
    whole = ctx.msm(K.KG_G1, g.ptr, 0, m.ptr, n)
    cuts = [0, (1 << 24), (1 << 24) + (1 << 23) + 77, 3 * (1 << 24) - 5, (1 << 26) - 1, n]
    parts, infs = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        xy, inf = ctx.commit(K.KG_G1, g.ptr + lo * 64, 0, m.ptr + lo * 32, hi - lo)
        parts.append(xy); infs.append(inf)
    xy, inf = ctx.points_sum_affine(K.KG_G1, np.stack(parts), np.array(infs, dtype=np.uint8))
    assert not inf and (whole[:8] == xy).all()
    # and a length just past the two-pass limit, unsliced (kg_msm_begin never slices): one pass, c = 16
    n1 = (1 << 24) + 3
    ctx.msm_begin(K.KG_G1, g.ptr, 0, m.ptr, n1, 0)
    one_pass = ctx.msm_end(K.KG_G1, 0)
    a = ctx.commit(K.KG_G1, g.ptr, 0, m.ptr, 1 << 24)
    b = ctx.commit(K.KG_G1, g.ptr + (1 << 24) * 64, 0, m.ptr + (1 << 24) * 32, 3)
    xy, inf = ctx.points_sum_affine(K.KG_G1, np.stack([a[0], b[0]]), np.array([a[1], b[1]], dtype=np.uint8))
    assert not inf and (one_pass[:8] == xy).all()


# ---- kg_ctx_set_stream -----------------------------------------------------------------------------------------------------------
def test_caller_stream_orders_producers_before_the_msm_and_the_ntt(oracle):
    """A torch host hands the library its current stream (kg_ctx_set_stream) and enqueues, WITHOUT synchronising: a long torch kernel
    chain, a device copy that produces the inputs, kg_field_vec_op (the producer kernel on the caller's stream), then kg_msm and
    kg_ntt on its output.  The MSM's scalar side runs on the library's own queues: it must be ordered behind everything the caller's
    stream holds (stream semantics, include/kogarashi_amd.h kg_ctx_set_inputs_complete) -- a missing hand-over would read the
    zero-filled buffers.  Then back to the context's own stream."""
    import torch
    import kogarashi_amd as K
    O, n, k = oracle, (1 << 17) + 11, 14
    ctx = K.Context(0)
    dev = torch.device("cuda", 0)
    bases = O.gen_bases(0, SEED + 70, 0, n)
    a, b = O.gen_scalars(0, SEED + 71, 0, n), O.gen_scalars(0, SEED + 72, 0, n)
    want_s = O.f_vec_mul(0, a, b)
    want_xy, want_inf = O.to_affine("g1", O.msm("g1", bases, want_s, None, threads=8))
    want_ntt = O.Fft(k).coset_dft(want_s[: 1 << k])
    t_bases = torch.from_numpy(bases.view(np.int64).reshape(-1)).to(dev)
    src_a = torch.from_numpy(a.view(np.int64).reshape(-1)).to(dev)
    src_b = torch.from_numpy(b.view(np.int64).reshape(-1)).to(dev)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(device=dev)
    for rep in range(3):
        t_a, t_b = torch.zeros_like(src_a), torch.zeros_like(src_b)
        t_s = torch.zeros_like(src_a)
        t_v = torch.zeros(4 << k, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ctx.set_stream(s.cuda_stream)
        with torch.cuda.stream(s):
            x = torch.randn(4096, 4096, device=dev)
            for _ in range(40):                      # ~10 ms of matrix products in front of the producers
                x = (x @ x).clamp_(-1, 1)
            t_a.copy_(src_a, non_blocking=True)
            t_b.copy_(src_b, non_blocking=True)
            ctx.field_vec_op(K.KG_FR, "mul", t_a.data_ptr(), t_b.data_ptr(), t_s.data_ptr(), n)
            got = ctx.msm(K.KG_G1, t_bases.data_ptr(), 0, t_s.data_ptr(), n)              # blocking: window groups, scalar queue
            ctx.msm_begin(K.KG_G1, t_bases.data_ptr(), 0, t_s.data_ptr(), n, 1)           # ticketed: scalar queue behind the main queue
            t_v.copy_(t_s[: 4 << k], non_blocking=True)                                     # torch op behind the library's producer
            ctx.ntt(t_v.data_ptr(), k, False, True)
            got_t = ctx.msm_end(K.KG_G1, 1)
            s.synchronize()
        assert not want_inf and (got[:8] == want_xy).all() and (got_t == got).all(), rep
        assert (t_v.cpu().numpy().view(np.uint64).reshape(-1, 4) == want_ntt).all(), rep
        ctx.set_stream(0)                            # the context's own stream again (drains the caller's)
        d = ctx.upload(want_s)
        assert (ctx.msm(K.KG_G1, t_bases.data_ptr(), 0, d.ptr, n) == got).all()
    ctx.close()


# ---- allocation failures ---------------------------------------------------------------------------------------------------------
def _hog(ctx, leave):
    """device allocations until at most `leave` bytes are free; returns them"""
    held = []
    for chunk in (64 << 30, 8 << 30, 1 << 30, 128 << 20, 16 << 20):
        while True:
            free, _ = ctx.mem_info()
            if free <= leave + chunk:
                break
            try:
                held.append(ctx.malloc(chunk))
            except Exception:
                break
    return held


def test_absurd_allocation_is_a_status_code_and_the_context_survives(ctx, oracle):
    import kogarashi_amd as K
    with pytest.raises(K.KogarashiError, match="out of device memory"):
        ctx.malloc(1 << 50)
    free, total = ctx.mem_info()
    assert 0 < free <= total and total > (200 << 30)                # an MI355X: 288 GB
    n = 3000
    b, s = oracle.gen_bases(0, SEED + 80, 0, n), oracle.gen_scalars(0, SEED + 81, 0, n)
    want_xy, _ = oracle.to_affine("g1", oracle.msm("g1", b, s, None, threads=4))
    assert (ctx.msm_host(K.KG_G1, b, None, s, n)[:8] == want_xy).all()


def test_released_blocks_are_kept_for_the_next_request_and_given_back_under_pressure():
    """kg_malloc / kg_free keep released blocks per size class (capi.cpp: the first copy into a FRESH allocation runs at 1-5 GB/s, into a used
    block at 56 GB/s): the same request gets the same block back, kept blocks count as free memory, KG_POOL_MB bounds what is kept, and a
    work-space request the device refuses releases them instead of failing."""
    import kogarashi_amd as K
    ctx = K.Context(0)
    free0, _ = ctx.mem_info()
    a = ctx.malloc(33 << 20)
    ctx.free(a)
    assert ctx.mem_info()[0] >= free0 - (1 << 20)                   # the kept 33 MiB count as free
    b = ctx.malloc((33 << 20) - 4096)                               # same size class (whole MiB above 1 MiB)
    assert b == a
    c = ctx.malloc(33 << 20)                                        # a second block of the class: a new allocation
    assert c != b
    ctx.free(b); ctx.free(c)
    big = [ctx.malloc(3 << 30) for _ in range(4)]                   # 12 GiB released with a 1 GiB pool: none of them is kept
    for p in big:
        ctx.free(p)
    # pressure: filling the device takes the kept blocks' memory too -- a request the device refuses releases them and is repeated
    held = _hog(ctx, 64 << 20)
    assert ctx.mem_info()[0] <= (64 << 20) + (16 << 20)             # nothing is left, kept or free
    with pytest.raises(K.KogarashiError, match="out of device memory"):
        ctx.malloc(1 << 30)
    for p in held:
        ctx.free(p)
    n = 1 << 18
    g, m = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, SEED + 90, 0, n, g.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 91, 0, n, m.ptr)
    assert ctx.msm(K.KG_G1, g.ptr, 0, m.ptr, n)[8:].any()
    ctx.close()


def test_kept_blocks_can_be_trimmed_and_a_block_freed_through_another_context_is_released():
    """kg_ctx_trim gives the kept blocks back to the driver (other allocators of the device -- torch, other contexts -- see them
    as used until then); a block handed out by context A and freed through context B is released, not kept under A's old entry; a
    work-space request that context B cannot satisfy releases context A's kept blocks too."""
    import os
    import torch
    import kogarashi_amd as K
    if os.environ.get("KG_POOL_MB"):
        pytest.skip("the test keeps a 256 MiB block: it assumes the default cap of the block pool (KG_POOL_MB is set)")
    a, b = K.Context(0), K.Context(0)
    torch.cuda.synchronize()
    p = a.malloc(256 << 20)
    a.free(p)                                                       # kept by A
    hip_free_kept = torch.cuda.mem_get_info(0)[0]
    assert a.mem_info()[0] >= hip_free_kept + (255 << 20)           # the library counts its kept bytes as free; the driver does not
    a.trim()
    assert torch.cuda.mem_get_info(0)[0] >= hip_free_kept + (200 << 20)
    assert abs(a.mem_info()[0] - torch.cuda.mem_get_info(0)[0]) < (64 << 20)
    q = a.malloc(64 << 20)
    b.free(q)                                                       # through the other context: released
    q2 = a.malloc(64 << 20)
    a.free(q2)
    # pressure on B releases what A keeps
    r = a.malloc(512 << 20)
    a.free(r)                                                       # 512 MiB kept by A
    held = _hog(b, 128 << 20)
    big = b.malloc(400 << 20)                                       # does not fit in the 128 MiB that are left: A's kept block is given back
    b.free(big)
    for h in held:
        b.free(h)
    b.trim(); a.trim()
    a.close(); b.close()


def test_queue_placement_entries_old_and_new(ctx):
    """kg_ctx_queue_placement is the version-4 entry again (the placement IS the return value: 0 off, 1 + j probed, -1 no clear picture);
    kg_ctx_queue_placement2 returns a status and the code through its out-parameter (0, 1, 2 + j)"""
    import ctypes as C
    L, h = ctx._lib, ctx._h
    old = L.kg_ctx_queue_placement(h)
    new = C.c_int(-7)
    assert L.kg_ctx_queue_placement2(h, C.byref(new)) == 0
    assert old >= -1 and new.value == (0 if old == 0 else (1 if old < 0 else 1 + old))
    assert L.kg_ctx_queue_placement2(h, None) == -2 and L.kg_ctx_queue_placement2(None, C.byref(new)) == -2


def test_work_space_refused_is_oom_and_the_call_succeeds_once_memory_is_back(oracle):
    """The device is filled up to 48 MiB: a 2^20-pair MSM's work space (hundreds of MiB), the host-scalar entry's upload buffer and
    a Groth16-size NTT buffer are refused -> KG_ERR_OOM each, nothing aborts; after the memory is released the same calls succeed
    and give the oracle's results."""
    import kogarashi_amd as K
    O, n, k = oracle, 1 << 20, 20
    ctx = K.Context(0)
    g, m = ctx.empty((n, 8)), ctx.empty((n, 4))
    ctx.gen_bases(K.KG_G1, SEED + 82, 0, n, g.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 83, 0, n, m.ptr)
    hm = m.numpy()
    v = ctx.empty((1 << 22, 4))                                    # a 2^22 transform needs a 128 MiB ping-pong buffer
    ctx.bases_register(K.KG_G1, g.ptr, 0, n)                       # (the resident copy exists before the device fills up)
    small = ctx.msm(K.KG_G1, g.ptr, 0, m.ptr, 1000)                # queues, events, small work spaces exist
    held = _hog(ctx, 48 << 20)
    assert ctx.mem_info()[0] <= (48 << 20) + (16 << 20)
    for call in (lambda: ctx.msm(K.KG_G1, g.ptr, 0, m.ptr, n),
                 lambda: ctx.msm_host_scalars(K.KG_G1, g.ptr, 0, hm, n),
                 lambda: ctx.msm_begin(K.KG_G1, g.ptr, 0, m.ptr, n, 0),
                 lambda: ctx.ntt(v.ptr, 22, False, False),
                 lambda: ctx.bases_precompute(g.ptr)):              # 15 windows x 64 MiB
        with pytest.raises(K.KogarashiError, match="out of device memory"):
            call()
    assert (ctx.msm(K.KG_G1, g.ptr, 0, m.ptr, 1000) == small).all()           # what fits still runs
    for p in held:
        ctx.free(p)
    got = ctx.msm(K.KG_G1, g.ptr, 0, m.ptr, n)
    want_xy, want_inf = O.to_affine("g1", O.msm("g1", g.numpy(), hm, None, threads=14))
    assert not want_inf and (got[:8] == want_xy).all()
    assert (ctx.msm_host_scalars(K.KG_G1, g.ptr, 0, hm, n) == got).all()
    ctx.msm_begin(K.KG_G1, g.ptr, 0, m.ptr, n, 0)
    assert (ctx.msm_end(K.KG_G1, 0) == got).all()
    ctx.ntt(m.ptr, k, False, False)
    assert (m.numpy() == O.Fft(k).dft(hm)).all()
    ctx.close()


def test_ntt_direct_tables_refused_fall_back_to_composed_twiddles(oracle):
    """2^22 transform with the device filled so that the 151 MB direct inter-step table cannot be allocated while the small tables
    can: the step composes its twiddles instead (ntt.hip get_tables) -- the transform succeeds and equals the oracle's."""
    import kogarashi_amd as K
    O, k = oracle, 22
    n = 1 << k
    ctx = K.Context(0)
    v = ctx.empty((n, 4))
    ctx.gen_scalars(K.KG_FR, SEED + 84, 0, n, v.ptr)
    hv = v.numpy()
    ctx.ntt(v.ptr, k, False, False)                                  # forward: ping-pong buffer and forward tables (direct) exist now
    fwd = v.numpy()
    assert (fwd == O.Fft(k).dft(hv)).all()
    held = _hog(ctx, 100 << 20)                                      # < 151 MB: the inverse's direct table is refused, its small tables fit
    ctx.ntt(v.ptr, k, True, False)
    for p in held:
        ctx.free(p)
    assert (v.numpy() == hv).all()                                   # idft(dft(v)) = v, through composed twiddles
    ctx.ntt(v.ptr, k, False, True)                                   # and the same context keeps working: coset_dft with the fwd tables
    assert (v.numpy() == O.Fft(k).coset_dft(hv)).all()
    ctx.close()


def test_contexts_on_one_device_from_concurrent_threads_and_a_context_handed_between_threads(oracle):
    """SURVEY 8b threading: the reference's call sites run on whatever thread calls them (rayon workers).  Four host threads, each with its
    own context on device 0, run blocking / in-flight / host-scalar MSMs and transforms at the same time (ctypes drops the GIL inside the
    calls); then ONE context is used from four threads in turn.  Every result must be the single-threaded one."""
    import threading
    import kogarashi_amd as K
    n = 1 << 16
    base = K.Context(0)
    db, ds, dv = base.empty((n, 8)), base.empty((n, 4)), base.empty((1 << 14, 4))
    base.gen_bases(K.KG_G1, SEED + 70, 0, n, db.ptr)
    base.gen_scalars(K.KG_FR, SEED + 71, 0, n, ds.ptr)
    base.gen_scalars(K.KG_FR, SEED + 72, 0, 1 << 14, dv.ptr)
    base.sync()
    hb, hs, hv = db.numpy(), ds.numpy(), dv.numpy()
    sizes = [1, 300, 1 << 10, 5000, 1 << 14, 40000, n]
    want = {m: base.msm(K.KG_G1, db.ptr, 0, ds.ptr, m).copy() for m in sizes}
    base.ntt(dv.ptr, 14, False, False)
    base.sync()
    want_ntt = dv.numpy().copy()
    oxy, oinf = oracle.to_affine("g1", oracle.msm("g1", hb[:5000], hs[:5000], None, threads=4))
    xy, inf = base.commit(K.KG_G1, db.ptr, 0, ds.ptr, 5000)
    assert not oinf and not inf and (xy == oxy).all()
    errors = []

    def work(ctx, tid, rounds):
        try:
            b, s = ctx.upload(hb), ctx.upload(hs)
            if tid & 1:
                ctx.bases_register(K.KG_G1, b.ptr, 0, n)
            for r in range(rounds):
                m = sizes[(r + tid) % len(sizes)]
                assert (ctx.msm(K.KG_G1, b.ptr, 0, s.ptr, m) == want[m]).all(), ("blocking", tid, m)
                assert (ctx.msm_host_scalars(K.KG_G1, b.ptr, 0, hs[:m], m) == want[m]).all(), ("host", tid, m)
                for t in range(4):
                    ctx.msm_begin(K.KG_G1, b.ptr, 0, s.ptr, sizes[(r + t) % len(sizes)], t)
                for t in range(4):
                    assert (ctx.msm_end(K.KG_G1, t) == want[sizes[(r + t) % len(sizes)]]).all(), ("ticket", tid, t)
                v = ctx.upload(hv)
                ctx.ntt(v.ptr, 14, False, False)
                ctx.sync()
                assert (v.numpy() == want_ntt).all(), ("ntt", tid)
            if tid & 1:
                ctx.bases_unregister(b.ptr)
        except BaseException as e:          # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    ctxs = [K.Context(0) for _ in range(4)]
    ths = [threading.Thread(target=work, args=(ctxs[i], i, 6)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    # one context, four threads in turn (no thread affinity: the creating thread never calls it again)
    shared = ctxs[0]
    for i in range(4):
        t = threading.Thread(target=work, args=(shared, i, 2))
        t.start()
        t.join()
    assert not errors, errors
    for c in ctxs:
        c.close()
    base.close()


_POOL_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import kogarashi_amd as K
K.init()
ctx = K.Context(0)
n = 1 << 19
db, ds = ctx.empty((n, 8)), ctx.empty((n, 4))
ctx.gen_bases(K.KG_G1, 1, 0, n, db.ptr); ctx.gen_scalars(K.KG_FR, 2, 0, n, ds.ptr); ctx.sync()
hs = ds.numpy()
want = ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)           # a blocking call on resident arrays needs no worker thread
out = []
for f in (lambda: ctx.msm_host_scalars(K.KG_G1, db.ptr, 0, hs, n), lambda: ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, n, 0),
          lambda: ctx.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, 100, 1)):
    try:
        f(); out.append("ok")
    except K.KogarashiError as e:
        out.append("status")
print("OUTCOMES", out, "THREADS", ctx.worker_threads(), flush=True)
ctx.sync()
assert (ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n) == want).all()
print("USABLE", flush=True)
"""


def test_a_worker_thread_that_cannot_be_started_is_a_status_not_an_abort():
    """never aborts.  The host side runs tasks on worker threads (the uploader of a host-scalar call, the host finishes of
    tickets): a thread that cannot be started is a std::system_error inside the library -- which must come back through the C ABI as a
    status (kg_guarded), with the context usable for what needs no thread.  The failure is injected through KG_POOL_MAX_THREADS=0 (the
    pool refuses every start exactly where the OS would: RLIMIT_NPROC does not bind root, which is what this pool's boxes run as)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _POOL_SCRIPT], cwd=root, env=dict(os.environ, KG_POOL_MAX_THREADS="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "OUTCOMES ['status', 'status', 'status'] THREADS 0" in r.stdout and "USABLE" in r.stdout, r.stdout[-1500:]
    r = subprocess.run([sys.executable, "-c", _POOL_SCRIPT], cwd=root, env=dict(os.environ, KG_POOL_MAX_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])       # one thread: a call gets through when its tasks find it idle in turn, or is a status
    assert "OUTCOMES [" in r.stdout and "THREADS 1" in r.stdout and "USABLE" in r.stdout, r.stdout[-1500:]


def test_worker_threads_are_started_once_and_reused(ctx, oracle):
    """No thread creation per call (round 5 started a std::async thread per MSM ticket, per slice and per proof part: 30-40 us each on the
    critical path of short calls): after one warm-up pass over tickets of long and short MSMs, a host-scalar call in slices and proofs in
    flight, repeating the same calls ~150 tasks later leaves kg_ctx_worker_threads at the handful of threads that were ever busy at once
    -- and the results are the same points."""
    import kogarashi_amd as K
    from kogarashi_amd import synthetic as syn
    from kogarashi_amd.api import Prover, groth16_setup
    c2 = K.Context(0)
    try:
        n = 1 << 19
        db, ds = c2.empty((n, 8)), c2.empty((n, 4))
        c2.gen_bases(K.KG_G1, 31, 0, n, db.ptr); c2.gen_scalars(K.KG_FR, 32, 0, n, ds.ptr); c2.sync()
        hs = ds.numpy()
        assert c2.worker_threads() == 0
        m = 1 << 8
        cc = syn.ChainCircuit(m)
        P = groth16_setup(cc.a, cc.b, cc.c, m, cc.l, cc.m_l_1, syn.fixed_toxic(), syn.FrOps, ctx=c2)
        pr = Prover(P, m, cc.l, cc.m_l_1, ctx=c2)
        r, s_ = syn.fixed_rs()

        def one_pass():
            res = []
            for i, cnt in enumerate((n, 1000, 1 << 16, 33)):
                c2.msm_begin(K.KG_G1, db.ptr, 0, ds.ptr, cnt, i)
            for i in range(4):
                res.append(c2.msm_end(K.KG_G1, i))
            res.append(c2.msm_host_scalars(K.KG_G1, db.ptr, 0, hs, n))
            res.append(np.concatenate([np.asarray(v).reshape(-1) for v in pr.create_proof(cc.a_eval, cc.b_eval, cc.c_eval, cc.x, cc.w, r, s_)[:3]]))
            return res
        first = one_pass()
        assert c2.worker_threads() >= 1
        for _ in range(8):
            again = one_pass()
            assert all((a == b).all() for a, b in zip(first, again))
        # nine passes of ~17 tasks each: the pool holds as many threads as tasks were ever alive at once (a handful), not one per task
        assert c2.worker_threads() <= 24, c2.worker_threads()
    finally:
        c2.close()


def test_profiling_entries_report_the_phases_of_the_calls_between_enable_and_summary(ctx):
    """kg_profile_enable / kg_profile_summary / kg_profile_last / kg_device_count: the library's own HIP-event phase timers (what bench.py's
    roofline.achieved is computed from).  Between enable and summary one blocking MSM and one transform: their phases appear once each
    per launch with positive durations that add up to less than the calls' wall time; after disable nothing accumulates."""
    import ctypes as C
    import time
    import kogarashi_amd as K
    from kogarashi_amd import lib as L
    assert L.load().kg_device_count() >= 1
    n = 1 << 18
    db, ds, dv = ctx.empty((n, 8)), ctx.empty((n, 4)), ctx.empty((1 << 16, 4))
    ctx.gen_bases(K.KG_G1, SEED + 80, 0, n, db.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 81, 0, n, ds.ptr)
    ctx.gen_scalars(K.KG_FR, SEED + 82, 0, 1 << 16, dv.ptr)
    ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
    ctx.ntt(dv.ptr, 16, False, False)
    ctx.sync()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    want = ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n)
    ctx.ntt(dv.ptr, 16, False, False)
    ctx.sync()
    wall_ms = (time.perf_counter() - t0) * 1e3
    s = ctx.profile_summary()
    for phase in ("prep_scalars", "sort", "accumulate", "reduce", "ntt"):
        assert phase in s and s[phase][0] > 0 and s[phase][1] >= 1, (phase, s)
    assert s["ntt"][1] == 1 and s["accumulate"][1] == s["sort"][1] >= 1          # one sort and one accumulation per window group
    assert max(v[0] for v in s.values()) < wall_ms * 1.5, (s, wall_ms)
    assert ctx.profile_last() == {k: v[0] for k, v in s.items()}
    names, ms = (C.c_char_p * 32)(), (C.c_float * 32)()
    assert L.load().kg_profile_last(ctx._h, names, ms, 32) == len(s)
    ctx.profile_enable(False)
    assert (ctx.msm(K.KG_G1, db.ptr, 0, ds.ptr, n) == want).all()
    assert ctx.profile_summary() == {} or all(v[1] == 0 for v in ctx.profile_summary().values())
