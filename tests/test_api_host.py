"""Host-side mirror of the reference interface (kogarashi_amd/api.py): argument validation that needs no device."""
import pytest


def test_fft_argument_checks():
    import kogarashi_amd as K
    with pytest.raises(AssertionError):          # fft.rs:28 assert!(k >= 1)
        K.Fft(0)
    with pytest.raises(ValueError):              # beyond the two-adicity S = 28 (bn254/src/fr.rs:53)
        K.Fft(29)


def test_window_rule_is_total():
    """every n gets a window whose top digit fits the bucket range (W*c >= 255) -- asks the library's own rule
    (kg_msm_pick_window, the function msm_sort calls), not a copy of it"""
    from kogarashi_amd import build, lib
    build.build()
    pick = lib.msm_pick_window
    for n in list(range(1, 70)) + [2 ** k + d for k in range(6, 31) for d in (-1, 0, 1)]:
        c = pick(n)
        w = (255 + c - 1) // c
        assert w * c >= 255 and 2 <= c <= 20
        top_bits = 254 - (w - 1) * c
        assert top_bits <= c - 1, (n, c)    # top digit (+1 carry of the bias) stays within the 2^(c-1) buckets
        if c >= 17:
            assert (1 << 16) <= n <= (1 << 24)        # needs the two-pass sort (msm_sort)
        if c >= 19:
            assert (1 << 23) <= n <= (1 << 24)        # the wide window (nine-bit fine field): the 2^23..2^24-pair commitments
    assert pick(1 << 20) == 16 and pick(1 << 18) == 15 and pick(1 << 22) == 17 and pick((1 << 23) - 1) == 17 and pick(1 << 23) == 20 and pick(1 << 24) == 20
    assert pick(1 << 9) == 8 and pick(100) == 5 and pick(1 << 10) == 15 and pick(1 << 12) == 15      # short inputs: widths whose top window is not a handful of buckets
    # bench.py's addition count uses the same rule
    import bench
    for lg in (10, 18, 20, 22, 24):
        c = pick(1 << lg)
        assert bench.window_adds(1 << lg) == ((255 + c - 1) // c) << lg


def test_shard_range_matches_dist_and_covers():
    """kg_shard_range (the cut of kg_commit_sharded / kg_sharded_key_*) == dist.shard_range (the multi-process cut)"""
    from kogarashi_amd import build, dist, lib
    build.build()
    for n in (0, 1, 7, 8, 9, 1000, (1 << 24) + 1):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = lib.shard_range(n, r, world)
                assert (lo, hi) == dist.shard_range(n, r, world) and lo == prev and hi - lo in (n // world, n // world + 1)
                prev = hi
            assert prev == n
    with pytest.raises(lib.KogarashiError):
        lib.shard_range(10, 3, 3)


def test_table_window_rule():
    """kg_msm_table_window: where kg_bases_precompute offers window tables, the window fits the merged sort -- the top digit
    stays inside the 2^(c-1) shared buckets, the bucket groups fit the first sort pass (<= 1024) and
    (window << ceil(log2 n)) | index fits the 24-bit index field of a sorted entry"""
    from kogarashi_amd import build, lib
    build.build()
    tw = lib.msm_table_window
    for n in [1, 100, (1 << 16) - 1, (1 << 20) + 1, 1 << 24]:
        assert tw(n) == 0
    for n in [1 << 16, (1 << 16) + 1, (1 << 17) - 1, 1 << 17, 200001, 1 << 18, (1 << 19) + 5, 1 << 20]:
        c = tw(n)
        assert c in (16, 17)
        w = (255 + c - 1) // c
        assert 254 - (w - 1) * c <= c - 1                     # unsigned top digit within the bucket range
        assert (1 << (c - 1)) >> 7 <= 1024                    # bucket groups of the first pass
        s = (n - 1).bit_length()
        assert (w << s) <= (1 << 24), (n, c)
    assert tw(1 << 18) == 17 and tw(1 << 16) == 16


def test_host_slice_plan_covers_the_range_at_every_length():
    """kg_msm_host_slices (no device): the index slices of kg_msm_host_scalars / kg_msm_host are contiguous, cover [0, n), are never empty, follow the
    documented counts, and the first slice of a host-scalar call is the short one (its upload is the one nothing hides)"""
    import os
    import random
    from kogarashi_amd.lib import msm_host_slices
    if any(k in os.environ for k in ("KG_HOST_SLICES", "KG_HOST_FIRST_DIV")):
        import pytest
        pytest.skip("slice knobs set in the environment")
    assert msm_host_slices(0) == [0]
    rnd = random.Random(5)
    sizes = [1, 2, 255, 256, 257, 1000, (1 << 19) - 1, 1 << 19, (1 << 19) + 1] + [1 << k for k in range(17, 27)] + [rnd.randrange(1, 1 << 25) for _ in range(200)]
    for n in sizes:
        for only in (True, False):
            lo = msm_host_slices(n, only)
            assert lo[0] == 0 and lo[-1] == n and all(a < b for a, b in zip(lo, lo[1:])), (n, only, lo)
            k = len(lo) - 1
            lg = n.bit_length() - 1
            if only:
                want = 8 if lg >= 24 else 6 if lg == 23 else 4 if lg == 22 else 3 if lg == 21 else 2 if lg >= 19 else 1
                assert k == want, (n, k)
                if k > 1:
                    rest = [b - a for a, b in zip(lo[1:], lo[2:])]
                    assert lo[1] <= min(rest) and 2 * lo[1] + 1024 >= min(rest), (n, lo)      # half a share, rounded down to 256 pairs
            else:
                assert k == (4 if n >= (1 << 20) else 2 if n >= (1 << 18) else 1), (n, k)
