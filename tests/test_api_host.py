"""Host-side mirror of the reference interface (kogarashi_amd/api.py): argument validation that needs no device."""
import pytest


def test_fft_argument_checks():
    import kogarashi_amd as K
    with pytest.raises(AssertionError):          # fft.rs:28 assert!(k >= 1)
        K.Fft(0)
    with pytest.raises(ValueError):              # beyond the two-adicity S = 28 (bn254/src/fr.rs:53)
        K.Fft(29)


def test_window_rule_is_total():
    """every n gets a window whose top digit fits the bucket range (W*c >= 255) -- mirrors pick_window in msm.hip"""
    def pick(n):
        lg = n.bit_length() - 1
        if lg >= 19:
            return 16
        if lg >= 14:
            return 15
        return min(max(lg - 3, 2), 10)
    for n in list(range(1, 70)) + [2 ** k + d for k in range(6, 31) for d in (-1, 0, 1)]:
        c = pick(n)
        w = (255 + c - 1) // c
        assert w * c >= 255 and 2 <= c <= 16
        top_bits = 254 - (w - 1) * c
        assert top_bits <= c - 1            # top digit (+1 carry) stays within 2^(c-1) buckets
