"""kogarashi_amd -- MI355X (gfx950) backend for Kogarashi's MSM + NTT + Groth16/Nova-commit hot path.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/kogarashi_amd.h), lib.py (ctypes binding),
api.py (host-side mirror of the reference's Rust call sites: msm_curve_addition, Fft, PedersenCommitment).
There is no CPU implementation in this package: without the built HIP library and a GPU every call raises."""

from . import lib  # noqa: F401
from .lib import (KG_FQ, KG_FR, KG_G1, KG_G2, KG_GRUMPKIN, Context, DeviceArray, KogarashiError, init, load)  # noqa: F401
from .api import Fft, NovaProver, PedersenCommitment, Prover, ShardedProver, msm_curve_addition, default_context  # noqa: F401
