"""Builds libkogarashi_amd.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m kogarashi_amd.build [--force] [--jobs N] [--experiments]

--experiments builds libkogarashi_amd_exp.so beside the product library: the same sources with -DKG_EXPERIMENTS, i.e. plus the three
kernels that lost their A/B runs (kg_experiments_built() == 1; loaded through KG_LIB_PATH by their parity tests and by A/B scripts).

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libkogarashi_amd.so")
OUT_EXP = os.path.join(HERE, "libkogarashi_amd_exp.so")
SOURCES = ["capi.cpp", "tuning.cpp", "sharded.cpp", "msm_host.cpp", "vec.hip", "msm_sort.hip", "msm_run.hip", "msm_small.hip", "ntt.hip", "groth16.hip", "setup.hip"]
EXTRA_DEPS = ["../../include/kogarashi_amd.h"]     # plus every header under csrc/ (see _compile)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result",
         "-ffp-contract=off", "-Xarch_host", "-march=x86-64-v3"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool, experiments: bool = False) -> str:
    obj = os.path.join(CSRC, ("exp_" if experiments else "") + src.rsplit(".", 1)[0] + ".o")
    headers = [h for h in os.listdir(CSRC) if h.endswith((".h", ".inc"))]
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in headers + EXTRA_DEPS]
    if force or _stale(obj, deps):
        cmd = ["hipcc", "-x", "hip"] + FLAGS + (["-DKG_EXPERIMENTS"] if experiments else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-6000:]}")
    return obj


def build(force: bool = False, jobs: int = 4, experiments: bool = False) -> str:
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    out = OUT_EXP if experiments else OUT
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, experiments), srcs))
    if force or _stale(out, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-6000:]}")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--experiments", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.jobs, a.experiments))
