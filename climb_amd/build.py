"""Build the C-ABI HIP library in-tree:  python -m climb_amd.build

hipcc cross-compiles for gfx950 without a GPU.  The .so stays next to the sources (git-ignored, but it travels to
the GPU box with the snapshot)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libclimb_hip.so")
LIB_F16 = os.path.join(CSRC, "libclimb_hip_f16.so")      # the same sources with IEEE-half operands (common.h: CLIMB_H16_F16)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]


# per-source flags; and sources compiled a second time under another object name (A/B builds of one kernel inside one library)
SRC_FLAGS = {}
TWICE = {"gemm_bf16_nt4.hip": ("gemm_bf16_nt4_slp", ["-DNT4_SLP_BUILD"])}      # (17 = 2 / 5: the r04 window boundary, kept as the A/B partner of r05's)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def units():
    """(source, object stem, extra flags)"""
    out = []
    for s in sources():
        out.append((s, s[:-4], SRC_FLAGS.get(s, [])))
        if s in TWICE:
            out.append((s, TWICE[s][0], TWICE[s][1]))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    _build(LIB_F16, os.path.join(CSRC, "build", "f16"), ["-DCLIMB_H16_F16=1"], force, verbose)
    return _build(LIB, os.path.join(CSRC, "build"), [], force, verbose)


def _build(LIB: str, objdir: str, extra, force: bool, verbose: bool) -> str:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s, stem, fl in units():
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, stem + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj, fl))

    def cc(job):
        src, obj, fl = job
        cmd = [HIPCC] + FLAGS + extra + fl + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, stem + ".o") for _, stem, _ in units()]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
