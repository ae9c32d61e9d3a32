"""Build libmdctgan_hip.so (gfx950) and the oracle's C restatement, in-tree.

    python -m mdctgan_amd.build            # incremental
    python -m mdctgan_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  The .so stays in-tree (git-ignored) so it
travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libmdctgan_hip.so")
REPO = os.path.dirname(HERE)
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=off"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(REPO, "include", "mdctgan_hip.h"))
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-I", os.path.join(REPO, "include"), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_oracle_c(force=False, verbose=True):
    """gcc build of oracle/c/*.c -> oracle/_build/liboracle_mdct.so (checker only)."""
    cdir = os.path.join(REPO, "oracle", "c")
    if not os.path.isdir(cdir):
        return None
    out_dir = os.path.join(REPO, "oracle", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "liboracle_mdct.so")
    srcs = [os.path.join(cdir, f) for f in sorted(os.listdir(cdir)) if f.endswith(".c")]
    if srcs and (force or _stale(out, srcs)):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-o", out] + srcs + ["-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_hip(force))
    print(build_oracle_c(force))
