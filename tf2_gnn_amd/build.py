"""Build driver for libtfgnn.so (HIP / gfx950 only).

``python -m tf2_gnn_amd.build`` compiles every ``csrc/*.hip`` translation unit with
``hipcc --offload-arch=gfx950`` (cross-compiles without a GPU) into ``csrc/_obj/*.o`` - in parallel,
skipping objects newer than their sources - and links ``tf2_gnn_amd/libtfgnn.so`` in-tree so that it
travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
OBJ_DIR = CSRC / "_obj"
LIB_PATH = PKG_DIR / "libtfgnn.so"
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _headers():
    return sorted(list(CSRC.glob("*.hpp")) + list((REPO_ROOT / "include").glob("*.h")))


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    OBJ_DIR.mkdir(exist_ok=True)
    hipcc = _hipcc()
    flags = [
        f"--offload-arch={ARCH}",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-Wno-comment",
        f"-I{REPO_ROOT / 'include'}",
        f"-I{CSRC}",
    ]
    hdrs = _headers()
    jobs = []
    objs = []
    for src in _sources():
        obj = OBJ_DIR / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[tfgnn build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(compile_one, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB_PATH)]
        if verbose:
            print("[tfgnn build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
