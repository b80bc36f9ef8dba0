"""Build libtheia_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the source snapshot).

Every csrc/*.hip file is compiled to its own object (in parallel, only when the source or a header is newer) and the
objects are linked into theia_amd/lib/libtheia_hip.so."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(os.path.dirname(HERE), "build", "obj")
OUT = os.path.join(HERE, "lib", "libtheia_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(os.path.dirname(HERE), "include", "theia_hip.h")]


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in sources() + headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr_t = max(os.path.getmtime(h) for h in headers())
    todo = [s for s in sources()
            if force or not os.path.exists(_obj(s)) or os.path.getmtime(_obj(s)) < max(os.path.getmtime(s), hdr_t)]

    def compile_one(src: str) -> None:
        cmd = [hipcc, *FLAGS, "-c", src, "-o", _obj(src)]
        if verbose:
            print("[theia_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj(s) for s in sources()], "-o", OUT]
    if verbose:
        print("[theia_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
