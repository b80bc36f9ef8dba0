"""Build libtheia_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the source snapshot)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("gemm.hip", "gemm_pp.hip", "gemm_wgrad_pp.hip", "norm.hip", "misc.hip", "attention.hip", "attention_mfma.hip")]
HDR = [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "csrc", "gemm_tile.h"), os.path.join(os.path.dirname(HERE), "include", "theia_hip.h")]
OUT = os.path.join(HERE, "lib", "libtheia_hip.so")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in SRC + HDR)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result", *SRC, "-o", OUT]
    if verbose:
        print("[theia_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
