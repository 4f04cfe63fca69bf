"""Build recipe for the gfx950 HIP library (in-tree, no JIT cache): hipcc -> morphganformer_amd/libmgf_hip.so.

hipcc cross-compiles without a GPU.  Objects are cached per source under csrc/_obj keyed on mtime so a rebuild
after touching one kernel takes seconds.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmgf_hip.so")
SOURCES = ["capi.cpp", "bias_act.hip", "upfirdn2d.hip", "conv_taps.hip", "latent_prep.hip", "attention.hip", "losses.hip", "lpips_stem.hip", "embed.hip", "backward.hip", "wino.hip", "wino3.hip", "pointwise.hip", "narrow_conv.hip", "warp.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=on: fma only inside one source expression (so `acc += a * b` still fuses) and never across statements --
# the kernels that must reproduce torch's two-rounding arithmetic bit for bit rely on this.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=on", "-Wno-unused-result",
         "-x", "hip"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "mgf_common.h"), os.path.join(HERE, "..", "include", "mgf.h")]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if force or _newer(sp, op) or any(_newer(h, op) for h in headers):
            cmd = [HIPCC] + FLAGS + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed = True
            sys.stderr.write(f"[mgf build] {src} FAILED:\n{out.decode(errors='replace')}\n")
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    if procs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
