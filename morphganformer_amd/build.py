"""Build recipe for the gfx950 HIP library (in-tree, no JIT cache): hipcc -> morphganformer_amd/libmgf_hip.so.

hipcc cross-compiles without a GPU.  Objects are cached under csrc/_obj as `<source>.<key>.o` where the key is a hash of everything
the object depends on -- the compiler's version banner, the full flag list, the source text and the text of the shared headers -- so
an object built with other flags (a timing-ablation macro, another compiler) can never be linked into the product: it has another
name.  A rebuild after touching one kernel takes seconds; stale objects of a source are deleted when its current one is built.

    python -m morphganformer_amd.build [--force] [--verbose]
    python -m morphganformer_amd.build --exp NAME --flags "-DMGF_EXP=3" [--source conv_taps.hip]     -> exp_build/libmgf_NAME.so

The second form is the experiment build of tools/: the extra flags apply to ONE source, every object is compiled into
exp_build/_obj (its own cache, nothing is copied from the product's), and the product library is not touched.
"""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmgf_hip.so")
SOURCES = ["capi.cpp", "bias_act.hip", "upfirdn2d.hip", "conv_taps.hip", "latent_prep.hip", "attention.hip", "losses.hip", "lpips_stem.hip",
           "embed.hip", "backward.hip", "wino.hip", "wino3.hip", "pointwise.hip", "narrow_conv.hip", "warp.hip", "lbp.hip"]
HEADERS = [os.path.join(CSRC, "mgf_common.h"), os.path.join(ROOT, "include", "mgf.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=on: fma only inside one source expression (so `acc += a * b` still fuses) and never across statements --
# the kernels that must reproduce torch's two-rounding arithmetic bit for bit rely on this.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=on", "-Wno-unused-result", "-x", "hip"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]

_compiler_id = None


def compiler_id() -> str:
    global _compiler_id
    if _compiler_id is None:
        _compiler_id = subprocess.run([HIPCC, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True).stdout.decode()
    return _compiler_id


def object_key(src_path: str, flags) -> str:
    h = hashlib.sha256()
    h.update(compiler_id().encode())
    h.update("\0".join(flags).encode())
    for p in [src_path] + HEADERS:
        with open(p, "rb") as f:
            h.update(b"\0" + os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _compile_all(obj_dir, flags_for, verbose, force, prune=True):
    """Compile every source into obj_dir (missing objects only, unless force); returns the object paths in SOURCES order.
    prune: delete a source's other objects once its current one is built (the experiment cache keeps them: variants alternate)."""
    os.makedirs(obj_dir, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        flags = flags_for(src)
        op = os.path.join(obj_dir, f"{src}.{object_key(sp, flags)}.o")
        objs.append(op)
        if force or not os.path.exists(op):
            cmd = [HIPCC] + flags + ["-c", sp, "-o", op + ".tmp"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, op, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, op, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed = True
            sys.stderr.write(f"[mgf build] {src} FAILED:\n{out.decode(errors='replace')}\n")
            continue
        os.replace(op + ".tmp", op)                         # an interrupted compile never leaves a half-written object under a valid name
        for stale in (glob.glob(os.path.join(obj_dir, src + ".*.o")) + glob.glob(os.path.join(obj_dir, src + ".o"))) if prune else []:
            if stale != op:
                os.remove(stale)
        if verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    return objs, bool(procs)


def _link(lib, objs, verbose):
    cmd = [HIPCC] + LINK_FLAGS + ["-o", lib + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(lib + ".tmp", lib)
    with open(lib + ".objs", "w") as f:                     # what the library was linked from (read back below: relink when it changes)
        f.write("\n".join(os.path.basename(o) for o in objs) + "\n")


def _linked_from(lib):
    try:
        with open(lib + ".objs") as f:
            return f.read().split()
    except OSError:
        return None


def build(verbose: bool = False, force: bool = False) -> str:
    """The product library.  Returns its path."""
    objs, compiled = _compile_all(OBJ, lambda src: list(FLAGS), verbose, force)
    if compiled or not os.path.exists(LIB) or _linked_from(LIB) != [os.path.basename(o) for o in objs]:
        _link(LIB, objs, verbose)
    return LIB


def build_experiment(name: str, extra_flags, source: str = "conv_taps.hip", verbose: bool = False) -> str:
    """exp_build/libmgf_<name>.so: the library with `extra_flags` applied to `source` only -- or to every source with "all" -- (timing ablations, instruction-mix probes).
    Its objects live in exp_build/_obj, keyed like the product's -- the flagged object has its own name, the others are compiled there
    once and shared between experiments -- and nothing is read from or written to the product's cache."""
    assert source == "all" or source in SOURCES, source
    out_dir = os.path.join(ROOT, "exp_build")
    objs, _ = _compile_all(os.path.join(out_dir, "_obj"), lambda src: FLAGS[:-2] + list(extra_flags) + FLAGS[-2:] if (src == source or source == "all") else list(FLAGS),
                           verbose, False, prune=False)
    lib = os.path.join(out_dir, f"libmgf_{name}.so")
    _link(lib, objs, verbose)
    return lib


if __name__ == "__main__":
    import argparse
    import shlex
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--exp")
    ap.add_argument("--flags", default="")
    ap.add_argument("--source", default="conv_taps.hip")
    a = ap.parse_args()
    if a.exp:
        print(build_experiment(a.exp, shlex.split(a.flags), a.source, verbose=True))
    else:
        print(build(verbose=True, force=a.force))
