"""Builds libvcengine.so (and libvccodec objects) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
.so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libvcengine.so"
SOURCES = ["vc_gemm.hip", "vc_gemm_pf.hip", "vc_gemm_wd.hip", "vc_attn.hip", "vc_tokens.hip", "vc_engine.hip", "vc_codec.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def sources() -> list[Path]:
    return [CSRC / s for s in SOURCES if (CSRC / s).exists()]


def build(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=()) -> Path:
    """Compile every HIP translation unit and link libvcengine.so. Returns the library path.

    variant/extra_flags build a diagnostic twin (libvcengine_<variant>.so, e.g. -DVC_KERNEL_TS for the
    in-kernel time stamps of tools/kernel_ts.py); select it at run time with VC_ENGINE_LIB=<path>."""
    srcs = sources()
    deps = srcs + list(CSRC.glob("*.h")) + list((HERE.parent / "include").glob("*.h"))
    LIB = HERE / (f"libvcengine_{variant}.so" if variant else "libvcengine.so")
    FLAGS = [*globals()["FLAGS"], *extra_flags]
    stamp = HERE / (f".build_stamp_{variant}" if variant else ".build_stamp")
    digest = _digest(deps) + " ".join(extra_flags)
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == digest:
        return LIB
    objdir = HERE / (f"build_{variant}" if variant else "build")
    objdir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    procs = []
    objs = []
    headers = [d for d in deps if d.suffix == ".h"]
    for src in srcs:
        obj = objdir / (src.stem + ".o")
        objs.append(obj)
        # one translation unit is recompiled only when it, a header or the flags changed (its own stamp next to the object)
        tu_stamp = objdir / (src.stem + ".stamp")
        tu_digest = _digest([src, *headers]) + " ".join(FLAGS)
        if not force and obj.exists() and tu_stamp.exists() and tu_stamp.read_text() == tu_digest:
            continue
        tu_stamp.unlink(missing_ok=True)
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, tu_stamp, tu_digest, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, tu_stamp, tu_digest, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{out}")
        tu_stamp.write_text(tu_digest)
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    stamp.write_text(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--ts" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, variant="ts", extra_flags=("-DVC_KERNEL_TS",)))
