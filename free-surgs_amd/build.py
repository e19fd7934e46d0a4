"""Builds libfsgs_hip.so (all HIP kernels + the C ABI of include/fsgs.h) for gfx950, in-tree.

    python free-surgs_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so lands in free-surgs_amd/fsgs_amd/lib/ and
travels to the GPU box with the repo snapshot (git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "fsgs_amd", "lib")
LIB = os.path.join(OUT_DIR, "libfsgs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
    "-ffp-contract=on", "-Wno-unused-result", "-DNDEBUG",
    # no SLP vectoriser: where two FMAs belong in one v_pk_fma_f32 the kernels say so (float2v); the pairs the vectoriser
    # forms on its own need register shuffles -- and, in streaming loops, copies of loads still in flight -- that cost
    # blend_bwd 5 % and exposed a memory round trip per row in photometric_bwd (DESIGN s3, round 3)
    "-fno-slp-vectorize",
]
# FSGS_DIAG=1: the diagnostics hooks of csrc/raster_kernels.h (diag_env) -- experiments only, objects kept apart
DIAG = os.environ.get("FSGS_DIAG") == "1"
if DIAG:
    FLAGS.append("-DFSGS_DIAG_HOOKS")
# A/B experiments on one GPU box: FSGS_CFLAGS="-DFOO=1" FSGS_LIB_TAG=foo builds lib/libfsgs_hip.foo.so from objects of
# its own (the product library is untouched); FSGS_LIB_PATH=<that file> makes fsgs_amd._lib load it.
EXTRA = os.environ.get("FSGS_CFLAGS", "").split()
TAG = os.environ.get("FSGS_LIB_TAG", "")
if EXTRA and not TAG:
    raise SystemExit("FSGS_CFLAGS needs FSGS_LIB_TAG (the product library is only ever built with the default flags)")
FLAGS += EXTRA
if TAG:
    LIB = os.path.join(OUT_DIR, "libfsgs_hip.%s.so" % TAG)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "fsgs.h"))
    hs.append(os.path.abspath(__file__))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OUT_DIR, os.path.basename(src).replace(".hip", (".%s.o" % TAG) if TAG else (".diag.o" if DIAG else ".o")))
    if _stale(obj, [src] + headers()):
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OUT_DIR):
            if f.endswith((".o", ".so")):
                os.remove(os.path.join(OUT_DIR, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, sources()))
    stamp, flavour = LIB + ".flavour", "diag" if DIAG else "product"
    linked_as = open(stamp).read().strip() if os.path.exists(stamp) else "product"
    if _stale(LIB, objs) or linked_as != flavour:  # (the two flavours keep separate objects but share the library name)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        with open(stamp, "w") as f:
            f.write(flavour + "\n")
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
