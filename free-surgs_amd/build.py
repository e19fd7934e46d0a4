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
# FSGS_DIAG=1: the diagnostics flavour -- the hooks of csrc/raster_kernels.h (diag_env: per-tile stamps, injected dispatch
# orders, lane-utilisation counters) and the experiment switches (FSGS_EXP_*, honoured ONLY together with the hooks).  A
# library of its own in a directory of its own (lib/diag/libfsgs_hip.diag.so, objects beside it): the product library and
# its objects are never touched, nothing of the flavour is loaded unless FSGS_LIB_PATH names it, and `build.py --clean-diag`
# removes it again (nothing of it is meant to be left in the tree that travels to the GPU box at round end).
DIAG = os.environ.get("FSGS_DIAG") == "1"
# A/B experiments on one GPU box: FSGS_DIAG=1 FSGS_CFLAGS="-DFSGS_EXP_FOO=1" FSGS_LIB_TAG=foo builds
# lib/diag/libfsgs_hip.foo.so from objects of its own; FSGS_LIB_PATH=<that file> makes fsgs_amd._lib load it.
EXTRA = os.environ.get("FSGS_CFLAGS", "").split()
TAG = os.environ.get("FSGS_LIB_TAG", "")
if EXTRA and not TAG:
    raise SystemExit("FSGS_CFLAGS needs FSGS_LIB_TAG (the product library is only ever built with the default flags)")
DIAG_DIR = os.path.join(OUT_DIR, "diag")
if DIAG or TAG:
    # FSGS_LIB_TAG alone (no FSGS_DIAG): the PRODUCT kernels + FSGS_CFLAGS, for a clean A/B of a code variant
    if DIAG:
        FLAGS.append("-DFSGS_DIAG_HOOKS")
    FLAGS += EXTRA
    OBJ_DIR = DIAG_DIR
    LIB = os.path.join(DIAG_DIR, "libfsgs_hip.%s.so" % (TAG or "diag"))
else:
    OBJ_DIR = OUT_DIR


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
    obj = os.path.join(OBJ_DIR, os.path.basename(src).replace(".hip", (".%s.o" % TAG) if TAG else ".o"))
    # the flag list an object was built with sits beside it: the same TAG rebuilt with other FSGS_CFLAGS (or with
    # FSGS_DIAG toggled) must not reuse the old objects -- an A/B of two identical libraries proves nothing
    stamp = obj + ".flags"
    want = " ".join(FLAGS)
    try:
        with open(stamp) as f:
            same_flags = f.read() == want
    except OSError:
        same_flags = False
    if not same_flags or _stale(obj, [src] + headers()):
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        with open(stamp, "w") as f:
            f.write(want)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            if f.endswith((".o", ".so", ".flags")):
                os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, sources()))
    if _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


def clean_diag():
    import shutil

    shutil.rmtree(DIAG_DIR, ignore_errors=True)
    for f in os.listdir(OUT_DIR) if os.path.isdir(OUT_DIR) else []:  # leftovers of the round-3 layout
        if f.endswith((".diag.o", ".flavour")) or (f.startswith("libfsgs_hip.") and f != "libfsgs_hip.so"):
            os.remove(os.path.join(OUT_DIR, f))


if __name__ == "__main__":
    if "--clean-diag" in sys.argv:
        clean_diag()
    else:
        build(force="--force" in sys.argv, verbose=True)
