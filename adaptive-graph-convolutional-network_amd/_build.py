"""Builds csrc/ into lib/libagcn_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libagcn_hip.so")
ARCH = "gfx950"
# -Wno-pass-failed: "loop not unrolled" remarks for unroll hints on runtime-bounded loops of the per-graph kernels
# -fno-slp-vectorize: keeps adjacent scalar fp32 adds / fmas as v_add_f32 / v_fma_f32; the packed v_pk_*_f32 forms the
# SLP vectoriser would build issue several times slower on gfx950 (measured in the pairwise-distance loop)
# -fvisibility=hidden: the dynamic symbol table is the functions of include/agcn.h (AGCN_API) and nothing else
FLAGS = ["-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17", "-Wall",
         "-Wno-unused-function", "-Wno-pass-failed", "-fvisibility=hidden", "-fvisibility-inlines-hidden"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _deps():
    root = os.path.dirname(HERE)
    return (sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(root, "include", "agcn.h"),
                                                                os.path.join(CSRC, "exports.map")])


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force=False, verbose=False, jobs=None):
    """Compile every HIP source for gfx950 into one shared library (in-tree): one object per source, compiled side by
    side (the specialised kernels take ~40 s each), then one link."""
    if not force and not is_stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "hipcc")
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"]
    jobs = jobs or min(8, os.cpu_count() or 1)

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        cmd = [hipcc, "--offload-arch=" + ARCH] + cflags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (os.path.basename(src), res.stdout))
        return obj

    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(compile_one, sources()))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
           "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout)
    return LIB_PATH
