"""Build libmotif_hip.so for gfx950 with hipcc (in-tree; cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "libmotif_hip.so")
SOURCES = ["api.hip", "conv_igemm.hip", "conv_split.hip", "conv_split2.hip", "conv_wino.hip", "conv_pw.hip", "conv_ig16.hip", "conv_direct.hip", "siren.hip", "siren_split.hip", "splat.hip", "misc.hip", "corr.hip", "dcn.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value",
         "-Wno-pass-failed"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Incremental in-tree build.  Safe when several processes call it at once (`bench.py --gpus N`, torchrun: every rank loads the
    library): the whole build runs under an exclusive file lock, staleness is re-checked under the lock, objects and the library
    are written to temporary names and renamed into place, so nobody ever maps a half-written file."""
    import fcntl
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(objdir, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(objdir, force, verbose):
    hdrs = [os.path.join(HERE, "common.h"), os.path.join(HERE, "conv_common.h"), os.path.join(HERE, "conv_split_common.h"), os.path.join(HERE, "conv_wave_epilogue.h"), os.path.join(HERE, "siren_common.h"), os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "motif_hip.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append((["hipcc"] + FLAGS + ["-c", src, "-o", obj + ".tmp.o"], obj + ".tmp.o", obj))

    def run(job):
        cmd, tmp, final = job
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, final)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        tmp = OUT + ".tmp.%d" % os.getpid()
        run((["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs, tmp, OUT))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
