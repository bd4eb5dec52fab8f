"""In-tree build of libunitspeech_hip.so with hipcc for gfx950 (no torch types cross the C ABI)."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libunitspeech_hip.so")
SOURCES = ["conv_igemm.hip", "ops.hip", "attn.hip", "wino.hip", "wino4.hip", "train.hip", "optim.hip", "glue.hip", "frontend.hip", "decoder.hip"]
# every source includes kernels.h; decoder.hip also includes the two .inc files (one stale check for all: a header edit is rare)
HEADERS = ["kernels.h", "pack_f16.h", "wino4_coef.h", "train_host.inc", "train_abi.inc", os.path.join("..", "..", "include", "unitspeech_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# experiment builds only (timing ablations, -DUS_EXP_...): never set in production
FLAGS += os.environ.get("UNITSPEECH_AMD_EXTRA_FLAGS", "").split()


def source_fingerprint(names) -> str:
    """sha256 over the named files of csrc/ (name and contents, in the given order): what ties a committed measurement to the kernels it
    was taken on without needing .git (bench.py: pmc_traffic; tools/pmc_summary.py)."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        h.update(n.encode() + b"\0")
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cc):
        raise RuntimeError("hipcc not found: the HIP decoder cannot be built (there is no CPU fallback)")
    return cc


def have_hipcc() -> bool:
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library; returns its path.  Serialised across processes by a
    lock file: the N ranks of a multi-GPU launch all come through here and must not rebuild the same objects side by side."""
    import fcntl
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    cc = _hipcc()
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def run(job):
        s, o = job
        cmd = [cc] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
