"""Builds the gfx950 shared library in-tree: ``deephumor_amd/lib/libdeephumor_hip.so``.

``hipcc`` cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting ``.so`` is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdeephumor_hip.so")
SOURCES = ("abi.hip", "gemm.hip", "gemm_bf16.hip", "rowops.hip", "attention.hip", "lstm.hip", "lstm_fused.hip", "lstm_wreg.hip", "linear_wreg.hip", "decode_layers.hip", "conv1x1_wreg.hip", "vocab_wreg.hip", "beam.hip", "conv.hip", "stem.hip", "conv3x3.hip", "conv_s1.hip", "conv_s2.hip", "conv_s3.hip", "conv_s4.hip", "preproc.hip", "gemm_f32x.hip", "gemm_f32xp.hip", "conv1x1_f32x.hip", "linear_f32x_wreg.hip", "runtime.hip")
ARCH = "gfx950"


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "deephumor_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compiles every HIP source for gfx950 and links the C-ABI shared library.  Serialised across processes by a file
    lock; the library is linked under a temporary name and moved into place atomically, so a concurrent loader never
    maps a half-written file."""
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIB_PATH
        return _build_locked(verbose)


def _build_locked(verbose):
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wno-comment",
               "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout.decode()}")
    os.replace(tmp, LIB_PATH)
    _stamp_commit()
    return LIB_PATH


def _stamp_commit():
    """Leaves the commit (+ "-dirty") of the tree the library was built from next to it: the GPU box gets a snapshot without
    ``.git``, and bench.py records this stamp in its line instead."""
    root = os.path.dirname(HERE)
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
        if not head:
            return
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True,
                               timeout=10).stdout.strip()
        with open(os.path.join(LIB_DIR, "BUILD_COMMIT"), "w") as f:
            f.write(head + ("-dirty" if dirty else "") + "\n")
    except Exception:
        pass


if __name__ == "__main__":
    print(build(force=True, verbose=True))
