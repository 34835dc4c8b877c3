"""Builds the HIP library in-tree (semantic-superpoint_amd/csrc/libssp_hip.so) for gfx950.
hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libssp_hip.so")
SOURCES = ["ssp.hip", "pk_math.hip.h", "conv_mfma.hip.h", "conv_wino.hip.h", "conv_wino_pipe.hip.h", "conv_wino_p2.hip.h", "conv_wino4.hip.h", "conv_wino_bf16.hip.h", "dense_loss.hip.h", "bn_kernels.hip.h", "loss_kernels.hip.h", "sem_kernels.hip.h", "pair_kernels.hip.h", "export_kernels.hip.h",
           os.path.join("..", "..", "include", "ssp_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def build(force=False, verbose=False):
    """Compile csrc/ssp.hip -> csrc/libssp_hip.so (gfx950). Returns the library path."""
    if not force and not _stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
           "ssp.hip", "-o", "libssp_hip.so"]
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 or verbose:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed (%d): %s" % (r.returncode, " ".join(cmd)))
    return LIB


def build_locked():
    """build() when stale, with an exclusive file lock so that the ranks of one node do not compile concurrently."""
    if not _stale():
        return LIB
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return build()  # re-checks _stale() under the lock
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
