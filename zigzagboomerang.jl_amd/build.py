"""Build recipe of libpdmp_mi355.so (hipcc, gfx950 only) -- used by __graft_entry__.build() and by hand.

    python zigzagboomerang.jl_amd/build.py

The shared object is written in-tree (lib/libpdmp_mi355.so) so that it travels to the GPU box with the
repository snapshot; it is git-ignored.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpdmp_mi355.so")
SOURCES = ["pdmp_capi.hip", "pdmp_kernels.hip", "pdmp_bps.hip", "pdmp_general.hip"]
HEADERS = [os.path.join(CSRC, "pdmp_engine.hpp"),
           os.path.join(PKG_DIR, "..", "include", "pdmp_mi355.h"),
           os.path.join(PKG_DIR, "..", "include", "pdmp_detmath.h")]

# -ffp-contract=off: the kernels must round exactly like the CPU oracle (no fused multiply-add).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
               "-shared", "-Wall", "-Wno-unused-function"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpdmp_mi355.so cannot be built (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return os.path.getmtime(LIB_PATH) < max(os.path.getmtime(p) for p in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    # several ranks of one node may arrive here together (torchrun): serialise, and let the late ones find the fresh .so
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():
                tmp = LIB_PATH + ".tmp.%d" % os.getpid()
                cmd = [find_hipcc()] + HIPCC_FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                os.replace(tmp, LIB_PATH)  # atomic: a concurrent dlopen never sees a half-written file
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


EXAMPLES_DIR = os.path.join(os.path.dirname(PKG_DIR), "examples")


def build_examples(verbose=False):
    """Compile the C++ host programs of examples/ (g++, C++17) against include/pdmp_mi355.hpp and the in-tree library."""
    out_dir = os.path.join(EXAMPLES_DIR, "_build")
    os.makedirs(out_dir, exist_ok=True)
    built = []
    for name in sorted(os.listdir(EXAMPLES_DIR)):
        if not name.endswith(".cpp"):
            continue
        src = os.path.join(EXAMPLES_DIR, name)
        exe = os.path.join(out_dir, name[:-4])
        hdrs = [os.path.join(os.path.dirname(PKG_DIR), "include", h) for h in ("pdmp_mi355.hpp", "pdmp_mi355.h")]
        if os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(f) for f in [src, LIB_PATH] + hdrs):
            built.append(exe)
            continue
        tmp = exe + ".tmp.%d" % os.getpid()
        cmd = [shutil.which("g++") or "g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(os.path.dirname(PKG_DIR), "include"),
               src, "-L", LIB_DIR, "-lpdmp_mi355", "-Wl,-rpath,$ORIGIN/../../zigzagboomerang.jl_amd/lib",
               "-Wl,-rpath-link,/opt/rocm/lib", "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, exe)
        built.append(exe)
    return built


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_examples(verbose=True))
