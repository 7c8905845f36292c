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
SOURCES = ["pdmp_capi.hip", "pdmp_kernels.hip", "pdmp_bps.hip", "pdmp_general.hip", "pdmp_partition.hip", "pdmp_trackp.hip", "pdmp_trackl.hip",
           "pdmp_consume.hip", "pdmp_logistic.hip", "pdmp_comm.hip", "pdmp_1d.hip", "pdmp_place.hip"]
# Measured-slower cross-implementations of two event loops (zz_local_exactp_kernel: the moving evaluation with one proposal per lane;
# zz_logistic_rows_kernel: several chains of config C4 per wavefront).  They are NOT in the default library: `build.py --variant parity`
# (-DPDMP_EXTRA_KERNELS) makes lib/libpdmp_mi355.parity.so with them, which the parity suite loads beside the default one
# (tests/conftest.py: gpu_pkg_parity) to hold them to the same oracle.
EXTRA_SOURCES = ["pdmp_exactp.hip", "pdmp_logrows.hip"]
PARITY_DEFINES = ("PDMP_EXTRA_KERNELS",)
HEADERS = [os.path.join(CSRC, "pdmp_engine.hpp"),
           os.path.join(CSRC, "pdmp_spec8g.inc"),  # (included by pdmp_kernels.hip)
           os.path.join(PKG_DIR, "..", "include", "pdmp_mi355.h"),
           os.path.join(PKG_DIR, "..", "include", "pdmp_debug.h"),
           os.path.join(PKG_DIR, "..", "include", "pdmp_detmath.h")]

# -ffp-contract=off: the kernels must round exactly like the CPU oracle (no fused multiply-add).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
               "-shared", "-Wall", "-Wno-unused-function"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpdmp_mi355.so cannot be built (there is no CPU fallback)")


def needs_build(lib_path=LIB_PATH):
    if not os.path.exists(lib_path):
        return True
    srcs = SOURCES + (EXTRA_SOURCES if lib_path.endswith(".parity.so") else [])
    deps = [os.path.join(CSRC, s) for s in srcs] + HEADERS
    return os.path.getmtime(lib_path) < max(os.path.getmtime(p) for p in deps)


def variant_path(variant):
    return LIB_PATH if not variant else os.path.join(LIB_DIR, "libpdmp_mi355.%s.so" % variant)


def build(force=False, verbose=False, variant=None, defines=()):
    """Compile every source to an object file (in parallel, re-used while newer than the source and the headers) and link them.

    variant / defines: an experimental build `lib/libpdmp_mi355.<variant>.so` with extra -D flags, selected at run time with the
    environment variable PDMP_MI355_LIB (A/B timing of kernel versions inside one GPU session, tools/ab.sh)."""
    if defines and not variant:
        raise ValueError("build(defines=...) needs a variant name: an experimental -D build must not replace the default library "
                         "(needs_build() compares time stamps only and would keep it)")
    if variant == "parity" and not defines:
        defines = PARITY_DEFINES
    lib_path = variant_path(variant)
    parity = variant == "parity" and tuple(defines) == PARITY_DEFINES
    if not force and (not defines or parity) and not needs_build(lib_path):
        return lib_path
    sources = SOURCES + (EXTRA_SOURCES if "PDMP_EXTRA_KERNELS" in defines else [])
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj" + ("." + variant if variant else ""))
    os.makedirs(obj_dir, exist_ok=True)
    # several ranks of one node may arrive here together (torchrun): serialise, and let the late ones find the fresh .so
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or (defines and not parity) or needs_build(lib_path):
                hipcc = find_hipcc()
                cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-D" + d for d in defines]
                hdr_time = max(os.path.getmtime(p) for p in HEADERS)

                def compile_one(src):
                    srcp = os.path.join(CSRC, src)
                    obj = os.path.join(obj_dir, src + ".o")
                    stamp = obj + ".flags"
                    flags_txt = " ".join(cflags)
                    fresh = (os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), hdr_time)
                             and os.path.exists(stamp) and open(stamp).read() == flags_txt)
                    if force or not fresh:
                        cmd = [hipcc] + cflags + ["-c", srcp, "-o", obj]
                        if verbose:
                            print(" ".join(cmd))
                        subprocess.check_call(cmd)
                        with open(stamp, "w") as f:
                            f.write(flags_txt)
                    return obj

                with ThreadPoolExecutor(max_workers=len(sources)) as pool:
                    objs = list(pool.map(compile_one, sources))
                tmp = lib_path + ".tmp.%d" % os.getpid()
                # (librccl for pdmp_comm.hip: the post-run gather / reduce links RCCL directly)
                cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                os.replace(tmp, lib_path)  # atomic: a concurrent dlopen never sees a half-written file
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib_path


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
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--variant", default=None, help="experimental build lib/libpdmp_mi355.<variant>.so")
    ap.add_argument("-D", dest="defines", action="append", default=[])
    a = ap.parse_args()
    print(build(force=a.force, verbose=True, variant=a.variant, defines=tuple(a.defines)))
    if not a.variant:
        print(build(force=a.force, verbose=True, variant="parity"))  # the parity suite's library (default + the opt-in cross-implementations)
        print(build_examples(verbose=True))
