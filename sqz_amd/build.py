"""Build the native pieces in-tree.

  libsqz_amd.so   sqz_amd/csrc/*.hip  -> sqz_amd/lib/   (hipcc, gfx950 only)
  liboracle.so    oracle/sqz_oracle.c -> oracle/        (gcc; TEST INFRASTRUCTURE)
  libsqz_ref.so   reference sources where they lie under /root/reference
                  -> oracle/_ref/ (only when the reference is mounted)

hipcc cross-compiles without a GPU, so this runs in the build container; the
.so files travel to the GPU box with the snapshot (git-ignored, not
gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sqz_amd", "csrc")
LIBDIR = os.path.join(ROOT, "sqz_amd", "lib")
LIB = os.path.join(LIBDIR, "libsqz_amd.so")
SOURCES = ["abi.hip", "lz77_scan.hip", "lz77_index.hip", "huffman_emit.hip", "decode.hip", "zipf.hip", "blocks.hip", "range_coder.hip"]
HEADERS = ["sqz_device.h", "sqz_tree.h", "sqz_kernels.h", "zipf_cdf.h"]
PUBLIC = [os.path.join(ROOT, "include", "sqz", h) for h in ("sqz.h", "sqz_workload.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libsqz_amd.so")


def build_native(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + PUBLIC
    if not force and not _stale(LIB, deps):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-o", LIB] + srcs
    if verbose:
        print("[sqz_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


# Device builds with LOWERED thresholds: the code paths no real stream reaches (a stream giving up its
# leaf intervals at the token limit, trees too deep for one lane per level, the reference's freeze)
# executed on the GPU and held against the oracle by tests/test_gpu_variants.py.  Test builds: nothing
# loads them except that test (through SQZ_AMD_LIB).
VARIANTS = {
    "wide": ["-DSQZ_BATCH_TOKENS=1500"],
    "shallow": ["-DSQZ_AUX_DEPTH=7", "-DSQZ_MAX_FAST_DEPTH=12"],
    "freeze": ["-DSQZ_FREEZE_DEPTH=9"],
}


def variant_path(name):
    return os.path.join(LIBDIR, f"libsqz_amd_{name}.so")


def build_variants(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + PUBLIC
    for name, flags in VARIANTS.items():
        lib = variant_path(name)
        if not force and not _stale(lib, deps):
            continue
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fvisibility=hidden", *flags, "-I" + os.path.join(ROOT, "include"), "-o", lib] + srcs
        if verbose:
            print("[sqz_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)


def build_oracle(verbose=True):
    """TEST INFRASTRUCTURE: the CPU restatement and, when possible, the reference."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir, "-s", "all", "freeze9"])
    if os.path.isdir("/root/reference/attic/map_experiment"):
        subprocess.check_call(["make", "-C", odir, "-s", "ref"])
    elif verbose:
        print("[sqz_amd.build] reference not mounted: oracle/_ref left as is", flush=True)


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
    if "--variants" in sys.argv:
        build_variants(force="--force" in sys.argv)
    build_oracle()
