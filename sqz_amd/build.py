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


def _lib_command(lib, flags=()):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    return [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
            "-fvisibility=hidden", *flags, "-I" + os.path.join(ROOT, "include"), "-o", lib] + srcs


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in HEADERS] + PUBLIC


def _run_all(cmds, verbose):
    """the library builds are independent hipcc runs of ~80 s each: start them together.

    Every library is linked under a temporary name and moved into place (os.replace) once its own
    hipcc has succeeded, so nothing can load a half-written .so; on the first failure the other
    compilers are terminated and ALL of them are waited for before the error is raised."""
    procs = []
    for cmd in cmds:
        final = cmd[cmd.index("-o") + 1]
        tmp = final + f".tmp{os.getpid()}"
        run = list(cmd)
        run[run.index("-o") + 1] = tmp
        if verbose:
            print("[sqz_amd.build]", " ".join(cmd), flush=True)
        procs.append((cmd, tmp, final, subprocess.Popen(run)))
    failed = None
    pending = list(procs)
    while pending:
        for item in list(pending):
            cmd, tmp, final, p = item
            try:
                rc = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            pending.remove(item)
            if rc == 0 and failed is None:
                os.replace(tmp, final)
            elif rc != 0 and failed is None:
                failed = (rc, cmd)
                for _, _, _, q in pending:          # stop the rest; the loop still waits for each
                    q.terminate()
    for _, tmp, _, _ in procs:
        if os.path.exists(tmp):
            os.remove(tmp)
    if failed is not None:
        raise subprocess.CalledProcessError(failed[0], failed[1])


def build_native(force=False, verbose=True, variants=False):
    """libsqz_amd.so (and, with variants=True, the lowered-threshold test builds) if stale"""
    os.makedirs(LIBDIR, exist_ok=True)
    todo = []
    if force or _stale(LIB, _deps()):
        todo.append(_lib_command(LIB))
    if variants:
        for name, flags in VARIANTS.items():
            if force or _stale(variant_path(name), _deps()):
                todo.append(_lib_command(variant_path(name), flags))
    _run_all(todo, verbose)
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
    build_native(force=force, verbose=verbose, variants=True)


def build_oracle(verbose=True):
    """TEST INFRASTRUCTURE: the CPU restatement and, when possible, the reference."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", odir, "-s", "all", "freeze9"])
    if os.path.isdir("/root/reference/attic/map_experiment"):
        subprocess.check_call(["make", "-C", odir, "-s", "ref-all"])
    elif verbose:
        print("[sqz_amd.build] reference not mounted: oracle/_ref left as is", flush=True)


if __name__ == "__main__":
    build_native(force="--force" in sys.argv, variants="--variants" in sys.argv)
    build_oracle()
