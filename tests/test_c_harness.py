"""The C99 host side: tests/harness/sqz_harness.c is the reference's H0 harness
(attic/map_experiment/test.c) re-created over the C ABI.  CPU: it must compile as
plain C99 against include/sqz/sqz.h and report ENODEV without a device.  GPU: it must
pass every case of the reference's main() (test.c:195-236)."""
import errno
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "harness", "sqz_harness.c")
EXE = os.path.join(ROOT, "tests", "harness", "sqz_harness")
SRC_B = os.path.join(ROOT, "tests", "harness", "sqz_boundary.c")
EXE_B = os.path.join(ROOT, "tests", "harness", "sqz_boundary")
SRC_RC = os.path.join(ROOT, "tests", "harness", "sqz_rc_harness.c")
EXE_RC = os.path.join(ROOT, "tests", "harness", "sqz_rc_harness")


@pytest.fixture(scope="module")
def harness():
    from sqz_amd import build
    build.build_native()
    libdir = os.path.join(ROOT, "sqz_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror",
                           "-I" + os.path.join(ROOT, "include"), SRC, "-L" + libdir, "-lsqz_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])
    return EXE


@pytest.fixture(scope="module")
def boundary():
    from sqz_amd import build
    build.build_native()
    libdir = os.path.join(ROOT, "sqz_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror",
                           "-I" + os.path.join(ROOT, "include"), SRC_B, "-L" + libdir, "-lsqz_amd", "-lpthread",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE_B])
    return EXE_B


@pytest.fixture(scope="module")
def rc_harness():
    from sqz_amd import build
    build.build_native()
    libdir = os.path.join(ROOT, "sqz_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror",
                           "-I" + os.path.join(ROOT, "include"), SRC_RC, "-L" + libdir, "-lsqz_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE_RC])
    return EXE_RC


def test_boundary_names_compile_and_are_loud_without_gpu(boundary):
    """squeeze_* constants / squeeze_sizeof of squeeze.h:9-25,94-107 compile as C99 (static asserts on
    their values inside the program); without a device the program reports ENODEV"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([boundary], cwd=ROOT, capture_output=True, text=True)
    assert p.returncode == errno.ENODEV and "no gfx950 device" in p.stdout


@pytest.mark.gpu
def test_boundary_trailer_init_with_and_two_threads(boundary, tmp_path):
    """a stream followed by other data through `.input` (bs.read = the reference's figure, the shim's
    over-read within its documented bound), init_with on a caller block of squeeze_sizeof(0) bytes,
    two threads compressing at once"""
    p = subprocess.run([boundary], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().splitlines()[-1] == "ok", p.stdout + p.stderr


def test_harness_is_c99_and_loud_without_gpu(harness):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([harness], cwd=ROOT, capture_output=True, text=True)
    assert p.returncode == errno.ENODEV and "no gfx950 device" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("win_bits", [10, 15])
def test_harness_passes_reference_cases(harness, win_bits):
    p = subprocess.run([harness, str(win_bits)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.strip().splitlines()
    assert lines[-1] == "ok"
    # sizes printed in the reference's format (test.c:85) equal the golden fingerprints
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        gold = json.load(fh)
    want = {c["file"]: c["out_bytes"] for c in gold["corpus"] if c["win_bits"] == win_bits}
    seen = 0
    for ln in lines:
        for f, size in want.items():
            if ln.endswith(f'of "{f}"'):
                assert int(ln.split("->")[1].split()[0]) == size, ln
                seen += 1
    assert seen == len(want)


@pytest.mark.gpu
def test_harness_writes_the_reference_file_image(harness, tmp_path):
    """file mode (attic test.c:39-42): the kept file equals the one the compiled reference wrote"""
    env = dict(os.environ, SQZ_HARNESS_KEEP=str(tmp_path))
    p = subprocess.run([harness, "10"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    kept = tmp_path / "laozi.txt.w10.file"
    with open(os.path.join(ROOT, "tests", "golden", "laozi.txt.w10.file"), "rb") as fh:
        assert kept.read_bytes() == fh.read()
    assert not os.path.exists(os.path.join(ROOT, "~compressed~.bin"))      # removed like test.c:170


def test_rc_harness_is_c99_under_the_reference_names_and_loud_without_gpu(rc_harness):
    """HEAD's harness (/root/reference/test.c) re-created over <sqz/sqz_rc.h> with the reference's own
    spellings (SQZ_RC_REFERENCE_NAMES): compiles as C99; without a device rc.error = ENODEV"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([rc_harness], cwd=ROOT, capture_output=True, text=True)
    assert p.returncode == errno.ENODEV and "no gfx950 device" in p.stdout


@pytest.mark.gpu
def test_rc_harness_files_are_the_reference_streams_behind_its_header(rc_harness, tmp_path):
    """every file of HEAD's main() that exists goes into "~compressed~.bin" through rc.write and comes back
    through rc.read (test.c:57-182); the kept images are "squeeze4" + the size as a host-order uint64 +
    the stream the compiled reference produces (golden_rc.json fingerprints), the printed lines are the
    reference's (test.c:91-98)"""
    import json
    import struct
    import oracle_lib as O
    env = dict(os.environ, SQZ_HARNESS_KEEP=str(tmp_path))
    corpus = os.path.join(ROOT, "tests", "corpus")
    p = subprocess.run([rc_harness, corpus, "x64.elf", "mandrill.png"], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().splitlines()[-1] == "ok", p.stdout + p.stderr
    with open(os.path.join(ROOT, "tests", "golden", "golden_rc.json")) as fh:
        gold = json.load(fh)
    seen = 0
    for c in gold["corpus"]:
        kept = tmp_path / (c["file"] + ".rc.file")
        if not os.path.exists(os.path.join(corpus, c["file"])):
            continue
        image = kept.read_bytes()
        assert image[:8] == b"squeeze4" and struct.unpack("<Q", image[8:16])[0] == c["in_bytes"]
        assert len(image) == 16 + c["out_bytes"] and O.fnv(image[16:]) == c["out_fnv"], c["file"]
        pc = (16 + c["out_bytes"]) * 100.0 / c["in_bytes"]
        line = '%7d -> %7d %6.2f%% of "%s"' % (c["in_bytes"], 16 + c["out_bytes"], pc, c["file"])
        assert any(ln.endswith(line) and ln.startswith("bps: ") for ln in p.stdout.splitlines()), line
        seen += 1
    assert seen >= 3
    assert not os.path.exists(tmp_path / "~compressed~.bin")              # removed like test.c:170
