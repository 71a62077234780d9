"""TEST INFRASTRUCTURE: ctypes access to oracle/liboracle.so (the CPU restatement)
and, when it exists, oracle/_ref/libsqz_ref.so (the reference itself, built in the
build container).  Imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
GOLD = os.path.join(ROOT, "tests", "golden")
CORPUS = os.path.join(ROOT, "tests", "corpus")


def _load_oracle():
    path = os.path.join(ODIR, "liboracle.so")
    src = [os.path.join(ODIR, f) for f in ("sqz_oracle.c", "sqz_oracle.h", "zipf_cdf.inc")]
    if (not os.path.exists(path)) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in src):
        subprocess.check_call(["make", "-C", ODIR, "-s", "all"])
    L = C.CDLL(path)
    L.sqzo_encode.restype = C.c_int
    L.sqzo_encode.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p,
                              C.c_uint64, C.POINTER(C.c_uint64)]
    L.sqzo_decode.restype = C.c_int
    L.sqzo_decode.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64,
                              C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.sqzo_tokens.restype = C.c_int
    L.sqzo_tokens.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                              C.POINTER(C.c_uint64)]
    L.sqzo_match_at.restype = None
    L.sqzo_match_at.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32,
                                C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.sqzo_fnv1a64.restype = C.c_uint64
    L.sqzo_fnv1a64.argtypes = [C.c_char_p, C.c_uint64]
    L.sqzo_zipf_block.restype = None
    L.sqzo_zipf_block.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64]
    L.sqzo_zipf_cdf.restype = C.POINTER(C.c_uint32)
    return L


ORACLE = _load_oracle()
_ref_path = os.path.join(ODIR, "_ref", "libsqz_ref.so")
REF = C.CDLL(_ref_path) if os.path.exists(_ref_path) else None
if REF is not None:
    REF.sqz_ref_compress.restype = C.c_int64
    REF.sqz_ref_compress.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p,
                                     C.c_uint64]
    REF.sqz_ref_decompress.restype = C.c_int
    REF.sqz_ref_decompress.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_int)]


class OracleError(OSError):
    pass


def encode(data: bytes, win_bits: int, header: bool = True, capacity: int = None,
           window: int = None) -> bytes:
    cap = 2 * len(data) + 1088 if capacity is None else capacity
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_uint64(0)
    e = ORACLE.sqzo_encode(data, len(data), window if window is not None else 1 << win_bits,
                           win_bits if header else 0, out, cap, C.byref(n))
    if e != 0:
        raise OracleError(e, "oracle encode")
    return out.raw[:n.value]


def encode_err(data: bytes, win_bits: int, header: bool, capacity: int):
    out = C.create_string_buffer(max(capacity, 1))
    n = C.c_uint64(0)
    e = ORACLE.sqzo_encode(data, len(data), 1 << win_bits, win_bits if header else 0, out,
                           capacity, C.byref(n))
    return e, out.raw[:n.value]


def header_length(comp: bytes) -> int:
    """The 64-bit length field of the H0 header (stream bit b = value bit b)."""
    w = int.from_bytes(comp[:8], "big")
    return int(f"{w:064b}"[::-1], 2)


def decode(comp: bytes, header: bool = True, nbytes: int = None):
    """returns (errno, bytes, win_bits)"""
    n = C.c_uint64(0 if nbytes is None else nbytes)
    if header:
        cap = min(header_length(comp), 1 << 28) if len(comp) >= 8 else 0
    else:
        cap = nbytes
    out = C.create_string_buffer(max(cap, 1))
    wb = C.c_int(0)
    e = ORACLE.sqzo_decode(comp, len(comp), int(header), out, cap, C.byref(n), C.byref(wb))
    return e, (out.raw[:n.value] if e == 0 else b""), wb.value


def tokens(data: bytes, window: int) -> np.ndarray:
    toks = np.zeros(max(len(data), 1), np.uint32)
    cnt = C.c_uint64(0)
    e = ORACLE.sqzo_tokens(data, len(data), window, toks.ctypes.data_as(C.c_void_p), len(toks),
                           C.byref(cnt))
    if e != 0:
        raise OracleError(e, "oracle tokens")
    return toks[:cnt.value].copy()


def match_at(data: bytes, i: int, window: int):
    ln, ds = C.c_uint32(0), C.c_uint32(0)
    ORACLE.sqzo_match_at(data, len(data), i, window, C.byref(ln), C.byref(ds))
    return ln.value, ds.value


def fnv(b: bytes) -> str:
    return f"{ORACLE.sqzo_fnv1a64(b, len(b)):016x}"


def zipf_block(index: int, nbytes: int) -> bytes:
    buf = C.create_string_buffer(nbytes)
    ORACLE.sqzo_zipf_block(index, buf, nbytes)
    return buf.raw


def zipf_cdf() -> np.ndarray:
    return np.ctypeslib.as_array(ORACLE.sqzo_zipf_cdf(), shape=(256,)).copy()


def tree_run(lib, fn, n, syms):
    m = 2 * n - 1
    s = np.asarray(syms, dtype=np.int32)
    arrs = [np.zeros(m, np.uint64), np.zeros(m, np.uint64)] + [np.zeros(m, np.int32) for _ in range(4)]
    info = np.zeros(4, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    r = getattr(lib, fn)(C.c_int32(n), p(s), C.c_uint64(len(s)), *[p(a) for a in arrs], p(info))
    assert r == 0
    return arrs, info


def ref_compress(data: bytes, win_bits: int, header: bool) -> bytes:
    assert REF is not None
    out = C.create_string_buffer(2 * len(data) + 1088)
    n = REF.sqz_ref_compress(data, len(data), win_bits, int(header), out, len(out))
    if n < 0:
        raise OracleError(-n, "reference compress")
    return out.raw[:n]


class Stats(C.Structure):
    """sqzo_stats of oracle/sqz_oracle.h"""
    _fields_ = [(n, C.c_uint64) for n in ("lit_updates lit_swaps lit_moves pos_updates pos_swaps "
                                          "pos_moves literal_bytes backref_bytes").split()] + \
               [("lit_entropy", C.c_double), ("pos_entropy", C.c_double),
                ("lit_depth", C.c_int32), ("pos_depth", C.c_int32)]


def encode_tokens(tokens):
    """stage 2 of the restatement alone on caller-made token words -> (errno, bytes, Stats)"""
    toks = np.ascontiguousarray(tokens, dtype=np.uint32)
    cap = 8 * len(toks) + 64
    out = C.create_string_buffer(cap)
    n, st = C.c_uint64(0), Stats()
    e = ORACLE.sqzo_encode_tokens(toks.ctypes.data_as(C.c_void_p), C.c_uint64(len(toks)), out,
                                  C.c_uint64(cap), C.byref(n), C.byref(st))
    return e, out.raw[:n.value], st


def encode_stats(data: bytes, win_bits: int):
    """payload-only encode + the counters of SURVEY.md section 8f-4 -> (bytes, Stats)"""
    out = C.create_string_buffer(2 * len(data) + 1088)
    n, st = C.c_uint64(0), Stats()
    e = ORACLE.sqzo_encode_stats(data, C.c_uint64(len(data)), C.c_uint32(1 << win_bits), 0, out,
                                 C.c_uint64(len(out)), C.byref(n), C.byref(st))
    if e != 0:
        raise OracleError(e, "oracle encode")
    return out.raw[:n.value], st


def golden_r2():
    with open(os.path.join(GOLD, "golden_r2.json")) as fh:
        return json.load(fh)


def golden():
    with open(os.path.join(GOLD, "golden.json")) as fh:
        return json.load(fh)


def corpus(name: str) -> bytes:
    with open(os.path.join(CORPUS, name), "rb") as fh:
        return fh.read()
