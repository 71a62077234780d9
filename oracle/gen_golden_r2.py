#!/usr/bin/env python3
"""oracle/gen_golden_r2.py -- TEST INFRASTRUCTURE ONLY (round-2 additions to tests/golden/).

Writes tests/golden/golden_r2.json from the REFERENCE ITSELF (oracle/_ref/libsqz_ref.so,
`make -C oracle ref`); runs only in the build container.

  zipf_fullsize   size + FNV-1a-64 of the reference's output for full-size blocks of the
                  benchmark workload (262,144 B Zipf, window 2^15, payload only), block ids
                  spread over 0..4095 -- pins BASELINE.json configs[2] beyond blocks 0 and 1
  stats           the reference's own counters after a compress (huffman.h:29-33
                  updates/swaps/moves per tree, huffman_entropy :237-249, depth marks) plus
                  the literal / back-reference byte split (squeeze.h:327-328,397-403 -- the
                  reference only prints those under SQUEEZE_MAP_STATS, so they come from
                  the restatement after its compressed bytes were checked equal)
"""
import ctypes as C
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCKS = [2, 3, 7, 64, 127, 128, 255, 256, 511, 512, 513, 777, 1000, 1023, 1024, 1025, 1365,
          1536, 1777, 2047, 2048, 2049, 2222, 2560, 2730, 3000, 3071, 3072, 3333, 3500, 3583,
          3584, 3585, 3800, 3999, 4000, 4064, 4093, 4094, 4095]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("lit_updates lit_swaps lit_moves pos_updates pos_swaps "
                                          "pos_moves literal_bytes backref_bytes").split()] + \
               [("lit_entropy", C.c_double), ("pos_entropy", C.c_double),
                ("lit_depth", C.c_int32), ("pos_depth", C.c_int32)]


def zipf_one(idx):
    import oracle_lib as O
    data = O.zipf_block(idx, 262144)
    out = O.ref_compress(data, 15, False)
    return {"block": idx, "in_bytes": 262144, "in_fnv": O.fnv(data), "win_bits": 15,
            "header": False, "out_bytes": len(out), "out_fnv": O.fnv(out)}


def stats_one(name, data, wb):
    import oracle_lib as O
    O.REF.sqz_ref_compress_stats.restype = C.c_int64
    out = C.create_string_buffer(2 * len(data) + 1088)
    cnt, ent, dep = (C.c_uint64 * 6)(), (C.c_double * 2)(), (C.c_int32 * 2)()
    n = O.REF.sqz_ref_compress_stats(data, C.c_uint64(len(data)), wb, out, C.c_uint64(len(out)),
                                     cnt, ent, dep)
    assert n >= 0
    ref_bytes = out.raw[:n]
    st, nb = Stats(), C.c_uint64()
    e = O.ORACLE.sqzo_encode_stats(data, C.c_uint64(len(data)), C.c_uint32(1 << wb), 0, out,
                                   C.c_uint64(len(out)), C.byref(nb), C.byref(st))
    assert e == 0 and out.raw[:nb.value] == ref_bytes
    mine = [st.lit_updates, st.lit_swaps, st.lit_moves, st.pos_updates, st.pos_swaps, st.pos_moves]
    assert mine == list(cnt) and [st.lit_entropy, st.pos_entropy] == list(ent)
    assert [st.lit_depth, st.pos_depth] == list(dep)
    return {"name": name, "win_bits": wb, "in_bytes": len(data), "out_bytes": n,
            "lit": {"updates": cnt[0], "swaps": cnt[1], "moves": cnt[2], "entropy": ent[0], "depth": dep[0]},
            "pos": {"updates": cnt[3], "swaps": cnt[4], "moves": cnt[5], "entropy": ent[1], "depth": dep[1]},
            "literal_bytes": st.literal_bytes, "backref_bytes": st.backref_bytes}


def main():
    import oracle_lib as O
    assert O.REF is not None, "needs oracle/_ref (make -C oracle ref)"
    gold = {"format": 1, "source": "reference H0 (attic/map_experiment) compiled into oracle/_ref"}
    with ProcessPoolExecutor(max_workers=6) as ex:
        gold["zipf_fullsize"] = list(ex.map(zipf_one, BLOCKS))
    gold["stats"] = []
    for f in ("laozi.txt", "confucius.txt"):
        for wb in (12, 15):
            gold["stats"].append(stats_one(f, O.corpus(f), wb))
    for idx, nb, wb in ((0, 16384, 12), (5, 16384, 12), (7, 40000, 15)):
        gold["stats"].append(stats_one(f"zipf{idx}x{nb}", O.zipf_block(idx, nb), wb))
    with open(os.path.join(ROOT, "tests", "golden", "golden_r2.json"), "w") as fh:
        json.dump(gold, fh, indent=1)
    print("wrote", len(gold["zipf_fullsize"]), "zipf fingerprints,", len(gold["stats"]), "stats")


if __name__ == "__main__":
    main()
