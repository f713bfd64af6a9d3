"""Pins the CPU oracle (oracle/) before anything is compared with it.

The reference's own tests hold no literal digests for this path (they assert equality between
implementations, TestChecksumImplsComputeSameValues.java:43-101, and structure, TestChecksum.java:47-80),
so the oracle is pinned by: the digests of the reference's two Java CRC tables, published known-answer
vectors, vectors produced by zlib / hashlib / the parsed Java CRC32C table (tests/golden/gen_golden.py), and -- for
SHA-256 -- the one literal digest of known content that another test of the reference holds for the same JDK
primitive (tests/golden/gen_reference_held.py).
"""
import hashlib
import struct
import zlib

import numpy as np
import pytest

import oracle
from conftest import make_data


def test_tables_equal_reference_java_tables(reference_tables):
    for name, poly in (("CRC32", 0xEDB88320), ("CRC32C", 0x82F63B78)):
        t = oracle.crc_table(poly)
        assert len(t) == 2048  # ChecksumByteBuffer.java:57
        rec = reference_tables[name]
        assert hashlib.sha256(struct.pack("<2048I", *t)).hexdigest() == rec["sha256"]
        for idx, val in rec["samples"].items():
            assert t[int(idx)] == int(val, 16)


def test_sha256_equals_the_literal_a_reference_test_holds(reference_held):
    # TestStringToSignProducer.java:86 asserts MessageDigest SHA-256 of a canonical request; the same message through
    # the oracle, as one window and as the ragged last window of a chunk
    for rec in reference_held:
        t = getattr(oracle, rec["type"])
        msg = np.frombuffer(rec["message_utf8"].encode("utf-8"), dtype=np.uint8)
        assert hashlib.sha256(msg.tobytes()).hexdigest() == rec["digest_hex"]
        assert oracle.compute(t, len(msg), msg).hex() == rec["digest_hex"], rec["held_by"]
        assert oracle.compute(t, 1 << 20, msg).hex() == rec["digest_hex"]
        two = np.concatenate([np.zeros(256, dtype=np.uint8), msg])   # window 0: 256 zero bytes, window 1: the message
        assert oracle.compute(t, 256, two)[32:].hex() == rec["digest_hex"]


def test_kat(vectors):
    for rec in vectors["kat"]:
        d = make_data(rec["data"])
        assert oracle.compute(rec["type"], rec["bpc"], d).hex() == "".join(rec["sums_hex"]), rec["name"]


def test_compute_golden(vectors):
    for rec in vectors["compute"]:
        d = make_data(rec["data"])
        got = oracle.compute(rec["type"], rec["bpc"], d)
        assert got.hex() == "".join(rec["sums_hex"]), rec["name"]


def test_verify_golden(vectors):
    for v in vectors["verify"]:
        sd = make_data(v["stored_data"])
        stored = oracle.compute(v["type"], v["bpc"], sd)
        ds = oracle.DIGEST_SIZE[v["type"]]
        if v["stored_count"] >= 0:
            stored = stored[: v["stored_count"] * ds]
        d = sd[v["data_off"]: v["data_off"] + v["data_len"]].copy()
        for o in v["flips"]:
            d[o] ^= 0xFF
        rc, first = oracle.verify(v["type"], v["bpc"], d, stored, v["start_index"])
        assert (rc, first) == (v["rc"], v["first_mismatch"]), v["name"]


def test_incremental_update_semantics():
    # TestChecksumByteBuffer.java:78-91: byte-at-a-time and cumulative updates agree with one-shot
    L = oracle.lib()
    import ctypes
    tab = (ctypes.c_uint32 * 2048)()
    L.oracle_crc_make_table(0xEDB88320, tab)
    rng = np.random.default_rng(7)
    for _ in range(50):
        d = rng.integers(0, 256, 1024, dtype=np.uint8).tobytes()
        crc = 0xFFFFFFFF
        for i in range(len(d)):
            crc = L.oracle_crc_update(crc, d[i:i + 1], 1, tab)
            if i % 97 == 0:
                assert (~crc) & 0xFFFFFFFF == zlib.crc32(d[: i + 1])
        assert (~crc) & 0xFFFFFFFF == zlib.crc32(d)
        cut = int(rng.integers(0, 1024))
        c2 = L.oracle_crc_update(L.oracle_crc_update(0xFFFFFFFF, d[:cut], cut, tab), d[cut:], 1024 - cut, tab)
        assert c2 == crc


def test_structure_like_reference_tests():
    # TestChecksum.java:47-62: 55 bytes at bpc 10 -> 6 checksums; :67-80 a changed byte changes them
    d = oracle.splitmix64(3, 55)
    for t in (oracle.CRC32, oracle.CRC32C, oracle.SHA256, oracle.MD5):
        s = oracle.compute(t, 10, d)
        assert len(s) == 6 * oracle.DIGEST_SIZE[t]
        d2 = d.copy()
        d2[50] ^= 1
        s2 = oracle.compute(t, 10, d2)
        assert s2 != s and s2[: 5 * oracle.DIGEST_SIZE[t]] == s[: 5 * oracle.DIGEST_SIZE[t]]
    assert oracle.compute(oracle.NONE, 10, d) == b""
    assert oracle.compute(oracle.CRC32, 10, np.zeros(0, np.uint8)) == b""
    with pytest.raises(ValueError):
        oracle.compute(9, 10, d)
    with pytest.raises(ValueError):
        oracle.compute(oracle.CRC32, 0, d)


def test_random_against_zlib_hashlib():
    rng = np.random.default_rng(11)
    for _ in range(40):
        n = int(rng.integers(0, 70000))
        bpc = int(rng.integers(1, 20000))
        d = rng.integers(0, 256, n, dtype=np.uint8)
        b = d.tobytes()
        exp = b"".join(struct.pack(">I", zlib.crc32(b[o:o + bpc])) for o in range(0, n, bpc))
        assert oracle.compute(oracle.CRC32, bpc, d) == exp
        exp = b"".join(hashlib.sha256(b[o:o + bpc]).digest() for o in range(0, n, bpc))
        assert oracle.compute(oracle.SHA256, bpc, d) == exp
        exp = b"".join(hashlib.md5(b[o:o + bpc]).digest() for o in range(0, n, bpc))
        assert oracle.compute(oracle.MD5, bpc, d) == exp


def test_clmul_baseline_engine_equals_the_oracle(vectors):
    # the PCLMULQDQ engine timed as the CPU baseline must give the oracle's digests (it falls back to the table
    # method on a CPU without the instruction, so this holds everywhere)
    rng = np.random.default_rng(12)
    for rec in vectors["compute"]:
        if rec["type"] in (oracle.CRC32, oracle.CRC32C):
            assert oracle.compute_clmul(rec["type"], rec["bpc"], make_data(rec["data"])).hex() == "".join(rec["sums_hex"]), rec["name"]
    for i in range(120):
        n = int(rng.choice([0, 1, 15, 16, 17, 63, 64, 65, 79, 80, 127, 128, 129, 191, 192, 1000, 4097, int(rng.integers(0, 200000))]))
        off = int(rng.integers(0, 17))
        d = oracle.splitmix64(900 + i, n + off)[off:]
        for t in (oracle.CRC32, oracle.CRC32C):
            bpc = int(rng.choice([64, 100, 4096, 1 << 20]))
            assert oracle.compute_clmul(t, bpc, d) == oracle.compute(t, bpc, d), (t, bpc, n, off)


def test_gf2_arithmetic_of_the_kernels_against_the_bit_serial_definition():
    """csrc/gf2_crc.h on the CPU (tests/cpp/test_gf2.cpp): check values, operator algebra, the operator bases behind
    the LDS tables, and the merge identities the kernels rely on (pieces summed independently, initial register
    folded into the first bytes, a window entered in the middle and wrapped round)."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call(["make", "-s", "-C", os.path.join(here, "cpp"), "gf2_test"])
    r = subprocess.run([os.path.join(here, "cpp", "gf2_test")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all passed" in r.stdout, r.stdout + r.stderr
