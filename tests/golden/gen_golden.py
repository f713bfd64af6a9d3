#!/usr/bin/env python3
"""Generates tests/golden/vectors.json and tests/golden/reference_tables.json.

Run in the authoring container (the reference tree is only available there):

    python tests/golden/gen_golden.py

Sources of truth, all independent of this repository's oracle and kernels:
  * CRC32      : Python zlib.crc32 (same IEEE 802.3 CRC as java.util.zip.CRC32, which
                 the reference selects at ChecksumByteBufferFactory.java:74-76).
  * CRC32C     : a pure-Python run of the slice-by-8 recurrence over the 2048-entry table
                 parsed AS TEXT from the reference's PureJavaCrc32CByteBuffer.java:38-559
                 (the reference proves this equal to java.util.zip.CRC32C in
                 TestChecksumImplsComputeSameValues.java:59-79), plus RFC 3720 B.4 vectors.
  * SHA-256/MD5: Python hashlib (FIPS 180-4 / RFC 1321, what MessageDigest implements,
                 Checksum.java:46-60,83-84).  The tests of the checksum path hold no literal digest
                 for these; the one SHA-256 literal another test of the reference holds for the same
                 MessageDigest is handled by gen_reference_held.py.
  * reference_tables.json holds only SHA-256 digests (and four sample entries) of the two Java
    tables, never the tables: enough to prove that a regenerated table is identical.

Only seeds and expected digests are stored; data is regenerated from the seed (SplitMix64).
"""
import hashlib
import json
import os
import re
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/common"

NONE, CRC32, CRC32C, SHA256, MD5 = 1, 2, 3, 4, 5
NAMES = {1: "NONE", 2: "CRC32", 3: "CRC32C", 4: "SHA256", 5: "MD5"}


def splitmix64_bytes(seed, n):
    k = (n + 7) // 8
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, k + 1, dtype=np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.astype("<u8").tobytes()[:n]


def make_data(spec):
    kind, n = spec["kind"], spec["len"]
    if kind == "splitmix64":
        return splitmix64_bytes(spec["seed"], n)
    if kind == "zeros":
        return bytes(n)
    if kind == "ones":
        return b"\xff" * n
    if kind == "incr":
        return bytes(i & 0xFF for i in range(n))
    if kind == "decr":
        return bytes((n - 1 - i) & 0xFF for i in range(n))
    if kind == "ascii":
        return spec["text"].encode("ascii")
    raise ValueError(kind)


def parse_java_table(path):
    """All 0x........ int literals inside `int[] T = { ... };`."""
    text = open(path, encoding="utf-8").read()
    body = text[text.index("int[] T = {"):]
    body = body[body.index("{") + 1: body.index("};")]
    vals = [int(m, 16) for m in re.findall(r"0x([0-9A-Fa-f]{1,8})", body)]
    assert len(vals) == 2048, len(vals)
    return vals


def table_digest(vals):
    return hashlib.sha256(struct.pack("<2048I", *vals)).hexdigest()


def crc_slice8(data, table):
    """ChecksumByteBuffer.java:82-120 over a parsed table; returns getValue()."""
    crc = 0xFFFFFFFF
    n = len(data)
    pos = 0
    while n - pos > 7:
        c0 = (data[pos] ^ crc) & 0xFF
        c1 = (data[pos + 1] ^ (crc >> 8)) & 0xFF
        c2 = (data[pos + 2] ^ (crc >> 16)) & 0xFF
        c3 = (data[pos + 3] ^ (crc >> 24)) & 0xFF
        crc = table[0x700 + c0] ^ table[0x600 + c1] ^ table[0x500 + c2] ^ table[0x400 + c3]
        crc ^= table[0x300 + data[pos + 4]] ^ table[0x200 + data[pos + 5]] ^ table[0x100 + data[pos + 6]] ^ table[data[pos + 7]]
        pos += 8
    while pos < n:
        crc = (crc >> 8) ^ table[(crc ^ data[pos]) & 0xFF]
        pos += 1
    return (~crc) & 0xFFFFFFFF


_T32C = None


def digest(type_, window):
    if type_ == CRC32:
        return struct.pack(">I", zlib.crc32(window) & 0xFFFFFFFF)
    if type_ == CRC32C:
        return struct.pack(">I", crc_slice8(window, _T32C))
    if type_ == SHA256:
        return hashlib.sha256(window).digest()
    if type_ == MD5:
        return hashlib.md5(window).digest()
    raise ValueError(type_)


def compute(type_, bpc, data):
    if type_ == NONE:
        return []
    return [digest(type_, data[o:o + bpc]) for o in range(0, len(data), bpc)]


def main():
    global _T32C
    t32 = parse_java_table(os.path.join(REF, "PureJavaCrc32ByteBuffer.java"))
    _T32C = parse_java_table(os.path.join(REF, "PureJavaCrc32CByteBuffer.java"))

    # the parsed CRC32 table must reproduce zlib before we trust the CRC32C one
    for n in (0, 1, 7, 8, 9, 55, 1000, 4097):
        d = splitmix64_bytes(99, n)
        assert crc_slice8(d, t32) == zlib.crc32(d) & 0xFFFFFFFF
    assert crc_slice8(b"123456789", _T32C) == 0xE3069283

    tables = {
        "note": "SHA-256 over the 2048 table ints packed little-endian, plus 4 sample entries each; "
                "parsed as text from PureJavaCrc32ByteBuffer.java / PureJavaCrc32CByteBuffer.java",
        "CRC32": {"poly": "0xEDB88320", "sha256": table_digest(t32),
                  "samples": {str(i): "0x%08X" % t32[i] for i in (1, 255, 256, 2047)}},
        "CRC32C": {"poly": "0x82F63B78", "sha256": table_digest(_T32C),
                   "samples": {str(i): "0x%08X" % _T32C[i] for i in (1, 255, 256, 2047)}},
    }
    with open(os.path.join(HERE, "reference_tables.json"), "w") as f:
        json.dump(tables, f, indent=1)
        f.write("\n")

    kat = [
        # name, type, data spec, expected hex (literal published values)
        ("crc32-check", CRC32, {"kind": "ascii", "text": "123456789", "len": 9}, "cbf43926"),
        ("crc32c-check", CRC32C, {"kind": "ascii", "text": "123456789", "len": 9}, "e3069283"),
        ("crc32c-rfc3720-zeros", CRC32C, {"kind": "zeros", "len": 32}, "8a9136aa"),
        ("crc32c-rfc3720-ones", CRC32C, {"kind": "ones", "len": 32}, "62a8ab43"),
        ("crc32c-rfc3720-incr", CRC32C, {"kind": "incr", "len": 32}, "46dd794e"),
        ("crc32c-rfc3720-decr", CRC32C, {"kind": "decr", "len": 32}, "113fdb5c"),
        ("sha256-abc", SHA256, {"kind": "ascii", "text": "abc", "len": 3},
         "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"),
        ("sha256-empty-window", SHA256, {"kind": "ascii", "text": "abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", "len": 56},
         "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"),
        ("md5-a", MD5, {"kind": "ascii", "text": "a", "len": 1}, "0cc175b9c0f1b6a831c399e269772661"),
        ("md5-abc", MD5, {"kind": "ascii", "text": "abc", "len": 3}, "900150983cd24fb0d6963f7d28e17f72"),
        ("md5-alphabet", MD5, {"kind": "ascii", "text": "abcdefghijklmnopqrstuvwxyz", "len": 26},
         "c3fcd3d76192e4007dfb496cca67e13b"),
        ("md5-80digits", MD5, {"kind": "ascii", "text": "1234567890" * 8, "len": 80},
         "57edf4a22be3c955ac49da2e2107b67a"),
    ]
    kat_records = []
    for name, t, spec, hexv in kat:
        d = make_data(spec)
        got = digest(t, d).hex()
        assert got == hexv, (name, got, hexv)
        kat_records.append({"name": name, "type": t, "bpc": max(1, len(d)), "data": spec, "sums_hex": [hexv]})

    cases = []

    def add(name, t, bpc, spec):
        d = make_data(spec)
        cases.append({"name": name, "type": t, "bpc": bpc, "data": spec,
                      "sums_hex": [s.hex() for s in compute(t, bpc, d)]})

    ALL = (CRC32, CRC32C, SHA256, MD5)
    # TestChecksum.java:47-62 shape: 55 bytes, bpc 10 -> 6 checksums
    for t in ALL:
        add("TestChecksum-55-10-%s" % NAMES[t], t, 10, {"kind": "splitmix64", "seed": 55, "len": 55})
    # TestChunkInputStream.java:50-76 shape: 100 bytes, bpc 20, CRC32
    add("TestChunkInputStream-100-20", CRC32, 20, {"kind": "splitmix64", "seed": 100, "len": 100})
    # TestKeyValueContainerCheck.java:211-217 shape: 3072 bytes, bpc 2048, SHA-256
    add("TestKeyValueContainerCheck-3072-2048", SHA256, 2048, {"kind": "splitmix64", "seed": 3072, "len": 3072})
    # ContainerProtocolCalls.java:336-337 writeSmallFile shape: CRC32 at bpc 256
    add("writeSmallFile-1000-256", CRC32, 256, {"kind": "splitmix64", "seed": 256, "len": 1000})
    # the unit-scale configuration: one 4 MiB chunk at 1 MiB windows
    for t in ALL:
        add("chunk4M-1M-%s" % NAMES[t], t, 1 << 20, {"kind": "splitmix64", "seed": 1, "len": 4 << 20})
    add("chunk4M-1M-zeros-CRC32", CRC32, 1 << 20, {"kind": "zeros", "len": 4 << 20})
    # last window partial, not a multiple of 16
    for t in ALL:
        add("chunk-ragged-%s" % NAMES[t], t, 1 << 20, {"kind": "splitmix64", "seed": 7, "len": (2 << 20) + 12345})
    # 16 KiB windows (the configured minimum, OzoneClientConfig.java:143-151)
    for t in (CRC32, CRC32C):
        add("win16K-%s" % NAMES[t], t, 16384, {"kind": "splitmix64", "seed": 16, "len": 20 * 16384 + 100})
    # window sizes of BenchMarkCRCStreaming.java:69
    for bpc in (512, 1024, 2048, 4096, 32768, 1048576):
        for t in (CRC32, CRC32C):
            add("genesis-%d-%s" % (bpc, NAMES[t]), t, bpc, {"kind": "splitmix64", "seed": bpc, "len": (2 << 20) + 777})
    # tiny lengths around every internal boundary (first 4 bytes, 8-byte slice, 16-byte lane unit, 64-byte block)
    for n in (0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 129):
        for bpc in (1, 8, 16, 64, 1000):
            for t in ALL:
                add("tiny-%d-%d-%s" % (n, bpc, NAMES[t]), t, bpc, {"kind": "splitmix64", "seed": 1000 + n, "len": n})
    # odd window sizes
    for bpc in (3, 17, 100, 1023, 1025, 4095, 16400, 65552):
        for t in ALL:
            add("odd-%d-%s" % (bpc, NAMES[t]), t, bpc, {"kind": "splitmix64", "seed": bpc, "len": 3 * bpc + bpc // 2})
    # NONE -> no checksums (Checksum.java:157-160)
    add("none", NONE, 1024, {"kind": "splitmix64", "seed": 5, "len": 5000})

    # verify cases: stored list computed over `stored_data`; the data handed to verify is the slice
    # [data_off, data_off+data_len) with bytes at `flips` (offsets inside the slice) inverted.
    verify = []

    def addv(name, t, bpc, total, data_off, data_len, flips, start_index, rc, first, stored_count=-1):
        verify.append({"name": name, "type": t, "bpc": bpc,
                       "stored_data": {"kind": "splitmix64", "seed": 4242, "len": total},
                       "stored_count": stored_count,
                       "data_off": data_off, "data_len": data_len, "flips": flips,
                       "start_index": start_index, "rc": rc, "first_mismatch": first})

    OK, MISMATCH, ORIG_EMPTY, COMPUTED_EMPTY, OUT_OF_RANGE = 0, 1, 2, 3, 4
    for t in ALL:
        addv("ok-%s" % NAMES[t], t, 20, 100, 0, 100, [], 0, OK, -1)
        addv("flip-w2-%s" % NAMES[t], t, 20, 100, 0, 100, [50], 0, MISMATCH, 2)
        addv("flip-two-%s" % NAMES[t], t, 20, 100, 0, 100, [70, 21], 0, MISMATCH, 1)
        # ChunkInputStream.java:497-500: read [20,60) of the chunk -> startIndex 1
        addv("sub-ok-%s" % NAMES[t], t, 20, 100, 20, 40, [], 1, OK, -1)
        addv("sub-flip-%s" % NAMES[t], t, 20, 100, 20, 40, [25], 1, MISMATCH, 1)
        addv("sub-wrong-start-%s" % NAMES[t], t, 20, 100, 20, 40, [], 0, MISMATCH, 0)
        addv("last-partial-%s" % NAMES[t], t, 2048, 3072, 0, 3072, [3071], 0, MISMATCH, 1)
        addv("overrun-%s" % NAMES[t], t, 20, 100, 0, 100, [], 2, MISMATCH, 0)
        addv("computed-empty-%s" % NAMES[t], t, 20, 100, 0, 0, [], 0, COMPUTED_EMPTY, -1)
    # stored list shorter than the computed one (only its first 3 entries kept): ChecksumData.java:142-147
    addv("overrun-range", CRC32, 20, 100, 0, 100, [], 0, OUT_OF_RANGE, -1, stored_count=3)
    addv("overrun-range-sub", SHA256, 20, 100, 40, 60, [], 2, OUT_OF_RANGE, -1, stored_count=4)
    addv("orig-empty", MD5, 20, 100, 0, 100, [], 0, ORIG_EMPTY, -1, stored_count=0)
    addv("big-1M-ok", CRC32, 1 << 20, 4 << 20, 0, 4 << 20, [], 0, OK, -1)
    addv("big-1M-flip", CRC32, 1 << 20, 4 << 20, 0, 4 << 20, [(3 << 20) + 17], 0, MISMATCH, 3)
    addv("big-16K-flip", CRC32C, 16384, 1 << 20, 0, 1 << 20, [16384 * 37 + 5, 16384 * 50], 0, MISMATCH, 37)

    # self-check the verify expectations with the same independent primitives
    for v in verify:
        sd = make_data(v["stored_data"])
        stored = compute(v["type"], v["bpc"], sd)
        if v["stored_count"] >= 0:
            stored = stored[: v["stored_count"]]
        d = bytearray(sd[v["data_off"]: v["data_off"] + v["data_len"]])
        for o in v["flips"]:
            d[o] ^= 0xFF
        comp = compute(v["type"], v["bpc"], bytes(d))
        rc, first = OK, -1
        if not stored:
            rc = ORIG_EMPTY
        elif not comp:
            rc = COMPUTED_EMPTY
        else:
            for i, c in enumerate(comp):
                if v["start_index"] + i >= len(stored):
                    rc = OUT_OF_RANGE
                    break
                if stored[v["start_index"] + i] != c:
                    rc, first = MISMATCH, i
                    break
        assert (rc, first) == (v["rc"], v["first_mismatch"]), (v["name"], rc, first)

    out = {"format": 1,
           "data_kinds": "splitmix64(seed,len): state=seed; each step state+=0x9E3779B97F4A7C15, "
                         "z=mix(state), 8 little-endian bytes per step; zeros; ones(0xFF); incr(i&255); "
                         "decr((len-1-i)&255); ascii(text)",
           "kat": kat_records, "compute": cases, "verify": verify}
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote %d KATs, %d compute cases, %d verify cases" % (len(kat_records), len(cases), len(verify)))


if __name__ == "__main__":
    main()
