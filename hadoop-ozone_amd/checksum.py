"""Host-side mirror of the reference's checksum interface, over the C-ABI HIP library.

Mirrors, name for name and with the same argument meaning and error behaviour:
  Checksum                 hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/common/Checksum.java:43-299
  ChecksumData             .../ozone/common/ChecksumData.java:34-192
  OzoneChecksumException   .../ozone/common/OzoneChecksumException.java:29-66
  ChecksumType             hadoop-hdds/interface-client/src/main/proto/DatanodeClientProtocol.proto:384-390

Every digest is computed by the GPU library (native.py); nothing here hashes on the CPU.
"""
import ctypes
import enum

from . import native
from .chunk_buffer import ByteBuffer, ChunkBuffer


class ChecksumType(enum.IntEnum):
    NONE = 1
    CRC32 = 2
    CRC32C = 3
    SHA256 = 4
    MD5 = 5


OZONE_CLIENT_BYTES_PER_CHECKSUM_MIN_SIZE = 16 * 1024  # OzoneConfigKeys.java:340


def _parse_size(v):
    """ConfigType.SIZE values: a number with an optional B / KB / MB / GB unit (binary multiples), e.g. "1MB"."""
    if isinstance(v, int):
        return v
    txt = str(v).strip().upper()
    for suffix, mul in (("KB", 1 << 10), ("MB", 1 << 20), ("GB", 1 << 30), ("B", 1)):
        if txt.endswith(suffix):
            return int(float(txt[:-len(suffix)].strip()) * mul)
    return int(txt)


class OzoneClientConfig:
    """The three checksum keys of the client configuration group "ozone.client" (OzoneClientConfig.java:103-125)
    with the validation of :143-151: checksum.type defaults to CRC32, bytes.per.checksum to 1MB and is raised
    to 16 KiB with a warning when set lower, verify.checksum defaults to true."""

    PREFIX = "ozone.client"

    def __init__(self, conf=None):
        import logging
        conf = conf or {}
        self.checksumType = str(conf.get(self.PREFIX + ".checksum.type", ChecksumType.CRC32.name))
        self.bytesPerChecksum = _parse_size(conf.get(self.PREFIX + ".bytes.per.checksum", "1MB"))
        v = conf.get(self.PREFIX + ".verify.checksum", True)
        self.checksumVerify = v if isinstance(v, bool) else str(v).strip().lower() == "true"
        if self.bytesPerChecksum < OZONE_CLIENT_BYTES_PER_CHECKSUM_MIN_SIZE:  # :143-151
            logging.getLogger("OzoneClientConfig").warning(
                "The checksum size (%d) is not allowed to be less than the minimum size (%d), resetting to the "
                "minimum size.", self.bytesPerChecksum, OZONE_CLIENT_BYTES_PER_CHECKSUM_MIN_SIZE)
            self.bytesPerChecksum = OZONE_CLIENT_BYTES_PER_CHECKSUM_MIN_SIZE

    def getChecksumType(self):
        """ChecksumType.valueOf(checksumType) (OzoneClientConfig.java:199-201): an unknown name is an error."""
        try:
            return ChecksumType[self.checksumType]
        except KeyError:
            raise ValueError("No enum constant ChecksumType." + self.checksumType)

    def setChecksumType(self, t):
        self.checksumType = t.name

    def getBytesPerChecksum(self):
        return self.bytesPerChecksum

    def setBytesPerChecksum(self, n):
        """The setter does not clamp (OzoneClientConfig.java:207-209): only validate() does, once, after injection."""
        self.bytesPerChecksum = n

    def isChecksumVerify(self):
        return self.checksumVerify

    def setChecksumVerify(self, v):
        self.checksumVerify = v

    def newChecksum(self):
        """What BlockOutputStream builds once per stream (BlockOutputStream.java:174-175)."""
        return Checksum(self.getChecksumType(), self.bytesPerChecksum)


def reference_list_windows(lengths):
    """Which elements of a buffer list ChunkBufferImplWithByteBufferList.iterate visits, in order
    (ChunkBufferImplWithByteBufferList.java:59-78, 192-201), given the elements' lengths.

    iterate().next() calls findCurrent() every time, and findCurrent() adds the limits of the consumed buffers to
    limitPrecedingCurrent without resetting it first.  position() = limitPrecedingCurrent + current.position()
    therefore runs ahead of the bytes really consumed and hasRemaining() turns false early: of four equal
    elements three are visited, of 256 only 24."""
    n = len(lengths)
    if n == 0:
        return []
    limit = sum(lengths)
    pos = [0] * n
    state = {"preceding": 0, "cur": 0}

    def find_current():  # :59-78
        found = False
        for i in range(n):
            if found:
                continue  # (checks position == 0, which holds here)
            if pos[i] < lengths[i]:
                found = True
                state["cur"] = i
            else:
                state["preceding"] += lengths[i]
        if not found:
            state["cur"] = n - 1
            state["preceding"] -= lengths[n - 1]

    find_current()  # constructor, :56
    visited = []
    while limit - (state["preceding"] + pos[state["cur"]]) > 0:  # hasNext(): hasRemaining()
        find_current()
        visited.append(state["cur"])
        pos[state["cur"]] = lengths[state["cur"]]
    return visited


class OzoneChecksumException(IOError):
    """OzoneChecksumException.java:29-66 (an IOException there, so callers fail over to the next replica)."""

    def __init__(self, arg):
        if isinstance(arg, ChecksumType) or (isinstance(arg, tuple) and arg and arg[0] == "type"):
            t = arg if isinstance(arg, ChecksumType) else arg[1]
            msg = "Unrecognized ChecksumType: %s" % (t.name if isinstance(t, ChecksumType) else t)  # :43-47
            self.index = None
        elif isinstance(arg, int):
            msg = "Checksum mismatch at index %d" % arg  # :35-37
            self.index = arg
        else:
            msg = str(arg)  # :63-65
            self.index = None
        super().__init__(msg)


class ChecksumData:
    """ChecksumData.java:34-192: {type, bytesPerChecksum, list of digests}."""

    def __init__(self, checksumType, bytesPerChecksum, checksums=()):
        self.type = checksumType
        self.bytesPerChecksum = bytesPerChecksum
        self.checksums = [bytes(c) for c in checksums]

    def getChecksumType(self):
        return self.type

    def getBytesPerChecksum(self):
        return self.bytesPerChecksum

    def getChecksums(self):
        return self.checksums

    def verifyChecksumDataMatches(self, that, startIndex):
        """ChecksumData.java:117-149: this[startIndex + i] must equal that[i] for every i."""
        if len(self.checksums) == 0:
            raise OzoneChecksumException("Original checksumData has no checksums")
        if len(that.checksums) == 0:
            raise OzoneChecksumException("Computed checksumData has no checksums")
        n = len(that.checksums)
        for index in range(n):
            if startIndex + index >= len(self.checksums) or startIndex + index < 0:
                raise OzoneChecksumException(
                    "Computed checksum has %d number of checksums. Original checksum has %d number of checksums "
                    "starting from index %d" % (n, len(self.checksums) - startIndex, startIndex))
            if self.checksums[startIndex + index] != that.checksums[index]:
                raise OzoneChecksumException(index)
        return True

    def __eq__(self, other):  # :156-182
        return (isinstance(other, ChecksumData) and self.type == other.type
                and self.bytesPerChecksum == other.bytesPerChecksum and self.checksums == other.checksums)

    def __hash__(self):
        return hash((int(self.type), self.bytesPerChecksum, tuple(self.checksums)))

    def __repr__(self):
        return "ChecksumData(%s, %d, %d checksums)" % (getattr(self.type, "name", self.type),
                                                      self.bytesPerChecksum, len(self.checksums))

    # ---- ContainerProtos.ChecksumData wire form (DatanodeClientProtocol.proto:378-382) ----
    def getProtoBufMessage(self):
        """Serialised ContainerProtos.ChecksumData: type = 1 (enum), bytesPerChecksum = 2 (uint32),
        checksums = 3 (repeated bytes) -- what ChecksumData.getProtoBufMessage().toByteArray() yields."""
        out = bytearray()
        out += b"\x08" + _varint(int(self.type))
        out += b"\x10" + _varint(self.bytesPerChecksum)
        for c in self.checksums:
            out += b"\x1a" + _varint(len(c)) + c
        return bytes(out)

    @staticmethod
    def getFromProtoBuf(buf):
        """Inverse of getProtoBufMessage (ChecksumData.java:94-102)."""
        if buf is None:
            raise ValueError("checksumDataProto is null")  # Preconditions.checkNotNull, :96
        t, bpc, sums, pos, n = None, None, [], 0, len(buf)
        while pos < n:
            key, pos = _read_varint(buf, pos)
            field, wt = key >> 3, key & 7
            if wt == 0:
                v, pos = _read_varint(buf, pos)
                if field == 1:
                    t = v
                elif field == 2:
                    bpc = v
            elif wt == 2:
                ln, pos = _read_varint(buf, pos)
                if field == 3:
                    sums.append(bytes(buf[pos:pos + ln]))
                pos += ln
            else:
                raise ValueError("unsupported wire type %d" % wt)
        if t is None or bpc is None:
            raise ValueError("required field missing")  # both are `required` in the proto
        return ChecksumData(ChecksumType(t), bpc, sums)


class ChunkInfo:
    """The record that carries a chunk's ChecksumData from the client to the datanode and onto its disk:
    hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/container/common/helpers/ChunkInfo.java:34-196, wire form
    ContainerProtos.ChunkInfo (DatanodeClientProtocol.proto:366-372: chunkName = 1, offset = 2, len = 3,
    repeated KeyValue metadata = 4, checksumData = 5; KeyValue :159-162)."""

    def __init__(self, chunkName, offset, len):  # noqa: A002 - the reference's parameter name
        self.chunkName = chunkName
        self.offset = offset
        self.len = len
        self.checksumData = None
        self.metadata = {}
        self.readDataIntoSingleBuffer = False

    def addMetadata(self, key, value):
        if key in self.metadata:
            raise IOError("This key already exists. Key " + key)  # ChunkInfo.java:72-79
        self.metadata[key] = value

    def getChunkName(self):
        return self.chunkName

    def getOffset(self):
        return self.offset

    def getLen(self):
        return self.len

    def getChecksumData(self):
        return self.checksumData

    def setChecksumData(self, cData):
        self.checksumData = cData

    def getMetadata(self):
        return self.metadata

    def getProtoBufMessage(self):
        """Serialised ContainerProtos.ChunkInfo (ChunkInfo.java:108-133): fields in number order, metadata in key
        order (a TreeMap), a missing ChecksumData written as the NONE message."""
        name = self.chunkName.encode("utf-8")
        out = bytearray()
        out += b"\x0a" + _varint(len(name)) + name
        out += b"\x10" + _varint(self.offset)
        out += b"\x18" + _varint(self.len)
        for k in sorted(self.metadata):
            kb, vb = k.encode("utf-8"), self.metadata[k].encode("utf-8")
            kv = b"\x0a" + _varint(len(kb)) + kb + b"\x12" + _varint(len(vb)) + vb
            out += b"\x22" + _varint(len(kv)) + kv
        cd = self.checksumData.getProtoBufMessage() if self.checksumData is not None else Checksum.getNoChecksumDataProto()
        out += b"\x2a" + _varint(len(cd)) + cd
        return bytes(out)

    @staticmethod
    def getFromProtoBuf(buf):
        """Inverse of getProtoBufMessage (ChunkInfo.java:88-103)."""
        if buf is None:
            raise ValueError("info is null")
        name = off = ln = cd = None
        meta, pos, n = [], 0, len(buf)
        while pos < n:
            key, pos = _read_varint(buf, pos)
            field, wt = key >> 3, key & 7
            if wt == 0:
                v, pos = _read_varint(buf, pos)
                if field == 2:
                    off = v
                elif field == 3:
                    ln = v
            elif wt == 2:
                size, pos = _read_varint(buf, pos)
                body = bytes(buf[pos:pos + size])
                pos += size
                if field == 1:
                    name = body.decode("utf-8")
                elif field == 4:
                    meta.append(_read_key_value(body))
                elif field == 5:
                    cd = ChecksumData.getFromProtoBuf(body)
            else:
                raise ValueError("unsupported wire type %d" % wt)
        if name is None or off is None or ln is None or cd is None:
            raise ValueError("required field missing")
        info = ChunkInfo(name, off, ln)
        for k, v in meta:
            info.addMetadata(k, v)
        info.setChecksumData(cd)
        return info

    def __str__(self):  # ChunkInfo.java:182-188 (the quote after chunkName= is not closed there either)
        return "ChunkInfo{chunkName='%s, offset=%d, len=%d}" % (self.chunkName, self.offset, self.len)


def _read_key_value(body):
    k, v, pos = "", "", 0
    while pos < len(body):
        key, pos = _read_varint(body, pos)
        size, pos = _read_varint(body, pos)
        text = body[pos:pos + size].decode("utf-8")
        pos += size
        if key >> 3 == 1:
            k = text
        elif key >> 3 == 2:
            v = text
    return k, v


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _split(raw, ds):
    return [raw[i:i + ds] for i in range(0, len(raw), ds)]


class Checksum:
    """Checksum.java:43-299.  Not thread safe in the reference (:41); this mirror keeps no state."""

    def __init__(self, type, bytesPerChecksum):  # noqa: A002 - the reference's parameter name
        self.checksumType = type
        self.bytesPerChecksum = bytesPerChecksum

    @staticmethod
    def int2ByteString(n):
        """Checksum.java:62-64: Guava Ints.toByteArray -> 4 bytes big-endian."""
        return (n & 0xFFFFFFFF).to_bytes(4, "big")

    def computeChecksum(self, data, off=None, length=None, reference_list_quirk=True):
        """The five computeChecksum overloads (Checksum.java:119-177).

        reference_list_quirk (list inputs only): True, the default, is the reference bit for bit -- the list
        overload wraps the list in a ChunkBuffer (Checksum.java:148-153) whose iterator stops early on lists of
        four or more elements, so only the elements reference_list_windows() names are checksummed (three of four,
        24 of 256).  It is the same default as ChunkBuffer.wrap(list), so computeChecksum(list) ==
        computeChecksum(ChunkBuffer.wrap(list)) as in the reference, and what a Checksum patched to call this
        library yields (the patch keeps the reference's iteration and only moves the per-window function to the
        GPU).  False checksums every element: the documented intent (Checksum.java:279-281).

        data: bytes-like (byte[] / ByteBuffer overloads; off/length as in computeChecksum(byte[],int,int)),
        a chunk_buffer.ByteBuffer or ChunkBuffer (windows as its iterate(bytesPerChecksum) cuts them),
        or a list of bytes-like (the List<ByteString> / ChunkBuffer-of-list overloads, one element per
        window; an element longer than bytesPerChecksum contributes only its first bytesPerChecksum
        bytes, Checksum.java:186-198 over ChunkBufferImplWithByteBufferList.java:180-204)."""
        t = self.checksumType
        if t == ChecksumType.NONE:
            return ChecksumData(t, self.bytesPerChecksum)  # :157-160
        try:
            ct = ChecksumType(int(t))
        except (ValueError, TypeError):
            raise OzoneChecksumException(("type", t))  # :163-167
        ds = native.DIGEST_SIZE[int(ct)]
        bpc = self.bytesPerChecksum
        if bpc <= 0:
            raise ValueError("bytesPerChecksum must be positive")  # ChunkBuffer.iterate rejects it
        if isinstance(data, ByteBuffer):  # :140-146 (the read-only view changes nothing here)
            data = ChunkBuffer.wrap(data)
        if isinstance(data, ChunkBuffer):
            return ChecksumData(ct, bpc, self._compute_windows(int(ct), bpc, ds, data))
        if isinstance(data, (list, tuple)):
            if reference_list_quirk:
                data = [data[i] for i in reference_list_windows([len(memoryview(b).cast("B")) for b in data])]
            return ChecksumData(ct, bpc, self._compute_list(int(ct), bpc, ds, data))
        mv = memoryview(data).cast("B") if not isinstance(data, (bytes, bytearray)) else memoryview(data)
        if off is not None:
            mv = mv[off:off + length]
        if len(mv) == 0:
            return ChecksumData(ct, bpc, [])
        return ChecksumData(ct, bpc, _split(native.compute(int(ct), bpc, mv), ds))

    # what Algorithm.newChecksumFunction() returns for a buffer with nothing remaining: the CRC register after
    # reset() (Checksum.java:70-74), MessageDigest.digest() of no input (:55-59)
    _EMPTY_DIGEST = {
        int(ChecksumType.CRC32): bytes(4),
        int(ChecksumType.CRC32C): bytes(4),
        int(ChecksumType.SHA256): bytes.fromhex("e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"),
        int(ChecksumType.MD5): bytes.fromhex("d41d8cd98f00b204e9800998ecf8427e"),
    }

    @classmethod
    def _compute_windows(cls, t, bpc, ds, chunkBuffer):
        """computeChecksum(ChunkBuffer), Checksum.java:155-177: one digest per buffer data.iterate(bytesPerChecksum)
        yields, each over at most bytesPerChecksum bytes from the buffer's position (:186-198).  The iteration
        runs first (it moves the ChunkBuffer's position as in the reference); then all windows go to the GPU in
        one call.  A buffer with nothing remaining -- IncrementalChunkBuffer hands out its buffers as they are,
        consumed ones included -- gets the digest of the empty message, as the reference's function yields."""
        windows = [b.view()[:bpc] for b in chunkBuffer.iterate(bpc)]
        filled = [w for w in windows if len(w)]
        sums = iter(cls._compute_list(t, bpc, ds, filled))
        return [next(sums) if len(w) else cls._EMPTY_DIGEST[t] for w in windows]

    @staticmethod
    def _compute_list(t, bpc, ds, buffers):
        bufs = [memoryview(b).cast("B") if not isinstance(b, (bytes, bytearray)) else memoryview(b) for b in buffers]
        bufs = [b[:bpc] for b in bufs if len(b) > 0]  # empty elements yield nothing to iterate over
        if not bufs:
            return []
        if all(len(b) == bpc for b in bufs[:-1]):
            # the usual layout (ChunkManager.java:113-125 sizes read buffers to bpc): one pass
            return _split(native.compute(t, bpc, b"".join(bufs)), ds)
        # ragged elements: each is its own window -> one batched call, one descriptor per element
        import numpy as np
        n = len(bufs)
        arr = (native.OzcsChunk * n)()
        keep = []
        outs = []
        for i, b in enumerate(bufs):
            a = np.frombuffer(b, dtype=np.uint8)
            o = ctypes.create_string_buffer(ds)
            keep.append(a)
            outs.append(o)
            arr[i].data = a.ctypes.data
            arr[i].len = a.nbytes
            arr[i].sums = ctypes.cast(o, ctypes.c_void_p).value
            arr[i].sums_cap = ds
            arr[i].bpc = bpc
            arr[i].type = t
        native.check(native.lib().ozcs_compute_batch(arr, n))
        return [o.raw for o in outs]

    @staticmethod
    def verifyChecksum(data, checksumData, startIndex=0, isSingleByteString=None, reference_list_quirk=True):
        """The three verifyChecksum overloads (Checksum.java:210-290).

        reference_list_quirk: as in computeChecksum, for the one-element-per-window layout (:282-288).

        bytes-like data: verifyChecksum(ByteString, ChecksumData, int) / (byte[], ChecksumData).
        list data + isSingleByteString: verifyChecksum(List<ByteString>, ChecksumData, int, boolean)."""
        t = checksumData.getChecksumType()
        if t == ChecksumType.NONE:
            return True  # :241-244, :269-272
        bpc = checksumData.getBytesPerChecksum()
        if isinstance(data, (list, tuple)):
            if isSingleByteString:
                data = data[0]  # :274-277
            else:
                computed = Checksum(t, bpc).computeChecksum(list(data), reference_list_quirk=reference_list_quirk)  # :282-288
                return checksumData.verifyChecksumDataMatches(computed, startIndex)
        stored = checksumData.getChecksums()
        if len(stored) == 0:
            raise OzoneChecksumException("Original checksumData has no checksums")
        mv = memoryview(data).cast("B") if not isinstance(data, (bytes, bytearray)) else memoryview(data)
        if len(mv) == 0:
            raise OzoneChecksumException("Computed checksumData has no checksums")
        if startIndex < 0:
            raise IndexError(startIndex)
        # compare on the device: only the verdict comes back
        rc, first = native.verify(int(t), bpc, mv, b"".join(stored), startIndex)
        if rc == native.ERANGE:
            n = (len(mv) + bpc - 1) // bpc
            raise OzoneChecksumException(
                "Computed checksum has %d number of checksums. Original checksum has %d number of checksums "
                "starting from index %d" % (n, len(stored) - startIndex, startIndex))
        if rc == native.EEMPTY:
            raise OzoneChecksumException(native.lib().ozcs_last_error().decode())
        if first >= 0:
            raise OzoneChecksumException(first)
        return True

    @staticmethod
    def getNoChecksumDataProto():
        """Checksum.java:295-298."""
        return ChecksumData(ChecksumType.NONE, 0).getProtoBufMessage()
