"""Batched form of the datanode scrubber's per-chunk check.

Mirrors KeyValueContainerCheck.verifyChecksum
(hadoop-hdds/container-service/src/main/java/org/apache/hadoop/ozone/container/keyvalue/KeyValueContainerCheck.java:262-322):
the reference reads one bytes-per-checksum window at a time, throttles, checksums it on the CPU and compares.
Here the chunk's windows are read in one go and checked by ONE ozcs_verify call (or, for many chunks, one
ozcs_verify_device launch); the observable behaviour is kept:

  * FILE_PER_BLOCK positions the file at the chunk's offset and never reads past the chunk (:276-285);
  * the throttler sees every window's byte count, in order, before that window is judged (:294);
  * the first failing window raises "Inconsistent read for chunk=... checksum item %d expected checksum ...
    actual checksum ... for block ..." with Java's Arrays.toString byte formatting (:299-311);
  * a file that ends early at a window boundary raises the "expected length=%d actual length=%d" form (:314-320),
    one that ends inside a window fails that window's comparison first, as in the reference.
"""
from . import native
from .checksum import ChecksumType, OzoneChecksumException

FILE_PER_CHUNK = "FILE_PER_CHUNK"
FILE_PER_BLOCK = "FILE_PER_BLOCK"


def _java_bytes(b):
    """Arrays.toString(byte[]): signed decimal bytes."""
    return "[" + ", ".join(str(x - 256 if x > 127 else x) for x in b) + "]"


def chunk_info_str(chunk_name, offset, length):
    """ChunkInfo.toString() (hadoop-hdds/common/.../container/common/helpers/ChunkInfo.java:182-188)."""
    return "ChunkInfo{chunkName='%s, offset=%d, len=%d}" % (chunk_name, offset, length)


class Canceled(Exception):
    pass


def verifyChecksum(block_id, chunk_name, chunk_offset, chunk_len, checksum_data, chunk_file, layout,
                   throttler=None, canceler=None):
    """Checks one chunk of a closed container against its stored ChecksumData.

    throttler: callable(num_bytes, canceler) or None (DataTransferThrottler.throttle);
    canceler: object with is_cancelled() or None.  Returns the number of bytes read."""
    sums = checksum_data.getChecksums()
    count = len(sums)
    bpc = checksum_data.getBytesPerChecksum()
    t = checksum_data.getChecksumType()
    # bytes the reference's loop would ask for: `count` windows of bpc, the last one cut for FILE_PER_BLOCK
    want = count * bpc
    if layout == FILE_PER_BLOCK and count and chunk_len % bpc != 0:
        want = (count - 1) * bpc + chunk_len % bpc
    with open(chunk_file, "rb") as f:
        if layout == FILE_PER_BLOCK:
            f.seek(chunk_offset)
        data = f.read(want) if want else b""
    n_read = len(data)
    windows_read = (n_read + bpc - 1) // bpc if bpc else 0
    if throttler is not None:
        for i in range(windows_read):
            throttler(min(bpc, n_read - i * bpc), canceler)
    if t != ChecksumType.NONE and windows_read:
        ds = native.DIGEST_SIZE[int(t)]
        rc, first = native.verify(int(t), bpc, data, b"".join(sums[:windows_read]), 0)
        if rc != native.OK:
            raise OzoneChecksumException(native.lib().ozcs_last_error().decode())
        if first >= 0:
            actual = native.compute(int(t), bpc, data[first * bpc:(first + 1) * bpc])[:ds]
            raise OzoneChecksumException(
                "Inconsistent read for chunk=%s checksum item %d expected checksum %s actual checksum %s for block %s"
                % (chunk_info_str(chunk_name, chunk_offset, chunk_len), first, _java_bytes(sums[first]),
                   _java_bytes(actual), block_id))
    elif t == ChecksumType.NONE and windows_read:
        # Checksum.computeChecksum yields no checksums for NONE, so getChecksums().get(0) fails in the
        # reference; a NONE chunk with a non-empty stored list cannot occur (the list is empty, count == 0)
        pass
    if n_read != chunk_len:
        raise OzoneChecksumException("Inconsistent read for chunk=%s expected length=%d actual length=%d for block %s"
                                     % (chunk_name, chunk_len, n_read, block_id))
    return n_read


def verifyBlock(block_id, chunks, layout, throttler=None, canceler=None):
    """The `for chunk in block.getChunks()` loop of KeyValueContainerCheck.scanData (:235-257) with ONE
    ozcs_verify_batch call for all chunks instead of one checksum call per window.

    chunks: list of (chunk_name, chunk_offset, chunk_len, checksum_data, chunk_file).  Chunks are read and
    throttled in order; the first chunk (in order) that fails raises what verifyChecksum would raise for it.
    Returns the number of bytes read."""
    import ctypes
    import numpy as np
    reads = []
    for name, off, ln, cd, path in chunks:
        sums = cd.getChecksums()
        bpc = cd.getBytesPerChecksum()
        want = len(sums) * bpc
        if layout == FILE_PER_BLOCK and sums and ln % bpc != 0:
            want = (len(sums) - 1) * bpc + ln % bpc
        with open(path, "rb") as f:
            if layout == FILE_PER_BLOCK:
                f.seek(off)
            data = f.read(want) if want else b""
        nwin = (len(data) + bpc - 1) // bpc if bpc else 0
        if throttler is not None:
            for i in range(nwin):
                throttler(min(bpc, len(data) - i * bpc), canceler)
        reads.append((np.frombuffer(data, dtype=np.uint8), b"".join(sums[:nwin]), nwin))
    n = len(chunks)
    arr = (native.OzcsVChunk * n)()
    for i, ((name, off, ln, cd, path), (data, stored, nwin)) in enumerate(zip(chunks, reads)):
        t = cd.getChecksumType()
        arr[i].type = int(t) if nwin else int(ChecksumType.NONE)   # nothing read: only the length check applies
        arr[i].bpc = cd.getBytesPerChecksum()
        arr[i].data = data.ctypes.data if data.size else None
        arr[i].len = data.size
        arr[i].expected = ctypes.cast(ctypes.c_char_p(stored), ctypes.c_void_p).value
        arr[i].n_expected = nwin
        arr[i].start_index = 0
    rc = native.lib().ozcs_verify_batch(arr, n)
    total = 0
    for i, ((name, off, ln, cd, path), (data, stored, nwin)) in enumerate(zip(chunks, reads)):
        if arr[i].status != native.OK:
            if rc in (native.ENODEV, native.EHIP, native.ENOMEM):
                native.check(arr[i].status)
            raise OzoneChecksumException("verify failed for chunk %s: status %d" % (name, arr[i].status))
        first = arr[i].first_mismatch
        if first >= 0:
            t, bpc = cd.getChecksumType(), cd.getBytesPerChecksum()
            ds = native.DIGEST_SIZE[int(t)]
            actual = native.compute(int(t), bpc, data[first * bpc:(first + 1) * bpc])[:ds]
            raise OzoneChecksumException(
                "Inconsistent read for chunk=%s checksum item %d expected checksum %s actual checksum %s for block %s"
                % (chunk_info_str(name, off, ln), first, _java_bytes(cd.getChecksums()[first]), _java_bytes(actual), block_id))
        if data.size != ln:
            raise OzoneChecksumException("Inconsistent read for chunk=%s expected length=%d actual length=%d for block %s"
                                         % (name, ln, data.size, block_id))
        total += data.size
    return total
