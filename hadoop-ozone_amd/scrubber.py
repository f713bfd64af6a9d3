"""Batched form of the datanode scrubber's per-chunk check.

Mirrors KeyValueContainerCheck.verifyChecksum
(hadoop-hdds/container-service/src/main/java/org/apache/hadoop/ozone/container/keyvalue/KeyValueContainerCheck.java:262-322):
the reference reads one bytes-per-checksum window at a time, throttles, checksums it on the CPU and compares.
Here the chunk's windows are read in one go and checked by ONE ozcs_verify call (or, for many chunks, one
ozcs_verify_device launch); the observable behaviour is kept:

  * FILE_PER_BLOCK positions the file at the chunk's offset and never reads past the chunk (:276-285);
  * window i is throttled, then judged, and the first failing window ends the scan: the throttler sees exactly
    the windows the reference would have read up to the failure, in order, with the canceler passed through (:294);
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


def _read_chunk(chunk_offset, chunk_len, checksum_data, chunk_file, layout):
    """The bytes the reference's loop would read for this chunk, in one go: `count` windows of bpc, the last one
    cut for FILE_PER_BLOCK so that the next chunk is not read (KeyValueContainerCheck.java:276-285)."""
    count = len(checksum_data.getChecksums())
    bpc = checksum_data.getBytesPerChecksum()
    want = count * bpc
    if layout == FILE_PER_BLOCK and count and chunk_len % bpc != 0:
        want = (count - 1) * bpc + chunk_len % bpc
    with open(chunk_file, "rb") as f:
        if layout == FILE_PER_BLOCK:
            f.seek(chunk_offset)
        return f.read(want) if want else b""


def _replay(block_id, chunk_name, chunk_offset, chunk_len, checksum_data, data, first, throttler, canceler):
    """Replays the reference's per-window order of events (KeyValueContainerCheck.java:279-320) over a chunk whose
    windows were checked in bulk (`first` = first failing window, or -1): window i is throttled, then judged, and
    the first failing window ends the loop -- later windows are neither throttled nor looked at.  The throttler
    is DataTransferThrottler.throttle(numBytes, canceler): it may return early when the canceler says so, the
    reference's loop goes on regardless, and so does this one."""
    sums = checksum_data.getChecksums()
    bpc = checksum_data.getBytesPerChecksum()
    t = checksum_data.getChecksumType()
    n_read = len(data)
    windows_read = (n_read + bpc - 1) // bpc if bpc else 0
    for i in range(windows_read):
        if throttler is not None:
            throttler(min(bpc, n_read - i * bpc), canceler)   # :294
        if t == ChecksumType.NONE:
            # Checksum.computeChecksum yields no checksums for NONE, so getChecksums().get(0) (:297-298) throws in
            # the reference for a NONE chunk that carries stored checksums (one without them has no windows)
            raise IndexError("Index: 0, Size: 0")
        if i == first:                                        # :297-311
            ds = native.DIGEST_SIZE[int(t)]
            actual = native.compute(int(t), bpc, data[i * bpc:(i + 1) * bpc])[:ds]
            raise OzoneChecksumException(
                "Inconsistent read for chunk=%s checksum item %d expected checksum %s actual checksum %s for block %s"
                % (chunk_info_str(chunk_name, chunk_offset, chunk_len), i, _java_bytes(sums[i]), _java_bytes(actual),
                   block_id))
    if n_read != chunk_len:                                   # :314-320
        raise OzoneChecksumException("Inconsistent read for chunk=%s expected length=%d actual length=%d for block %s"
                                     % (chunk_name, chunk_len, n_read, block_id))
    return n_read


def verifyChecksum(block_id, chunk_name, chunk_offset, chunk_len, checksum_data, chunk_file, layout,
                   throttler=None, canceler=None):
    """Checks one chunk of a closed container against its stored ChecksumData.

    throttler: callable(num_bytes, canceler) or None (DataTransferThrottler.throttle);
    canceler: passed through to the throttler, as in the reference.  Returns the number of bytes read."""
    sums = checksum_data.getChecksums()
    bpc = checksum_data.getBytesPerChecksum()
    t = checksum_data.getChecksumType()
    data = _read_chunk(chunk_offset, chunk_len, checksum_data, chunk_file, layout)
    windows_read = (len(data) + bpc - 1) // bpc if bpc else 0
    first = -1
    if windows_read and t != ChecksumType.NONE:
        rc, first = native.verify(int(t), bpc, data, b"".join(sums[:windows_read]), 0)
        if rc != native.OK:
            raise OzoneChecksumException(native.lib().ozcs_last_error().decode())
    return _replay(block_id, chunk_name, chunk_offset, chunk_len, checksum_data, data, first, throttler, canceler)


def verifyChunkProto(block_id, chunk_proto, chunk_file, layout, throttler=None, canceler=None):
    """verifyChecksum with the arguments the reference's function takes (KeyValueContainerCheck.java:262-267): the
    chunk as a serialised ContainerProtos.ChunkInfo, from which name, offset, length and the stored ChecksumData are
    read (:266-269, :305)."""
    from .checksum import ChunkInfo
    info = ChunkInfo.getFromProtoBuf(chunk_proto)
    return verifyChecksum(block_id, info.getChunkName(), info.getOffset(), info.getLen(), info.getChecksumData(),
                          chunk_file, layout, throttler, canceler)


_MISSING, _SKIPPED = "missing", "skipped"


def _verify_chunks(flat, layout, throttler, canceler, block_still_exists=None):
    """flat: list of (block_id, (chunk_name, chunk_offset, chunk_len, checksum_data, chunk_file)).  All chunks are
    read and hashed by ONE ozcs_verify_batch call; then the reference's order of events is replayed chunk by chunk.

    The per-chunk gate of scanData (KeyValueContainerCheck.java:241-256) is applied first: a chunk whose file does not
    exist is not read -- at its turn in the replay the block is looked up again (block_still_exists(block_id); the
    reference asks the block table, :243-247) and "Missing chunk file <path>" is raised if it is still there, else
    the chunk is passed over (the block was deleted meanwhile); a chunk whose stored checksum type is NONE is passed
    over as well: not read, not throttled, not judged (:252-253)."""
    import ctypes
    import os
    import numpy as np
    reads = []
    empty = np.zeros(0, dtype=np.uint8)
    for bid, (name, off, ln, cd, path) in flat:
        if not os.path.exists(path):
            reads.append((empty, b"", _MISSING))
            continue
        if cd.getChecksumType() == ChecksumType.NONE:
            reads.append((empty, b"", _SKIPPED))
            continue
        data = _read_chunk(off, ln, cd, path, layout)
        bpc = cd.getBytesPerChecksum()
        nwin = (len(data) + bpc - 1) // bpc if bpc else 0
        reads.append((np.frombuffer(data, dtype=np.uint8), b"".join(cd.getChecksums()[:nwin]), nwin))
    n = len(flat)
    arr = (native.OzcsVChunk * max(1, n))()
    for i, ((bid, (name, off, ln, cd, path)), (data, stored, nwin)) in enumerate(zip(flat, reads)):
        t = cd.getChecksumType()
        if nwin in (_MISSING, _SKIPPED):
            arr[i].type = int(ChecksumType.NONE)                   # nothing for the GPU to do
            arr[i].bpc = 1
            continue
        arr[i].type = int(t) if nwin else int(ChecksumType.NONE)   # nothing read: only the length check applies
        arr[i].bpc = cd.getBytesPerChecksum()
        arr[i].data = data.ctypes.data if data.size else None
        arr[i].len = data.size
        arr[i].expected = ctypes.cast(ctypes.c_char_p(stored), ctypes.c_void_p).value
        arr[i].n_expected = nwin
        arr[i].start_index = 0
    rc = native.lib().ozcs_verify_batch(arr, n) if n else native.OK
    total = 0
    for i, ((bid, (name, off, ln, cd, path)), (data, stored, nwin)) in enumerate(zip(flat, reads)):
        if nwin == _SKIPPED:
            continue                                               # :252-253
        if nwin == _MISSING:
            # "concurrent mutation in Block DB? lookup the block again" (:242-251)
            if block_still_exists is None or block_still_exists(bid):
                raise IOError("Missing chunk file " + os.path.abspath(path))
            continue
        if arr[i].status != native.OK:
            if rc in (native.ENODEV, native.EHIP, native.ENOMEM):
                native.check(arr[i].status)
            raise OzoneChecksumException("verify failed for chunk %s: status %d" % (name, arr[i].status))
        total += _replay(bid, name, off, ln, cd, data, arr[i].first_mismatch, throttler, canceler)
    return total


def verifyBlock(block_id, chunks, layout, throttler=None, canceler=None, block_still_exists=None):
    """The `for chunk in block.getChunks()` loop of KeyValueContainerCheck.scanData (:235-257) with ONE
    ozcs_verify_batch call for all chunks instead of one checksum call per window, the loop's gate included
    (:241-256): a chunk with stored type NONE is passed over unread and unthrottled; a chunk whose file is missing
    raises IOError("Missing chunk file <absolute path>") at its turn if block_still_exists(block_id) says the block
    is still in the table (default: it is), and is passed over otherwise.

    chunks: list of (chunk_name, chunk_offset, chunk_len, checksum_data, chunk_file).  All chunks are read and
    hashed in bulk first; then the reference's order of events is replayed chunk by chunk, window by window
    (throttle, judge, stop at the first failure): the throttler sees exactly the windows the reference would
    have read up to the failure, and the first chunk (in order) that fails raises what verifyChecksum raises
    for it.  Returns the number of bytes read."""
    return _verify_chunks([(block_id, c) for c in chunks], layout, throttler, canceler, block_still_exists)


def verifyContainer(blocks, layout, throttler=None, canceler=None, batch_bytes=1 << 30, block_still_exists=None):
    """The data scan of one container, KeyValueContainerCheck.scanData (:226-257): every chunk of every block, in
    order.  Chunks are gathered across block boundaries into batches of about `batch_bytes` of stored length, each
    batch one ozcs_verify_batch call; the reference's order of events is then replayed chunk by chunk, so the first
    failing chunk of the first failing block raises what verifyChecksum raises for it and nothing behind it is
    throttled.  (A chunk behind a failure may already have been read: reads run ahead by at most one batch.)  The
    per-chunk gate is the reference's (:241-256, see verifyBlock): NONE chunks are passed over, a missing chunk file
    of a block that block_still_exists(block_id) confirms raises "Missing chunk file ...".

    blocks: list of (block_id, chunks) with chunks as in verifyBlock.  Returns the number of bytes read."""
    total = 0
    batch, size = [], 0

    def run():
        nonlocal total, batch, size
        # one call for the whole batch; replay needs every chunk's block id
        flat = [(bid, c) for bid, chunks in batch for c in chunks]
        total += _verify_chunks(flat, layout, throttler, canceler, block_still_exists)
        batch, size = [], 0

    for bid, chunks in blocks:
        batch.append((bid, chunks))
        size += sum(c[2] for c in chunks)
        if size >= batch_bytes:
            run()
    if batch:
        run()
    return total
