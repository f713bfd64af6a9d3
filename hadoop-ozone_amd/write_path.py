"""Client write-path pieces around the checksum call, mirrored from BlockOutputStream
(hadoop-hdds/client/src/main/java/org/apache/hadoop/hdds/scm/storage/BlockOutputStream.java).

  write / writeChunkIfNeeded   :225-266   bytes go into the current chunk buffer; a full buffer becomes a chunk
  writeChunkToContainer        :645-690   offset bookkeeping, Checksum.computeChecksum(chunk), the ChunkInfo record
                                          (name "<localID>_chunk_<n>", offset, len, checksumData), writeChunkAsync
  flush / close                :405-473   the partly filled buffer goes out as a last, shorter chunk

The reference computes the checksum synchronously on the writer thread before the chunk is shipped.  Here the chunk
buffers are the pinned slots of an ozcs pipe (in place of BufferPool's heap buffers, BufferPool.java:74-85): a full
buffer is submitted and the writer goes on filling the next one while the first is copied and hashed; chunks are
handed to `writeChunkAsync` strictly in order, each as soon as its digests have arrived.  What comes out -- the
ChunkInfo records and the bytes -- is what the reference would ship.

The reference turns the chunk buffer into the request's ByteString before it computes the checksum
(chunk.toByteString(bufferPool.byteStringConversion()), :648-649): a copy by default, a zero-copy wrap when
ozone.client.stream.buffer... unsafe conversion is configured.  Both are here:
  * "safe" (default): the chunk's bytes are copied out of the pinned slot when it is submitted, and the pipe runs in
    in-order mode -- a slot goes back to the pool the moment the GPU is done with it, so a CRC chunk does not hold its
    slot behind an older SHA-256 / MD5 chunk, while ChunkInfo records still leave in the order of the writes;
  * "unsafe": the slot itself is what is shipped, so it stays the chunk's until its turn to be shipped has come.
"""
import ctypes

from . import native
from .checksum import Checksum, ChecksumData, ChecksumType, ChunkInfo, OzoneChecksumException, OzoneClientConfig


def writeSmallFileChunkInfo(localID, data):
    """The ChunkInfo ContainerProtocolCalls.writeSmallFile builds for a small file that travels in one
    PutSmallFile request (hadoop-hdds/common/src/main/java/org/apache/hadoop/hdds/scm/storage/
    ContainerProtocolCalls.java:334-345): name "<localID>_chunk", offset 0, the whole length, the metadata pair
    OverWriteRequested=true, and CRC32 checksums over 256-byte windows whatever the client configuration says."""
    info = ChunkInfo("%d_chunk" % localID, 0, len(data))
    info.addMetadata("OverWriteRequested", "true")
    info.setChecksumData(Checksum(ChecksumType.CRC32, 256).computeChecksum(data))
    return info


class BlockOutputStream:
    def __init__(self, localID, writeChunkAsync, config=None, streamBufferSize=4 << 20, capacity=8, device=0,
                 byteStringConversion="safe"):
        """writeChunkAsync(chunkInfo: ChunkInfo, data: bytes) receives every chunk, in order (ContainerProtocolCalls
        .writeChunkAsync in the reference).  capacity = streamBufferMaxSize / streamBufferSize, the number of chunk
        buffers (BlockOutputStreamEntryPool.java:91-96).  byteStringConversion: "safe" or "unsafe", see the module text."""
        if byteStringConversion not in ("safe", "unsafe"):
            raise ValueError(byteStringConversion)
        self.safe = byteStringConversion == "safe"
        self.config = config or OzoneClientConfig()
        self.type = self.config.getChecksumType()
        self.bpc = self.config.getBytesPerChecksum()
        self.localID = localID
        self.sink = writeChunkAsync
        self.bufferSize = streamBufferSize
        self.chunkOffset = 0
        self.chunkIndex = 0
        self.writtenDataLength = 0
        self.chunks = []       # containerBlockData's chunk list (ChunkInfo records), in order
        self._pending = []     # submitted chunks not yet shipped: (ticket, ChunkInfo, slot address, digest buffer)
        self._cur = None       # (slot address, ticket) of the buffer being filled
        self._pos = 0
        self._closed = False
        self._L = native.lib()
        if int(self.type) not in native.DIGEST_SIZE:
            raise OzoneChecksumException(("type", self.type))
        self._pipe = ctypes.c_void_p()
        native.check(self._L.ozcs_pipe_create(device, capacity, streamBufferSize, ctypes.byref(self._pipe)))
        if self.safe:
            native.check(self._L.ozcs_pipe_set_in_order(self._pipe, 1))

    # ---- BlockOutputStream.write(byte[], off, len), :241-266
    def write(self, b, off=0, length=None):
        self._check_open()
        mv = memoryview(b).cast("B")
        if length is None:
            length = len(mv) - off
        if off < 0 or length < 0 or off + length > len(mv):
            raise IndexError("off=%d len=%d of %d" % (off, length, len(mv)))
        while length > 0:
            self._allocate_new_buffer_if_needed()
            n = min(self.bufferSize - self._pos, length)
            ctypes.memmove(self._cur[0] + self._pos, (ctypes.c_char * n).from_buffer_copy(mv[off:off + n]), n)
            self._pos += n
            if self._pos == self.bufferSize:
                self._write_chunk()
            off += n
            length -= n
            self.writtenDataLength += n

    def flush(self):
        """The data buffered so far goes out (handleFlush: a partly filled buffer becomes a chunk of its own) and every
        submitted chunk is shipped."""
        self._check_open()
        if self._cur is not None and self._pos > 0:
            self._write_chunk()
        while self._pending:
            self._ship_oldest()

    def close(self):
        if self._closed:
            return
        try:
            self.flush()
        finally:
            self._closed = True
            if self._cur is not None:   # acquired, never filled: give it back
                self._L.ozcs_pipe_release(self._pipe, self._cur[1])
                self._cur = None
            self._L.ozcs_pipe_destroy(self._pipe)
            self._pipe = None

    def getWrittenDataLength(self):
        return self.writtenDataLength

    # ---- internals
    def _check_open(self):
        if self._closed:
            raise IOError("BlockOutputStream has been closed.")

    def _allocate_new_buffer_if_needed(self):  # :281-286 over BufferPool.allocateBuffer
        if self._cur is not None:
            return
        buf, ticket = ctypes.c_void_p(), ctypes.c_int()
        while True:
            # in-order pipe: block for a slot (any completion gives one back); otherwise a slot is free only once its
            # chunk has been shipped
            acquire = self._L.ozcs_pipe_acquire_wait if self.safe else self._L.ozcs_pipe_acquire
            rc = acquire(self._pipe, ctypes.byref(buf), ctypes.byref(ticket))
            if rc != native.EBUSY:
                break
            if not self._pending:   # nothing of ours is in flight, yet no buffer is to be had: report it, do not spin
                native.check(rc)
            self._ship_oldest()   # every buffer is in flight: the reference waits for acknowledgements here (handleFullBuffer)
        native.check(rc)
        self._cur, self._pos = (buf.value, ticket.value), 0
        if self.safe:
            self._ship_ready()

    def _write_chunk(self):  # writeChunk + writeChunkToContainer, :495-507, :645-690
        addr, ticket = self._cur
        n = self._pos
        offset = self.chunkOffset
        self.chunkOffset += n
        self.chunkIndex += 1
        info = ChunkInfo("%d_chunk_%d" % (self.localID, self.chunkIndex), offset, n)
        ds = native.DIGEST_SIZE[int(self.type)]
        nsums = 0 if self.type == ChecksumType.NONE else (n + self.bpc - 1) // self.bpc
        out = ctypes.create_string_buffer(max(1, nsums * ds))
        data = ctypes.string_at(addr, n) if self.safe else None   # chunk.toByteString's copy (:648-649)
        # whatever the outcome, the buffer is no longer this stream's current one: a rejected submission has released
        # the slot, an accepted one owns it
        self._cur, self._pos = None, 0
        native.check(self._L.ozcs_pipe_submit_slot(self._pipe, ticket, int(self.type), self.bpc, n, out, nsums * ds))
        self._pending.append((ticket, info, addr, out, nsums, ds, data))

    def _ship_ready(self):
        """In-order pipe: ships every chunk at the head of the queue whose digests have arrived, without waiting."""
        while self._pending and self._ship_oldest(block=0):
            pass

    def _ship_oldest(self, block=1):
        if not self._pending:
            raise IOError("no chunk is in flight")
        ticket, info, addr, out, nsums, ds, data = self._pending[0]
        got = ctypes.c_uint64()
        if self.safe:
            done, status = ctypes.c_int(-1), ctypes.c_int(0)
            rc = self._L.ozcs_pipe_next_in_order(self._pipe, block, ctypes.byref(done), ctypes.byref(status), ctypes.byref(got))
            if rc == native.EBUSY and not block:
                return False
            native.check(rc)
            assert done.value == ticket, (done.value, ticket)
            native.check(status.value)
        else:
            native.check(self._L.ozcs_pipe_wait(self._pipe, ticket, ctypes.byref(got)))
            data = ctypes.string_at(addr, info.getLen())   # the slot is free now, but only this stream acquires from the pipe
        self._pending.pop(0)
        assert got.value == nsums
        raw = out.raw[:nsums * ds]
        info.setChecksumData(ChecksumData(self.type, self.bpc, [raw[i:i + ds] for i in range(0, len(raw), ds)] if ds else []))
        self.chunks.append(info)
        self.sink(info, data)
        return True
