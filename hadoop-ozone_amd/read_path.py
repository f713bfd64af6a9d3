"""Client read-path pieces around the checksum call, mirrored from ChunkInputStream
(hadoop-hdds/client/src/main/java/org/apache/hadoop/hdds/scm/storage/ChunkInputStream.java).

  computeChecksumBoundaries   :522-535  widen a read to bytes-per-checksum boundaries
  validate                    :455-504  the validator run on every ReadChunk response: length check, start index,
                                        Checksum.verifyChecksum (here: one GPU call for all windows of the response,
                                        without the per-window copy of ChecksumByteBufferImpl.java:42-46)

and the datanode side that shapes what the validator receives (the V0 / V1 layouts of a ReadChunk response):

  getBufferCapacityForChunkRead   hadoop-hdds/container-service/.../keyvalue/interfaces/ChunkManager.java:107-134
  assignByteBuffers               hadoop-hdds/common/.../ozone/common/utils/BufferUtils.java:42-63, 127-129
  readChunkBuffers                the cut FilePerBlockStrategy.readChunk makes (FilePerBlockStrategy.java:163-173)

and what the client does with the validator's verdict:

  readChunkWithFailover           the loop over a pipeline's datanodes of XceiverClientGrpc.sendCommandWithRetry
                                  (hadoop-hdds/client/.../hdds/scm/XceiverClientGrpc.java:368-418): an IOException from a
                                  validator -- a checksum mismatch is one -- sends the read to the next replica
"""
from .checksum import Checksum, ChecksumType, OzoneChecksumException


def computeChecksumBoundaries(startByteIndex, dataLen, bytesPerChecksum, chunkLength):
    """-> (adjustedChunkOffset, adjustedChunkLen): the byte range to fetch so that whole windows can be verified."""
    endByteIndex = startByteIndex + dataLen - 1
    adjustedChunkOffset = (startByteIndex // bytesPerChecksum) * bytesPerChecksum        # inclusive
    endIndex = ((endByteIndex // bytesPerChecksum) + 1) * bytesPerChecksum              # exclusive
    return adjustedChunkOffset, min(endIndex, chunkLength) - adjustedChunkOffset


def validate(chunkName, chunkOffset, checksumData, reqOffset, reqLen, data=None, dataBuffers=None, verifyChecksum=True):
    """The ReadChunk response validator.

    chunkOffset / checksumData describe the whole chunk (ChunkInfo); reqOffset / reqLen the (boundary-widened)
    range that was requested.  Exactly one of `data` (V0: one ByteString) and `dataBuffers` (V1: one ByteString
    per window) is given.  Raises OzoneChecksumException, which the reference treats as "try the next replica"
    (XceiverClientGrpc.java:368-409)."""
    if data is not None:
        byteStrings, isV0, got = [data], True, len(data)
    else:
        byteStrings, isV0 = list(dataBuffers), False
        got = sum(len(b) for b in byteStrings)   # BufferUtils.getBuffersLen
    if got != reqLen:
        raise OzoneChecksumException("Inconsistent read for chunk=%s len=%d bytesRead=%d" % (chunkName, reqLen, got))
    if verifyChecksum:
        relativeOffset = reqOffset - chunkOffset
        startIndex = int(relativeOffset // checksumData.getBytesPerChecksum())
        Checksum.verifyChecksum(byteStrings, checksumData, startIndex, isV0)
    return True



OZONE_CHUNK_READ_BUFFER_DEFAULT_SIZE = 64 * 1024   # ozone.chunk.read.buffer.default.size


def getBufferCapacityForChunkRead(chunkInfo, defaultReadBufferCapacity=OZONE_CHUNK_READ_BUFFER_DEFAULT_SIZE):
    """Capacity of the buffers a datanode reads a chunk into: the whole chunk for a client that asked for one buffer
    (V0), else one buffer per bytes-per-checksum window so that each buffer corresponds to one checksum (V1) -- the
    default capacity when the chunk carries no checksums -- and the whole chunk when that leaves 0."""
    bufferCapacity = 0
    if chunkInfo.readDataIntoSingleBuffer:
        bufferCapacity = chunkInfo.getLen()
    else:
        checksumData = chunkInfo.getChecksumData()
        if checksumData is not None:
            if checksumData.getChecksumType() == ChecksumType.NONE:
                bufferCapacity = defaultReadBufferCapacity
            else:
                bufferCapacity = checksumData.getBytesPerChecksum()
    if bufferCapacity == 0:
        bufferCapacity = chunkInfo.getLen()
    return bufferCapacity


def assignByteBuffers(totalLen, bufferCapacity):
    """Lengths of the buffers that hold totalLen bytes: ceil(totalLen / bufferCapacity) of them, all of
    bufferCapacity bytes but the last, which takes what remains."""
    if totalLen <= 0:
        raise ValueError("Buffer Length should be a positive integer.")
    if bufferCapacity <= 0:
        raise ValueError("Buffer Capacity should be a positive integer.")
    numBuffers = -(-totalLen // bufferCapacity)
    return [bufferCapacity] * (numBuffers - 1) + [totalLen - bufferCapacity * (numBuffers - 1)]


def readChunkBuffers(chunkData, chunkInfo, defaultReadBufferCapacity=OZONE_CHUNK_READ_BUFFER_DEFAULT_SIZE):
    """The bytes of a chunk (or of the requested range of it) cut the way a datanode returns them: a list with one
    element for a V0 client, one element per window otherwise.  An empty chunk gives one empty buffer
    (FilePerBlockStrategy.java:150-153)."""
    if len(chunkData) <= 0:
        return [bytes(chunkData)]
    out, pos = [], 0
    for n in assignByteBuffers(len(chunkData), getBufferCapacityForChunkRead(chunkInfo, defaultReadBufferCapacity)):
        out.append(bytes(chunkData[pos:pos + n]))
        pos += n
    return out


def readChunkWithFailover(datanodes, sendCommand, validators=()):
    """Asks the datanodes of a pipeline in turn until one answers with a response every validator accepts.

    sendCommand(dn) -> response; validators are callables of the response that raise IOError (OzoneChecksumException
    is one) to reject it.  Returns (response, datanodes tried in order).  When every datanode fails, the LAST
    datanode's exception is raised, as the reference keeps only the latest (`ioException = e`)."""
    ioException = None
    tried = []
    for dn in datanodes:
        tried.append(dn)
        try:
            response = sendCommand(dn)
            for validator in validators:
                validator(response)
            return response, tried
        except IOError as e:
            ioException = e
    if ioException is None:
        raise ValueError("no datanode to read from")   # Preconditions.checkNotNull(ioException)
    raise ioException
