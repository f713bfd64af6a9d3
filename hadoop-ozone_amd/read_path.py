"""Client read-path pieces around the checksum call, mirrored from ChunkInputStream
(hadoop-hdds/client/src/main/java/org/apache/hadoop/hdds/scm/storage/ChunkInputStream.java).

  computeChecksumBoundaries   :522-535  widen a read to bytes-per-checksum boundaries
  validate                    :455-504  the validator run on every ReadChunk response: length check, start index,
                                        Checksum.verifyChecksum (here: one GPU call for all windows of the response,
                                        without the per-window copy of ChecksumByteBufferImpl.java:42-46)
"""
from .checksum import Checksum, OzoneChecksumException


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
