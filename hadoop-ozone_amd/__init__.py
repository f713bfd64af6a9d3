"""MI355X-native per-chunk data checksumming for Ozone (CRC32 / CRC32C / SHA-256 / MD5 per
bytes-per-checksum window).

  native    ctypes view of the C-ABI HIP library (csrc/libozcs.so, include/ozone_checksum.h)
  checksum  host-side mirror of the reference interface: Checksum, ChecksumData,
            OzoneChecksumException, ChecksumType (hadoop-hdds/common/.../ozone/common/), the ChunkInfo
            record that carries ChecksumData on the wire, and the client configuration keys
  chunk_buffer  ChunkBuffer and its three implementations (the buffer a chunk is collected in, cut into windows
            by iterate()), over the part of java.nio.ByteBuffer they use
  read_path the validator and boundary arithmetic of ChunkInputStream around the verify call
  scrubber  the datanode scrubber's per-chunk check (KeyValueContainerCheck.verifyChecksum), batched
  write_path  BlockOutputStream's chunking and ChunkInfo bookkeeping over the pinned pipe
"""
from . import native  # noqa: F401
from .checksum import (Checksum, ChecksumData, ChecksumType, ChunkInfo,  # noqa: F401
                       OzoneChecksumException, OzoneClientConfig)
from .chunk_buffer import ByteBuffer, ChunkBuffer  # noqa: F401
from . import chunk_buffer, read_path, scrubber, write_path  # noqa: F401,E402
