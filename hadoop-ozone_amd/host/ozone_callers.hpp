// The three callers of the checksum path, restructured around batched GPU calls, in C++ over
// ozone_checksum.hpp.  Observable behaviour (what is read, what is raised, with which text) follows the
// reference; only the per-window CPU loop inside is replaced.
//
//   read path   ChunkInputStream.computeChecksumBoundaries / validator
//               hadoop-hdds/client/src/main/java/org/apache/hadoop/hdds/scm/storage/ChunkInputStream.java:455-535
//               and the datanode's cut of a ReadChunk response into buffers (what makes the V0 / V1 layouts):
//               hadoop-hdds/container-service/src/main/java/org/apache/hadoop/ozone/container/keyvalue/interfaces/ChunkManager.java:107-134
//               hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/common/utils/BufferUtils.java:42-63, 127-129
//   scrubber    KeyValueContainerCheck.verifyChecksum
//               hadoop-hdds/container-service/src/main/java/org/apache/hadoop/ozone/container/keyvalue/KeyValueContainerCheck.java:262-322
//   write path  BlockOutputStream.writeChunkToContainer over BufferPool buffers
//               hadoop-hdds/client/src/main/java/org/apache/hadoop/hdds/scm/storage/BlockOutputStream.java:495-507, 645-690
//               hadoop-hdds/client/src/main/java/org/apache/hadoop/hdds/scm/storage/BufferPool.java:74-85
//               BlockOutputStream.write / flush / close (:225-266, :405-473) over those buffers
// Header-only; link libozcs.so.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "ozone_checksum.hpp"

namespace ozone {

// ---------------------------------------------------------------------------------------------- read path

// ChunkInputStream.java:522-535: the byte range to fetch so that whole windows can be verified.
// Returns (adjustedChunkOffset, adjustedChunkLen).
inline std::pair<long long, long long> computeChecksumBoundaries(long long startByteIndex, int dataLen,
                                                                 int bytesPerChecksum, long long chunkLength) {
  const long long endByteIndex = startByteIndex + dataLen - 1;  // inclusive
  const long long adjustedChunkOffset = (startByteIndex / bytesPerChecksum) * bytesPerChecksum;
  const long long endIndex = ((endByteIndex / bytesPerChecksum) + 1) * bytesPerChecksum;  // exclusive
  return {adjustedChunkOffset, std::min(endIndex, chunkLength) - adjustedChunkOffset};
}

// ChunkManager.java:107-134: capacity of the buffers a datanode reads a chunk into -- the whole chunk for a client that
// asked for one buffer (V0), else one buffer per bytes-per-checksum window so that each buffer corresponds to one
// checksum (V1), the default capacity (ozone.chunk.read.buffer.default.size, 64 KiB) when the chunk carries no
// checksums, and the whole chunk when that leaves 0.
constexpr long long kChunkReadBufferDefaultSize = 64 * 1024;
inline long long getBufferCapacityForChunkRead(const ChunkInfo& chunkInfo,
                                               long long defaultReadBufferCapacity = kChunkReadBufferDefaultSize) {
  long long bufferCapacity = 0;
  if (chunkInfo.isReadDataIntoSingleBuffer()) {
    bufferCapacity = chunkInfo.getLen();
  } else if (chunkInfo.hasChecksumData()) {
    const ChecksumData& c = chunkInfo.getChecksumData();
    bufferCapacity = c.getChecksumType() == ChecksumType::NONE ? defaultReadBufferCapacity : c.getBytesPerChecksum();
  }
  if (bufferCapacity == 0) bufferCapacity = chunkInfo.getLen();
  return bufferCapacity;
}

// BufferUtils.java:42-63: lengths of the buffers that hold totalLen bytes -- all of bufferCapacity bytes but the last,
// which takes what remains.
inline std::vector<long long> assignByteBuffers(long long totalLen, long long bufferCapacity) {
  if (totalLen <= 0) throw std::invalid_argument("Buffer Length should be a positive integer.");
  if (bufferCapacity <= 0) throw std::invalid_argument("Buffer Capacity should be a positive integer.");
  const long long numBuffers = (totalLen + bufferCapacity - 1) / bufferCapacity;
  std::vector<long long> lens((size_t)(numBuffers - 1), bufferCapacity);
  lens.push_back(totalLen - bufferCapacity * (numBuffers - 1));
  return lens;
}

// The bytes of a chunk (or of the requested range of it) cut the way FilePerBlockStrategy.readChunk returns them
// (FilePerBlockStrategy.java:150-173): one element for a V0 client, one per window otherwise; an empty chunk gives one
// empty buffer.
inline std::vector<ByteString> readChunkBuffers(const ByteString& chunkData, const ChunkInfo& chunkInfo,
                                                long long defaultReadBufferCapacity = kChunkReadBufferDefaultSize) {
  if (chunkData.empty()) return {ByteString()};
  std::vector<ByteString> out;
  size_t pos = 0;
  for (long long n : assignByteBuffers((long long)chunkData.size(),
                                       getBufferCapacityForChunkRead(chunkInfo, defaultReadBufferCapacity))) {
    out.push_back(chunkData.substr(pos, (size_t)n));
    pos += (size_t)n;
  }
  return out;
}

// The ReadChunk response validator (ChunkInputStream.java:455-504).  chunkOffset / checksumData describe the
// whole chunk, reqOffset / reqLen the (boundary-widened) range that was requested; byteStrings is the
// response: one element for the V0 layout (isV0), one element per window otherwise.  All windows of the
// response are verified by one library call (V0) or one batched call (V1) instead of a CPU pass per window
// with its copy (ChecksumByteBufferImpl.java:42-46).
inline void validateReadChunk(const std::string& chunkName, long long chunkOffset, const ChecksumData& checksumData,
                              long long reqOffset, long long reqLen, const std::vector<ByteString>& byteStrings,
                              bool isV0, bool verifyChecksum = true) {
  long long got = 0;
  for (const ByteString& b : byteStrings) got += (long long)b.size();  // BufferUtils.getBuffersLen
  if (got != reqLen)
    throw OzoneChecksumException("Inconsistent read for chunk=" + chunkName + " len=" + std::to_string(reqLen) +
                                 " bytesRead=" + std::to_string(got));
  if (!verifyChecksum) return;
  const long long relativeOffset = reqOffset - chunkOffset;
  const int startIndex = (int)(relativeOffset / checksumData.getBytesPerChecksum());
  Checksum::verifyChecksum(byteStrings, checksumData, startIndex, isV0);
}

// ---------------------------------------------------------------------------------------------- scrubber

enum class ChunkLayOutVersion { FILE_PER_CHUNK = 1, FILE_PER_BLOCK = 2 };

// Arrays.toString(byte[]): signed decimal bytes.
inline std::string javaBytesToString(const ByteString& b) {
  std::string s = "[";
  for (size_t i = 0; i < b.size(); i++) {
    if (i) s += ", ";
    s += std::to_string((int)(signed char)b[i]);
  }
  return s + "]";
}

// ChunkInfo.toString() (hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/container/common/helpers/ChunkInfo.java:182-188).
inline std::string chunkInfoToString(const std::string& chunkName, long long offset, long long len) {
  return "ChunkInfo{chunkName='" + chunkName + ", offset=" + std::to_string(offset) + ", len=" + std::to_string(len) + "}";
}

// One chunk of a closed container against its stored ChecksumData (KeyValueContainerCheck.java:262-322).
// The reference reads, throttles and checks one window at a time; here the chunk's windows are read in one
// go and checked by one ozcs_verify call.  Kept: FILE_PER_BLOCK positions the file at the chunk's offset and
// never reads past the chunk (:276-285); window i is throttled, then judged, and the first failing window ends
// the scan with the reference's text (:294-311), so the throttler sees exactly the windows up to the failure; a file that ends early at a window boundary
// raises the length form (:314-320), one that ends inside a window fails that window first.
// throttler may be empty (a caller that cancels binds its canceler into it, as DataTransferThrottler.throttle(v,
// canceler) receives it).  Returns the number of bytes read.
namespace detail {

// What the scrubber reads of one chunk: whole windows up to the stored count, FILE_PER_BLOCK positioned at the
// chunk's offset and never past the chunk (KeyValueContainerCheck.java:276-285).
inline std::string readChunkWindows(long long chunkOffset, long long chunkLen, const ChecksumData& checksumData,
                                    const std::string& chunkFile, ChunkLayOutVersion layout) {
  const long long count = (long long)checksumData.getChecksums().size();
  const long long bpc = checksumData.getBytesPerChecksum();
  long long want = count * bpc;
  if (layout == ChunkLayOutVersion::FILE_PER_BLOCK && count && chunkLen % bpc != 0) want = (count - 1) * bpc + chunkLen % bpc;
  std::string data((size_t)want, '\0');
  long long nRead = 0;
  FILE* f = std::fopen(chunkFile.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + chunkFile);
  if (layout == ChunkLayOutVersion::FILE_PER_BLOCK) std::fseek(f, (long)chunkOffset, SEEK_SET);
  if (want) nRead = (long long)std::fread(&data[0], 1, (size_t)want, f);
  std::fclose(f);
  data.resize((size_t)nRead);
  return data;
}

// The reference's order of events, replayed over a verdict the GPU has already given (`first` = first failing
// window or -1): window i is throttled, then judged, and the first failing window ends the scan (:294-311) -- the
// throttler never sees a window behind the failure; then the length check (:314-320).
inline long long replayChunk(const std::string& blockId, const std::string& chunkName, long long chunkOffset, long long chunkLen,
                             const ChecksumData& checksumData, const std::string& data, int64_t first,
                             const std::function<void(long long)>& throttler) {
  const std::vector<ByteString>& sums = checksumData.getChecksums();
  const long long bpc = checksumData.getBytesPerChecksum();
  const long long nRead = (long long)data.size();
  const long long windowsRead = bpc ? (nRead + bpc - 1) / bpc : 0;
  const ChecksumType type = checksumData.getChecksumType();
  for (long long i = 0; i < windowsRead; i++) {
    if (throttler) throttler(std::min(bpc, nRead - i * bpc));
    if (type == ChecksumType::NONE)  // computeChecksum yields no checksums for NONE: getChecksums().get(0) throws (:297-298)
      throw std::out_of_range("Index: 0, Size: 0");
    if (i == first) {
      const size_t woff = (size_t)(first * bpc), wlen = (size_t)std::min(bpc, nRead - first * bpc);
      const ChecksumData actual = Checksum(type, (int)bpc).computeChecksum(data.data() + woff, wlen);
      throw OzoneChecksumException("Inconsistent read for chunk=" + chunkInfoToString(chunkName, chunkOffset, chunkLen) +
                                   " checksum item " + std::to_string(first) + " expected checksum " +
                                   javaBytesToString(sums[(size_t)first]) + " actual checksum " +
                                   javaBytesToString(actual.getChecksums().at(0)) + " for block " + blockId);
    }
  }
  if (nRead != chunkLen)
    throw OzoneChecksumException("Inconsistent read for chunk=" + chunkName + " expected length=" + std::to_string(chunkLen) +
                                 " actual length=" + std::to_string(nRead) + " for block " + blockId);
  return nRead;
}

}  // namespace detail

inline long long verifyChunkFile(const std::string& blockId, const std::string& chunkName, long long chunkOffset,
                                 long long chunkLen, const ChecksumData& checksumData, const std::string& chunkFile,
                                 ChunkLayOutVersion layout, const std::function<void(long long)>& throttler = {}) {
  const std::vector<ByteString>& sums = checksumData.getChecksums();
  const long long bpc = checksumData.getBytesPerChecksum();
  const std::string data = detail::readChunkWindows(chunkOffset, chunkLen, checksumData, chunkFile, layout);
  const long long nRead = (long long)data.size();
  const long long windowsRead = bpc ? (nRead + bpc - 1) / bpc : 0;
  const ChecksumType type = checksumData.getChecksumType();
  int64_t first = -1;
  if (type != ChecksumType::NONE && windowsRead) {  // all windows hashed and compared by one call
    std::string stored;
    for (long long i = 0; i < windowsRead; i++) stored += sums[(size_t)i];
    const int rc = ozcs_verify((int)type, (uint32_t)bpc, data.data(), (uint64_t)nRead, stored.data(), (uint64_t)windowsRead, 0, &first);
    if (rc != OZCS_OK) throw OzoneChecksumException(std::string(ozcs_last_error()));
  }
  return detail::replayChunk(blockId, chunkName, chunkOffset, chunkLen, checksumData, data, first, throttler);
}

// One chunk of a block as the scrubber sees it: the ChunkInfo record's fields and the file that holds it.
struct ScrubChunk {
  std::string chunkName;
  long long chunkOffset, chunkLen;
  ChecksumData checksumData;
  std::string chunkFile;
};

namespace detail {

inline bool fileExists(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (f) std::fclose(f);
  return f != nullptr;
}

// All chunks (each with the id of its block) read and hashed by ONE ozcs_verify_batch call; then the reference's
// order of events replayed chunk by chunk.  The per-chunk gate of scanData (KeyValueContainerCheck.java:241-256)
// comes first: a chunk whose file does not exist is not read -- at its turn in the replay the block is looked up
// again (blockStillExists; the reference asks the block table, :243-247) and IOException("Missing chunk file
// <path>") is thrown if it is still there, else the chunk is passed over; a chunk whose stored checksum type is
// NONE is passed over too: not read, not throttled, not judged (:252-253).
inline long long verifyChunks(const std::vector<std::pair<std::string, const ScrubChunk*>>& flat, ChunkLayOutVersion layout,
                              const std::function<void(long long)>& throttler,
                              const std::function<bool(const std::string&)>& blockStillExists = {}) {
  enum Gate : char { kVerify, kMissing, kSkipped };
  std::vector<std::string> reads, stored;
  std::vector<Gate> gate;
  std::vector<ozcs_vchunk> v(flat.size());
  for (const auto& bc : flat) {
    const ScrubChunk& c = *bc.second;
    gate.push_back(!fileExists(c.chunkFile) ? kMissing
                   : c.checksumData.getChecksumType() == ChecksumType::NONE ? kSkipped : kVerify);
    if (gate.back() != kVerify) {
      reads.emplace_back();
      stored.emplace_back();
      continue;
    }
    reads.push_back(readChunkWindows(c.chunkOffset, c.chunkLen, c.checksumData, c.chunkFile, layout));
    const long long bpc = c.checksumData.getBytesPerChecksum();
    const long long nwin = bpc ? ((long long)reads.back().size() + bpc - 1) / bpc : 0;
    std::string s;
    for (long long i = 0; i < nwin; i++) s += c.checksumData.getChecksums()[(size_t)i];
    stored.push_back(std::move(s));
  }
  for (size_t i = 0; i < flat.size(); i++) {
    const ScrubChunk& c = *flat[i].second;
    const long long bpc = c.checksumData.getBytesPerChecksum();
    const uint64_t nwin = bpc ? (reads[i].size() + (size_t)bpc - 1) / (size_t)bpc : 0;
    v[i] = ozcs_vchunk{};
    v[i].type = nwin && gate[i] == kVerify ? (int)c.checksumData.getChecksumType() : OZCS_NONE;  // nothing read: only the length check applies
    v[i].bpc = gate[i] == kVerify ? (uint32_t)bpc : 1u;
    v[i].data = reads[i].empty() ? nullptr : reads[i].data();
    v[i].len = reads[i].size();
    v[i].expected = stored[i].data();
    v[i].n_expected = nwin;
  }
  const int rc = flat.empty() ? OZCS_OK : ozcs_verify_batch(v.data(), (uint32_t)v.size());
  long long total = 0;
  for (size_t i = 0; i < flat.size(); i++) {
    const ScrubChunk& c = *flat[i].second;
    if (gate[i] == kSkipped) continue;  // :252-253
    if (gate[i] == kMissing) {          // "concurrent mutation in Block DB? lookup the block again" (:242-251)
      if (!blockStillExists || blockStillExists(flat[i].first)) throw IOException("Missing chunk file " + c.chunkFile);
      continue;
    }
    if (v[i].status != OZCS_OK) {
      if (rc == OZCS_ENODEV || rc == OZCS_EHIP || rc == OZCS_ENOMEM)
        throw GpuUnavailable(v[i].status, std::string("ozcs_verify_batch: ") + ozcs_last_error());
      throw OzoneChecksumException("verify failed for chunk " + c.chunkName + ": status " + std::to_string(v[i].status));
    }
    total += replayChunk(flat[i].first, c.chunkName, c.chunkOffset, c.chunkLen, c.checksumData, reads[i], v[i].first_mismatch, throttler);
  }
  return total;
}

}  // namespace detail

// The `for chunk in block.getChunks()` loop of KeyValueContainerCheck.scanData (:235-257) with ONE ozcs_verify_batch
// call for all chunks instead of one checksum call per window.  All chunks are read and hashed in bulk first; then
// the reference's order of events is replayed chunk by chunk, window by window, so the throttler sees exactly the
// windows the reference would have read up to the failure and the first chunk (in order) that fails raises what
// verifyChunkFile raises for it.  Returns the number of bytes read.
// The loop's gate is the reference's (:241-256): chunks with stored type NONE are passed over unread and unthrottled; a
// missing chunk file throws IOException("Missing chunk file <path>") at its turn unless blockStillExists(blockId)
// says the block has left the table meanwhile (no callback: it is taken to be there).
inline long long verifyBlock(const std::string& blockId, const std::vector<ScrubChunk>& chunks, ChunkLayOutVersion layout,
                             const std::function<void(long long)>& throttler = {},
                             const std::function<bool(const std::string&)>& blockStillExists = {}) {
  std::vector<std::pair<std::string, const ScrubChunk*>> flat;
  for (const ScrubChunk& c : chunks) flat.emplace_back(blockId, &c);
  return detail::verifyChunks(flat, layout, throttler, blockStillExists);
}

// The data scan of one container, scanData (:226-257): every chunk of every block, in order.  Chunks are gathered
// across block boundaries into batches of about batchBytes of stored length, each batch one ozcs_verify_batch call;
// the first failing chunk of the first failing block raises what verifyChunkFile raises for it and nothing behind it
// is throttled (reads run ahead by at most one batch).
struct ScrubBlock {
  std::string blockId;
  std::vector<ScrubChunk> chunks;
};
inline long long verifyContainer(const std::vector<ScrubBlock>& blocks, ChunkLayOutVersion layout,
                                 const std::function<void(long long)>& throttler = {}, long long batchBytes = 1ll << 30,
                                 const std::function<bool(const std::string&)>& blockStillExists = {}) {
  long long total = 0, size = 0;
  std::vector<std::pair<std::string, const ScrubChunk*>> flat;
  for (const ScrubBlock& b : blocks) {
    for (const ScrubChunk& c : b.chunks) {
      flat.emplace_back(b.blockId, &c);
      size += c.chunkLen;
    }
    if (size >= batchBytes) {
      total += detail::verifyChunks(flat, layout, throttler, blockStillExists);
      flat.clear();
      size = 0;
    }
  }
  if (!flat.empty()) total += detail::verifyChunks(flat, layout, throttler, blockStillExists);
  return total;
}

// ---------------------------------------------------------------------------------------------- write path

// The chunk buffers of a BlockOutputStream as pinned pipe slots: allocateBuffer() hands out a pinned buffer
// for the application's bytes (BufferPool.java:74-85), writeChunk() queues copy + checksum + copy-back and
// returns at once (where the reference computes the checksum synchronously, BlockOutputStream.java:645-656),
// and checksumOf() delivers the ChecksumData that goes into the chunk's ChunkInfo before writeChunkAsync.
class PinnedChunkPool {
 public:
  PinnedChunkPool(int device, uint32_t capacity, uint64_t bufferSize, ChecksumType type, int bytesPerChecksum)
      : type_(type), bpc_(bytesPerChecksum), outs_(capacity) {
    const int rc = ozcs_pipe_create(device, capacity, bufferSize, &pipe_);
    if (rc != OZCS_OK) throw GpuUnavailable(rc, std::string("ozcs_pipe_create: ") + ozcs_last_error());
  }
  ~PinnedChunkPool() { ozcs_pipe_destroy(pipe_); }
  PinnedChunkPool(const PinnedChunkPool&) = delete;
  PinnedChunkPool& operator=(const PinnedChunkPool&) = delete;

  uint64_t getBufferSize() const { return ozcs_pipe_slot_bytes(pipe_); }

  // A free pinned buffer and its ticket, or {nullptr, -1} when every buffer is held (the reference blocks in
  // waitOnFlushFutures then; here the caller finishes one of its tickets with checksumOf()).
  std::pair<void*, int> allocateBuffer() {
    void* buf = nullptr;
    int ticket = -1;
    const int rc = ozcs_pipe_acquire(pipe_, &buf, &ticket);
    if (rc == OZCS_EBUSY) return {nullptr, -1};
    if (rc != OZCS_OK) throw GpuUnavailable(rc, std::string("ozcs_pipe_acquire: ") + ozcs_last_error());
    return {buf, ticket};
  }

  // The first `len` bytes of the ticket's buffer are the chunk: start copy + checksum + copy-back.
  void writeChunk(int ticket, uint64_t len) {
    const int t = (int)type_;
    const uint64_t n = type_ == ChecksumType::NONE ? 0 : ozcs_num_checksums(len, (uint32_t)bpc_);
    std::string& o = outs_.at((size_t)ozcs_pipe_ticket_slot(ticket));
    o.assign((size_t)(n * ozcs_digest_size(t)), '\0');
    const int rc = ozcs_pipe_submit_slot(pipe_, ticket, t, (uint32_t)bpc_, len, o.empty() ? nullptr : &o[0], o.size());
    if (rc == OZCS_EINVAL && (t < OZCS_NONE || t > OZCS_MD5)) throw OzoneChecksumException(type_);
    if (rc != OZCS_OK) throw GpuUnavailable(rc, std::string("ozcs_pipe_submit_slot: ") + ozcs_last_error());
  }

  // Blocks until the ticket's digests are back; the buffer returns to the pool (releaseBuffer).
  ChecksumData checksumOf(int ticket) {
    uint64_t n = 0;
    const int rc = ozcs_pipe_wait(pipe_, ticket, &n);
    if (rc != OZCS_OK) throw GpuUnavailable(rc, std::string("ozcs_pipe_wait: ") + ozcs_last_error());
    const std::string& raw = outs_.at((size_t)ozcs_pipe_ticket_slot(ticket));
    const size_t ds = ozcs_digest_size((int)type_);
    std::vector<ByteString> v;
    for (uint64_t i = 0; i < n; i++) v.emplace_back(raw.substr((size_t)i * ds, ds));
    return ChecksumData(type_, bpc_, std::move(v));
  }

 private:
  ozcs_pipe* pipe_ = nullptr;
  ChecksumType type_;
  int bpc_;
  std::vector<std::string> outs_;  // per slot: where the pipe delivers the digests of the slot's current ticket
};

// The ChunkInfo ContainerProtocolCalls.writeSmallFile builds for a small file that travels in one PutSmallFile
// request (hadoop-hdds/common/src/main/java/org/apache/hadoop/hdds/scm/storage/ContainerProtocolCalls.java:334-345):
// name "<localID>_chunk", offset 0, the whole length, the metadata pair OverWriteRequested=true, and CRC32 checksums
// over 256-byte windows whatever the client configuration says.
inline ChunkInfo writeSmallFileChunkInfo(long long localID, const void* data, size_t len) {
  ChunkInfo info(std::to_string(localID) + "_chunk", 0, (long long)len);
  info.addMetadata("OverWriteRequested", "true");
  info.setChecksumData(Checksum(ChecksumType::CRC32, 256).computeChecksum(data, len));
  return info;
}

// BlockOutputStream's data path around the checksum call
// (hadoop-hdds/client/src/main/java/org/apache/hadoop/hdds/scm/storage/BlockOutputStream.java):
//   write / writeChunkIfNeeded   :225-266   bytes go into the current chunk buffer; a full buffer becomes a chunk
//   writeChunkToContainer        :645-690   offset bookkeeping, computeChecksum(chunk), the ChunkInfo record (name
//                                           "<localID>_chunk_<n>", offset, len, checksumData), writeChunkAsync
//   flush / close                :405-473   the partly filled buffer goes out as a last, shorter chunk
// The chunk buffers are the pinned slots of an ozcs pipe (in place of BufferPool's heap buffers): a full buffer is
// submitted and the writer goes on filling the next one while the first is copied and hashed; chunks are handed to
// writeChunkAsync strictly in order, each as soon as its digests have arrived.  What comes out -- the ChunkInfo
// records and the bytes -- is what the reference would ship.
class BlockOutputStream {
 public:
  using WriteChunkAsync = std::function<void(const ChunkInfo&, const char* data, size_t len)>;

  // capacity = streamBufferMaxSize / streamBufferSize, the number of chunk buffers
  // (BlockOutputStreamEntryPool.java:91-96).
  //
  // safeByteStringConversion: the reference turns the chunk buffer into the request's ByteString before it computes
  // the checksum (chunk.toByteString(bufferPool.byteStringConversion()), BlockOutputStream.java:648-649) -- a copy by
  // default, a zero-copy wrap when unsafe conversion is configured.  true (default): the chunk's bytes are copied out
  // of the pinned slot on submission and the pipe runs in in-order mode, so a slot goes back to the pool the moment
  // the GPU is done with it (a CRC chunk does not hold its slot behind an older SHA-256 / MD5 chunk) while ChunkInfo
  // records still leave in write order.  false: the slot itself is what is shipped and stays the chunk's until then.
  BlockOutputStream(long long localID, WriteChunkAsync writeChunkAsync, const OzoneClientConfig& config = OzoneClientConfig(),
                    uint64_t streamBufferSize = 4u << 20, uint32_t capacity = 8, int device = 0,
                    bool safeByteStringConversion = true)
      : localID_(localID), sink_(std::move(writeChunkAsync)), type_(config.getChecksumType()),
        bpc_(config.getBytesPerChecksum()), bufferSize_(streamBufferSize), safe_(safeByteStringConversion) {
    const int t = (int)type_;
    if (t < OZCS_NONE || t > OZCS_MD5) throw OzoneChecksumException(type_);
    int rc = ozcs_pipe_create(device, capacity, streamBufferSize, &pipe_);
    if (rc != OZCS_OK) throw GpuUnavailable(rc, std::string("ozcs_pipe_create: ") + ozcs_last_error());
    if (safe_ && (rc = ozcs_pipe_set_in_order(pipe_, 1)) != OZCS_OK) {
      ozcs_pipe_destroy(pipe_);
      throw GpuUnavailable(rc, std::string("ozcs_pipe_set_in_order: ") + ozcs_last_error());
    }
  }
  ~BlockOutputStream() {
    try {
      close();
    } catch (...) {
    }
  }
  BlockOutputStream(const BlockOutputStream&) = delete;
  BlockOutputStream& operator=(const BlockOutputStream&) = delete;

  // write(byte[], off, len), :241-266
  void write(const void* b, size_t len) {
    checkOpen();
    const char* src = static_cast<const char*>(b);
    while (len > 0) {
      allocateNewBufferIfNeeded();
      const size_t n = std::min<size_t>(bufferSize_ - pos_, len);
      std::memcpy(static_cast<char*>(cur_) + pos_, src, n);
      pos_ += n;
      if (pos_ == bufferSize_) writeChunk();
      src += n;
      len -= n;
      writtenDataLength_ += (long long)n;
    }
  }
  // The data buffered so far goes out (handleFlush: a partly filled buffer becomes a chunk of its own) and every
  // submitted chunk is shipped.
  void flush() {
    checkOpen();
    if (cur_ && pos_ > 0) writeChunk();
    while (!pending_.empty()) shipOldest();
  }
  void close() {
    if (closed_) return;
    struct Finally {
      BlockOutputStream* s;
      ~Finally() {
        s->closed_ = true;
        if (s->cur_) {  // acquired, never filled: give it back
          (void)ozcs_pipe_release(s->pipe_, s->curTicket_);
          s->cur_ = nullptr;
        }
        ozcs_pipe_destroy(s->pipe_);
        s->pipe_ = nullptr;
      }
    } fin{this};
    flush();
  }
  long long getWrittenDataLength() const { return writtenDataLength_; }
  // containerBlockData's chunk list, in order
  const std::vector<ChunkInfo>& getChunks() const { return chunks_; }

 private:
  struct Pending {
    int ticket;
    ChunkInfo info;
    const char* data;
    std::string sums;
    uint64_t nsums;
    std::string copy;  // safe conversion: the chunk's bytes, copied out of the slot on submission
  };
  void checkOpen() const {
    if (closed_) throw std::runtime_error("BlockOutputStream has been closed.");
  }
  void allocateNewBufferIfNeeded() {  // :281-286 over BufferPool.allocateBuffer
    if (cur_) return;
    int rc;
    // in-order pipe: block for a slot (any completion gives one back); otherwise a slot is free only once its chunk
    // has been shipped
    while ((rc = safe_ ? ozcs_pipe_acquire_wait(pipe_, &cur_, &curTicket_) : ozcs_pipe_acquire(pipe_, &cur_, &curTicket_)) == OZCS_EBUSY) {
      if (pending_.empty()) break;  // nothing of ours is in flight, yet no buffer is to be had: report it, do not spin
      shipOldest();  // every buffer is in flight: the reference waits for acknowledgements here (handleFullBuffer)
    }
    if (rc != OZCS_OK) {
      cur_ = nullptr;
      throw GpuUnavailable(rc, std::string("ozcs_pipe_acquire: ") + ozcs_last_error());
    }
    pos_ = 0;
    if (safe_)
      while (!pending_.empty() && shipOldest(false)) {
      }
  }
  void writeChunk() {  // writeChunk + writeChunkToContainer, :495-507, :645-690
    const uint64_t n = pos_;
    const long long offset = chunkOffset_;
    chunkOffset_ += (long long)n;
    chunkIndex_++;
    const int t = (int)type_;
    const uint64_t nsums = type_ == ChecksumType::NONE ? 0 : ozcs_num_checksums(n, (uint32_t)bpc_);
    pending_.push_back(Pending{curTicket_, ChunkInfo(std::to_string(localID_) + "_chunk_" + std::to_string(chunkIndex_), offset, (long long)n),
                               static_cast<const char*>(cur_), std::string((size_t)(nsums * ozcs_digest_size(t)), '\0'), nsums,
                               safe_ ? std::string(static_cast<const char*>(cur_), (size_t)n) : std::string()});
    Pending& p = pending_.back();  // std::deque: the digest buffer does not move while the kernel writes it
    // whatever the outcome, the buffer is no longer this stream's current one: a rejected submission has released
    // the slot, an accepted one owns it
    cur_ = nullptr;
    pos_ = 0;
    const int rc = ozcs_pipe_submit_slot(pipe_, p.ticket, t, (uint32_t)bpc_, n, p.sums.empty() ? nullptr : &p.sums[0], p.sums.size());
    if (rc != OZCS_OK) {
      pending_.pop_back();
      throw GpuUnavailable(rc, std::string("ozcs_pipe_submit_slot: ") + ozcs_last_error());
    }
  }
  // Ships the oldest submitted chunk once its digests are there; block = false (in-order pipe only) returns false
  // instead of waiting for them.
  bool shipOldest(bool block = true) {
    if (pending_.empty()) throw std::runtime_error("no chunk is in flight");
    Pending& p = pending_.front();
    uint64_t got = 0;
    if (safe_) {
      int done = -1, status = OZCS_OK;
      const int rc = ozcs_pipe_next_in_order(pipe_, block ? 1 : 0, &done, &status, &got);
      if (rc == OZCS_EBUSY && !block) return false;
      if (rc != OZCS_OK) throw GpuUnavailable(rc, std::string("ozcs_pipe_next_in_order: ") + ozcs_last_error());
      if (status != OZCS_OK) throw GpuUnavailable(status, "a chunk's checksum failed on the GPU");
      if (done != p.ticket) throw std::logic_error("the pipe delivered out of order");
    } else {
      const int rc = ozcs_pipe_wait(pipe_, p.ticket, &got);
      if (rc != OZCS_OK) throw GpuUnavailable(rc, std::string("ozcs_pipe_wait: ") + ozcs_last_error());
    }
    const size_t ds = ozcs_digest_size((int)type_);
    std::vector<ByteString> sums;
    for (uint64_t i = 0; i < got; i++) sums.emplace_back(p.sums.substr((size_t)i * ds, ds));
    p.info.setChecksumData(ChecksumData(type_, bpc_, std::move(sums)));
    chunks_.push_back(p.info);
    // unsafe conversion: the slot is free again, but only this stream acquires from the pipe, so its bytes stay
    // until the next write
    sink_(chunks_.back(), safe_ ? p.copy.data() : p.data, (size_t)p.info.getLen());
    pending_.pop_front();
    return true;
  }

  long long localID_;
  WriteChunkAsync sink_;
  ChecksumType type_;
  int bpc_;
  uint64_t bufferSize_;
  bool safe_;
  ozcs_pipe* pipe_ = nullptr;
  void* cur_ = nullptr;  // the buffer being filled and its ticket
  int curTicket_ = -1;
  uint64_t pos_ = 0;
  long long chunkOffset_ = 0, writtenDataLength_ = 0;
  int chunkIndex_ = 0;
  bool closed_ = false;
  std::deque<Pending> pending_;  // submitted chunks not yet shipped
  std::vector<ChunkInfo> chunks_;
};

}  // namespace ozone
