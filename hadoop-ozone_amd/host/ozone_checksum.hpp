// Host-side C++ mirror of the reference's checksum interface, over the C ABI (include/ozone_checksum.h).
//
// Same names, argument meaning and error behaviour as
//   Checksum                 hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/common/Checksum.java:43-299
//   ChecksumData             .../ozone/common/ChecksumData.java:34-192
//   OzoneChecksumException   .../ozone/common/OzoneChecksumException.java:29-66
//   ChecksumType             hadoop-hdds/interface-client/src/main/proto/DatanodeClientProtocol.proto:384-390
// (ChunkBuffer and its implementations: ozone_chunk_buffer.hpp.)
// Every digest is computed by the GPU library; nothing here hashes on the CPU.  Header-only; link libozcs.so.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ozone_checksum.h"
#include "ozone_chunk_buffer.hpp"

namespace ozone {

enum class ChecksumType : int { NONE = 1, CRC32 = 2, CRC32C = 3, SHA256 = 4, MD5 = 5 };

inline const char* name(ChecksumType t) {
  switch (t) {
    case ChecksumType::NONE: return "NONE";
    case ChecksumType::CRC32: return "CRC32";
    case ChecksumType::CRC32C: return "CRC32C";
    case ChecksumType::SHA256: return "SHA256";
    case ChecksumType::MD5: return "MD5";
  }
  return "?";
}

using ByteString = std::string;  // immutable byte sequence, as com.google.protobuf.ByteString

// java.io.IOException, for the failures the callers report as such (a missing chunk file in the scrubber,
// KeyValueContainerCheck.java:249-250) and as the base of OzoneChecksumException.
class IOException : public std::runtime_error {
 public:
  explicit IOException(const std::string& message) : std::runtime_error(message) {}
};

// An IOException in the reference (callers fail over to the next replica on it).
class OzoneChecksumException : public IOException {
 public:
  explicit OzoneChecksumException(int index)  // OzoneChecksumException.java:35-37
      : IOException("Checksum mismatch at index " + std::to_string(index)), index_(index) {}
  explicit OzoneChecksumException(ChecksumType unrecognized)  // :43-47
      : IOException(std::string("Unrecognized ChecksumType: ") + name(unrecognized)), index_(-1) {}
  explicit OzoneChecksumException(const std::string& message) : IOException(message), index_(-1) {}  // :63-65
  int index() const { return index_; }

 private:
  int index_;
};

// Raised when the GPU library itself cannot serve the call (no device, HIP failure): the caller keeps its
// CPU path, as with NativeCRC32Wrapper.isAvailable() == false.
class GpuUnavailable : public std::runtime_error {
 public:
  GpuUnavailable(int code, const std::string& msg) : std::runtime_error(msg), code(code) {}
  int code;
};

class ChecksumData {
 public:
  ChecksumData(ChecksumType type, int bytesPerChecksum, std::vector<ByteString> checksums = {})
      : type_(type), bytesPerChecksum_(bytesPerChecksum), checksums_(std::move(checksums)) {}

  ChecksumType getChecksumType() const { return type_; }
  int getBytesPerChecksum() const { return bytesPerChecksum_; }
  const std::vector<ByteString>& getChecksums() const { return checksums_; }

  // ChecksumData.java:117-149
  bool verifyChecksumDataMatches(const ChecksumData& that, int startIndex) const {
    if (checksums_.empty()) throw OzoneChecksumException("Original checksumData has no checksums");
    if (that.checksums_.empty()) throw OzoneChecksumException("Computed checksumData has no checksums");
    const size_t n = that.checksums_.size();
    for (size_t index = 0; index < n; index++) {
      const long long at = (long long)startIndex + (long long)index;
      if (at < 0 || (size_t)at >= checksums_.size())
        throw OzoneChecksumException("Computed checksum has " + std::to_string(n) +
                                     " number of checksums. Original checksum has " +
                                     std::to_string((long long)checksums_.size() - startIndex) +
                                     " number of checksums starting from index " + std::to_string(startIndex));
      if (checksums_[(size_t)at] != that.checksums_[index]) throw OzoneChecksumException((int)index);
    }
    return true;
  }

  bool operator==(const ChecksumData& o) const {  // :156-182
    return type_ == o.type_ && bytesPerChecksum_ == o.bytesPerChecksum_ && checksums_ == o.checksums_;
  }
  bool operator!=(const ChecksumData& o) const { return !(*this == o); }

  // Serialised ContainerProtos.ChecksumData (DatanodeClientProtocol.proto:378-382), ChecksumData.java:78-87.
  ByteString getProtoBufMessage() const {
    ByteString out;
    out.push_back(0x08); putVarint(out, (uint64_t)(int)type_);
    out.push_back(0x10); putVarint(out, (uint64_t)(uint32_t)bytesPerChecksum_);
    for (const ByteString& c : checksums_) { out.push_back(0x1a); putVarint(out, c.size()); out += c; }
    return out;
  }

  // ChecksumData.getFromProtoBuf (ChecksumData.java:94-102) from the serialised message: fields 1 (type,
  // varint), 2 (bytesPerChecksum, varint) and 3 (checksums, repeated bytes); other fields are skipped.
  static ChecksumData getFromProtoBuf(const ByteString& msg) {
    int type = 0;
    uint64_t bpc = 0;
    std::vector<ByteString> sums;
    size_t i = 0;
    while (i < msg.size()) {
      const uint64_t key = getVarint(msg, i);
      const uint64_t field = key >> 3, wire = key & 7;
      if (wire == 0) {
        const uint64_t v = getVarint(msg, i);
        if (field == 1) type = (int)v;
        if (field == 2) bpc = v;
      } else if (wire == 2) {
        const uint64_t n = getVarint(msg, i);
        if (n > msg.size() - i) throw std::invalid_argument("ChecksumData: truncated message");
        if (field == 3) sums.emplace_back(msg.substr(i, (size_t)n));
        i += (size_t)n;
      } else if (wire == 1) {
        i += 8;
      } else if (wire == 5) {
        i += 4;
      } else {
        throw std::invalid_argument("ChecksumData: unsupported wire type");
      }
    }
    if (i != msg.size()) throw std::invalid_argument("ChecksumData: truncated message");
    return ChecksumData((ChecksumType)type, (int)bpc, std::move(sums));
  }

 private:
  static void putVarint(ByteString& o, uint64_t v) {
    while (v >= 0x80) { o.push_back((char)(v | 0x80)); v >>= 7; }
    o.push_back((char)v);
  }
  static uint64_t getVarint(const ByteString& m, size_t& i) {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (i >= m.size()) throw std::invalid_argument("ChecksumData: truncated varint");
      const uint8_t b = (uint8_t)m[i++];
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    throw std::invalid_argument("ChecksumData: varint too long");
  }
  ChecksumType type_;
  int bytesPerChecksum_;
  std::vector<ByteString> checksums_;
};

// Which elements of a buffer list ChunkBufferImplWithByteBufferList.iterate visits, in order, given the elements'
// lengths (ChunkBufferImplWithByteBufferList.java:59-78, 192-201).  iterate().next() calls findCurrent() every
// time, and findCurrent() adds the limits of the consumed buffers to limitPrecedingCurrent without resetting it:
// position() runs ahead of the bytes really consumed and hasRemaining() turns false early -- of four equal
// elements three are visited, of 256 only 24.
inline std::vector<size_t> referenceListWindows(const std::vector<size_t>& lengths) {
  const size_t n = lengths.size();
  std::vector<size_t> visited;
  if (n == 0) return visited;
  long long limit = 0;
  for (size_t l : lengths) limit += (long long)l;
  std::vector<size_t> pos(n, 0);
  long long preceding = 0;
  size_t cur = 0;
  auto findCurrent = [&]() {  // :59-78
    bool found = false;
    for (size_t i = 0; i < n && !found; i++) {
      if (pos[i] < lengths[i]) { found = true; cur = i; }
      else preceding += (long long)lengths[i];
    }
    if (!found) { cur = n - 1; preceding -= (long long)lengths[cur]; }
  };
  findCurrent();  // the constructor's call, :56
  while (limit - (preceding + (long long)pos[cur]) > 0) {  // hasNext() = hasRemaining()
    findCurrent();
    visited.push_back(cur);
    pos[cur] = lengths[cur];
  }
  return visited;
}

// Checksum.java:43-299.
class Checksum {
 public:
  Checksum(ChecksumType type, int bytesPerChecksum) : checksumType_(type), bytesPerChecksum_(bytesPerChecksum) {}

  static ByteString int2ByteString(int32_t n) {  // Checksum.java:62-64
    const uint32_t u = (uint32_t)n;
    return ByteString{(char)(u >> 24), (char)(u >> 16), (char)(u >> 8), (char)u};
  }

  // computeChecksum(byte[] / byte[],off,len / ByteBuffer): Checksum.java:119-146
  ChecksumData computeChecksum(const void* data, size_t len) const {
    if (checksumType_ == ChecksumType::NONE) return ChecksumData(checksumType_, bytesPerChecksum_);  // :157-160
    const int t = (int)checksumType_;
    if (t < OZCS_NONE || t > OZCS_MD5) throw OzoneChecksumException(checksumType_);  // :163-167
    if (bytesPerChecksum_ <= 0) throw std::invalid_argument("bytesPerChecksum must be positive");
    const size_t ds = ozcs_digest_size(t);
    const uint64_t n = ozcs_num_checksums(len, (uint32_t)bytesPerChecksum_);
    std::string raw(n * ds, '\0');
    uint64_t got = 0;
    check(ozcs_compute(t, (uint32_t)bytesPerChecksum_, data, len, raw.empty() ? nullptr : &raw[0], raw.size(), &got));
    return ChecksumData(checksumType_, bytesPerChecksum_, split(raw, ds, got));
  }

  // computeChecksum(List<ByteString>) / computeChecksum(ChunkBuffer over a list): one element per window; an
  // element longer than bytesPerChecksum contributes only its first bytesPerChecksum bytes
  // (Checksum.java:148-153, 186-198).  By default (referenceListQuirk = true) the result is the reference's bit for
  // bit: its list iterator stops early on lists of four or more elements, so only the elements
  // referenceListWindows() names are checksummed -- the same default as ChunkBuffer::wrap(list), as both overloads
  // share one iterator in the reference, and what a Checksum patched to call this library yields (the patch keeps
  // the reference's iteration).  false checksums every element: the documented intent (Checksum.java:279-281).
  ChecksumData computeChecksum(const std::vector<ByteString>& all, bool referenceListQuirk = true) const {
    std::vector<ByteString> subset;
    if (referenceListQuirk) {
      std::vector<size_t> lengths;
      for (const ByteString& b : all) lengths.push_back(b.size());
      for (size_t i : referenceListWindows(lengths)) subset.push_back(all[i]);
    }
    const std::vector<ByteString>& byteStrings = referenceListQuirk ? subset : all;
    std::vector<std::pair<const void*, size_t>> windows;
    for (const ByteString& b : byteStrings)
      if (!b.empty()) windows.emplace_back(b.data(), b.size());  // empty elements yield nothing to iterate over
    return computeWindows(windows);
  }

  // computeChecksum(ChunkBuffer), Checksum.java:155-177: one digest per buffer data.iterate(bytesPerChecksum)
  // yields, each over at most bytesPerChecksum bytes from the buffer's position (:186-198).  The iteration runs
  // first (it moves the ChunkBuffer's position as in the reference); then all windows go to the GPU in one call.
  // A buffer with nothing remaining -- IncrementalChunkBuffer hands out its buffers as they are, consumed ones
  // included -- gets the digest of the empty message, as the reference's function yields.
  ChecksumData computeChecksum(ChunkBuffer& data) const {
    if (checksumType_ == ChecksumType::NONE) return ChecksumData(checksumType_, bytesPerChecksum_);
    if (bytesPerChecksum_ <= 0) throw std::invalid_argument("bytesPerChecksum must be positive");
    std::vector<ByteBufferPtr> keep;
    std::vector<std::pair<const void*, size_t>> windows;
    for (ByteBufferIterator it = data.iterate(bytesPerChecksum_); it.hasNext();) {
      keep.push_back(it.next());
      windows.emplace_back(keep.back()->data(), (size_t)keep.back()->remaining());
    }
    return computeWindows(windows);
  }
  // computeChecksum(ByteBuffer), :140-146.
  ChecksumData computeChecksum(const ByteBufferPtr& data) const {
    ChunkBufferPtr wrapped = ChunkBuffer::wrap(data->asReadOnlyBuffer());
    return computeChecksum(*wrapped);
  }

  // verifyChecksum(ByteString, ChecksumData, int) / (byte[], ChecksumData): Checksum.java:210-250.
  static bool verifyChecksum(const void* data, size_t len, const ChecksumData& checksumData, int startIndex = 0) {
    const ChecksumType type = checksumData.getChecksumType();
    if (type == ChecksumType::NONE) return true;  // :241-244
    const int t = (int)type;
    if (t < OZCS_NONE || t > OZCS_MD5) throw OzoneChecksumException(type);
    std::string stored;
    for (const ByteString& c : checksumData.getChecksums()) stored += c;
    const size_t ds = ozcs_digest_size(t);
    int64_t first = -1;
    if (startIndex < 0) throw std::out_of_range("startIndex");
    const int rc = ozcs_verify(t, (uint32_t)checksumData.getBytesPerChecksum(), data, len, stored.data(),
                               ds ? stored.size() / ds : 0, (uint64_t)startIndex, &first);
    if (rc == OZCS_EEMPTY || rc == OZCS_ERANGE) throw OzoneChecksumException(std::string(ozcs_last_error()));
    check(rc);
    if (first >= 0) throw OzoneChecksumException((int)first);
    return true;
  }

  // verifyChecksum(List<ByteString>, ChecksumData, int, boolean): Checksum.java:265-290.
  static bool verifyChecksum(const std::vector<ByteString>& byteStrings, const ChecksumData& checksumData,
                             int startIndex, bool isSingleByteString, bool referenceListQuirk = true) {
    if (checksumData.getChecksumType() == ChecksumType::NONE) return true;  // :269-272
    if (isSingleByteString)
      return verifyChecksum(byteStrings.at(0).data(), byteStrings.at(0).size(), checksumData, startIndex);  // :274-277
    const ChecksumData computed =
        Checksum(checksumData.getChecksumType(), checksumData.getBytesPerChecksum()).computeChecksum(byteStrings, referenceListQuirk);
    return checksumData.verifyChecksumDataMatches(computed, startIndex);  // :282-289
  }

 private:
  static void check(int rc) {
    if (rc == OZCS_OK) return;
    throw GpuUnavailable(rc, std::string("ozcs error ") + std::to_string(rc) + ": " + ozcs_last_error());
  }
  // One digest per window, each clamped to bytesPerChecksum bytes; all filled windows in one batched call.
  ChecksumData computeWindows(const std::vector<std::pair<const void*, size_t>>& windows) const {
    if (checksumType_ == ChecksumType::NONE) return ChecksumData(checksumType_, bytesPerChecksum_);
    const int t = (int)checksumType_;
    if (t < OZCS_NONE || t > OZCS_MD5) throw OzoneChecksumException(checksumType_);
    if (bytesPerChecksum_ <= 0) throw std::invalid_argument("bytesPerChecksum must be positive");
    const size_t ds = ozcs_digest_size(t);
    std::vector<ByteString> outs(windows.size(), ByteString(ds, '\0'));
    std::vector<ozcs_chunk> jobs;
    for (size_t i = 0; i < windows.size(); i++) {
      if (windows[i].second == 0) {
        outs[i] = emptyDigest(checksumType_);
        continue;
      }
      ozcs_chunk c{};
      c.data = windows[i].first;
      c.len = windows[i].second < (size_t)bytesPerChecksum_ ? windows[i].second : (size_t)bytesPerChecksum_;
      c.sums = &outs[i][0];
      c.sums_cap = ds;
      c.bpc = (uint32_t)bytesPerChecksum_;
      c.type = t;
      jobs.push_back(c);
    }
    if (!jobs.empty()) check(ozcs_compute_batch(jobs.data(), (uint32_t)jobs.size()));
    return ChecksumData(checksumType_, bytesPerChecksum_, outs);
  }
  // What Algorithm.newChecksumFunction() returns for a buffer with nothing remaining: the CRC register after
  // reset() (Checksum.java:70-74), MessageDigest.digest() of no input (:55-59).
  static ByteString emptyDigest(ChecksumType type) {
    static const unsigned char sha[32] = {0xe3, 0xb0, 0xc4, 0x42, 0x98, 0xfc, 0x1c, 0x14, 0x9a, 0xfb, 0xf4, 0xc8, 0x99, 0x6f, 0xb9, 0x24,
                                          0x27, 0xae, 0x41, 0xe4, 0x64, 0x9b, 0x93, 0x4c, 0xa4, 0x95, 0x99, 0x1b, 0x78, 0x52, 0xb8, 0x55};
    static const unsigned char md5[16] = {0xd4, 0x1d, 0x8c, 0xd9, 0x8f, 0x00, 0xb2, 0x04, 0xe9, 0x80, 0x09, 0x98, 0xec, 0xf8, 0x42, 0x7e};
    switch (type) {
      case ChecksumType::SHA256: return ByteString(reinterpret_cast<const char*>(sha), 32);
      case ChecksumType::MD5: return ByteString(reinterpret_cast<const char*>(md5), 16);
      default: return ByteString(4, '\0');
    }
  }
  static std::vector<ByteString> split(const std::string& raw, size_t ds, uint64_t n) {
    std::vector<ByteString> v;
    v.reserve(n);
    for (uint64_t i = 0; i < n; i++) v.emplace_back(raw.substr(i * ds, ds));
    return v;
  }
  ChecksumType checksumType_;
  int bytesPerChecksum_;
};

// The record that carries a chunk's ChecksumData from the client to the datanode and onto its disk
// (hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/container/common/helpers/ChunkInfo.java:34-196); wire form
// ContainerProtos.ChunkInfo (DatanodeClientProtocol.proto:366-372: chunkName = 1, offset = 2, len = 3, repeated
// KeyValue metadata = 4, checksumData = 5; KeyValue :159-162).
class ChunkInfo {
 public:
  ChunkInfo(std::string chunkName, long long offset, long long len)
      : chunkName_(std::move(chunkName)), offset_(offset), len_(len), checksumData_(ChecksumType::NONE, 0), hasChecksum_(false) {}

  void addMetadata(const std::string& key, const std::string& value) {  // ChunkInfo.java:72-79
    if (!metadata_.emplace(key, value).second) throw std::runtime_error("This key already exists. Key " + key);
  }
  const std::string& getChunkName() const { return chunkName_; }
  long long getOffset() const { return offset_; }
  long long getLen() const { return len_; }
  const ChecksumData& getChecksumData() const { return checksumData_; }
  void setChecksumData(const ChecksumData& c) { checksumData_ = c; hasChecksum_ = true; }
  bool hasChecksumData() const { return hasChecksum_; }  // the Java field is null until setChecksumData
  // ChunkInfo.java:41-44, 190-196: set for older clients, whose ReadChunk responses hold all data in one buffer (V0)
  void setReadDataIntoSingleBuffer(bool v) { readDataIntoSingleBuffer_ = v; }
  bool isReadDataIntoSingleBuffer() const { return readDataIntoSingleBuffer_; }
  const std::map<std::string, std::string>& getMetadata() const { return metadata_; }

  // ChunkInfo.java:108-133: fields in number order, metadata in key order (a TreeMap), a missing ChecksumData
  // written as the NONE message (Checksum.getNoChecksumDataProto, Checksum.java:295-298).
  ByteString getProtoBufMessage() const {
    ByteString out;
    putBytes(out, 0x0a, chunkName_);
    out.push_back(0x10); putVarint(out, (uint64_t)offset_);
    out.push_back(0x18); putVarint(out, (uint64_t)len_);
    for (const auto& kv : metadata_) {
      ByteString m;
      putBytes(m, 0x0a, kv.first);
      putBytes(m, 0x12, kv.second);
      putBytes(out, 0x22, m);
    }
    putBytes(out, 0x2a, (hasChecksum_ ? checksumData_ : ChecksumData(ChecksumType::NONE, 0)).getProtoBufMessage());
    return out;
  }

  static ChunkInfo getFromProtoBuf(const ByteString& msg) {  // ChunkInfo.java:88-103
    std::string name;
    uint64_t off = 0, len = 0;
    bool hasName = false, hasOff = false, hasLen = false, hasCd = false;
    std::vector<std::pair<std::string, std::string>> meta;
    ChecksumData cd(ChecksumType::NONE, 0);
    size_t i = 0;
    while (i < msg.size()) {
      const uint64_t key = getVarint(msg, i);
      const uint64_t field = key >> 3, wire = key & 7;
      if (wire == 0) {
        const uint64_t v = getVarint(msg, i);
        if (field == 2) { off = v; hasOff = true; }
        if (field == 3) { len = v; hasLen = true; }
      } else if (wire == 2) {
        const uint64_t n = getVarint(msg, i);
        if (n > msg.size() - i) throw std::invalid_argument("ChunkInfo: truncated message");
        const ByteString body = msg.substr(i, (size_t)n);
        i += (size_t)n;
        if (field == 1) { name = body; hasName = true; }
        if (field == 4) {
          std::string k, v;
          size_t j = 0;
          while (j < body.size()) {
            const uint64_t kk = getVarint(body, j), sz = getVarint(body, j);
            if (sz > body.size() - j) throw std::invalid_argument("ChunkInfo: truncated KeyValue");
            ((kk >> 3) == 1 ? k : v) = body.substr(j, (size_t)sz);
            j += (size_t)sz;
          }
          meta.emplace_back(k, v);
        }
        if (field == 5) { cd = ChecksumData::getFromProtoBuf(body); hasCd = true; }
      } else {
        throw std::invalid_argument("ChunkInfo: unsupported wire type");
      }
    }
    if (!(hasName && hasOff && hasLen && hasCd)) throw std::invalid_argument("ChunkInfo: required field missing");
    ChunkInfo info(name, (long long)off, (long long)len);
    for (const auto& kv : meta) info.addMetadata(kv.first, kv.second);
    info.setChecksumData(cd);
    return info;
  }

  // ChunkInfo.toString() (:182-188; the quote after chunkName= is not closed there either)
  std::string toString() const {
    return "ChunkInfo{chunkName='" + chunkName_ + ", offset=" + std::to_string(offset_) + ", len=" + std::to_string(len_) + "}";
  }

 private:
  static void putVarint(ByteString& o, uint64_t v) {
    while (v >= 0x80) { o.push_back((char)(v | 0x80)); v >>= 7; }
    o.push_back((char)v);
  }
  static void putBytes(ByteString& o, char tag, const ByteString& b) { o.push_back(tag); putVarint(o, b.size()); o += b; }
  static uint64_t getVarint(const ByteString& m, size_t& i) {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (i >= m.size()) throw std::invalid_argument("ChunkInfo: truncated varint");
      const uint8_t b = (uint8_t)m[i++];
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    throw std::invalid_argument("ChunkInfo: varint too long");
  }
  std::string chunkName_;
  long long offset_, len_;
  ChecksumData checksumData_;
  bool hasChecksum_;
  bool readDataIntoSingleBuffer_ = false;
  std::map<std::string, std::string> metadata_;
};

// The three checksum keys of the client configuration group "ozone.client" (OzoneClientConfig.java:103-125) with
// the validation of :143-151: checksum.type defaults to CRC32, bytes.per.checksum to 1MB and is raised to 16 KiB
// (OzoneConfigKeys.java:340) with a warning when set lower, verify.checksum defaults to true.
class OzoneClientConfig {
 public:
  static constexpr int OZONE_CLIENT_BYTES_PER_CHECKSUM_MIN_SIZE = 16 * 1024;

  explicit OzoneClientConfig(const std::map<std::string, std::string>& conf = {}) {
    auto get = [&](const char* key, const char* dflt) {
      const auto it = conf.find(std::string("ozone.client.") + key);
      return it == conf.end() ? std::string(dflt) : it->second;
    };
    checksumType_ = get("checksum.type", "CRC32");
    bytesPerChecksum_ = (int)parseSize(get("bytes.per.checksum", "1MB"));
    checksumVerify_ = lower(get("verify.checksum", "true")) == "true";
    if (bytesPerChecksum_ < OZONE_CLIENT_BYTES_PER_CHECKSUM_MIN_SIZE) {
      char w[200];
      std::snprintf(w, sizeof w, "The checksum size (%d) is not allowed to be less than the minimum size (%d), "
                    "resetting to the minimum size.", bytesPerChecksum_, OZONE_CLIENT_BYTES_PER_CHECKSUM_MIN_SIZE);
      warning_ = w;
      std::fprintf(stderr, "WARN OzoneClientConfig: %s\n", w);
      bytesPerChecksum_ = OZONE_CLIENT_BYTES_PER_CHECKSUM_MIN_SIZE;
    }
  }

  // ChecksumType.valueOf(checksumType): an unknown name is an error (OzoneClientConfig.java:199-201)
  ChecksumType getChecksumType() const {
    for (ChecksumType t : {ChecksumType::NONE, ChecksumType::CRC32, ChecksumType::CRC32C, ChecksumType::SHA256, ChecksumType::MD5})
      if (checksumType_ == name(t)) return t;
    throw std::invalid_argument("No enum constant ChecksumType." + checksumType_);
  }
  void setChecksumType(ChecksumType t) { checksumType_ = name(t); }
  int getBytesPerChecksum() const { return bytesPerChecksum_; }
  void setBytesPerChecksum(int n) { bytesPerChecksum_ = n; }  // the setter does not clamp, only validation does
  bool isChecksumVerify() const { return checksumVerify_; }
  void setChecksumVerify(bool v) { checksumVerify_ = v; }
  const std::string& validationWarning() const { return warning_; }
  // what BlockOutputStream builds once per stream (BlockOutputStream.java:174-175)
  Checksum newChecksum() const { return Checksum(getChecksumType(), bytesPerChecksum_); }

 private:
  static std::string lower(std::string s) {
    for (char& c : s) c = (char)std::tolower((unsigned char)c);
    return s;
  }
  // ConfigType.SIZE: a number with an optional B / KB / MB / GB unit (binary multiples)
  static long long parseSize(const std::string& text) {
    std::string t;
    for (char c : text) if (c != ' ') t.push_back((char)std::toupper((unsigned char)c));
    long long mul = 1;
    auto ends = [&](const char* suf) {
      const std::string s(suf);
      return t.size() >= s.size() && t.compare(t.size() - s.size(), s.size(), s) == 0;
    };
    if (ends("KB")) { mul = 1ll << 10; t.resize(t.size() - 2); }
    else if (ends("MB")) { mul = 1ll << 20; t.resize(t.size() - 2); }
    else if (ends("GB")) { mul = 1ll << 30; t.resize(t.size() - 2); }
    else if (ends("B")) { t.resize(t.size() - 1); }
    return (long long)(std::strtod(t.c_str(), nullptr) * (double)mul);
  }
  std::string checksumType_;
  int bytesPerChecksum_;
  bool checksumVerify_;
  std::string warning_;
};

}  // namespace ozone
