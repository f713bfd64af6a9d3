// The reference's TestChecksum.java cases, written against the C++ mirror (hadoop-ozone_amd/host/ozone_checksum.hpp)
// and checked digest by digest against the CPU oracle.  Runs in a process of its own, without PyTorch: this is
// how a JNI host would load the library.  Exit code 0 = all passed.
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <utility>
#include <map>
#include <string>
#include <vector>

#include "../../hadoop-ozone_amd/host/ozone_callers.hpp"
#include "../../oracle/ozcs_oracle.h"

using namespace ozone;

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

static std::vector<ByteString> oracle_sums(ChecksumType t, int bpc, const std::string& d) {
  const size_t ds = oracle_digest_size((int)t);
  const size_t n = (d.size() + bpc - 1) / bpc;
  std::string raw(n * ds + 1, '\0');
  const long long got = oracle_compute((int)t, (uint32_t)bpc, (const uint8_t*)d.data(), d.size(), (uint8_t*)&raw[0], n * ds);
  std::vector<ByteString> v;
  for (long long i = 0; i < got; i++) v.push_back(raw.substr(i * ds, ds));
  return v;
}

int main() {
  if (!ozcs_available()) { std::printf("no GPU: %s\n", ozcs_last_error()); return 2; }
  std::string data(55, '\0');
  oracle_fill_splitmix64(1, (uint8_t*)&data[0], data.size());
  for (auto& c : data) c = (char)('a' + ((unsigned char)c % 26));  // RandomStringUtils.randomAlphabetic(55)

  const ChecksumType types[] = {ChecksumType::SHA256, ChecksumType::CRC32, ChecksumType::CRC32C, ChecksumType::MD5};
  for (ChecksumType t : types) {
    Checksum checksum(t, 10);
    const ChecksumData cd = checksum.computeChecksum(data.data(), data.size());
    EXPECT(cd.getChecksums().size() == 6);                                  // TestChecksum.java:56-58
    EXPECT(cd.getChecksums() == oracle_sums(t, 10, data));
    EXPECT(Checksum::verifyChecksum(data.data(), data.size(), cd));         // :60-61
    std::string bad = data;
    bad[50] = (char)(bad[50] + 1);
    EXPECT(checksum.computeChecksum(bad.data(), bad.size()) != cd);         // :67-80
    try {
      Checksum::verifyChecksum(bad.data(), bad.size(), cd);
      EXPECT(!"mismatch not reported");
    } catch (const OzoneChecksumException& e) {
      EXPECT(e.index() == 5);
      EXPECT(std::string(e.what()) == "Checksum mismatch at index 5");
    }
  }
  // different types give different ChecksumData (TestChecksum.java:86-101)
  EXPECT(Checksum(ChecksumType::SHA256, 10).computeChecksum(data.data(), data.size()) !=
         Checksum(ChecksumType::CRC32, 10).computeChecksum(data.data(), data.size()));
  // NONE: no checksums, verify is true without looking (Checksum.java:157-160, 241-244)
  EXPECT(Checksum(ChecksumType::NONE, 10).computeChecksum(data.data(), data.size()).getChecksums().empty());
  EXPECT(Checksum::verifyChecksum("x", 1, ChecksumData(ChecksumType::NONE, 10)));
  // list overloads and startIndex (ChunkInputStream.java:497-503): 100 bytes at bpc 20
  std::string whole(100, '\0');
  oracle_fill_splitmix64(2, (uint8_t*)&whole[0], whole.size());
  Checksum c20(ChecksumType::CRC32, 20);
  const ChecksumData cd = c20.computeChecksum(whole.data(), whole.size());
  std::vector<ByteString> windows;
  for (int i = 0; i < 100; i += 20) windows.push_back(whole.substr(i, 20));
  EXPECT(c20.computeChecksum(windows, false) == cd);   // every element; the default follows the reference's iterator
  EXPECT(Checksum::verifyChecksum(std::vector<ByteString>(windows.begin() + 1, windows.begin() + 3), cd, 1, false));
  EXPECT(Checksum::verifyChecksum(std::vector<ByteString>{whole.substr(20, 40)}, cd, 1, true));
  try {
    Checksum::verifyChecksum(std::vector<ByteString>(windows.begin() + 1, windows.begin() + 3), cd, 2, false);
    EXPECT(!"wrong start index not reported");
  } catch (const OzoneChecksumException& e) { EXPECT(e.index() == 0); }
  try {
    Checksum::verifyChecksum(whole.data(), whole.size(), ChecksumData(ChecksumType::CRC32, 20, {cd.getChecksums()[0], cd.getChecksums()[1]}));
    EXPECT(!"overrun not reported");
  } catch (const OzoneChecksumException& e) {
    EXPECT(std::string(e.what()).find("Computed checksum has 5 number of checksums. Original checksum has 2") == 0);
  }
  try {
    Checksum::verifyChecksum(whole.data(), whole.size(), ChecksumData(ChecksumType::CRC32, 20));
    EXPECT(!"empty stored list not reported");
  } catch (const OzoneChecksumException& e) { EXPECT(std::string(e.what()) == "Original checksumData has no checksums"); }
  // proto wire form
  const ChecksumData one(ChecksumType::CRC32, 1 << 20, {Checksum::int2ByteString((int32_t)0xCBF43926)});
  const unsigned char want[] = {0x08, 0x02, 0x10, 0x80, 0x80, 0x40, 0x1a, 0x04, 0xcb, 0xf4, 0x39, 0x26};
  EXPECT(one.getProtoBufMessage() == std::string((const char*)want, sizeof want));
  // config 1: one 4 MiB chunk at 1 MiB windows, all types
  std::string chunk(4u << 20, '\0');
  oracle_fill_splitmix64(1, (uint8_t*)&chunk[0], chunk.size());
  for (ChecksumType t : types)
    EXPECT(Checksum(t, 1 << 20).computeChecksum(chunk.data(), chunk.size()).getChecksums() == oracle_sums(t, 1 << 20, chunk));
  // proto round trip (ChecksumData.java:78-102)
  EXPECT(ChecksumData::getFromProtoBuf(one.getProtoBufMessage()) == one);
  EXPECT(ChecksumData::getFromProtoBuf(cd.getProtoBufMessage()) == cd);
  try {
    ChecksumData::getFromProtoBuf(one.getProtoBufMessage().substr(0, 9));
    EXPECT(!"truncated message accepted");
  } catch (const std::invalid_argument&) {}

  // ---- read path: TestChunkInputStream shapes (100-byte chunk, 20 bytes per checksum) ----
  EXPECT(computeChecksumBoundaries(25, 30, 20, 100) == std::make_pair(20LL, 40LL));   // bytes 25..54 -> 20..59
  EXPECT(computeChecksumBoundaries(0, 100, 20, 100) == std::make_pair(0LL, 100LL));
  EXPECT(computeChecksumBoundaries(95, 5, 20, 100) == std::make_pair(80LL, 20LL));
  EXPECT(computeChecksumBoundaries(70, 30, 20, 95) == std::make_pair(60LL, 35LL));   // capped at the chunk length
  {
    const auto range = computeChecksumBoundaries(25, 30, 20, 100);
    const std::string resp = whole.substr((size_t)range.first, (size_t)range.second);
    validateReadChunk("chunk1", 0, cd, range.first, range.second, {resp}, true);                        // V0
    validateReadChunk("chunk1", 0, cd, range.first, range.second, {resp.substr(0, 20), resp.substr(20)}, false);  // V1
    try {
      validateReadChunk("chunk1", 0, cd, range.first, range.second, {resp.substr(0, 39)}, true);
      EXPECT(!"short response accepted");
    } catch (const OzoneChecksumException& e) {
      EXPECT(std::string(e.what()) == "Inconsistent read for chunk=chunk1 len=40 bytesRead=39");
    }
    std::string flipped = resp;
    flipped[25] ^= 1;   // second window of the response = window 2 of the chunk
    try {
      validateReadChunk("chunk1", 0, cd, range.first, range.second, {flipped}, true);
      EXPECT(!"corrupt response accepted");
    } catch (const OzoneChecksumException& e) { EXPECT(e.index() == 1); }
  }

  // ---- the datanode's cut of a response into buffers (ChunkManager.java:107-134, BufferUtils.java:42-63) ----
  {
    ChunkInfo info("chunk1", 0, 100);
    info.setChecksumData(cd);                                     // CRC32 at 20 bytes per checksum
    EXPECT(getBufferCapacityForChunkRead(info) == 20);            // one buffer per checksum
    EXPECT(assignByteBuffers(90, 20) == (std::vector<long long>{20, 20, 20, 20, 10}));
    const auto range = computeChecksumBoundaries(25, 30, 20, 100);
    const std::string resp = whole.substr((size_t)range.first, (size_t)range.second);
    const std::vector<ByteString> v1 = readChunkBuffers(resp, info);
    EXPECT(v1.size() == 2 && v1[0].size() == 20 && v1[1].size() == 20);
    validateReadChunk("chunk1", 0, cd, range.first, range.second, v1, false);
    info.setReadDataIntoSingleBuffer(true);                       // an older client: everything in one buffer
    EXPECT(getBufferCapacityForChunkRead(info) == 100);
    const std::vector<ByteString> v0 = readChunkBuffers(resp, info);
    EXPECT(v0.size() == 1 && v0[0] == resp);
    validateReadChunk("chunk1", 0, cd, range.first, range.second, v0, true);
    ChunkInfo none("n", 0, 200000);
    none.setChecksumData(ChecksumData(ChecksumType::NONE, 0));
    EXPECT(getBufferCapacityForChunkRead(none) == 64 * 1024);     // no checksums: the default capacity
    EXPECT(assignByteBuffers(200000, 65536) == (std::vector<long long>{65536, 65536, 65536, 3392}));
    ChunkInfo bare("b", 0, 77);                                   // no checksum data at all: one buffer
    EXPECT(getBufferCapacityForChunkRead(bare) == 77);
    EXPECT(readChunkBuffers(ByteString(), info).size() == 1);
    try { assignByteBuffers(0, 4); EXPECT(!"zero length accepted"); } catch (const std::invalid_argument&) {}
  }

  // ---- ChunkInfo wire form (ChunkInfo.java:88-133, DatanodeClientProtocol.proto:366-372) ----
  {
    const ChecksumData cd(ChecksumType::CRC32, 1 << 20, {ByteString("\x01\x02\x03\x04", 4), ByteString("\xff\xfe\xfd\xfc", 4)});
    ChunkInfo info("105_chunk_3", 8ll << 20, 4ll << 20);
    info.setChecksumData(cd);
    info.addMetadata("zeta", "1");
    info.addMetadata("alpha", "x");
    try { info.addMetadata("alpha", "y"); EXPECT(!"duplicate key accepted"); } catch (const std::runtime_error&) {}
    const ByteString cdm = cd.getProtoBufMessage();
    ByteString want = ByteString("\x0a\x0b") + "105_chunk_3" + ByteString("\x10\x80\x80\x80\x04", 5) + ByteString("\x18\x80\x80\x80\x02", 5) +
                      ByteString("\x22\x0a\x0a\x05") + "alpha" + ByteString("\x12\x01x") + ByteString("\x22\x09\x0a\x04") + "zeta" +
                      ByteString("\x12\x01") + "1" + ByteString("\x2a") + ByteString(1, (char)cdm.size()) + cdm;
    EXPECT(info.getProtoBufMessage() == want);
    const ChunkInfo back = ChunkInfo::getFromProtoBuf(want);
    EXPECT(back.getChunkName() == "105_chunk_3" && back.getOffset() == (8ll << 20) && back.getLen() == (4ll << 20));
    EXPECT(back.getChecksumData() == cd && back.getMetadata().size() == 2 && back.getMetadata().at("alpha") == "x");
    EXPECT(back.toString() == "ChunkInfo{chunkName='105_chunk_3, offset=8388608, len=4194304}");
    EXPECT(ChunkInfo::getFromProtoBuf(ChunkInfo("c", 0, 0).getProtoBufMessage()).getChecksumData() == ChecksumData(ChecksumType::NONE, 0));
  }

  // ---- client configuration keys (OzoneClientConfig.java:103-125, 143-151) ----
  {
    const OzoneClientConfig dflt;
    EXPECT(dflt.getChecksumType() == ChecksumType::CRC32);
    EXPECT(dflt.getBytesPerChecksum() == 1024 * 1024);
    EXPECT(dflt.isChecksumVerify());
    typedef std::map<std::string, std::string> Conf;
    OzoneClientConfig low(Conf{{"ozone.client.bytes.per.checksum", "4KB"}, {"ozone.client.checksum.type", "MD5"},
                               {"ozone.client.verify.checksum", "false"}});
    EXPECT(low.getBytesPerChecksum() == 16 * 1024);   // raised to the minimum, with the reference's warning
    EXPECT(low.validationWarning() == "The checksum size (4096) is not allowed to be less than the minimum size (16384), "
                                      "resetting to the minimum size.");
    EXPECT(low.getChecksumType() == ChecksumType::MD5 && !low.isChecksumVerify());
    EXPECT(OzoneClientConfig(Conf{{"ozone.client.bytes.per.checksum", "256KB"}}).getBytesPerChecksum() == 262144);
    low.setBytesPerChecksum(10);                       // the setter does not clamp
    EXPECT(low.getBytesPerChecksum() == 10);
    try {
      OzoneClientConfig(Conf{{"ozone.client.checksum.type", "CRC64"}}).getChecksumType();
      EXPECT(!"unknown type name accepted");
    } catch (const std::invalid_argument&) {}
  }

  // ---- list inputs: every element by default, the reference's subset on request ----
  {
    EXPECT((referenceListWindows({4, 4, 4, 4}) == std::vector<size_t>{0, 1, 2}));
    EXPECT((referenceListWindows({20, 20, 20, 20, 20}) == std::vector<size_t>{0, 1, 2, 3}));
    EXPECT(referenceListWindows(std::vector<size_t>(256, 16384)).size() == 24);
    EXPECT((referenceListWindows({0, 5, 0, 5, 5, 5}) == std::vector<size_t>{1, 3, 4}));
    EXPECT(referenceListWindows({}).empty());
    std::string bytes(100, '\0');
    oracle_fill_splitmix64(77, (uint8_t*)&bytes[0], bytes.size());
    std::vector<ByteString> parts;
    for (int o = 0; o < 100; o += 20) parts.push_back(bytes.substr(o, 20));
    const std::vector<ByteString> full = oracle_sums(ChecksumType::CRC32, 20, bytes);
    const Checksum ck(ChecksumType::CRC32, 20);
    EXPECT(ck.computeChecksum(parts, false).getChecksums() == full);
    // the default is the reference's result, and equal to the ChunkBuffer route (Checksum.java:148-153 share one iterator)
    EXPECT(ck.computeChecksum(parts).getChecksums() == std::vector<ByteString>(full.begin(), full.begin() + 4));
    EXPECT(ck.computeChecksum(parts, true).getChecksums() == std::vector<ByteString>(full.begin(), full.begin() + 4));
    {
      std::vector<ByteBufferPtr> bufs;
      for (const ByteString& p : parts) bufs.push_back(ByteBuffer::wrapReadOnly(p.data(), p.size()));
      EXPECT(ck.computeChecksum(*ChunkBuffer::wrap(bufs)) == ck.computeChecksum(parts));
    }
    std::vector<ByteString> bad = parts;
    bad[4][0] ^= 1;
    const ChecksumData stored(ChecksumType::CRC32, 20, full);
    try {
      Checksum::verifyChecksum(bad, stored, 0, false, false);
      EXPECT(!"corrupt fifth window accepted");
    } catch (const OzoneChecksumException& e) { EXPECT(e.index() == 4); }
    EXPECT(Checksum::verifyChecksum(bad, stored, 0, false));         // the reference never looks at it: the default
    EXPECT(Checksum::verifyChecksum(bad, stored, 0, false, true));
  }

  // ---- scrubber: TestKeyValueContainerCheck shape (SHA-256, ragged last window), both layouts ----
  {
    const int bpc = 2048;
    const long long chunkLen = 3 * 2048 + 1024, chunkOffset = 5000;
    std::string file(chunkOffset + chunkLen + 777, '\0');
    oracle_fill_splitmix64(9, (uint8_t*)&file[0], file.size());
    const std::string path = "/tmp/ozcs_cpp_chunk_" + std::to_string((long long)getpid());
    auto put = [&](const std::string& bytes) {
      FILE* f = std::fopen(path.c_str(), "wb");
      std::fwrite(bytes.data(), 1, bytes.size(), f);
      std::fclose(f);
    };
    put(file);
    const ChecksumData stored = Checksum(ChecksumType::SHA256, bpc).computeChecksum(file.data() + chunkOffset, (size_t)chunkLen);
    std::vector<long long> seen;
    EXPECT(verifyChunkFile("blk#1", "c1", chunkOffset, chunkLen, stored, path, ChunkLayOutVersion::FILE_PER_BLOCK,
                           [&](long long n) { seen.push_back(n); }) == chunkLen);
    EXPECT((seen == std::vector<long long>{2048, 2048, 2048, 1024}));
    std::string bad = file;
    bad[(size_t)chunkOffset + 2048 + 7] ^= 0x40;
    put(bad);
    try {
      verifyChunkFile("blk#1", "c1", chunkOffset, chunkLen, stored, path, ChunkLayOutVersion::FILE_PER_BLOCK);
      EXPECT(!"corrupt chunk accepted");
    } catch (const OzoneChecksumException& e) {
      const std::string w = e.what();
      EXPECT(w.find("Inconsistent read for chunk=ChunkInfo{chunkName='c1, offset=5000, len=7168} checksum item 1 expected checksum [") == 0);
      EXPECT(w.find(" for block blk#1") == w.size() - 16);
    }
    seen.clear();   // window i is throttled, then judged; the failure in window 1 ends the scan
    try {
      verifyChunkFile("blk#1", "c1", chunkOffset, chunkLen, stored, path, ChunkLayOutVersion::FILE_PER_BLOCK,
                      [&](long long n) { seen.push_back(n); });
      EXPECT(!"corrupt chunk accepted");
    } catch (const OzoneChecksumException&) {
      EXPECT((seen == std::vector<long long>{2048, 2048}));
    }
    put(file.substr(0, (size_t)chunkOffset + 4096));   // truncated at a window boundary
    try {
      verifyChunkFile("blk#1", "c1", chunkOffset, chunkLen, stored, path, ChunkLayOutVersion::FILE_PER_BLOCK);
      EXPECT(!"truncated chunk accepted");
    } catch (const OzoneChecksumException& e) {
      EXPECT(std::string(e.what()) == "Inconsistent read for chunk=c1 expected length=7168 actual length=4096 for block blk#1");
    }
    put(file.substr((size_t)chunkOffset, (size_t)chunkLen));   // FILE_PER_CHUNK: the file is the chunk
    EXPECT(verifyChunkFile("blk#1", "c1", chunkOffset, chunkLen, stored, path, ChunkLayOutVersion::FILE_PER_CHUNK) == chunkLen);
    std::remove(path.c_str());
  }

  // ---- scrubber, a whole block in one batched call: three chunks in one block file, the second corrupt in two
  // windows; the throttler sees chunk 0 whole and chunk 1 up to its first failing window, nothing behind it ----
  {
    const int bpc = 4096;
    const long long chunkLen = 10 * 4096;
    std::string file((size_t)(3 * chunkLen), '\0');
    oracle_fill_splitmix64(600, (uint8_t*)&file[0], file.size());
    const std::string path = "/tmp/ozcs_cpp_block_" + std::to_string((long long)getpid());
    auto put = [&](const std::string& bytes) {
      FILE* f = std::fopen(path.c_str(), "wb");
      std::fwrite(bytes.data(), 1, bytes.size(), f);
      std::fclose(f);
    };
    std::vector<ScrubChunk> chunks;
    for (int i = 0; i < 3; i++)
      chunks.push_back(ScrubChunk{"c" + std::to_string(i), i * chunkLen, chunkLen,
                                  ChecksumData(ChecksumType::CRC32C, bpc, oracle_sums(ChecksumType::CRC32C, bpc, file.substr((size_t)(i * chunkLen), (size_t)chunkLen))),
                                  path});
    put(file);
    long long throttled = 0;
    EXPECT(verifyBlock("blk#9", chunks, ChunkLayOutVersion::FILE_PER_BLOCK, [&](long long n) { throttled += n; }) == 3 * chunkLen);
    EXPECT(throttled == 3 * chunkLen);
    std::string bad = file;
    bad[(size_t)(chunkLen + 3 * bpc + 17)] ^= 0x80;  // chunk 1, window 3
    bad[(size_t)(chunkLen + 7 * bpc)] ^= 0x80;       // and a later window, never reached
    put(bad);
    std::vector<long long> seen;
    try {
      verifyBlock("blk#9", chunks, ChunkLayOutVersion::FILE_PER_BLOCK, [&](long long n) { seen.push_back(n); });
      EXPECT(!"corrupt block accepted");
    } catch (const OzoneChecksumException& e) {
      const std::string w = e.what();
      EXPECT(w.find("Inconsistent read for chunk=ChunkInfo{chunkName='c1, offset=40960, len=40960} checksum item 3 expected checksum [") == 0);
      EXPECT(seen.size() == 10 + 4);   // chunk 0's ten windows, chunk 1's windows 0..3
    }
    // the same chunks as a container of three one-chunk blocks, batches cut after every block
    std::vector<ScrubBlock> blocks;
    for (int i = 0; i < 3; i++) blocks.push_back(ScrubBlock{"blk#" + std::to_string(i), {chunks[(size_t)i]}});
    seen.clear();
    try {
      verifyContainer(blocks, ChunkLayOutVersion::FILE_PER_BLOCK, [&](long long n) { seen.push_back(n); }, chunkLen);
      EXPECT(!"corrupt container accepted");
    } catch (const OzoneChecksumException& e) {
      const std::string w = e.what();
      EXPECT(w.find(" for block blk#1") == w.size() - 16);
      EXPECT(seen.size() == 10 + 4);
    }
    put(file);
    EXPECT(verifyContainer(blocks, ChunkLayOutVersion::FILE_PER_BLOCK) == 3 * chunkLen);
    // scanData's gate (KeyValueContainerCheck.java:241-256): a chunk of stored type NONE between two CRC32C chunks is
    // passed over (its bytes never reach the throttler); a missing file is an IOException unless the block has gone
    std::vector<ScrubChunk> gated = chunks;
    gated[1].checksumData = ChecksumData(ChecksumType::NONE, bpc, {"stale"});
    throttled = 0;
    EXPECT(verifyBlock("blk#9", gated, ChunkLayOutVersion::FILE_PER_BLOCK, [&](long long n) { throttled += n; }) == 2 * chunkLen);
    EXPECT(throttled == 2 * chunkLen);
    gated[2].chunkFile = path + ".gone";
    seen.clear();
    try {
      verifyBlock("blk#9", gated, ChunkLayOutVersion::FILE_PER_BLOCK, [&](long long n) { seen.push_back(n); });
      EXPECT(!"missing chunk file accepted");
    } catch (const OzoneChecksumException&) {
      EXPECT(!"a missing file is a plain IOException");
    } catch (const IOException& e) {
      EXPECT(std::string(e.what()) == "Missing chunk file " + path + ".gone");
      EXPECT(seen.size() == 10);   // chunk 0 was scanned first
    }
    std::vector<std::string> asked;
    EXPECT(verifyContainer({ScrubBlock{"blk#9", gated}}, ChunkLayOutVersion::FILE_PER_BLOCK, {}, 1ll << 30,
                           [&](const std::string& id) { asked.push_back(id); return false; }) == chunkLen);
    EXPECT(asked == std::vector<std::string>{"blk#9"});
    std::remove(path.c_str());
  }

  // ---- write path: chunks through pinned buffers, checksum overlapped with the next fill ----
  {
    PinnedChunkPool pool(0, 3, 1 << 20, ChecksumType::CRC32C, 65536);
    EXPECT(pool.getBufferSize() == (1u << 20));
    std::vector<std::pair<int, std::string>> inflight;
    int checked = 0;
    for (int i = 0; i < 10; i++) {
      std::pair<void*, int> b = pool.allocateBuffer();
      if (!b.first) {   // every buffer held: finish the oldest chunk, as BufferPool makes the writer wait
        EXPECT(pool.checksumOf(inflight.front().first).getChecksums() == oracle_sums(ChecksumType::CRC32C, 65536, inflight.front().second));
        inflight.erase(inflight.begin());
        checked++;
        b = pool.allocateBuffer();
      }
      EXPECT(b.first != nullptr);
      std::string bytes((1u << 20) - 1000 * i, '\0');
      oracle_fill_splitmix64(100 + i, (uint8_t*)&bytes[0], bytes.size());
      std::memcpy(b.first, bytes.data(), bytes.size());
      pool.writeChunk(b.second, bytes.size());
      inflight.emplace_back(b.second, bytes);
    }
    for (auto& f : inflight) {
      EXPECT(pool.checksumOf(f.first).getChecksums() == oracle_sums(ChecksumType::CRC32C, 65536, f.second));
      checked++;
    }
    EXPECT(checked == 10);
  }
  // computeChecksum(ChunkBuffer) over the three implementations (Checksum.java:155-177 over ChunkBuffer.iterate)
  for (ChecksumType t : types) {
    const int n = 100000, bpc = 16384;
    std::string bytes((size_t)n, '\0');
    oracle_fill_splitmix64(97, (uint8_t*)&bytes[0], bytes.size());
    const std::vector<ByteString> want = oracle_sums(t, bpc, bytes);
    Checksum ck(t, bpc);
    ChunkBufferPtr one = ChunkBuffer::allocate(n);
    one->put(bytes);
    ChunkBufferPtr dup = one->duplicate(0, one->position());
    EXPECT(ck.computeChecksum(*dup).getChecksums() == want);
    EXPECT(dup->position() == n && dup->limit() == n);   // "upon the iteration complete, position == limit"
    EXPECT(ck.computeChecksum(ByteBuffer::wrapReadOnly(bytes.data(), bytes.size())).getChecksums() == want);
    ChunkBufferPtr mid = one->duplicate(bpc, 70000);
    EXPECT(ck.computeChecksum(*mid).getChecksums() == oracle_sums(t, bpc, bytes.substr(bpc, 70000 - bpc)));
    ChunkBufferPtr inc = ChunkBuffer::allocate(n, bpc);   // stream.buffer.increment == bytes.per.checksum
    for (int o = 0; o < n; o += 7777) inc->put(bytes.substr((size_t)o, 7777));
    EXPECT(inc->position() == n);
    ChunkBufferPtr incAll = inc->duplicate(0, n);
    EXPECT(ck.computeChecksum(*incAll).getChecksums() == want);
    try {
      ChunkBufferPtr d = inc->duplicate(0, n);
      Checksum(t, 8192).computeChecksum(*d);
      EXPECT(!"mismatched increment accepted");
    } catch (const UnsupportedOperationException& e) {
      EXPECT(std::string(e.what()) == "Buffer size and increment mismatched: bufferSize = 8192 but increment = 16384");
    }
    // consumed buffers of a duplicate are handed out as they are: digests of the empty message lead the list
    ChunkBufferPtr tailBuf = inc->duplicate(2 * bpc, n);
    const std::vector<ByteString> tail = ck.computeChecksum(*tailBuf).getChecksums();
    EXPECT(tail.size() == want.size());
    for (size_t i = 2; i < want.size() && i < tail.size(); i++) EXPECT(tail[i] == want[i]);
    std::string emptyDigest(oracle_digest_size((int)t), '\0');
    if (t == ChecksumType::SHA256) oracle_sha256((const uint8_t*)"", 0, (uint8_t*)&emptyDigest[0]);
    if (t == ChecksumType::MD5) oracle_md5((const uint8_t*)"", 0, (uint8_t*)&emptyDigest[0]);
    EXPECT(tail.size() > 1 && tail[0] == emptyDigest && tail[1] == emptyDigest);
    // one buffer per window (a ReadChunk V1 response): the reference's iteration visits 4 of these 7 elements
    std::vector<ByteBufferPtr> parts;
    std::vector<size_t> lengths;
    for (int o = 0; o < n; o += bpc) {
      parts.push_back(ByteBuffer::wrapReadOnly(bytes.data() + o, (size_t)std::min(bpc, n - o)));
      lengths.push_back((size_t)parts.back()->remaining());
    }
    ChunkBufferPtr lst = ChunkBuffer::wrap(parts);
    std::vector<ByteString> subset;
    for (size_t i : referenceListWindows(lengths)) subset.push_back(want[i]);
    EXPECT(ck.computeChecksum(*lst).getChecksums() == subset);
    for (auto& b : parts) b->rewind();
    ChunkBufferImplWithByteBufferList every(parts, false);
    EXPECT(ck.computeChecksum(every).getChecksums() == want);
  }
  // BlockOutputStream over the pinned pipe, in the shape of TestBlockOutputStreamCorrectness.java: seeded bytes
  // written in odd pieces; what reaches writeChunkAsync is the same bytes cut into chunks of the buffer size, each
  // with the ChunkInfo the reference builds (BlockOutputStream.java:645-656), checksums included
  for (const char* typeSpec : {"CRC32", "SHA256", "NONE", "CRC32/unsafe", "SHA256/unsafe"}) {
    typedef std::map<std::string, std::string> Conf;
    const std::string spec(typeSpec);
    const bool safe = spec.find("/unsafe") == std::string::npos;   // chunk.toByteString: a copy (default) or a wrap
    const std::string typeName = spec.substr(0, spec.find('/'));
    OzoneClientConfig cfg(Conf{{"ozone.client.checksum.type", typeName}, {"ozone.client.bytes.per.checksum", "256KB"}});
    const size_t total = 6 * (1u << 20) + 12345;
    std::string payload(total, '\0');
    oracle_fill_splitmix64(2024, (uint8_t*)&payload[0], payload.size());
    std::vector<std::pair<ChunkInfo, std::string>> shipped;
    {
      BlockOutputStream out(777, [&](const ChunkInfo& info, const char* d, size_t n) { shipped.emplace_back(info, std::string(d, n)); },
                            cfg, 1u << 20, 3, 0, safe);
      const size_t pieces[5] = {1, 4099, 600001, 77, 262144};
      size_t pos = 0;
      for (int k = 0; pos < total; k++) {
        const size_t n = std::min(total - pos, pieces[k % 5]);
        out.write(payload.data() + pos, n);
        pos += n;
      }
      out.close();
      EXPECT(out.getWrittenDataLength() == (long long)total);
      EXPECT(out.getChunks().size() == 7);
      try {
        out.write("x", 1);
        EXPECT(!"write after close accepted");
      } catch (const std::runtime_error&) {
      }
    }
    EXPECT(shipped.size() == 7);
    std::string joined;
    long long off = 0;
    for (size_t i = 0; i < shipped.size(); i++) {
      const ChunkInfo& info = shipped[i].first;
      const std::string& d = shipped[i].second;
      joined += d;
      EXPECT(d.size() == (i < 6 ? (1u << 20) : 12345u));
      EXPECT(info.getChunkName() == "777_chunk_" + std::to_string(i + 1));
      EXPECT(info.getOffset() == off && info.getLen() == (long long)d.size());
      const ChecksumData& cd = info.getChecksumData();
      EXPECT(cd.getChecksumType() == cfg.getChecksumType() && cd.getBytesPerChecksum() == 262144);
      if (cfg.getChecksumType() == ChecksumType::NONE) EXPECT(cd.getChecksums().empty());
      else EXPECT(cd.getChecksums() == oracle_sums(cfg.getChecksumType(), 262144, d));
      const ChunkInfo back = ChunkInfo::getFromProtoBuf(info.getProtoBufMessage());  // WriteChunkRequestProto.chunkData
      EXPECT(back.getChecksumData() == cd && back.getOffset() == off);
      off += (long long)d.size();
    }
    EXPECT(joined == payload);
  }
  // writeSmallFile's ChunkInfo (ContainerProtocolCalls.java:334-345): CRC32 at 256-byte windows, fixed
  for (size_t n : {1u, 256u, 257u, 70000u}) {
    std::string bytes(n, '\0');
    oracle_fill_splitmix64(40 + n, (uint8_t*)&bytes[0], n);
    const ChunkInfo info = writeSmallFileChunkInfo(91, bytes.data(), n);
    EXPECT(info.getChunkName() == "91_chunk" && info.getOffset() == 0 && info.getLen() == (long long)n);
    EXPECT(info.getMetadata().size() == 1 && info.getMetadata().at("OverWriteRequested") == "true");
    EXPECT(info.getChecksumData().getBytesPerChecksum() == 256);
    EXPECT(info.getChecksumData().getChecksums() == oracle_sums(ChecksumType::CRC32, 256, bytes));
    EXPECT(ChunkInfo::getFromProtoBuf(info.getProtoBufMessage()).getChecksumData() == info.getChecksumData());
  }
  std::printf(failures ? "%d FAILURES\n" : "host mirror: all passed\n", failures);
  ozcs_shutdown();
  return failures ? 1 : 0;
}
