// The reference's TestChecksum.java cases, written against the C++ mirror (hadoop-ozone_amd/host/ozone_checksum.hpp)
// and checked digest by digest against the CPU oracle.  Runs in a process of its own, without PyTorch: this is
// how a JNI host would load the library.  Exit code 0 = all passed.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../hadoop-ozone_amd/host/ozone_checksum.hpp"
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
  EXPECT(c20.computeChecksum(windows) == cd);
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
  std::printf(failures ? "%d FAILURES\n" : "host mirror: all passed\n", failures);
  ozcs_shutdown();
  return failures ? 1 : 0;
}
