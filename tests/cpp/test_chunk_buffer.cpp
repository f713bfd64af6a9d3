// The C++ ChunkBuffer mirror against the reference's own test of the class
// (hadoop-hdds/common/src/test/java/org/apache/hadoop/ozone/common/TestChunkBuffer.java:40-238): same cases, same
// assertions, a seeded generator in place of ThreadLocalRandom.  No GPU and no library: nothing here hashes.
#include <cstdio>
#include <random>
#include <string>

#include "../../hadoop-ozone_amd/host/ozone_chunk_buffer.hpp"

using namespace ozone;

static int failures = 0;
#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);            \
      failures++;                                                            \
    }                                                                        \
  } while (0)

static std::mt19937 rng(20240607);
static int nextInt(int n) { return (int)(rng() % (unsigned)n); }
static std::string randomBytes(int n) {
  std::string s((size_t)n, '\0');
  for (auto& c : s) c = (char)(rng() & 0xff);
  return s;
}

static void assertToByteString(const std::string& expected, int offset, int length, ChunkBuffer& impl) {  // :184-199
  ChunkBufferPtr duplicated = impl.duplicate(offset, offset + length);
  EXPECT(duplicated->position() == offset);
  EXPECT(duplicated->remaining() == length);
  const std::string computed = duplicated->toByteString([](ByteBuffer& b) { return b.str(); });
  EXPECT(duplicated->position() == offset);
  EXPECT(duplicated->remaining() == length);
  EXPECT(computed == expected.substr((size_t)offset, (size_t)length));
}

static void assertDuplicate(const std::string& expected, ChunkBuffer& impl) {  // :147-155
  const int n = (int)expected.size();
  assertToByteString(expected, 0, n, impl);
  for (int i = 0; i < 10; i++) {
    const int offset = nextInt(n);
    assertToByteString(expected, offset, nextInt(n - offset + 1), impl);
  }
}

static void assertIterate(const std::string& expected, ChunkBuffer& impl, int bpc) {  // :157-182
  const int n = (int)expected.size();
  ChunkBufferPtr duplicated = impl.duplicate(0, n);
  EXPECT(duplicated->position() == 0);
  EXPECT(duplicated->remaining() == n);
  const int numChecksums = (n + bpc - 1) / bpc;
  ByteBufferIterator i = duplicated->iterate(bpc);
  int count = 0;
  for (int j = 0; j < numChecksums; j++) {
    ByteBufferPtr b = i.next();
    const int expectedRemaining = j < numChecksums - 1 ? bpc : n - bpc * (numChecksums - 1);
    EXPECT(b->remaining() == expectedRemaining);
    const int offset = j * bpc;
    for (int k = 0; k < expectedRemaining; k++) {
      EXPECT(b->get() == (int8_t)expected[(size_t)(offset + k)]);
      count++;
    }
  }
  EXPECT(count == n);
  EXPECT(!i.hasNext());
  try {
    i.next();
    EXPECT(!"no NoSuchElementException");
  } catch (const NoSuchElementException&) {
  }
}

static void assertWrite(const std::string& expected, ChunkBuffer& impl) {  // :201-216
  impl.rewind();
  EXPECT(impl.position() == 0);
  std::string out;
  impl.writeTo([&](const uint8_t* p, size_t n) { out.append(reinterpret_cast<const char*>(p), n); });
  EXPECT(out == expected);
  EXPECT(!impl.hasRemaining());
}

static void runTestImpl(const std::string& expected, int bpc, ChunkBuffer& impl) {  // :110-145
  const int n = (int)expected.size();
  EXPECT(impl.position() == 0);
  EXPECT(impl.remaining() == n);
  EXPECT(impl.limit() == n);
  impl.put(expected);
  EXPECT(impl.position() == n);
  EXPECT(impl.remaining() == 0);
  EXPECT(impl.limit() == n);
  assertDuplicate(expected, impl);
  if (bpc > 0) {
    assertIterate(expected, impl, bpc);
  } else if (bpc == 0) {
    for (int d = 1; d < 5; d++)
      if (n / d > 0) assertIterate(expected, impl, n / d);
    for (int d = 1; d < 10; d++) assertIterate(expected, impl, nextInt(n) + 1);
  }
  assertWrite(expected, impl);
}

int main() {
  // testImplWithByteBuffer, :45-58
  {
    std::vector<int> ns = {1, 1 << 10};
    for (int i = 0; i < 10; i++) ns.push_back(nextInt(100) + 1);
    for (int n : ns) {
      ChunkBufferPtr impl = ChunkBuffer::allocate(n);
      runTestImpl(randomBytes(n), 0, *impl);
    }
  }
  // testIncrementalChunkBuffer, :60-77
  {
    std::vector<std::pair<int, int>> cases = {{1, 1}, {4, 8}, {16, 1 << 10}};
    for (int i = 0; i < 10; i++) {
      const int a = nextInt(100) + 1, b = nextInt(100) + 1;
      cases.push_back({std::min(a, b), std::max(a, b)});
    }
    for (auto& c : cases) {
      IncrementalChunkBuffer impl(c.second, c.first, false);
      runTestImpl(randomBytes(c.second), c.first, impl);
    }
  }
  // testImplWithList, :79-108
  {
    std::vector<std::pair<int, int>> cases = {{4, 8}, {16, 1 << 10}};
    for (int i = 0; i < 10; i++) {
      const int a = nextInt(10) + 1, b = nextInt(100) + 1;
      cases.push_back({std::min(a, b), std::max(a, b)});
    }
    for (auto& c : cases) {
      const int count = c.first, n = c.second, avg = n / count;
      std::vector<ByteBufferPtr> buffers;
      int offset = 0;
      for (int i = 0; i < count - 1; i++) {
        const int length = nextInt(avg) + 1;
        buffers.push_back(ByteBuffer::allocate(length));
        offset += length;
      }
      if (n > offset) buffers.push_back(ByteBuffer::allocate(n - offset));
      ChunkBufferPtr impl = ChunkBuffer::wrap(buffers);
      runTestImpl(randomBytes(n), -1, *impl);
    }
  }
  // factories and texts (ChunkBuffer.java:35-62; toString of the three classes)
  EXPECT(ChunkBuffer::allocate(10)->toString() == "ChunkBufferImplWithByteBuffer:limit=10");
  EXPECT(ChunkBuffer::allocate(10, 10)->toString() == "ChunkBufferImplWithByteBuffer:limit=10");
  EXPECT(ChunkBuffer::allocate(10, 4)->toString() == "IncrementalChunkBuffer:limit=10,increment=4");
  EXPECT(ChunkBuffer::wrap(std::vector<ByteBufferPtr>{ByteBuffer::allocate(3), ByteBuffer::allocate(2)})->toString() ==
         "ChunkBufferImplWithByteBufferList:n=2:p=0:l=5");
  {
    ChunkBufferPtr inc = ChunkBuffer::allocate(10, 4);
    inc->put(std::string("1234567"));
    try {
      inc->put(std::string("abcd"));
      EXPECT(!"no overflow");
    } catch (const BufferOverflowException& e) {  // IncrementalChunkBuffer.java:202-208
      EXPECT(std::string(e.what()) == "Failed to put since that.remaining() = 4 > this.remaining() = 3");
    }
    try {
      inc->iterate(5);
      EXPECT(!"no mismatch");
    } catch (const UnsupportedOperationException& e) {  // :252-256
      EXPECT(std::string(e.what()) == "Buffer size and increment mismatched: bufferSize = 5 but increment = 4");
    }
    inc->put(std::string("890"));
    EXPECT((inc->duplicate(0, 10)->toByteStringList() == std::vector<std::string>{"1234", "5678", "90"}));
    EXPECT((inc->duplicate(3, 9)->toByteStringList() == std::vector<std::string>{"4", "5678", "9"}));
  }
  // the list iteration as the reference runs it (three of four equal elements) and as intended
  {
    const std::string data = randomBytes(20);
    auto make = [&](bool quirk) {
      std::vector<ByteBufferPtr> parts;
      for (int i = 0; i < 4; i++) parts.push_back(ByteBuffer::wrapReadOnly(data.data() + 5 * i, 5));
      return ChunkBufferImplWithByteBufferList(parts, quirk);
    };
    ChunkBufferImplWithByteBufferList asIs = make(true), fixed = make(false);
    int seen = 0;
    for (ByteBufferIterator it = asIs.iterate(5); it.hasNext(); seen++) EXPECT(it.next()->str() == data.substr(5 * (size_t)seen, 5));
    EXPECT(seen == 3);
    seen = 0;
    for (ByteBufferIterator it = fixed.iterate(5); it.hasNext(); seen++) EXPECT(it.next()->str() == data.substr(5 * (size_t)seen, 5));
    EXPECT(seen == 4);
    EXPECT(fixed.position() == 20);
  }
  if (failures) {
    std::printf("%d failure(s)\n", failures);
    return 1;
  }
  std::printf("all passed\n");
  return 0;
}
