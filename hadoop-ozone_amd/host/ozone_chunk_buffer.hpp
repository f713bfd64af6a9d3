// Host-side mirror of the reference's ChunkBuffer: the buffer a chunk is collected in on the write path and
// handed to Checksum.computeChecksum, which cuts it into bytes-per-checksum windows through iterate().
//
//   ChunkBuffer (interface + factories)   hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/common/ChunkBuffer.java:31-185
//   ChunkBufferImplWithByteBuffer         .../ozone/common/ChunkBufferImplWithByteBuffer.java:32-152
//   ChunkBufferImplWithByteBufferList     .../ozone/common/ChunkBufferImplWithByteBufferList.java:42-256
//   IncrementalChunkBuffer                .../ozone/common/IncrementalChunkBuffer.java:38-312
//
// ByteBuffer is the part of java.nio.ByteBuffer those classes use.  Java code shares ByteBuffer objects by
// reference (a list element consumed by an iterator is consumed for the list's owner too), so buffers travel as
// ByteBufferPtr here; duplicate() makes a new object over the same storage.  Header-only, no GPU code: the
// digests of the windows are Checksum::computeChecksum's business (ozone_checksum.hpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ozone {

struct BufferOverflowException : std::runtime_error {  // java.nio.BufferOverflowException (+ initCause text)
  explicit BufferOverflowException(const std::string& cause = "") : std::runtime_error(cause) {}
};
struct ReadOnlyBufferException : std::runtime_error {
  ReadOnlyBufferException() : std::runtime_error("ReadOnlyBufferException") {}
};
struct NoSuchElementException : std::runtime_error {
  NoSuchElementException() : std::runtime_error("NoSuchElementException") {}
};
struct UnsupportedOperationException : std::runtime_error {
  explicit UnsupportedOperationException(const std::string& m) : std::runtime_error(m) {}
};
struct IllegalStateException : std::runtime_error {
  explicit IllegalStateException(const std::string& m) : std::runtime_error(m) {}
};

inline void checkArgument(bool cond, const char* msg = "") {  // Guava Preconditions.checkArgument
  if (!cond) throw std::invalid_argument(msg);
}

class ByteBuffer;
using ByteBufferPtr = std::shared_ptr<ByteBuffer>;

class ByteBuffer {
 public:
  static ByteBufferPtr allocate(int capacity) {
    checkArgument(capacity >= 0, "capacity < 0");
    auto mem = std::make_shared<std::vector<uint8_t>>((size_t)capacity, (uint8_t)0);
    uint8_t* base = mem->data();
    return ByteBufferPtr(new ByteBuffer(std::move(mem), base, capacity, 0, capacity, false));
  }
  // ByteBuffer.wrap(byte[] copy of the caller's bytes, offset, length): capacity = n, position = offset.
  static ByteBufferPtr copyOf(const void* data, size_t n, int offset = 0, int length = -1) {
    ByteBufferPtr b = allocate((int)n);
    if (n) std::memcpy(b->base_, data, n);
    if (length < 0) length = (int)n - offset;
    if (offset < 0 || offset + length > (int)n) throw std::out_of_range("offset/length");
    b->pos_ = offset;
    b->lim_ = offset + length;
    return b;
  }
  // A read-only view of memory the caller keeps alive (ByteString.asReadOnlyByteBuffer): no copy.
  static ByteBufferPtr wrapReadOnly(const void* data, size_t n) {
    return ByteBufferPtr(new ByteBuffer(nullptr, const_cast<uint8_t*>(static_cast<const uint8_t*>(data)), (int)n, 0, (int)n, true));
  }

  int capacity() const { return cap_; }
  int position() const { return pos_; }
  ByteBuffer& position(int p) {
    checkArgument(p >= 0 && p <= lim_, "newPosition > limit");
    pos_ = p;
    return *this;
  }
  int limit() const { return lim_; }
  ByteBuffer& limit(int l) {
    checkArgument(l >= 0 && l <= cap_, "newLimit > capacity");
    lim_ = l;
    if (pos_ > l) pos_ = l;
    return *this;
  }
  int remaining() const { return lim_ - pos_; }
  bool hasRemaining() const { return pos_ < lim_; }
  ByteBuffer& rewind() { pos_ = 0; return *this; }
  ByteBuffer& clear() { pos_ = 0; lim_ = cap_; return *this; }
  ByteBufferPtr duplicate() const { return ByteBufferPtr(new ByteBuffer(mem_, base_, cap_, pos_, lim_, ro_)); }
  ByteBufferPtr asReadOnlyBuffer() const { return ByteBufferPtr(new ByteBuffer(mem_, base_, cap_, pos_, lim_, true)); }
  bool isReadOnly() const { return ro_; }
  const uint8_t* data() const { return base_ + pos_; }  // the remaining bytes [position, limit)
  std::string str() const { return std::string(reinterpret_cast<const char*>(data()), (size_t)remaining()); }

  int8_t get() {  // relative get
    if (pos_ >= lim_) throw std::out_of_range("BufferUnderflowException");
    return (int8_t)base_[pos_++];
  }
  ByteBuffer& put(uint8_t b) {
    if (ro_) throw ReadOnlyBufferException();
    if (pos_ >= lim_) throw BufferOverflowException();
    base_[pos_++] = b;
    return *this;
  }
  ByteBuffer& put(ByteBuffer& src) {  // relative bulk put: both positions advance
    if (ro_) throw ReadOnlyBufferException();
    const int n = src.remaining();
    if (n > remaining()) throw BufferOverflowException();
    if (n) std::memmove(base_ + pos_, src.base_ + src.pos_, (size_t)n);
    pos_ += n;
    src.pos_ += n;
    return *this;
  }
  bool equals(const ByteBuffer& o) const {  // ByteBuffer.equals: the same remaining bytes
    return remaining() == o.remaining() && (remaining() == 0 || std::memcmp(data(), o.data(), (size_t)remaining()) == 0);
  }

 private:
  ByteBuffer(std::shared_ptr<std::vector<uint8_t>> mem, uint8_t* base, int cap, int pos, int lim, bool ro)
      : mem_(std::move(mem)), base_(base), cap_(cap), pos_(pos), lim_(lim), ro_(ro) {}
  std::shared_ptr<std::vector<uint8_t>> mem_;  // null for a view of the caller's memory
  uint8_t* base_;
  int cap_, pos_, lim_;
  bool ro_;
};

// ChunkBuffer.java:168-173.
inline void assertInt(int expected, int computed, const std::function<std::string()>& prefix) {
  if (expected != computed)
    throw IllegalStateException(prefix() + ": expected = " + std::to_string(expected) + " but computed = " + std::to_string(computed));
}

class ChunkBuffer;
using ChunkBufferPtr = std::unique_ptr<ChunkBuffer>;
using ByteStringFunction = std::function<std::string(ByteBuffer&)>;

// Iterator<ByteBuffer> of iterate(): next() on an exhausted iterator throws NoSuchElementException.
class ByteBufferIterator {
 public:
  ByteBufferIterator(std::function<bool()> hasNext, std::function<ByteBufferPtr()> take)
      : hasNext_(std::move(hasNext)), take_(std::move(take)) {}
  bool hasNext() const { return hasNext_(); }
  ByteBufferPtr next() {
    if (!hasNext_()) throw NoSuchElementException();
    return take_();
  }

 private:
  std::function<bool()> hasNext_;
  std::function<ByteBufferPtr()> take_;
};

class ChunkBuffer {
 public:
  virtual ~ChunkBuffer() {}
  // ChunkBuffer.java:35-62
  static ChunkBufferPtr allocate(int capacity, int increment = 0);
  static ChunkBufferPtr wrap(ByteBufferPtr buffer);
  static ChunkBufferPtr wrap(const std::vector<ByteBufferPtr>& buffers);

  virtual int position() = 0;
  virtual int remaining() = 0;
  virtual int limit() = 0;
  virtual ChunkBuffer& rewind() = 0;
  virtual bool hasRemaining() { return remaining() > 0; }
  virtual ChunkBuffer& clear() = 0;
  virtual ChunkBuffer& put(ByteBuffer& b) = 0;
  ChunkBuffer& put(const void* b, size_t n) {  // put(byte[]) / put(byte[], offset, length) / put(ByteString)
    ByteBufferPtr v = ByteBuffer::wrapReadOnly(b, n);
    return put(*v);
  }
  ChunkBuffer& put(const std::string& b) { return put(b.data(), b.size()); }
  ChunkBuffer& put(uint8_t b) { return put(&b, 1); }
  virtual ChunkBufferPtr duplicate(int newPosition, int newLimit) = 0;
  // From the current position to the current limit, bufferSize bytes per step (the implementations differ in
  // what they make of bufferSize).
  virtual ByteBufferIterator iterate(int bufferSize) = 0;
  virtual std::vector<ByteBufferPtr> asByteBufferList() = 0;
  // writeTo(GatheringByteChannel): position..limit of every buffer goes to `sink`; the positions advance.
  virtual long long writeTo(const std::function<void(const uint8_t*, size_t)>& sink) {
    long long total = 0;
    for (const ByteBufferPtr& b : asByteBufferList()) {
      const int n = b->remaining();
      if (n) sink(b->data(), (size_t)n);
      b->position(b->limit());
      total += n;
    }
    return total;
  }
  // toByteString / toByteStringList (:143-156): `function` must leave position and limit alone.
  std::string toByteString(const ByteStringFunction& function = safeWrap) {
    return toByteStringImpl([&](ByteBuffer& b) { return applyAndAssert(b, function); });
  }
  std::vector<std::string> toByteStringList(const ByteStringFunction& function = safeWrap) {
    return toByteStringListImpl([&](ByteBuffer& b) { return applyAndAssert(b, function); });
  }
  virtual std::string toString() = 0;

 protected:
  virtual std::string toByteStringImpl(const ByteStringFunction& f) = 0;
  virtual std::vector<std::string> toByteStringListImpl(const ByteStringFunction& f) = 0;
  static std::string safeWrap(ByteBuffer& b) { return b.str(); }  // ByteStringConversion.safeWrap: a copy

 private:
  std::string applyAndAssert(ByteBuffer& b, const ByteStringFunction& function) {  // :176-184
    const int pos = b.position(), lim = b.limit();
    std::string out = function(b);
    assertInt(pos, b.position(), [&] { return toString() + ": Unexpected position"; });
    assertInt(lim, b.limit(), [&] { return toString() + ": Unexpected limit"; });
    return out;
  }
};

// ChunkBufferImplWithByteBuffer.java:32-152: one ByteBuffer.
class ChunkBufferImplWithByteBuffer : public ChunkBuffer {
 public:
  explicit ChunkBufferImplWithByteBuffer(ByteBufferPtr buffer) : buffer_(std::move(buffer)) {
    if (!buffer_) throw std::invalid_argument("buffer == null");
  }
  int position() override { return buffer_->position(); }
  int remaining() override { return buffer_->remaining(); }
  int limit() override { return buffer_->limit(); }
  ChunkBuffer& rewind() override { buffer_->rewind(); return *this; }
  ChunkBuffer& clear() override { buffer_->clear(); return *this; }
  using ChunkBuffer::put;
  ChunkBuffer& put(ByteBuffer& b) override { buffer_->put(b); return *this; }
  // :62-83: duplicates of at most bufferSize bytes each, the last possibly shorter; the position moves along.
  ByteBufferIterator iterate(int bufferSize) override {
    ByteBufferPtr buf = buffer_;
    return ByteBufferIterator([buf] { return buf->hasRemaining(); },
                              [buf, bufferSize] {
                                ByteBufferPtr d = buf->duplicate();
                                const int m = std::min(buf->position() + bufferSize, buf->limit());
                                d->limit(m);
                                buf->position(m);
                                return d;
                              });
  }
  std::vector<ByteBufferPtr> asByteBufferList() override { return {buffer_}; }
  ChunkBufferPtr duplicate(int newPosition, int newLimit) override {
    ByteBufferPtr d = buffer_->duplicate();
    d->position(newPosition).limit(newLimit);  // position first, checked against the old limit, as in the reference
    return ChunkBufferPtr(new ChunkBufferImplWithByteBuffer(d));
  }
  std::string toString() override { return "ChunkBufferImplWithByteBuffer:limit=" + std::to_string(buffer_->limit()); }

 protected:
  std::string toByteStringImpl(const ByteStringFunction& f) override { return f(*buffer_); }
  std::vector<std::string> toByteStringListImpl(const ByteStringFunction& f) override { return {f(*buffer_)}; }

 private:
  ByteBufferPtr buffer_;
};

// ChunkBufferImplWithByteBufferList.java:42-256: a list of ByteBuffers; iterate() yields one element per step
// whatever bufferSize is.
//
// referenceListQuirk.  findCurrent() adds the limits of the consumed buffers to limitPrecedingCurrent without
// resetting it (:59-78), and iterate().next() calls it on every step (:192-201), so position() runs ahead of the
// bytes consumed and the iteration ends early: three of four equal elements are visited.  With referenceListQuirk
// = true (the default of this class, which mirrors the reference as it is) that is reproduced; with false,
// findCurrent() starts its sum from zero and every element is visited.
class ChunkBufferImplWithByteBufferList : public ChunkBuffer {
 public:
  explicit ChunkBufferImplWithByteBufferList(const std::vector<ByteBufferPtr>& buffers, bool referenceListQuirk = true)
      : buffers_(buffers), quirk_(referenceListQuirk) {
    if (buffers_.empty()) buffers_.push_back(ByteBuffer::allocate(0));
    for (const ByteBufferPtr& b : buffers) limit_ += b->limit();
    findCurrent();
  }
  int position() override { return limitPrecedingCurrent_ + current().position(); }
  int remaining() override { return limit_ - position(); }
  int limit() override { return limit_; }
  bool hasRemaining() override { return position() < limit_; }
  ChunkBuffer& rewind() override {
    for (const ByteBufferPtr& b : buffers_) b->rewind();
    currentIndex_ = 0;
    limitPrecedingCurrent_ = 0;
    return *this;
  }
  ChunkBuffer& clear() override {
    for (const ByteBufferPtr& b : buffers_) b->clear();
    currentIndex_ = 0;
    limitPrecedingCurrent_ = 0;
    return *this;
  }
  using ChunkBuffer::put;
  ChunkBuffer& put(ByteBuffer& that) override {  // :128-150
    const int thisRemaining = remaining();
    int thatRemaining = that.remaining();
    if (thatRemaining > thisRemaining)
      throw BufferOverflowException("Failed to put since that.remaining() = " + std::to_string(thatRemaining) +
                                    " > this.remaining() = " + std::to_string(thisRemaining));
    while (thatRemaining > 0) {
      ByteBuffer& b = current();
      const int n = std::min(b.remaining(), thatRemaining);
      that.limit(that.position() + n);
      b.put(that);
      thatRemaining -= n;
      advanceCurrent();
    }
    return *this;
  }
  ChunkBufferPtr duplicate(int newPosition, int newLimit) override {  // :153-174
    checkArgument(newPosition >= 0);
    checkArgument(newPosition <= newLimit);
    checkArgument(newLimit <= limit());
    std::vector<ByteBufferPtr> duplicates;
    int lo = 0;
    for (const ByteBufferPtr& buf : buffers_) {
      const int hi = lo + buf->limit();
      ByteBufferPtr d = buf->duplicate();
      d->position(relativeToRange(newPosition, lo, hi)).limit(relativeToRange(newLimit, lo, hi));
      duplicates.push_back(d);
      lo = hi;
    }
    return ChunkBufferPtr(new ChunkBufferImplWithByteBufferList(duplicates, quirk_));
  }
  ByteBufferIterator iterate(int) override {  // :180-204
    return ByteBufferIterator([this] { return hasRemaining(); },
                              [this] {
                                findCurrent();
                                ByteBuffer& cur = *buffers_[(size_t)currentIndex_];
                                ByteBufferPtr d = cur.duplicate();
                                d->limit(cur.limit());
                                cur.position(cur.limit());
                                return d;
                              });
  }
  std::vector<ByteBufferPtr> asByteBufferList() override { return buffers_; }
  long long writeTo(const std::function<void(const uint8_t*, size_t)>& sink) override {
    const long long n = ChunkBuffer::writeTo(sink);
    findCurrent();  // :213
    return n;
  }
  std::string toString() override {
    return "ChunkBufferImplWithByteBufferList:n=" + std::to_string(buffers_.size()) + ":p=" + std::to_string(position()) +
           ":l=" + std::to_string(limit());
  }

 protected:
  std::string toByteStringImpl(const ByteStringFunction& f) override {
    std::string out;
    for (const ByteBufferPtr& b : buffers_) out += f(*b);
    return out;
  }
  std::vector<std::string> toByteStringListImpl(const ByteStringFunction& f) override {
    std::vector<std::string> out;
    for (const ByteBufferPtr& b : buffers_) out.push_back(f(*b));
    return out;
  }

 private:
  ByteBuffer& current() { return *buffers_[(size_t)currentIndex_]; }
  void findCurrent() {  // :59-78
    if (!quirk_) limitPrecedingCurrent_ = 0;
    bool found = false;
    for (size_t i = 0; i < buffers_.size(); i++) {
      const ByteBuffer& buf = *buffers_[i];
      if (found) {
        checkArgument(buf.position() == 0, "all buffers after current one should have position=0");
      } else if (buf.position() < buf.limit()) {
        found = true;
        currentIndex_ = (int)i;
      } else {
        limitPrecedingCurrent_ += buf.limit();
      }
    }
    if (!found) {
      currentIndex_ = (int)buffers_.size() - 1;
      limitPrecedingCurrent_ -= current().limit();
    }
  }
  void advanceCurrent() {  // :84-92
    if (currentIndex_ < (int)buffers_.size() - 1 && !current().hasRemaining()) {
      limitPrecedingCurrent_ += current().limit();
      currentIndex_++;
    }
  }
  static int relativeToRange(int value, int lo, int hi) {  // :245-255
    return value <= lo ? 0 : value <= hi ? value - lo : hi - lo;
  }
  std::vector<ByteBufferPtr> buffers_;
  bool quirk_;
  int limit_ = 0, limitPrecedingCurrent_ = 0, currentIndex_ = 0;
};

// IncrementalChunkBuffer.java:38-312: buffers of `increment` bytes each, allocated as data arrives.  iterate()
// is legal only with bufferSize == increment and hands out the underlying buffers (:250-259).
class IncrementalChunkBuffer : public ChunkBuffer {
 public:
  IncrementalChunkBuffer(int limit, int increment, bool isDuplicated)
      : limit_(limit), increment_(increment), limitIndex_(increment > 0 ? limit / increment : 0), isDuplicated_(isDuplicated) {
    checkArgument(limit >= 0);
    checkArgument(increment > 0);
  }
  int position() override {  // :147-157
    const int i = firstNonFullIndex();
    const ByteBufferPtr ith = getAtIndex(i);
    return i * increment_ + (ith ? ith->position() : 0);
  }
  int remaining() override { return limit_ - position(); }
  int limit() override { return limit_; }
  ChunkBuffer& rewind() override {
    for (const ByteBufferPtr& b : buffers_) b->rewind();
    firstNonFullIndex_ = 0;
    return *this;
  }
  ChunkBuffer& clear() override {
    for (const ByteBufferPtr& b : buffers_) b->clear();
    firstNonFullIndex_ = 0;
    return *this;
  }
  using ChunkBuffer::put;
  ChunkBuffer& put(ByteBuffer& that) override {  // :201-220
    if (that.remaining() > remaining())
      throw BufferOverflowException("Failed to put since that.remaining() = " + std::to_string(that.remaining()) +
                                    " > this.remaining() = " + std::to_string(remaining()));
    const int thatLimit = that.limit();
    for (int p = position(); that.position() < thatLimit;) {
      ByteBufferPtr b = getAndAllocateAtPosition(p);
      const int n = std::min(b->remaining(), thatLimit - that.position());
      that.limit(that.position() + n);
      b->put(that);
      that.limit(thatLimit);
      p += n;
    }
    return *this;
  }
  ChunkBufferPtr duplicate(int newPosition, int newLimit) override {  // :223-247
    checkArgument(newPosition >= 0);
    checkArgument(newPosition <= newLimit);
    checkArgument(newLimit <= limit_);
    std::unique_ptr<IncrementalChunkBuffer> d(new IncrementalChunkBuffer(newLimit, increment_, true));
    const int pi = newPosition / increment_, pr = newPosition % increment_;
    const int li = newLimit / increment_, lr = newLimit % increment_;
    const int newSize = lr == 0 ? li : li + 1;
    for (int i = 0; i < newSize; i++) {
      const int pos = i < pi ? increment_ : i == pi ? pr : 0;
      const int lim = i < li ? increment_ : i == li ? lr : 0;
      const ByteBufferPtr ith = getAtIndex(i);
      if (!ith) throw std::invalid_argument("buffers[" + std::to_string(i) + "] == null");
      ByteBufferPtr b = ith->duplicate();
      b->position(pos).limit(lim);
      d->buffers_.push_back(b);
    }
    return ChunkBufferPtr(d.release());
  }
  ByteBufferIterator iterate(int bufferSize) override {  // :250-259
    if (bufferSize != increment_)
      throw UnsupportedOperationException("Buffer size and increment mismatched: bufferSize = " + std::to_string(bufferSize) +
                                          " but increment = " + std::to_string(increment_));
    auto list = std::make_shared<std::vector<ByteBufferPtr>>(buffers_);
    auto at = std::make_shared<size_t>(0);
    return ByteBufferIterator([list, at] { return *at < list->size(); }, [list, at] { return (*list)[(*at)++]; });
  }
  std::vector<ByteBufferPtr> asByteBufferList() override { return buffers_; }
  std::string toString() override {
    return "IncrementalChunkBuffer:limit=" + std::to_string(limit_) + ",increment=" + std::to_string(increment_);
  }

 protected:
  std::string toByteStringImpl(const ByteStringFunction& f) override {
    std::string out;
    for (const ByteBufferPtr& b : buffers_) out += f(*b);
    return out;
  }
  std::vector<std::string> toByteStringListImpl(const ByteStringFunction& f) override {
    std::vector<std::string> out;
    for (const ByteBufferPtr& b : buffers_) out.push_back(f(*b));
    return out;
  }

 private:
  int capacityAtIndex(int i) const {  // :66-70
    checkArgument(i >= 0);
    checkArgument(i <= limitIndex_);
    return i < limitIndex_ ? increment_ : limit_ % increment_;
  }
  void assertAt(int expected, int computed, const char* name, int i) {
    assertInt(expected, computed, [&] { return toString() + ": Unexpected " + name + " at index " + std::to_string(i); });
  }
  ByteBufferPtr getAtIndex(int i) {  // :78-98
    checkArgument(i >= 0);
    checkArgument(i <= limitIndex_);
    ByteBufferPtr ith = (size_t)i < buffers_.size() ? buffers_[(size_t)i] : nullptr;
    if (ith) {
      if (!isDuplicated_) assertAt(capacityAtIndex(i), ith->capacity(), "capacity", i);
      else if (i < limitIndex_) assertAt(increment_, ith->capacity(), "capacity", i);
      else if (i == limitIndex_) assertAt(capacityAtIndex(i), ith->limit(), "capacity", i);
      else assertAt(0, ith->limit(), "capacity", i);
    }
    return ith;
  }
  ByteBufferPtr getAndAllocateAtIndex(int index) {  // :101-122
    checkArgument(index >= 0);
    if (limit_ % increment_ == 0) checkArgument(index < limitIndex_);
    else checkArgument(index <= limitIndex_);
    int i = (int)buffers_.size();
    if (index < i) return getAtIndex(index);
    ByteBufferPtr b;
    for (; i <= index; i++) {
      b = ByteBuffer::allocate(capacityAtIndex(i));
      buffers_.push_back(b);
    }
    return b;
  }
  ByteBufferPtr getAndAllocateAtPosition(int position) {  // :125-132
    checkArgument(position >= 0);
    checkArgument(position < limit_);
    const int i = position / increment_;
    ByteBufferPtr ith = getAndAllocateAtIndex(i);
    assertAt(position % increment_, ith->position(), "position", i);
    return ith;
  }
  int firstNonFullIndex() {  // :135-144
    for (int i = firstNonFullIndex_; i < (int)buffers_.size(); i++) {
      if (getAtIndex(i)->position() != increment_) {
        firstNonFullIndex_ = i;
        return i;
      }
    }
    firstNonFullIndex_ = (int)buffers_.size();
    return firstNonFullIndex_;
  }
  int limit_, increment_, limitIndex_;
  std::vector<ByteBufferPtr> buffers_;
  bool isDuplicated_;
  int firstNonFullIndex_ = 0;
};

inline ChunkBufferPtr ChunkBuffer::allocate(int capacity, int increment) {  // ChunkBuffer.java:47-52
  if (increment > 0 && increment < capacity) return ChunkBufferPtr(new IncrementalChunkBuffer(capacity, increment, false));
  return ChunkBufferPtr(new ChunkBufferImplWithByteBuffer(ByteBuffer::allocate(capacity)));
}
inline ChunkBufferPtr ChunkBuffer::wrap(ByteBufferPtr buffer) { return ChunkBufferPtr(new ChunkBufferImplWithByteBuffer(std::move(buffer))); }
inline ChunkBufferPtr ChunkBuffer::wrap(const std::vector<ByteBufferPtr>& buffers) {
  return ChunkBufferPtr(new ChunkBufferImplWithByteBufferList(buffers));
}

}  // namespace ozone
