"""Host-side mirror of the reference's ChunkBuffer: the buffer a chunk is collected in on the write path and
handed to Checksum.computeChecksum, which cuts it into bytes-per-checksum windows through iterate().

Mirrors, name for name and with the same argument meaning and error behaviour:
  ChunkBuffer (interface + factories)   hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/common/ChunkBuffer.java:31-185
  ChunkBufferImplWithByteBuffer         .../ozone/common/ChunkBufferImplWithByteBuffer.java:32-152
  ChunkBufferImplWithByteBufferList     .../ozone/common/ChunkBufferImplWithByteBufferList.java:42-256
  IncrementalChunkBuffer                .../ozone/common/IncrementalChunkBuffer.java:38-312
ByteBuffer is the part of java.nio.ByteBuffer those classes use (position / limit / capacity over shared
storage, duplicate, relative put / get).  Nothing here hashes: Checksum.computeChecksum (checksum.py) takes the
windows iterate() yields to the GPU library in one call.
"""


class BufferOverflowException(Exception):
    """java.nio.BufferOverflowException; `cause` is what initCause() attached."""

    def __init__(self, cause=None):
        super().__init__(str(cause) if cause else "")
        self.cause = cause


class ReadOnlyBufferException(Exception):
    """java.nio.ReadOnlyBufferException."""


class NoSuchElementException(Exception):
    """java.util.NoSuchElementException (an exhausted iterate() iterator)."""


class UnsupportedOperationException(Exception):
    """java.lang.UnsupportedOperationException."""


class IllegalStateException(Exception):
    """java.lang.IllegalStateException (ChunkBuffer.assertInt)."""


def _check_argument(cond, msg=""):
    """Guava Preconditions.checkArgument."""
    if not cond:
        raise ValueError(msg)


class ByteBuffer:
    """What ChunkBuffer needs of java.nio.ByteBuffer: a window [0, capacity) onto shared storage with a position
    and a limit.  duplicate() shares the storage and copies position and limit."""

    __slots__ = ("_mem", "_off", "_cap", "_pos", "_lim", "_ro")

    def __init__(self, mem, off, cap, pos, lim, read_only=False):
        self._mem, self._off, self._cap, self._pos, self._lim, self._ro = mem, off, cap, pos, lim, read_only

    @staticmethod
    def allocate(capacity):
        _check_argument(capacity >= 0, "capacity < 0: (%d < 0)" % capacity)
        return ByteBuffer(memoryview(bytearray(capacity)), 0, capacity, 0, capacity)

    @staticmethod
    def wrap(array, offset=0, length=None):
        """ByteBuffer.wrap(byte[]) / wrap(byte[], offset, length): capacity = the whole array, position = offset,
        limit = offset + length.  A bytes object (immutable) yields a read-only buffer, as a ByteString's
        asReadOnlyByteBuffer() does."""
        mv = memoryview(array).cast("B") if not isinstance(array, memoryview) else array.cast("B")
        n = len(mv)
        length = n - offset if length is None else length
        if offset < 0 or length < 0 or offset + length > n:
            raise IndexError("offset=%d length=%d array=%d" % (offset, length, n))
        return ByteBuffer(mv, 0, n, offset, offset + length, read_only=mv.readonly)

    def capacity(self):
        return self._cap

    def position(self, newPosition=None):
        if newPosition is None:
            return self._pos
        _check_argument(0 <= newPosition <= self._lim, "newPosition > limit: (%d > %d)" % (newPosition, self._lim))
        self._pos = newPosition
        return self

    def limit(self, newLimit=None):
        if newLimit is None:
            return self._lim
        _check_argument(0 <= newLimit <= self._cap, "newLimit > capacity: (%d > %d)" % (newLimit, self._cap))
        self._lim = newLimit
        if self._pos > newLimit:
            self._pos = newLimit
        return self

    def remaining(self):
        return self._lim - self._pos

    def hasRemaining(self):
        return self._pos < self._lim

    def rewind(self):
        self._pos = 0
        return self

    def clear(self):
        self._pos, self._lim = 0, self._cap
        return self

    def flip(self):
        self._lim, self._pos = self._pos, 0
        return self

    def duplicate(self):
        return ByteBuffer(self._mem, self._off, self._cap, self._pos, self._lim, self._ro)

    def asReadOnlyBuffer(self):
        return ByteBuffer(self._mem, self._off, self._cap, self._pos, self._lim, True)

    def isReadOnly(self):
        return self._ro

    def view(self):
        """The remaining bytes [position, limit) as a memoryview onto the shared storage (no copy)."""
        return self._mem[self._off + self._pos:self._off + self._lim]

    def get(self):
        """Relative get of one byte (signed, as Java's byte)."""
        if self._pos >= self._lim:
            raise IndexError("BufferUnderflowException")
        b = self._mem[self._off + self._pos]
        self._pos += 1
        return b - 256 if b > 127 else b

    def put(self, src):
        """Relative bulk put of another buffer's remaining bytes (both positions advance), or of one byte."""
        if self._ro:
            raise ReadOnlyBufferException()
        if isinstance(src, int):
            if self._pos >= self._lim:
                raise BufferOverflowException()
            self._mem[self._off + self._pos] = src & 0xff
            self._pos += 1
            return self
        n = src.remaining()
        if n > self.remaining():
            raise BufferOverflowException()
        self._mem[self._off + self._pos:self._off + self._pos + n] = src.view()
        self._pos += n
        src._pos += n
        return self

    def __eq__(self, other):  # ByteBuffer.equals: same remaining bytes
        return isinstance(other, ByteBuffer) and self.view() == other.view()

    def __hash__(self):
        return hash(bytes(self.view()))

    def __repr__(self):
        return "java.nio.HeapByteBuffer[pos=%d lim=%d cap=%d]" % (self._pos, self._lim, self._cap)


def _as_byte_buffer(b, offset=None, length=None):
    """The put() overloads: ByteBuffer, byte[] (+ offset, length), ByteString (read-only view), one byte."""
    if isinstance(b, ByteBuffer):
        return b
    if offset is None:
        return ByteBuffer.wrap(b)
    return ByteBuffer.wrap(b, offset, length)


def _safe_wrap(buffer):
    """ByteStringConversion.safeWrap: a copy of the remaining bytes; position and limit stay."""
    return bytes(buffer.view())


class ChunkBuffer:
    """ChunkBuffer.java:31-185: the interface with its factories and default methods."""

    @staticmethod
    def allocate(capacity, increment=0):
        """:35-52: an IncrementalChunkBuffer when 0 < increment < capacity, else one buffer of `capacity`."""
        if 0 < increment < capacity:
            return IncrementalChunkBuffer(capacity, increment, False)
        return ChunkBufferImplWithByteBuffer(ByteBuffer.allocate(capacity))

    @staticmethod
    def wrap(buffer):
        """:55-62: one ByteBuffer, or a list of them."""
        if isinstance(buffer, (list, tuple)):
            return ChunkBufferImplWithByteBufferList(buffer)
        return ChunkBufferImplWithByteBuffer(buffer)

    # ---- what an implementation provides
    def position(self):
        raise NotImplementedError

    def remaining(self):
        raise NotImplementedError

    def limit(self):
        raise NotImplementedError

    def rewind(self):
        raise NotImplementedError

    def clear(self):
        raise NotImplementedError

    def _put(self, b):
        raise NotImplementedError

    def duplicate(self, newPosition, newLimit):
        raise NotImplementedError

    def iterate(self, bufferSize):
        raise NotImplementedError

    def asByteBufferList(self):
        raise NotImplementedError

    def toByteStringImpl(self, function):
        raise NotImplementedError

    def toByteStringListImpl(self, function):
        raise NotImplementedError

    # ---- defaults (:77-166)
    def hasRemaining(self):
        return self.remaining() > 0

    def put(self, b, offset=None, length=None):
        """put(ByteBuffer) / put(byte[]) / put(byte[], offset, length) / put(ByteString) / put(byte)."""
        if isinstance(b, int):
            return self._put(ByteBuffer.wrap(bytes([b & 0xff])))
        return self._put(_as_byte_buffer(b, offset, length))

    def writeTo(self, channel):
        """Writes position..limit to `channel` (anything with write(bytes-like)); returns the byte count.  The
        buffers' positions advance as a GatheringByteChannel.write advances them."""
        total = 0
        for b in self.asByteBufferList():
            n = b.remaining()
            if n:
                channel.write(b.view())
                b.position(b.limit())
                total += n
        return total

    def toByteString(self, function=_safe_wrap):
        return self.toByteStringImpl(lambda b: _apply_and_assert(b, function, self))

    def toByteStringList(self, function=_safe_wrap):
        return self.toByteStringListImpl(lambda b: _apply_and_assert(b, function, self))


def assertInt(expected, computed, prefix):
    """ChunkBuffer.java:168-173."""
    if expected != computed:
        raise IllegalStateException("%s: expected = %d but computed = %d" % (prefix(), expected, computed))


def _apply_and_assert(buffer, function, name):
    """ChunkBuffer.java:176-184: the conversion function must leave position and limit alone."""
    pos, lim = buffer.position(), buffer.limit()
    out = function(buffer)
    assertInt(pos, buffer.position(), lambda: "%s: Unexpected position" % name)
    assertInt(lim, buffer.limit(), lambda: "%s: Unexpected limit" % name)
    return out


class _Iter:
    """An Iterator<ByteBuffer> over has_next / take, raising NoSuchElementException when exhausted through
    next() and ending a Python for-loop through __next__."""

    def __init__(self, has_next, take):
        self._has_next, self._take = has_next, take

    def hasNext(self):
        return self._has_next()

    def next(self):
        if not self._has_next():
            raise NoSuchElementException()
        return self._take()

    def __iter__(self):
        return self

    def __next__(self):
        if not self._has_next():
            raise StopIteration
        return self._take()


class _Iterable:
    def __init__(self, make):
        self._make = make

    def iterator(self):
        return self._make()

    def __iter__(self):
        return self._make()


class ChunkBufferImplWithByteBuffer(ChunkBuffer):
    """ChunkBufferImplWithByteBuffer.java:32-152: one ByteBuffer."""

    def __init__(self, buffer):
        if buffer is None:
            raise TypeError("buffer == null")
        self.buffer = buffer

    def position(self):
        return self.buffer.position()

    def remaining(self):
        return self.buffer.remaining()

    def limit(self):
        return self.buffer.limit()

    def rewind(self):
        self.buffer.rewind()
        return self

    def clear(self):
        self.buffer.clear()
        return self

    def iterate(self, bufferSize):
        """:62-83: duplicates of at most bufferSize bytes each from position to limit, the last possibly
        shorter; the buffer's position moves along."""
        buf = self.buffer

        def take():
            d = buf.duplicate()
            m = min(buf.position() + bufferSize, buf.limit())
            d.limit(m)
            buf.position(m)
            return d

        return _Iterable(lambda: _Iter(buf.hasRemaining, take))

    def asByteBufferList(self):
        return [self.buffer]

    def duplicate(self, newPosition, newLimit):
        d = self.buffer.duplicate()
        # Java: duplicated.position(newPosition).limit(newLimit) -- position first, checked against the old limit
        d.position(newPosition).limit(newLimit)
        return ChunkBufferImplWithByteBuffer(d)

    def _put(self, b):
        self.buffer.put(b)
        return self

    def toByteStringImpl(self, function):
        return function(self.buffer)

    def toByteStringListImpl(self, function):
        return [function(self.buffer)]

    def __eq__(self, other):
        return isinstance(other, ChunkBufferImplWithByteBuffer) and self.buffer == other.buffer

    def __hash__(self):
        return hash(self.buffer)

    def __str__(self):
        return "ChunkBufferImplWithByteBuffer:limit=%d" % self.buffer.limit()


class ChunkBufferImplWithByteBufferList(ChunkBuffer):
    """ChunkBufferImplWithByteBufferList.java:42-256: a list of ByteBuffers; iterate() yields one element per
    step whatever bufferSize is.

    reference_list_quirk.  findCurrent() adds the limits of the consumed buffers to limitPrecedingCurrent without
    resetting it (:59-78), and iterate().next() calls it on every step (:192-201), so position() runs ahead of
    the bytes consumed and the iteration ends early: three of four equal elements are visited.  With
    reference_list_quirk=True (the default of this class, which mirrors the reference as it is) that is
    reproduced; with False, findCurrent() starts its sum from zero and every element is visited."""

    def __init__(self, buffers, reference_list_quirk=True):
        _check_argument(buffers is not None, "buffer == null")
        self.buffers = list(buffers) if len(buffers) else [ByteBuffer.allocate(0)]
        self._limit = sum(b.limit() for b in buffers)
        self.reference_list_quirk = reference_list_quirk
        self.limitPrecedingCurrent = 0
        self.currentIndex = 0
        self._findCurrent()

    def _findCurrent(self):  # :59-78
        if not self.reference_list_quirk:
            self.limitPrecedingCurrent = 0
        found = False
        for i, buf in enumerate(self.buffers):
            pos = buf.position()
            if found:
                _check_argument(pos == 0, "all buffers after current one should have position=0")
            elif pos < buf.limit():
                found = True
                self.currentIndex = i
            else:
                self.limitPrecedingCurrent += buf.limit()
        if not found:
            self.currentIndex = len(self.buffers) - 1
            self.limitPrecedingCurrent -= self._current().limit()

    def _current(self):
        return self.buffers[self.currentIndex]

    def _advanceCurrent(self):  # :84-92
        if self.currentIndex < len(self.buffers) - 1:
            cur = self.buffers[self.currentIndex]
            if not cur.hasRemaining():
                self.currentIndex += 1
                self.limitPrecedingCurrent += cur.limit()

    def _rewindCurrent(self):
        self.currentIndex = 0
        self.limitPrecedingCurrent = 0

    def position(self):
        return self.limitPrecedingCurrent + self._current().position()

    def remaining(self):
        return self._limit - self.position()

    def limit(self):
        return self._limit

    def hasRemaining(self):
        return self.position() < self._limit

    def rewind(self):
        for b in self.buffers:
            b.rewind()
        self._rewindCurrent()
        return self

    def clear(self):
        for b in self.buffers:
            b.clear()
        self._rewindCurrent()
        return self

    def _put(self, that):  # :128-150
        thisRemaining = self.remaining()
        thatRemaining = that.remaining()
        if thatRemaining > thisRemaining:
            raise BufferOverflowException(ValueError(
                "Failed to put since that.remaining() = %d > this.remaining() = %d" % (thatRemaining, thisRemaining)))
        while thatRemaining > 0:
            b = self._current()
            n = min(b.remaining(), thatRemaining)
            that.limit(that.position() + n)
            b.put(that)
            thatRemaining -= n
            self._advanceCurrent()
        return self

    def duplicate(self, newPosition, newLimit):  # :153-174
        _check_argument(newPosition >= 0)
        _check_argument(newPosition <= newLimit)
        _check_argument(newLimit <= self.limit())
        duplicates = []
        lo = 0
        for buf in self.buffers:
            hi = lo + buf.limit()
            d = buf.duplicate()
            # position first, as the reference does: it is checked against the element's old limit
            d.position(_relative_to_range(newPosition, lo, hi)).limit(_relative_to_range(newLimit, lo, hi))
            duplicates.append(d)
            lo = hi
        return ChunkBufferImplWithByteBufferList(duplicates, self.reference_list_quirk)

    def iterate(self, bufferSize):  # :180-204
        def take():
            self._findCurrent()
            cur = self.buffers[self.currentIndex]
            d = cur.duplicate()
            d.limit(cur.limit())
            cur.position(cur.limit())
            return d

        return _Iterable(lambda: _Iter(self.hasRemaining, take))

    def asByteBufferList(self):
        return self.buffers

    def writeTo(self, channel):
        n = super().writeTo(channel)
        self._findCurrent()  # :213
        return n

    def toByteStringImpl(self, function):
        return b"".join(function(b) for b in self.buffers)

    def toByteStringListImpl(self, function):
        return [function(b) for b in self.buffers]

    def __str__(self):
        return "ChunkBufferImplWithByteBufferList:n=%d:p=%d:l=%d" % (len(self.buffers), self.position(), self.limit())


def _relative_to_range(value, lo, hi):  # :245-255
    if value <= lo:
        return 0
    if value <= hi:
        return value - lo
    return hi - lo


class IncrementalChunkBuffer(ChunkBuffer):
    """IncrementalChunkBuffer.java:38-312: buffers of `increment` bytes each, allocated as data arrives.
    iterate() is legal only with bufferSize == increment and hands out the underlying buffers (:250-259)."""

    def __init__(self, limit, increment, isDuplicated):
        _check_argument(limit >= 0)
        _check_argument(increment > 0)
        self._limit = limit
        self.increment = increment
        self.limitIndex = limit // increment
        self.buffers = []
        self.isDuplicated = isDuplicated
        self._firstNonFullIndex = 0

    def _capacity_at(self, i):  # :66-70
        _check_argument(i >= 0)
        _check_argument(i <= self.limitIndex)
        return self.increment if i < self.limitIndex else self._limit % self.increment

    def _assert_int(self, expected, computed, name, i):
        assertInt(expected, computed, lambda: "%s: Unexpected %s at index %d" % (self, name, i))

    def _get_at(self, i):  # :78-98
        _check_argument(i >= 0)
        _check_argument(i <= self.limitIndex)
        ith = self.buffers[i] if i < len(self.buffers) else None
        if ith is not None:
            if not self.isDuplicated:
                self._assert_int(self._capacity_at(i), ith.capacity(), "capacity", i)
            elif i < self.limitIndex:
                self._assert_int(self.increment, ith.capacity(), "capacity", i)
            elif i == self.limitIndex:
                self._assert_int(self._capacity_at(i), ith.limit(), "capacity", i)
            else:
                self._assert_int(0, ith.limit(), "capacity", i)
        return ith

    def _get_and_allocate_at_index(self, index):  # :101-122
        _check_argument(index >= 0)
        if self._limit % self.increment == 0:
            _check_argument(index < self.limitIndex)
        else:
            _check_argument(index <= self.limitIndex)
        i = len(self.buffers)
        if index < i:
            return self._get_at(index)
        b = None
        while i <= index:
            b = ByteBuffer.allocate(self._capacity_at(i))
            self.buffers.append(b)
            i += 1
        return b

    def _get_and_allocate_at_position(self, position):  # :125-132
        _check_argument(position >= 0)
        _check_argument(position < self._limit)
        i = position // self.increment
        ith = self._get_and_allocate_at_index(i)
        self._assert_int(position % self.increment, ith.position(), "position", i)
        return ith

    def _first_non_full_index(self):  # :135-144
        for i in range(self._firstNonFullIndex, len(self.buffers)):
            if self._get_at(i).position() != self.increment:
                self._firstNonFullIndex = i
                return i
        self._firstNonFullIndex = len(self.buffers)
        return self._firstNonFullIndex

    def position(self):  # :147-157
        i = self._first_non_full_index()
        ith = self._get_at(i) if i <= self.limitIndex else None
        return i * self.increment + (ith.position() if ith is not None else 0)

    def remaining(self):
        return self._limit - self.position()

    def limit(self):
        return self._limit

    def rewind(self):
        for b in self.buffers:
            b.rewind()
        self._firstNonFullIndex = 0
        return self

    def clear(self):
        for b in self.buffers:
            b.clear()
        self._firstNonFullIndex = 0
        return self

    def _put(self, that):  # :201-220
        if that.remaining() > self.remaining():
            raise BufferOverflowException(ValueError(
                "Failed to put since that.remaining() = %d > this.remaining() = %d" % (that.remaining(), self.remaining())))
        thatLimit = that.limit()
        p = self.position()
        while that.position() < thatLimit:
            b = self._get_and_allocate_at_position(p)
            n = min(b.remaining(), thatLimit - that.position())
            that.limit(that.position() + n)
            b.put(that)
            that.limit(thatLimit)  # ByteBuffer.limit() is restored by the next that.limit(...) in the reference
            p += n
        return self

    def duplicate(self, newPosition, newLimit):  # :223-247
        _check_argument(newPosition >= 0)
        _check_argument(newPosition <= newLimit)
        _check_argument(newLimit <= self._limit)
        d = IncrementalChunkBuffer(newLimit, self.increment, True)
        pi, pr = divmod(newPosition, self.increment)
        li, lr = divmod(newLimit, self.increment)
        newSize = li if lr == 0 else li + 1
        for i in range(newSize):
            pos = self.increment if i < pi else pr if i == pi else 0
            lim = self.increment if i < li else lr if i == li else 0
            ith = self._get_at(i)
            if ith is None:
                raise TypeError("buffers[%d] == null" % i)
            b = ith.duplicate()
            # the reference sets the position first (checked against the old limit), then the limit (which pulls
            # a position beyond it back)
            b.position(pos).limit(lim)
            d.buffers.append(b)
        return d

    def iterate(self, bufferSize):  # :250-259
        if bufferSize != self.increment:
            raise UnsupportedOperationException(
                "Buffer size and increment mismatched: bufferSize = %d but increment = %d" % (bufferSize, self.increment))
        return self.asByteBufferList()

    def asByteBufferList(self):
        return tuple(self.buffers)  # Collections.unmodifiableList

    def toByteStringImpl(self, function):
        return b"".join(function(b) for b in self.buffers)

    def toByteStringListImpl(self, function):
        return [function(b) for b in self.buffers]

    def __eq__(self, other):
        return isinstance(other, IncrementalChunkBuffer) and self._limit == other._limit and self.buffers == other.buffers

    def __hash__(self):
        return hash(tuple(self.buffers))

    def __str__(self):
        return "IncrementalChunkBuffer:limit=%d,increment=%d" % (self._limit, self.increment)
