"""TEST INFRASTRUCTURE ONLY: a range ENCODER for the token streams the reference's HEAD decoder accepts
(/root/reference/src/sqz.c:793-839) but HEAD's own encoder never writes -- back references.

HEAD's sqz_compress is literal-only (its finders are compiled out), so no stream any
encoder in the reference produces reaches the decoder's back-reference branch: the size / bits / pm_dist
models, the byte-serial overlapping copy and the ERANGE / ENOBUFS ordering of :809-833.  This module writes
such streams with the reference's own arithmetic (rc_encode :506-521, rc_emit :474-479, rc_flush :492-497,
the models of sqz_init :550-565, pm_update :466-472), so that the compiled reference, the oracle, the CPU
wave emulator and the GPU kernel can be compared on them (tests/test_rc.py).  The token format is the
DECODER's (:809-821): flag 0, size (2..254), bits, then bits-1 distance bits, least significant first, each
through its own two-symbol model; distance = 1 << bits | those bits.  Nothing here is product code."""

M64 = (1 << 64) - 1


class Model:
    def __init__(self, n):
        self.f = [1 if k < n else 0 for k in range(256)]
        self.total = n

    def below(self, sym):
        return sum(self.f[:sym])

    def update(self, sym):
        if self.total < (1 << 56):
            self.f[sym] += 1
            self.total += 1


class TokenEncoder:
    def __init__(self):
        self.low, self.range = 0, M64
        self.out = bytearray()
        self.lit, self.size, self.byte, self.bits = Model(2), Model(256), Model(256), Model(32)
        self.dist = [Model(2) for _ in range(32)]

    def _emit(self):                                   # rc_emit :474-479
        self.out.append(self.low >> 56)
        self.low = (self.low << 8) & M64
        self.range = (self.range << 8) & M64

    def _same_top(self):                               # :481-483 (low + range wraps like the reference's uint64_t)
        return (self.low >> 56) == (((self.low + self.range) & M64) >> 56)

    def encode(self, m, sym):                          # rc_encode :506-521
        total, start, size = m.total, m.below(sym), m.f[sym]
        self.range //= total
        self.low = (self.low + start * self.range) & M64
        self.range = (self.range * size) & M64
        m.update(sym)
        while self._same_top():
            self._emit()
        if self.range < total + 1:
            self._emit()
            self._emit()
            self.range = M64 - self.low

    def literal(self, b):                              # :722-723
        self.encode(self.lit, 1)
        self.encode(self.byte, b)

    def match(self, size, bits, low_bits=0):
        """what the decoder reads at :809-821; the distance it forms is (1 << bits | low_bits) for bits > 0
        (low_bits < 2^(bits-1)), and 0 for bits == 0"""
        self.encode(self.lit, 0)
        self.encode(self.size, size)
        if size == 0xFF or size < 2 or size > 254:
            return                                     # end of stream / ERANGE before anything else is read
        self.encode(self.bits, bits)
        for b in range(bits - 1):
            self.encode(self.dist[b], (low_bits >> b) & 1)

    def finish(self):                                  # :741-743
        self.encode(self.lit, 0)
        self.encode(self.size, 0xFF)
        for _ in range(8):                             # rc_flush :492-497
            self.range = M64
            self._emit()
        return bytes(self.out)


def distance(bits, low_bits=0):
    return ((1 << bits) | low_bits) if bits > 0 else 0


def expand(tokens):
    """what a conforming decoder writes for ("lit", b) / ("match", size, bits, low_bits) tokens with every
    copy in range: the byte-serial overlap rule of :826-830"""
    out = bytearray()
    for t in tokens:
        if t[0] == "lit":
            out.append(t[1])
        else:
            d = distance(t[2], t[3])
            for _ in range(t[1]):
                out.append(out[len(out) - d])
    return bytes(out)


def encode_tokens(tokens):
    e = TokenEncoder()
    for t in tokens:
        if t[0] == "lit":
            e.literal(t[1])
        else:
            e.match(t[1], t[2], t[3])
    return e.finish()
