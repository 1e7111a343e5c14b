"""The reference's source of randomness on the host side of the binding: `ark_std::test_rng()` and
`Fr::rand` (plonk/src/common/kzg.rs:25-28 draws the trapdoor from it; plonk/src/common/protocols.rs:239-253,
341-356, 457-471, 567-591 the extra interpolation points of the gadgets).

`test_rng()` is rand's `StdRng` -- the ChaCha stream cipher with 12 rounds -- keyed with a fixed seed; `Fr::rand`
reads four 64-bit words as the limbs of the element's Montgomery representation, clears the top bit and redraws
while the value is not below r.  The limbs drawn are therefore exactly what the C ABI takes: no conversion.
Insecure by design, like the reference's (a test generator with a public seed)."""
import numpy as np

_SEED = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0]) + bytes(16)
_SIGMA = np.frombuffer(b"expand 32-byte k", dtype="<u4")
_R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class TestRng:
    """`ark_std::test_rng()`: every instance starts the same stream."""

    __test__ = False

    def __init__(self):
        self._key = np.frombuffer(_SEED, dtype="<u4")
        self._block = 0
        self._words = []

    def _refill(self):
        start = np.concatenate([_SIGMA, self._key, np.array(
            [self._block & 0xFFFFFFFF, self._block >> 32, 0, 0], dtype=np.uint32)]).astype(np.uint64)
        s = [int(v) for v in start]

        def qr(a, b, c, d):
            for rot_d, rot_b in ((16, 12), (8, 7)):
                s[a] = (s[a] + s[b]) & 0xFFFFFFFF
                x = s[d] ^ s[a]
                s[d] = ((x << rot_d) | (x >> (32 - rot_d))) & 0xFFFFFFFF
                s[c] = (s[c] + s[d]) & 0xFFFFFFFF
                x = s[b] ^ s[c]
                s[b] = ((x << rot_b) | (x >> (32 - rot_b))) & 0xFFFFFFFF

        for _ in range(6):  # 12 rounds: six column / diagonal pairs
            qr(0, 4, 8, 12), qr(1, 5, 9, 13), qr(2, 6, 10, 14), qr(3, 7, 11, 15)
            qr(0, 5, 10, 15), qr(1, 6, 11, 12), qr(2, 7, 8, 13), qr(3, 4, 9, 14)
        self._words = [(x + int(y)) & 0xFFFFFFFF for x, y in zip(s, start)]
        self._block += 1

    def next_u32(self):
        if not self._words:
            self._refill()
        return self._words.pop(0)

    def next_u64(self):
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)

    def fr(self):
        """`Fr::rand(&mut rng)` as (4,) Montgomery limbs."""
        while True:
            limbs = [self.next_u64() for _ in range(4)]
            limbs[3] &= 0x7FFFFFFFFFFFFFFF
            if sum(l << (64 * i) for i, l in enumerate(limbs)) < _R_MOD:
                return np.array(limbs, dtype=np.uint64)

    def fr_vec(self, count):
        """`count` consecutive draws as (count, 4) limbs."""
        return np.stack([self.fr() for _ in range(count)]) if count else np.zeros((0, 4), dtype=np.uint64)
