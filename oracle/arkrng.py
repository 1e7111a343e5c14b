"""TEST INFRASTRUCTURE (oracle): `ark_std::test_rng()` and `Fr::rand` restated in Python.

The reference draws its trapdoor and the extra interpolation points of its gadgets from
`ark_std::test_rng()` (plonk/src/common/kzg.rs:25-28; plonk/src/common/protocols.rs:239,251-253,
341,354-356, 457,469-471, 567,589-591), and its tests draw everything from it
(plonk/tests/protocols_tests.rs).  Those crates are third-party and absent here (ark-std 0.5.0, ark-ff 0.5.0,
rand 0.8, rand_chacha 0.3, plonk/Cargo.toml:14-19), so their published algorithms are restated:

* `ark_std::test_rng()` with ark-std's default features: `StdRng::from_seed` of a fixed 32-byte seed.
* rand 0.8 `StdRng` = `ChaCha12Rng`: the ChaCha block function with 12 rounds, key = seed, 64-bit block
  counter in state words 12-13 starting at 0, stream id 0 in words 14-15; output words are consumed in
  order, `next_u64` takes two consecutive 32-bit words, low half first.
* ark-ff `Fr::rand`: four `next_u64` as limbs (least significant first), the top 64 - 63 = 1 unused bit of
  the last limb masked off, rejected and redrawn while the value is >= the modulus; the limbs ARE the
  element's internal (Montgomery) representation, so the element is limbs * 2^-256 mod r.

Pinned on the reference's own data: with tau drawn this way, [tau^i]G reproduces all 22 points of
plonk/data/srs.json (tests/test_oracle_rng.py).  The seed, the round count and the Montgomery reading were
each decided by that file, not by memory.
"""
from . import pyref as o

MASK32 = 0xFFFFFFFF

# ark-std 0.5.0 src/rand_helper.rs: the seed of test_rng()
TEST_RNG_SEED = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)


def _rotl(v, n):
    return ((v << n) & MASK32) | (v >> (32 - n))


def _quarter(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & MASK32
    s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & MASK32
    s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & MASK32
    s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & MASK32
    s[b] = _rotl(s[b] ^ s[c], 7)


def chacha_block(key_words, counter, stream, rounds):
    """One 64-byte ChaCha block as sixteen 32-bit words (64-bit counter, 64-bit stream id: the layout of
    rand_chacha)."""
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [
        counter & MASK32, (counter >> 32) & MASK32, stream & MASK32, (stream >> 32) & MASK32]
    s = list(init)
    for _ in range(rounds // 2):
        _quarter(s, 0, 4, 8, 12)
        _quarter(s, 1, 5, 9, 13)
        _quarter(s, 2, 6, 10, 14)
        _quarter(s, 3, 7, 11, 15)
        _quarter(s, 0, 5, 10, 15)
        _quarter(s, 1, 6, 11, 12)
        _quarter(s, 2, 7, 8, 13)
        _quarter(s, 3, 4, 9, 14)
    return [(a + b) & MASK32 for a, b in zip(s, init)]


class StdRng:
    """rand 0.8 `StdRng::from_seed(seed)` (ChaCha12) as a stream of 32-bit words."""

    def __init__(self, seed=TEST_RNG_SEED, rounds=12):
        assert len(seed) == 32
        self.key = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8)]
        self.rounds = rounds
        self.counter = 0
        self.words = []

    def next_u32(self):
        if not self.words:
            self.words = chacha_block(self.key, self.counter, 0, self.rounds)
            self.counter += 1
        return self.words.pop(0)

    def next_u64(self):
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)


def gen_range_u32(rng, bound):
    """rand 0.8 `rng.gen_range(0..bound)` for a u32 bound (UniformInt::sample_single): a 32-bit draw v, the
    64-bit product v * bound, accepted when its low word lies in the zone (bound << leading_zeros(bound)) - 1;
    the result is the high word."""
    zone = ((bound << (32 - bound.bit_length())) - 1) & MASK32
    while True:
        m = rng.next_u32() * bound
        if (m & MASK32) <= zone:
            return m >> 32


def shuffle(rng, items):
    """rand 0.8 `SliceRandom::shuffle` (in place): for i = len - 1 down to 1, swap(i, gen_range(0..i + 1)).
    UNPINNED: restated from the published algorithm; nothing the reference holds depends on which permutation
    comes out (plonk/tests/protocols_tests.rs:665, 794 only need SOME permutation), so no fixture can confirm the
    draw-by-draw agreement of what follows a shuffle."""
    for i in range(len(items) - 1, 0, -1):
        j = gen_range_u32(rng, i + 1)
        items[i], items[j] = items[j], items[i]


def test_rng():
    """`ark_std::test_rng()`: a fresh generator from the fixed seed on every call."""
    return StdRng()


test_rng.__test__ = False  # not a pytest test


def fr_rand_montgomery_limbs(rng):
    """The four limbs `Fr::rand` accepts (the element's internal representation)."""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= (1 << 63) - 1  # 256 - 255 = 1 bit to shave
        value = sum(l << (64 * i) for i, l in enumerate(limbs))
        if value < o.R:
            return value


R_INV = pow(1 << 256, -1, o.R)


def fr_rand(rng):
    """`Fr::rand(&mut rng)` as the integer the element stands for."""
    return fr_rand_montgomery_limbs(rng) * R_INV % o.R


def random_polynomial(rng, degree):
    """plonk/src/common/polynomials.rs:9-11: `DensePolynomial::rand(degree, rng)` = degree + 1 draws, lowest
    coefficient first, trailing zeros trimmed."""
    return o.poly_trim([fr_rand(rng) for _ in range(degree + 1)])


def kzg_setup(degree):
    """plonk/src/common/kzg.rs:24-45 with its own tau: tau = Fr::rand(test_rng())."""
    tau = fr_rand(test_rng())
    return o.kzg_setup_with_tau(degree, tau), tau
