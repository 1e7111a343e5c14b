"""GPU parity of the compressed-G1 codec and of SRS ingest from its compressed encodings
(plonk/src/setup_global_params/json.rs:22-48; plonk/src/common/utils.rs:70-72) against the
reference's golden files and the CPU oracle, through the C ABI."""
import random

import numpy as np
import pytest

import plonk_amd
from oracle import capi as c
from oracle import pyref as o
from tests.gpu_util import backend, compress, same_point  # noqa: F401

pytestmark = pytest.mark.gpu


def test_golden_srs_decodes_like_the_reference(backend, golden):
    hexes = golden["srs"]["gp"]["tau_powers_g1"]  # plonk/data/srs.json:8-29
    raw = [bytes.fromhex(h) for h in hexes]
    pts = backend.g1_decompress(raw)
    want = c.g1_to_limbs([o.g1_decompress(b) for b in raw])
    for i in range(len(raw)):
        assert same_point(pts[i], want[i])
        assert np.array_equal(pts[i, 12:], c.fq_to_mont([1])[0])  # Z = 1
    assert backend.g1_compress(pts) == raw


def test_srs_ingest_from_compressed_reproduces_the_verification_key(backend, golden):
    raw = [bytes.fromhex(h) for h in golden["srs"]["gp"]["tau_powers_g1"]]
    backend.srs_load_compressed(raw)
    assert backend.srs_size() == 22
    for name in "SW":  # plonk/data/verification_key.json:2-3
        sc = c.fr_to_mont([int(x) for x in golden["proving_key"][name]])
        assert compress(backend.kzg_commit(sc)).hex() == golden["verification_key"]["com_" + name]


def test_proof_commitments_round_trip(backend, golden):
    # every 96-hex-digit string of plonk/data/proof.json is a compressed G1 written by utils.rs:70-72
    def walk(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from walk(v)
        elif isinstance(x, list):
            for v in x:
                yield from walk(v)
        elif isinstance(x, str) and len(x) == 96:
            yield x
    raw = [bytes.fromhex(h) for h in walk(golden["proof"])]
    assert len(raw) >= 20
    pts = backend.g1_decompress(raw)
    assert backend.g1_compress(pts) == raw
    for b, p in zip(raw, pts):
        assert compress(p) == b


def test_random_points_and_identity(backend):
    rnd = random.Random(90)
    n = 3000
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    srs = backend.kzg_setup(tau, n - 1)  # Jacobian, arbitrary Z
    srs[5] = c.g1_identity()
    srs[77] = c.g1_neg(srs[78])
    enc = backend.g1_compress(srs)
    assert enc == [bytes(b) for b in c.g1_compress(srs)]
    assert enc[5] == bytes([0xC0] + [0] * 47)
    back = backend.g1_decompress(enc)
    assert backend.g1_compress(back) == enc
    assert np.array_equal(back[5, 12:], np.zeros(6, dtype=np.uint64))  # identity: Z = 0


def _curve_point_outside_the_subgroup(rnd):
    while True:
        x = rnd.randrange(o.Q)
        rhs = (x * x * x + 4) % o.Q
        y = pow(rhs, (o.Q + 1) // 4, o.Q)
        if y * y % o.Q != rhs:
            continue
        # [r]P != O  <=>  [r-1]P != -P  (the oracle's scalar multiplication reduces its scalar mod r)
        if not o.g1_eq(o.g1_mul((x, y, 1), o.R - 1), o.g1_neg((x, y, 1))):
            return x, y


def _encode(x, y):
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if y > (o.Q - 1) // 2 else 0)
    return bytes(b)


def test_rejected_encodings(backend, golden):
    rnd = random.Random(91)
    good = [bytes.fromhex(h) for h in golden["srs"]["gp"]["tau_powers_g1"]]
    # x^3 + 4 not a square
    while True:
        x = rnd.randrange(o.Q)
        rhs = (x * x * x + 4) % o.Q
        if pow(rhs, (o.Q - 1) // 2, o.Q) != 1:
            break
    not_on_curve = _encode(x, 0)
    too_large = bytearray(o.Q.to_bytes(48, "big"))  # x = q
    too_large[0] |= 0x80
    uncompressed_flag = bytearray(good[3])
    uncompressed_flag[0] &= 0x7F
    xs, ys = _curve_point_outside_the_subgroup(rnd)
    outside = _encode(xs, ys)
    for bad in (not_on_curve, bytes(too_large), bytes(uncompressed_flag), outside):
        batch = good[:7] + [bad] + good[7:]
        with pytest.raises(plonk_amd.ReferencePanic):  # json.rs:30-33 `.expect("Failed to deserialize G1")`
            backend.g1_decompress(batch)
        assert backend.last_bad_index == 7
        with pytest.raises(plonk_amd.ReferencePanic):
            backend.srs_load_compressed(batch)
        assert backend.srs_size() == 0
    # without the subgroup test the curve point is accepted and decodes to itself
    pts = backend.g1_decompress([outside], check_subgroup=False)
    assert backend.g1_compress(pts) == [outside]
    # the sign bit selects the other root
    flipped = bytearray(good[1])
    flipped[0] ^= 0x20
    p, q = backend.g1_decompress([good[1], bytes(flipped)])
    assert same_point(q, c.g1_neg(p))
