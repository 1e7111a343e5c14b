"""Host-only parts of plonk_amd/prover.py (no GPU): limb conversions, trimming, the circuit generators
against the restatement's, the example's literals."""
import numpy as np
import pytest

from oracle import capi as c
from oracle import pyref as o
from plonk_amd import prover as pv


def test_montgomery_conversions_match_the_oracle():
    vals = [0, 1, 2, o.R - 1, o.R, o.R + 5, 0x123456789ABCDEF << 100]
    limbs = pv.to_mont(vals)
    assert np.array_equal(limbs, c.fr_to_mont([v % o.R for v in vals]))
    assert pv.from_mont(limbs) == [v % o.R for v in vals]
    assert pv.from_mont(limbs[3]) == [o.R - 1]


def test_trim_drops_only_trailing_zeros():
    a = pv.to_mont([0, 5, 0, 7, 0, 0])
    assert pv.trim(a).shape[0] == 4 and pv.trim(a[:1]).shape[0] == 0 and pv.trim(a[:0]).shape[0] == 0
    big = np.zeros((10000, 4), dtype=np.uint64)
    assert pv.trim(big).shape[0] == 0
    big[17] = pv.to_mont([3])[0]
    assert pv.trim(big).shape[0] == 18


def test_root_of_unity_is_construct_omega():
    for n in (2, 12, 24, 1 << 10, 3 << 20):
        assert pv.root_of_unity(n) == pow(o.FR_GENERATOR, (o.R - 1) // n, o.R)
    assert pv.root_of_unity(12) == o.construct_Omega(12)[1]
    with pytest.raises(ValueError):
        pv.root_of_unity(7)


def test_circuit_generators_match_the_restatement():
    ex, oex = pv.example_circuit(), o.example_circuit()
    assert pv.from_mont(ex.trace) == oex.trace and list(ex.wiring) == oex.wiring and list(ex.gates) == oex.gates
    assert (ex.pub_inputs, ex.output, ex.d, ex.number_gates) == ([5, 6], 77, 12, 3)
    for k in (2, 3, 6):
        a, b = pv.synthetic_circuit(k, 33), o.synthetic_circuit(k, 33)
        assert pv.from_mont(a.trace) == b.trace and list(a.wiring) == b.wiring and list(a.gates) == b.gates
        assert a.pub_inputs == b.pub_inputs and a.output == b.output and a.d == 3 << k
    with pytest.raises(ValueError):
        pv.Circuit([1] * 5, pv.to_mont([1] * 12), list(range(12)), [1, 2], 3)


def test_tasks_are_dealt_by_size():
    sizes = [10, 3, 3, 20, 10, 10, 1]
    assert pv.deal_tasks(sizes, 1) == [0] * 7
    owner = pv.deal_tasks(sizes, 3)
    load = [sum(s for s, o_ in zip(sizes, owner) if o_ == r) for r in range(3)]
    assert sorted(load) == [17, 20, 20] and owner[3] == 0
    assert pv.deal_tasks(sizes, 3) == owner  # deterministic: every rank computes the same deal


def test_dealing_tasks_together_with_the_polynomials_they_read():
    """Sharing a proof between ranks: every task gets an owner, the deal is the same on every rank, one rank
    takes everything, and from four ranks on no rank builds all three polynomial groups (the point of dealing
    by need: the replicated building was about half of an 8-rank proof's latency)."""
    d = 3 << 10
    sizes = [d - 2, d, d - 2, 2 * d, d, d, d, d // 3, 2 * d, d, d - 1, d, d, d, d - 1, d, d, d]
    needs = [None, None, None, "q3", None, None, None, None, "q3", "t", "q4", "t", "t", "t", "q4", None, None, None]
    assert pv.deal_tasks_by_need(sizes, needs, 1, d) == [0] * 18
    for world in (2, 3, 4, 8):
        owner = pv.deal_tasks_by_need(sizes, needs, world, d)
        assert owner == pv.deal_tasks_by_need(sizes, needs, world, d)
        assert len(owner) == 18 and set(owner) <= set(range(world))
        built = [set() for _ in range(world)]
        load = [0.0] * world
        for i, r in enumerate(owner):
            built[r].update(pv.BUILD_NEEDS[needs[i]])
            load[r] += sizes[i] / d
        finish = [load[r] + sum(pv.BUILD_COST_IN_D[g] for g in built[r]) for r in range(world)]
        replicated = sum(sizes) / d / world + sum(pv.BUILD_COST_IN_D.values())
        assert max(finish) <= replicated + 0.35, (world, finish, replicated)
        if world >= 4:
            assert all(len(b) < 3 for b in built)
            assert max(finish) < 0.85 * replicated
