"""Host side of the prover on limb arrays: `setup_proving_key`, `setup_verification_key` and
`prover::run` of the reference (plonk/src/setup_proving_key/mod.rs:32-103,
plonk/src/setup_verification_key/mod.rs:37-113, plonk/src/prover/mod.rs:18-74, part1-5.rs) with every
polynomial and every group operation done by the HIP library through `Backend`.

Polynomials and value vectors are (n, 4) uint64 arrays of Montgomery limbs (the memory layout of
`Vec<Fr>`) from end to end; Python integers appear only for the handful of scalars of a proof
(challenges, opened values, the two coefficients of v).  The reference hard-codes its 3-gate example;
this module takes the circuit as data in the same trace layout (gate l on rows 3l, 3l+1, 3l+2 of
Omega, public input i on row d-1-i, witnesses below them) for any d = 3 * 2^k or 2^k that divides
r - 1.  The example is `example_circuit()`.

Nothing here falls back to a CPU implementation: without the HIP library `Backend` cannot be
constructed."""
import hashlib

import numpy as np

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Q_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FR_GENERATOR = 7  # ark_bls12_381::Fr::GENERATOR (common/utils.rs:20-21)
_MONT = (1 << 256) % R_MOD
_MONT_INV = pow(_MONT, -1, R_MOD)


# ------------------------------------------------------------------------------ scalars <-> limbs
def to_mont(values):
    """Python integers -> (n, 4) Montgomery limbs."""
    raw = b"".join((v % R_MOD * _MONT % R_MOD).to_bytes(32, "little") for v in values)
    return np.frombuffer(raw, dtype=np.uint64).reshape(-1, 4).copy()


def from_mont(limbs):
    """(n, 4) or (4,) Montgomery limbs -> list of canonical Python integers."""
    raw = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 4).tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") * _MONT_INV % R_MOD for i in range(0, len(raw), 32)]


def _m(value):
    return to_mont([value])[0]


def trim(poly):
    """Drops trailing zero coefficients (ark's DensePolynomial invariant).  Scans from the top: the
    leading coefficient is almost always there already."""
    n = poly.shape[0]
    while n:
        lo = max(0, n - 4096)
        nz = np.flatnonzero(poly[lo:n].any(axis=1))
        if nz.size:
            return poly[: lo + int(nz[-1]) + 1]
        n = lo
    return poly[:0]


def root_of_unity(n):
    """omega of `construct_Omega(n)` (common/utils.rs:11-38)."""
    if n < 2 or (R_MOD - 1) % n:
        raise ValueError("k must divide r - 1 (multiplicative group order)")
    return pow(FR_GENERATOR, (R_MOD - 1) // n, R_MOD)


def deal_tasks(sizes, world):
    """Owner of each task when tasks of the given sizes (coefficients of the MSM) are dealt to `world`
    ranks: largest first, each to the rank with the least work so far.  The same on every rank."""
    load = [0] * world
    owner = [0] * len(sizes)
    for i in sorted(range(len(sizes)), key=lambda j: (-sizes[j], j)):
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += sizes[i]
    return owner


# What building a polynomial group costs next to a multi-scalar multiplication, in coefficients of an MSM of the
# same duration (measured at 2^20 gates on one GPU: about 8.4 ms per d coefficients of MSM; the gate quotient
# and the running product about 12 ms each, the wiring quotient 12 ms on top of the running product).
BUILD_COST_IN_D = {"q3": 1.4, "t": 1.4, "q4": 1.4}
BUILD_NEEDS = {"q3": ("q3",), "t": ("t",), "q4": ("t", "q4"), None: ()}


def deal_tasks_by_need(sizes, needs, world, d):
    """Owner of each task when the ranks also share the BUILDING of the polynomials: task i is an MSM over
    sizes[i] coefficients of a polynomial of group needs[i] (None: one every rank has anyway; "q3": the gate
    quotient; "t": the running product; "q4": the wiring quotient, which is built from t).  A rank builds only
    the groups its own tasks need.  Largest tasks first, each to the rank whose finishing time (MSMs so far +
    the builds it would have to add) stays lowest.  The same on every rank."""
    load = [0.0] * world
    built = [set() for _ in range(world)]
    owner = [0] * len(sizes)

    def added(rank, need):
        return sum(BUILD_COST_IN_D[g] * d for g in BUILD_NEEDS[need] if g not in built[rank])

    # tasks that need a group first (they pin where the groups are built), then the free ones
    order = sorted(range(len(sizes)), key=lambda j: (needs[j] is None, -sizes[j], j))
    for i in order:
        r = min(range(world), key=lambda k: (load[k] + added(k, needs[i]) + sizes[i], k))
        load[r] += added(r, needs[i]) + sizes[i]
        built[r].update(BUILD_NEEDS[needs[i]])
        owner[i] = r
    return owner


# ------------------------------------------------------------------------------ circuit description
class Circuit:
    """gates[l]: 1 addition, 0 multiplication; trace: (d, 4) limbs, the value on every row of Omega;
    wiring: (d,) integers, W(Omega[j]) = Omega[wiring[j]]; pub_inputs, output: integers."""

    def __init__(self, gates, trace, wiring, pub_inputs, output):
        self.gates = np.asarray(gates, dtype=np.uint8)
        self.trace = np.ascontiguousarray(trace, dtype=np.uint64).reshape(-1, 4)
        self.wiring = np.asarray(wiring, dtype=np.uint32)
        self.pub_inputs = [int(x) for x in pub_inputs]
        self.output = int(output)
        self.d = self.trace.shape[0]
        self.number_gates = self.gates.shape[0]
        if self.wiring.shape[0] != self.d or 3 * self.number_gates + len(self.pub_inputs) > self.d:
            raise ValueError("circuit does not fit its trace")


def example_circuit():
    """The reference's circuit (prover/part1.rs:16-59, setup_proving_key/mod.rs:45-95): inputs 5, 6,
    witness 1; 5 + 6 = 11, 6 + 1 = 7, 11 * 7 = 77."""
    d = 12
    src = [d - 2, 1, 3, d - 1, 0, 2, 6, d - 3, 4, 5, 7, 8]
    dst = [1, 3, d - 2, 0, d - 1, 6, 2, 4, d - 3, 7, 5, 8]
    wiring = [0] * d
    for a, b in zip(src, dst):
        wiring[a] = b
    return Circuit([1, 1, 0], to_mont([5, 6, 11, 6, 1, 7, 11, 7, 77, 1, 6, 5]), wiring, [5, 6], 77)


def synthetic_circuit(log_gates, seed, number_gates=None):
    """A random circuit of the example's family: 2^log_gates - 1 gates, 2 public inputs, 1 witness,
    d = 3 * 2^log_gates rows.  Every gate adds or multiplies two earlier values.  number_gates: fewer gates on
    the same rows (the rows between the last gate and the witness stay empty: value 0, wired to themselves)."""
    import random
    rnd = random.Random(seed)
    ng = (1 << log_gates) - 1
    d = 3 * ng + 3
    if number_gates is not None:
        if not 1 <= number_gates <= ng:
            raise ValueError("number_gates out of range")
        ng = number_gates
    trace = [0] * d
    values, positions = [], []
    for pos in (d - 1, d - 2, d - 3):
        v = rnd.randrange(1, R_MOD)
        trace[pos] = v
        values.append(v)
        positions.append([pos])
    gates = []
    for l in range(ng):
        ia = max(0, len(values) - 1 - rnd.randrange(4)) if l else 0
        ib = rnd.randrange(len(values)) if l else 1
        kind = rnd.randrange(2)
        out = (values[ia] + values[ib]) % R_MOD if kind else values[ia] * values[ib] % R_MOD
        gates.append(kind)
        for idx, pos in ((ia, 3 * l), (ib, 3 * l + 1)):
            trace[pos] = values[idx]
            positions[idx].append(pos)
        trace[3 * l + 2] = out
        values.append(out)
        positions.append([3 * l + 2])
    wiring = list(range(d))
    for plist in positions:
        for i, pos in enumerate(plist):
            wiring[pos] = plist[(i + 1) % len(plist)]
    return Circuit(gates, to_mont(trace), wiring, [trace[d - 1], trace[d - 2]], trace[3 * ng - 1])


# ------------------------------------------------------------------------------ the prover
class Prover:
    """One circuit size on one GPU.  `setup(tau)` stands for setup_global_params (the SRS stays on
    the device), `keys(circuit)` for setup_proving_key + setup_verification_key, `prove(circuit, keys)`
    for prover::run; the proof comes back as the dictionary `ProofJson` serialises
    (common/proof.rs:58-268)."""

    def __init__(self, backend, d):
        self.be = backend
        self.d = d
        self.w = root_of_unity(d)
        self.one = _m(1)
        self.zero = np.zeros(4, dtype=np.uint64)
        # Omega as limbs: the powers of w (f(wX) of the all-ones polynomial)
        self.omega = backend.compose_with_scaling(np.tile(self.one, (d, 1)), _m(self.w))
        self.kzg_log = None  # set to [] to record every commitment and opening of the next prove()
        self._buffer_cache = {}  # size in bytes -> free device buffers (see _Pool)
        self.challenges = {}

    # -- helpers -----------------------------------------------------------------------------
    def setup(self, tau):
        """`kzg_setup(2d - 3)` (setup_global_params/mod.rs:60-61: 21 for d = 12)."""
        self.be.kzg_setup(_m(tau), 2 * self.d - 3, want_points=False)

    def _transform(self, data, inverse):
        if self.d & (self.d - 1) == 0:
            return self.be.fr_ntt(data, inverse=inverse)
        return self.be.fr_ntt_mixed(data, inverse=inverse)

    def interpolate_over_omega(self, values):
        return trim(self._transform(values, True))

    def evaluate_over_omega(self, poly):
        padded = np.zeros((self.d, 4), dtype=np.uint64)
        padded[: poly.shape[0]] = poly
        return self._transform(padded, False)

    def lin(self, x, y, a, b, c=0, broadcast=False):
        return self.be.fr_linear_combination(x, y, _m(a), _m(b), _m(c), broadcast)

    def shift(self, poly, factor):
        """f(factor * X) (`compose_polynomials(f, [0, factor])`)."""
        return self.be.compose_with_scaling(poly, _m(factor))

    def mul(self, a, b):
        return trim(self.be.fr_poly_mul(a, b))

    def hex_of(self, point):
        return bytes(self.be.g1_compress(np.asarray(point).reshape(1, 18))[0]).hex()

    def challenge(self, points, count=None):
        """common/utils.rs:65-82 (one challenge) and :85-117 (several): Blake2s-256 over the compressed
        commitments, digests read little-endian mod r."""
        encodings = self.be.g1_compress(np.stack([np.asarray(p, dtype=np.uint64).reshape(18) for p in points]))
        return self.challenge_from_encodings(encodings, count)

    @staticmethod
    def challenge_from_encodings(encodings, count=None):
        h = hashlib.blake2s()
        for enc in encodings:
            h.update(bytes(enc))
        if count is None:
            return int.from_bytes(h.digest(), "little") % R_MOD
        base = h.digest()
        out = []
        for i in range(count):
            sub = hashlib.blake2s()
            sub.update(base)
            sub.update(bytes([i & 0xFF]))
            out.append(int.from_bytes(sub.digest(), "little") % R_MOD)
        return out

    def open(self, poly, point):
        v, proof = self.be.kzg_evaluate(poly, _m(point))
        v = from_mont(v)[0]
        self._log(poly, proof, point, v)
        return v, proof

    def open_many(self, poly, points):
        vs, proofs = self.be.kzg_evaluate_many(poly, to_mont(points))
        vs = from_mont(vs)
        for point, v, proof in zip(points, vs, proofs):
            self._log(poly, proof, point, v)
        return vs, proofs

    def commit(self, *polys):
        coms = self.be.kzg_commit_batch(list(polys))
        for poly, com in zip(polys, coms):
            self._log(poly, com)
        return coms

    def _log(self, poly, point_g1, u=None, v=None):
        if self.kzg_log is not None:
            self.kzg_log.append((poly, np.array(point_g1), u, v))

    def audit_with_tau(self, tau):
        """With the trapdoor of a test SRS every logged group element can be checked without
        pairings: a commitment must be [f(tau)]G (plonk/tests/kzg_tests.rs:36-69) and an opening
        proof [(f(tau) - v) / (tau - u)]G (:95-146).  Returns the number of items checked."""
        for poly, point, u, v in self.kzg_log:
            f_tau = from_mont(self.be.fr_eval(poly, _m(tau)))[0]
            scalar = f_tau if u is None else (f_tau - v) * pow(tau - u, -1, R_MOD) % R_MOD
            want = self.be.kzg_commit(to_mont([scalar]))  # [scalar]G: the SRS starts with G
            if self.hex_of(want) != self.hex_of(point):
                raise AssertionError("commitment or opening proof does not match the trapdoor evaluation")
        return len(self.kzg_log)

    def opening_checks(self, known=()):
        """(coms, us, vs, proofs) of every logged opening, as `kzg_verify_batch` takes them.  The commitment of
        an opened polynomial is the logged one when the prover committed to it, one of `known` =
        [(coefficients, commitment), ...] when its coefficients equal those (S and W come with the keys), and
        is computed otherwise."""
        committed = {id(poly): point for poly, point, u, _ in self.kzg_log if u is None}
        coms, us, vs, proofs = [], [], [], []
        for poly, point, u, v in self.kzg_log:
            if u is None:
                continue
            if id(poly) not in committed:
                for coeffs, com in known:
                    a, b = trim(np.asarray(poly)), trim(np.asarray(coeffs))
                    if a.shape == b.shape and np.array_equal(a, b):
                        committed[id(poly)] = np.array(com)
                        break
                else:
                    committed[id(poly)] = np.array(self.be.kzg_commit(poly))
            coms.append(committed[id(poly)])
            us.append(u)
            vs.append(v)
            proofs.append(point)
        if not coms:
            return np.zeros((0, 18), np.uint64), np.zeros((0, 4), np.uint64), np.zeros((0, 4), np.uint64), \
                np.zeros((0, 18), np.uint64)
        return np.stack(coms), to_mont(us), to_mont(vs), np.stack(proofs)

    def verify_openings(self, tau_g2, rho=None, known=()):
        """Every logged opening checked as the reference's verifier does, by pairings (kzg_verify,
        plonk/src/common/kzg.rs:91-98), all of them folded into one pairing-product check
        (`kzg_verify_batch`).  tau_g2: gp.tau_g2, 96 bytes.  Needs the public parameters only, no trapdoor."""
        from .backend import kzg_verify_batch
        return kzg_verify_batch(tau_g2, *self.opening_checks(known), rho)

    def check_identities(self, circuit_shape, proof, keys):
        """The field equations of the verifier (verify_zero_on_roots_test protocols.rs:220-229,
        verify_T_S_zero_test :789-803, verify_prescribed_permutation_check :702-705) on the opened
        values of a proof, with the challenges re-derived from its commitments.  circuit_shape:
        (number_gates, pub_inputs).  The pairing checks are not part of this."""
        ng, pub_inputs = circuit_shape
        d, w = self.d, self.w
        P = R_MOD
        # part 2
        rows_in = [pow(w, d - 1 - i, P) for i in range(len(pub_inputs))]
        v = trim(self.be.interpolate_polynomial(to_mont(rows_in), to_mont(pub_inputs)))
        com_T = self.be.g1_decompress([bytes.fromhex(proof["com_T"])], check_subgroup=True)[0]
        com_T_minus_v = self._subtract_points(com_T, self.be.kzg_commit(v))
        r2 = self.challenge([com_T_minus_v])
        _, f_r, _, q_r, _ = proof["proof_T_minus_v_zero"]
        z = 1
        for x in rows_in:
            z = z * (r2 - x) % P
        ok = int(f_r) % P == int(q_r) * z % P
        # part 3
        h = hashlib.blake2s()
        h.update(bytes.fromhex(proof["com_T"]))
        h.update(bytes.fromhex(self.hex_of(keys["com_S"])))
        r3 = int.from_bytes(h.digest(), "little") % P
        _, T_r, _, T_w_r, _, T_w2_r, _, S_r, _, q3_r, _ = proof["proof_T_S_zero"]
        T_r, T_w_r, T_w2_r, S_r, q3_r = (int(x) for x in (T_r, T_w_r, T_w2_r, S_r, q3_r))
        n = ng + 1
        if d == 3 * n:  # the gate rows are the subgroup of order n without its last element
            z = (pow(r3, n, P) - 1) * pow(r3 - pow(w, 3 * ng, P), -1, P) % P
        else:
            z = 1
            for l in range(ng):
                z = z * (r3 - pow(w, 3 * l, P)) % P
        ok = ok and (S_r * (T_r + T_w_r) + (1 - S_r) * T_r * T_w_r - T_w2_r) % P == q3_r * z % P
        # part 4
        h = hashlib.blake2s()
        h.update(bytes.fromhex(proof["com_T"]))
        h.update(bytes.fromhex(self.hex_of(keys["com_W"])))
        base = h.digest()
        r, s_, rp = (int.from_bytes(hashlib.blake2s(base + bytes([i])).digest(), "little") % P for i in range(3))
        pp = proof["proof_T_W_prescribed_permutation"]
        t_last, t_rp, t_w_rp, q_rp, f_w_rp, g_w_rp, W_w_rp = (int(pp[i]) for i in (2, 4, 6, 8, 10, 12, 14))
        ok = ok and t_last == 1
        ok = ok and (t_w_rp * (r - s_ * w * rp - g_w_rp) - t_rp * (r - s_ * W_w_rp - f_w_rp)) % P == \
            q_rp * (pow(rp, d, P) - 1) % P
        return ok

    def _subtract_points(self, p, q):
        """p - q for two Jacobian points in limb form (`com_T - com_v`, prover/part2.rs:49)."""
        neg = np.array(q, dtype=np.uint64).reshape(18)
        y = int.from_bytes(neg[6:12].tobytes(), "little")
        neg[6:12] = np.frombuffer(((Q_MOD - y) % Q_MOD).to_bytes(48, "little"), dtype=np.uint64)
        d_in = self.be.to_device(np.stack([np.asarray(p, dtype=np.uint64).reshape(18), neg]))
        d_out = self.be.malloc(144)
        self.be.g1_sum_dev(d_in, 2, d_out)
        self.be.sync()
        out = self.be.from_device(d_out, (18,))
        self.be.free(d_in)
        self.be.free(d_out)
        return out

    # -- setup_proving_key / setup_verification_key ------------------------------------------
    def selector(self, circuit):
        """S = interpolate(Omega_gates, gates), degree < number_gates (setup_proving_key/mod.rs:36-43).
        The gate rows Omega[3l] = (w^3)^l are a geometric progression: any number of them is interpolated in
        O(n log n) by `interpolate_geometric`.  When they are all of the subgroup generated by w^3 but one row
        (the family of the example) the missing value is the one that makes the top coefficient of the subgroup
        interpolant vanish, and one inverse transform gives S."""
        ng, d = circuit.number_gates, self.d
        n = ng + 1
        if d % 3 or d // 3 != n or n & (n - 1):
            values = np.zeros((ng, 4), dtype=np.uint64)
            values[circuit.gates != 0] = self.one
            return trim(self.be.interpolate_geometric(_m(pow(self.w, 3, R_MOD)), values))
        w3 = pow(self.w, 3, R_MOD)
        powers = self.be.compose_with_scaling(np.tile(self.one, (n, 1)), _m(w3))
        values = np.zeros((n, 4), dtype=np.uint64)
        values[:ng][circuit.gates != 0] = self.one
        # top coefficient = (1/n) sum_j y_j w3^j; solve for y_ng
        masked = np.zeros((n, 4), dtype=np.uint64)
        masked[:ng][circuit.gates != 0] = powers[:ng][circuit.gates != 0]
        total = from_mont(self.be.fr_prefix_sum(masked)[n - 1])[0]
        missing = (-total) * pow(pow(w3, ng, R_MOD), -1, R_MOD) % R_MOD
        values[ng] = _m(missing)
        s_poly = self.be.fr_ntt(values, inverse=True)
        assert not s_poly[n - 1].any()
        return trim(s_poly)

    def keys(self, circuit):
        """S, W and their commitments (setup_proving_key/mod.rs:32-103, setup_verification_key/mod.rs:
        37-113)."""
        S = self.selector(circuit)
        W = self.interpolate_over_omega(self.omega[circuit.wiring])
        com = self.be.kzg_commit_batch([S, W])
        return {"S": S, "W": W, "com_S": com[0], "com_W": com[1]}

    def _check_wiring(self, circuit):
        """T(W(y)) = T(y) on every row (the assertion of prover/part4.rs:36-42): rows joined by the wiring
        carry the same value.  Checked on the device (a gather over the trace)."""
        d_trace = self.be.to_device(circuit.trace)
        try:
            if not self.be.fr_respects_permutation_dev(d_trace, circuit.wiring):
                raise ValueError("T_W should be equal to T on Omega (prover/part4.rs:36-42)")
        finally:
            self.be.free(d_trace)

    def _gate_rows_are_a_subgroup_less_one(self, ng):
        n = ng + 1
        return self.d == 3 * n and n & (n - 1) == 0

    def _gate_quotient_host(self, S_T, ng):
        """S_T / Z_gates (prover/part3.rs:66, protocols.rs:169-181).  When the gate rows Omega[3l] are
        the subgroup of order n = ng + 1 without its last element rho, Z_gates (X - rho) = X^n - 1 and
        the quotient is S_T (X - rho) / (X^n - 1): two linear passes instead of a product tree and a
        series division.  The remainder must vanish, which is the reference's assertion that S_T
        cancels on every gate row (prover/part3.rs:61-64)."""
        be = self.be
        if not self._gate_rows_are_a_subgroup_less_one(ng):
            return trim(be.compute_q_zero_test_from_roots(self.omega[0:3 * ng:3], S_T))
        n, rho = ng + 1, pow(self.w, 3 * ng, R_MOD)
        shifted = np.zeros((S_T.shape[0] + 1, 4), dtype=np.uint64)
        shifted[1:] = S_T
        product = self.lin(shifted, S_T, 1, -rho)
        q = be.compute_q_zero_test(n, product)
        if self.lin(product[:n], q[:n], 1, 1).any():
            raise ValueError("S_T should cancel on y (prover/part3.rs:61-64)")
        return trim(q)

    # -- device-resident polynomials ------------------------------------------------------------
    class _Pool:
        """Device buffers of one proof: (pointer, length in field elements) pairs.  Released buffers go
        back to the prover's cache (by size) instead of the allocator: the next proof of the same circuit
        size reuses them, and nothing is allocated or freed while a proof runs."""

        def __init__(self, be, cache):
            self.be, self.cache, self.taken = be, cache, []

        def new(self, n):
            nbytes = max(n, 1) * 32
            free = self.cache.get(nbytes)
            ptr = free.pop() if free else self.be.malloc(nbytes)
            self.taken.append((ptr, nbytes))
            return (ptr, n)

        def put(self, host):
            host = np.ascontiguousarray(host, dtype=np.uint64).reshape(-1, 4)
            buf = self.new(host.shape[0])
            if host.shape[0]:
                self.be.to_device(host, buf[0])
            return buf

        def release(self):
            self.be.sync()
            for ptr, nbytes in self.taken:
                self.cache.setdefault(nbytes, []).append(ptr)
            self.taken = []

    def release_buffers(self):
        """Frees the device buffers kept from earlier proofs."""
        self.be.sync()
        for ptrs in self._buffer_cache.values():
            for ptr in ptrs:
                self.be.free(ptr)
        self._buffer_cache = {}

    def prove_resident(self, circuit, keys, times=None, ranks=None):
        """`prove` with every polynomial kept in device memory between the steps: the trace, S and W go
        up once, commitments, opened values and proofs come down; same proof, bit for bit.
        ranks: None, or an object with `world`, `rank` and `exchange(rows)` (plonk_amd.sharding.
        TaskExchange): one process per GPU; the 18 independent commitments and openings are dealt out
        together with the building of the polynomials they read (`deal_tasks_by_need`: a rank builds the
        gate quotient, the running product or the wiring quotient only if one of its own tasks needs it) and
        their results gathered."""
        import time
        clock = [time.perf_counter()]

        def lap(name):
            if times is not None:
                now = time.perf_counter()
                times[name] = now - clock[0]
                clock[0] = now

        be, d, w = self.be, self.d, self.w
        ng = circuit.number_gates
        pool = self._Pool(be, self._buffer_cache)
        pow2 = d & (d - 1) == 0

        def transform(buf, inverse):
            if pow2:
                be.fr_ntt_dev(buf[0], d.bit_length() - 1, 1, inverse)
            else:
                be.fr_ntt_mixed_dev(buf[0], d, 1, inverse)

        def lin(x, y, a, b, c=0, broadcast=False, out=None):
            x = x or (None, 0)
            y = y or (None, 0)
            out = out or pool.new(max(x[1], y[1]))
            be.fr_linear_combination_dev(x[0], x[1], y[0], y[1], _m(a), _m(b), _m(c), broadcast, out[0])
            return out

        def mul(a, b):
            out = pool.new(a[1] + b[1] - 1)
            be.fr_poly_mul_dev(a[0], a[1], b[0], b[1], out[0])
            return out

        def copy_padded(src, n):
            out = pool.new(n)
            if n > src[1]:
                be.memset_zero(out[0], n * 32)
            be.memcpy_d2d(out[0], src[0], min(n, src[1]) * 32)
            return out

        def shift(src, factor):
            out = copy_padded(src, src[1])
            be.fr_scale_by_powers_dev(out[0], out[1], _m(factor))
            return out

        def commit(*bufs):
            coms = be.kzg_commit_many_dev(list(bufs))
            for buf, com in zip(bufs, coms):
                self._log(buf, com)
            return coms

        def open_many(buf, points):
            vs, proofs = be.kzg_open_many_dev(buf[0], buf[1], to_mont(points))
            vs = from_mont(vs)
            for point, v, proof in zip(points, vs, proofs):
                self._log(buf, proof, point, v)
            return vs, proofs

        def top_is_zero(buf):
            return not be.from_device(buf[0] + (buf[1] - 1) * 32, (4,)).any()

        world, rank = (ranks.world, ranks.rank) if ranks is not None else (1, 0)
        try:
            S, W = pool.put(keys["S"]), pool.put(keys["W"])
            # part 1, and the one commitment of part 2 that the transcript needs before anything else
            T = pool.put(circuit.trace)
            if not be.fr_respects_permutation_dev(T[0], circuit.wiring):
                raise ValueError("T_W should be equal to T on Omega (prover/part4.rs:36-42)")
            transform(T, True)
            if top_is_zero(T):
                raise ValueError("T must have degree d - 1 (prover/part1.rs:63-65)")
            n_in = len(circuit.pub_inputs)
            rows_in = to_mont([pow(w, d - 1 - i, R_MOD) for i in range(n_in)])
            v = pool.put(trim(be.interpolate_polynomial(rows_in, to_mont(circuit.pub_inputs))))
            com_T, com_v = commit(T, v)
            # every challenge depends only on com_T, com_v and the keys (prover/part2.rs:55, part3.rs:69,
            # part4.rs:48): all other commitments and all openings are independent of each other
            enc = be.g1_compress(np.stack([self._subtract_points(com_T, com_v), com_T, keys["com_S"], keys["com_W"]]))
            r2 = self.challenge_from_encodings(enc[0:1])
            r3 = self.challenge_from_encodings([enc[1], enc[2]])
            r, s, rp = self.challenge_from_encodings([enc[1], enc[3]], 3)
            self.challenges = {"r2": r2, "r3": r3, "r": r, "s": s, "rp": rp}
            lap("part1")

            # The polynomials of parts 2, 3 and 4, in groups that are built only where they are needed: one
            # process builds all of them; several processes deal the tasks out so that the gate quotient (q3),
            # the running product (t) and the wiring quotient (q4) are each built by the few ranks whose tasks
            # read them (deal_tasks_by_need), and the other ranks start their multi-scalar multiplications at once.
            T_minus_v = lin(T, v, 1, -1)
            q2 = pool.new(d - n_in)
            be.fr_div_by_roots_dev(T_minus_v[0], d, pool.put(rows_in)[0], n_in, q2[0])
            n_q3 = (ng - 1) + 2 * (d - 1) + 1 - ng   # coefficients of S_T / Z_gates
            n_q4 = 2 * (d - 1) + 1 - d               # coefficients of t1 / (X^d - 1)
            wr = w * r3 % R_MOD
            # (polynomial, its length, the group it belongs to, point or None for a commitment)
            plan = [("q2", d - n_in, None, None), ("T_minus_v", d, None, r2), ("q2", d - n_in, None, r2),
                    ("q3", n_q3, "q3", None), ("T", d, None, r3), ("T", d, None, wr), ("T", d, None, w * wr % R_MOD),
                    ("S", S[1], None, r3), ("q3", n_q3, "q3", r3),
                    ("t", d, "t", None), ("q4", n_q4, "q4", None), ("t", d, "t", pow(w, d - 1, R_MOD)), ("t", d, "t", rp),
                    ("t", d, "t", rp * w % R_MOD), ("q4", n_q4, "q4", rp),
                    ("T", d, None, rp * w % R_MOD), ("W", W[1], None, rp * w % R_MOD), ("T", d, None, pow(w, 3 * ng - 1, R_MOD))]
            if world > 1:
                owner = deal_tasks_by_need([p[1] for p in plan], [p[2] for p in plan], world, d)
            else:
                owner = [0] * len(plan)
            mine = [i for i in range(len(plan)) if owner[i] == rank]
            wanted = set()
            for i in mine:
                wanted.update(BUILD_NEEDS[plan[i][2]])
            polys = {"q2": q2, "T_minus_v": T_minus_v, "T": T, "S": S, "W": W}
            failure = None
            try:
                if "q3" in wanted:
                    T_w = shift(T, w)
                    T_w2 = shift(T, w * w % R_MOD)
                    one_minus_S = lin(S, None, -1, 0, 1)
                    S_T = lin(mul(S, lin(T, T_w, 1, 1)), mul(one_minus_S, mul(T, T_w)), 1, 1)
                    lin(S_T, T_w2, 1, -1, out=S_T)
                    q3 = pool.new(S_T[1] - ng)
                    if self._gate_rows_are_a_subgroup_less_one(ng):
                        # Z_gates (X - rho) = X^n - 1: multiply by the missing factor, divide by X^n - 1
                        n_sub, rho = ng + 1, pow(w, 3 * ng, R_MOD)
                        shifted = pool.new(S_T[1] + 1)
                        be.memset_zero(shifted[0], 32)
                        be.memcpy_d2d(shifted[0] + 32, S_T[0], S_T[1] * 32)
                        lin(shifted, S_T, 1, -rho, out=shifted)
                        be.fr_div_vanishing_dev(shifted[0], shifted[1], n_sub, q3[0])
                        rem = lin((shifted[0], n_sub), (q3[0], min(n_sub, q3[1])), 1, 1)
                        if not be.fr_is_zero_dev(rem[0], rem[1]):
                            raise ValueError("S_T should cancel on y (prover/part3.rs:61-64)")
                    else:
                        rows_gates = pool.put(self.omega[0:3 * ng:3])
                        be.fr_div_by_roots_dev(S_T[0], S_T[1], rows_gates[0], ng, q3[0])
                    assert q3[1] == n_q3
                    polys["q3"] = q3
                if "t" in wanted:
                    T_vals = copy_padded(T, d)
                    transform(T_vals, False)
                    W_vals = copy_padded(W, d)
                    transform(W_vals, False)
                    omega = pool.put(self.omega)
                    num = lin(W_vals, T_vals, -s, -1, r, broadcast=True)
                    den = lin(omega, T_vals, -s, -1, r, broadcast=True)
                    be.fr_ratio_prefix_product_dev(num[0], den[0], d)
                    t = num
                    transform(t, True)
                    if top_is_zero(t):
                        raise NotImplementedError(
                            "extra random interpolation points (protocols.rs:589-600) are not drawn here")
                    polys["t"] = t
                if "q4" in wanted:
                    T_w = shift(T, w)
                    t_w = shift(t, w)
                    left = lin(pool.put(to_mont([r, (-s * w) % R_MOD])), T_w, 1, -1)
                    right = lin(shift(W, w), T_w, -s, -1, r)
                    t1 = mul(t_w, left)
                    lin(t1, mul(t, right), 1, -1, out=t1)
                    q4 = pool.new(t1[1] - d)
                    be.fr_div_vanishing_dev(t1[0], t1[1], d, q4[0])
                    rem = lin((t1[0], d), (q4[0], min(d, q4[1])), 1, 1)  # t1 = q4 X^d - q4 + remainder
                    if not be.fr_is_zero_dev(rem[0], rem[1]):
                        raise ValueError("t1 must be divisible by Z_Omega (prover/part4.rs:58)")
                    assert q4[1] == n_q4
                    polys["q4"] = q4
            except (ValueError, NotImplementedError) as exc:
                if world == 1:
                    raise
                failure = exc  # the other ranks wait in the exchange: tell them there, then raise
            be.sync()
            lap("polynomials")

            # 4 commitments and 14 openings (T at w rp serves f and g)
            tasks = [(polys.get(name), point) for name, _, _, point in plan]
            rows = np.zeros((len(tasks) + world, 22), dtype=np.uint64)
            if mine and failure is None:  # one call: the multi-scalar multiplications are queued back to back
                vals, points = be.kzg_commit_open_many_dev(
                    [(tasks[i][0][0], tasks[i][0][1], None if tasks[i][1] is None else _m(tasks[i][1])) for i in mine])
                for k, i in enumerate(mine):
                    rows[i, :18] = points[k]
                    rows[i, 18:] = vals[k]
                    buf, point = tasks[i]
                    self._log(buf, points[k], point, None if point is None else from_mont(vals[k])[0])
            lap("commit_open")
            if world > 1:
                rows[len(tasks) + rank, 0] = 0 if failure is None else 1
                rows = ranks.exchange(rows, owner + list(range(world)))
                lap("exchange")
                if failure is not None:
                    raise failure
                bad = [k for k in range(world) if rows[len(tasks) + k, 0]]
                if bad:
                    raise ValueError("the witness failed a consistency check on rank %d" % bad[0])
            g1 = rows[:, :18]
            val = from_mont(rows[:, 18:])
            if val[17] != circuit.output % R_MOD:
                raise ValueError("Evaluation does not match with output")
            p2 = {"com_q": g1[0], "f_r": val[1], "proof_f_r": g1[1], "q_r": val[2], "proof_q_r": g1[2]}
            p3 = {"com_q": g1[3], "T_r": val[4], "proof_T_r": g1[4], "T_w_r": val[5], "proof_T_w_r": g1[5],
                  "T_w2_r": val[6], "proof_T_w2_r": g1[6], "S_r": val[7], "proof_S_r": g1[7], "q_r": val[8],
                  "proof_q_r": g1[8]}
            p4 = {"com_t": g1[9], "com_q": g1[10], "t_w_k_minus_1": val[11], "proof_t_w_k_minus_1": g1[11],
                  "t_rp": val[12], "proof_t_rp": g1[12], "t_w_rp": val[13], "proof_t_w_rp": g1[13],
                  "q_rp": val[14], "proof_q_rp": g1[14], "f_w_rp": val[15], "proof_f_w_rp": g1[15],
                  "g_w_rp": val[15], "proof_g_w_rp": g1[15], "W_w_rp": val[16], "proof_W_w_rp": g1[16]}
            out_proof = g1[17]
            if self.kzg_log is not None:  # the audit reads the polynomials after the buffers are gone
                host = {}  # one host array per device buffer: entries of one polynomial share it
                for buf, _, _, _ in self.kzg_log:
                    if (buf[0], buf[1]) not in host:
                        host[(buf[0], buf[1])] = be.from_device(buf[0], (buf[1], 4))
                self.kzg_log = [(host[(buf[0], buf[1])], g, u, v) for buf, g, u, v in self.kzg_log]
        finally:
            pool.release()
        return self._proof_json(circuit, com_T, p2, p3, p4, out_proof)

    # -- prover::run ---------------------------------------------------------------------------
    def prove(self, circuit, keys, times=None):
        """times: optional dict that receives the wall time of each part in seconds."""
        import time
        clock = [time.perf_counter()]

        def lap(name):
            if times is not None:
                now = time.perf_counter()
                times[name] = now - clock[0]
                clock[0] = now

        be, d, w = self.be, self.d, self.w
        ng = circuit.number_gates
        S, W = keys["S"], keys["W"]
        # part 1 (prover/part1.rs:9-69)
        self._check_wiring(circuit)
        T = self.interpolate_over_omega(circuit.trace)
        if T.shape[0] != d:
            raise ValueError("T must have degree d - 1 (prover/part1.rs:63-65)")
        com_T = self.commit(T)[0]
        lap("part1")

        # part 2 (prover/part2.rs:14-61): T - v vanishes on the public-input rows
        n_in = len(circuit.pub_inputs)
        rows_in = [pow(w, d - 1 - i, R_MOD) for i in range(n_in)]
        v = trim(be.interpolate_polynomial(to_mont(rows_in), to_mont(circuit.pub_inputs)))
        T_minus_v = trim(self.lin(T, v, 1, -1))
        com_v = self.commit(v)[0]
        com_T_minus_v = self._subtract_points(com_T, com_v)
        q2 = trim(be.compute_q_zero_test_from_roots(to_mont(rows_in), T_minus_v))
        r2 = self.challenge([com_T_minus_v])
        p2 = {"com_q": self.commit(q2)[0]}
        p2["f_r"], p2["proof_f_r"] = self.open(T_minus_v, r2)
        p2["q_r"], p2["proof_q_r"] = self.open(q2, r2)
        lap("part2")

        # part 3 (prover/part3.rs:16-75, protocols.rs:722-786): the gate constraint
        T_w = self.shift(T, w)
        T_w2 = self.shift(T, w * w % R_MOD)
        one_minus_S = self.lin(S, S[:0], -1, 0, 1)
        S_T = self.lin(self.lin(self.mul(S, self.lin(T, T_w, 1, 1)),
                                self.mul(one_minus_S, self.mul(T, T_w)), 1, 1), T_w2, 1, -1)
        S_T = trim(S_T)
        q3 = self._gate_quotient_host(S_T, ng)
        r3 = self.challenge([com_T, keys["com_S"]])
        p3 = {"com_q": self.commit(q3)[0]}
        vs, proofs = self.open_many(T, [r3, w * r3 % R_MOD, w * w % R_MOD * r3 % R_MOD])
        p3["T_r"], p3["T_w_r"], p3["T_w2_r"] = vs
        p3["proof_T_r"], p3["proof_T_w_r"], p3["proof_T_w2_r"] = proofs
        p3["S_r"], p3["proof_S_r"] = self.open(S, r3)
        p3["q_r"], p3["proof_q_r"] = self.open(q3, r3)
        lap("part3")

        # part 4 (prover/part4.rs:17-66, protocols.rs:559-687): the wiring, T(W(y)) = T(y) on Omega
        r, s, rp = self.challenge([com_T, keys["com_W"]], 3)
        self.challenges = {"r2": r2, "r3": r3, "r": r, "s": s, "rp": rp}
        T_vals = self.evaluate_over_omega(T)
        W_vals = self.evaluate_over_omega(W)
        num = self.lin(W_vals, T_vals, -s, -1, r, broadcast=True)   # r - s W(x) - f(x)
        den = self.lin(self.omega, T_vals, -s, -1, r, broadcast=True)  # r - s x - g(x)
        t = self.interpolate_over_omega(be.fr_ratio_prefix_product(num, den))
        if t.shape[0] != d:
            raise NotImplementedError("extra random interpolation points (protocols.rs:589-600) are not drawn here")
        t_w = self.shift(t, w)
        left = self.lin(to_mont([r, (-s * w) % R_MOD]), T_w, 1, -1)      # r - s w X - g(wX)
        right = self.lin(self.shift(W, w), T_w, -s, -1, r)                 # r - s W(wX) - f(wX)
        t1 = trim(self.lin(self.mul(t_w, left), self.mul(t, right), 1, -1))
        q4 = trim(be.compute_q_zero_test(d, t1))
        if self.lin(t1[:d], q4[:d], 1, 1).any():
            raise ValueError("t1 must be divisible by Z_Omega (prover/part4.rs:58)")
        p4 = {}
        com = self.commit(t, q4)
        p4["com_t"], p4["com_q"] = com[0], com[1]
        vs, proofs = self.open_many(t, [pow(w, d - 1, R_MOD), rp, rp * w % R_MOD])
        p4["t_w_k_minus_1"], p4["t_rp"], p4["t_w_rp"] = vs
        p4["proof_t_w_k_minus_1"], p4["proof_t_rp"], p4["proof_t_w_rp"] = proofs
        p4["q_rp"], p4["proof_q_rp"] = self.open(q4, rp)
        # f = g = T: the reference opens T at w rp twice (prover/part4.rs:63, protocols.rs:663-665)
        p4["f_w_rp"], p4["proof_f_w_rp"] = self.open(T, rp * w % R_MOD)
        p4["g_w_rp"], p4["proof_g_w_rp"] = p4["f_w_rp"], p4["proof_f_w_rp"]
        p4["W_w_rp"], p4["proof_W_w_rp"] = self.open(W, rp * w % R_MOD)
        lap("part4")

        # part 5 (prover/part5.rs:6-27): the output sits on the last gate's output row
        out_v, out_proof = self.open(T, pow(w, 3 * ng - 1, R_MOD))
        if out_v != circuit.output % R_MOD:
            raise ValueError("Evaluation does not match with output")
        lap("part5")

        return self._proof_json(circuit, com_T, p2, p3, p4, out_proof)

    def _proof_json(self, circuit, com_T, p2, p3, p4, out_proof):
        """Field order of common/proof.rs:220-266.  All group elements are compressed in one call (one
        field inversion per point, side by side)."""
        s_ = str
        points = [com_T, p2["com_q"], p2["proof_f_r"], p2["proof_q_r"],
                  p3["com_q"], p3["proof_T_r"], p3["proof_T_w_r"], p3["proof_T_w2_r"], p3["proof_S_r"], p3["proof_q_r"],
                  p4["com_t"], p4["com_q"], p4["proof_t_w_k_minus_1"], p4["proof_t_rp"], p4["proof_t_w_rp"],
                  p4["proof_q_rp"], p4["proof_f_w_rp"], p4["proof_g_w_rp"], p4["proof_W_w_rp"], out_proof]
        hx = [bytes(e).hex() for e in
              self.be.g1_compress(np.stack([np.asarray(p, dtype=np.uint64).reshape(18) for p in points]))]
        return {
            "pub_inputs": [s_(x % R_MOD) for x in circuit.pub_inputs],
            "output": s_(circuit.output % R_MOD),
            "com_T": hx[0],
            "proof_T_minus_v_zero": [hx[1], s_(p2["f_r"]), hx[2], s_(p2["q_r"]), hx[3]],
            "proof_T_S_zero": [hx[4], s_(p3["T_r"]), hx[5], s_(p3["T_w_r"]), hx[6], s_(p3["T_w2_r"]), hx[7],
                               s_(p3["S_r"]), hx[8], s_(p3["q_r"]), hx[9]],
            "proof_T_W_prescribed_permutation": [
                hx[10], hx[11], s_(p4["t_w_k_minus_1"]), hx[12], s_(p4["t_rp"]), hx[13], s_(p4["t_w_rp"]), hx[14],
                s_(p4["q_rp"]), hx[15], s_(p4["f_w_rp"]), hx[16], s_(p4["g_w_rp"]), hx[17], s_(p4["W_w_rp"]), hx[18]],
            "proof_last_gate_KZG": hx[19],
        }
