"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- Python big-integer restatement of the
reference KZG / polynomial path and of the prover that drives it.

This module is the checker, never the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  Nothing under plonk_amd/ imports it.

Parity status: PINNED for the example circuit.  `reproduce_verification_key` and
`reproduce_proof` regenerate the reference's committed plonk/data/verification_key.json
and plonk/data/proof.json byte for byte from plonk/data/srs.json and
plonk/data/proving_key.json (copies kept under tests/golden/).  At larger sizes no
reference-produced vector exists: results there are pinned by this restatement plus
size-independent algebraic known answers (commit == [f(tau)]G).

The arithmetic of the reference lives in arkworks 0.5 (ark-ff / ark-ec / ark-poly /
ark-bls12-381 / ark-serialize, caret "0.5.0", plonk/Cargo.toml:14-19) and blake2 0.10.6;
none of those sources is in /root/reference, so their published algorithms are restated
here: BLS12-381 (y^2 = x^3 + 4 over Fq, scalar field Fr, multiplicative generator 7 of
Fr), the zcash-style compressed point encoding, dense univariate polynomials with
trailing zero coefficients stripped, and Blake2s-256.

Every function cites the reference file:line it follows (paths relative to the
reference root).  Elements of Fr and Fq are plain Python ints in canonical (non
Montgomery) form; G1 points are Jacobian triples (X, Y, Z) with Z == 0 for the identity,
as ark's G1Projective.
"""
import hashlib

# ---------------------------------------------------------------------------
# BLS12-381 constants (standard; r-1 limbs are pinned by plonk/tests/field_tests.rs:7-17,
# the generator encoding by plonk/data/srs.json:8).
# ---------------------------------------------------------------------------
Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
FR_GENERATOR = 7          # ark_bls12_381::Fr::GENERATOR
FR_TWO_ADICITY = 32
CURVE_B = 4
G1_GEN_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_GEN_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G1_GENERATOR = (G1_GEN_X, G1_GEN_Y, 1)
G1_IDENTITY = (1, 1, 0)   # ark's G1Projective::zero()


# ---------------------------------------------------------------------------
# Fr helpers
# ---------------------------------------------------------------------------
def fr(x):
    return x % R


def fr_inv(x):
    x %= R
    if x == 0:
        # ark: Field::inverse() -> None, callers .unwrap() (polynomials.rs:42, protocols.rs:349,583)
        raise ZeroDivisionError("inverse of zero in Fr")
    return pow(x, -1, R)


def fr_to_string(x):
    """ark-ff Display of a prime-field element: canonical integer in decimal
    (used by setup_proving_key/mod.rs:27-28, common/proof.rs:62-63,226...)."""
    return str(x % R)


def fr_from_le_bytes_mod_order(b):
    """ark_ff::PrimeField::from_le_bytes_mod_order (common/utils.rs:81,112)."""
    return int.from_bytes(b, "little") % R


# ---------------------------------------------------------------------------
# G1 (Jacobian coordinates over Fq)
# ---------------------------------------------------------------------------
def g1_is_identity(P):
    return P[2] % Q == 0


def g1_double(P):
    X1, Y1, Z1 = P
    if Z1 % Q == 0:
        return G1_IDENTITY
    # dbl-2009-l (a = 0)
    A = X1 * X1 % Q
    B = Y1 * Y1 % Q
    C = B * B % Q
    D = 2 * ((X1 + B) * (X1 + B) - A - C) % Q
    E = 3 * A % Q
    F = E * E % Q
    X3 = (F - 2 * D) % Q
    Y3 = (E * (D - X3) - 8 * C) % Q
    Z3 = 2 * Y1 * Z1 % Q
    return (X3, Y3, Z3)


def g1_add(P, S):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = S
    if Z1 % Q == 0:
        return S
    if Z2 % Q == 0:
        return P
    Z1Z1 = Z1 * Z1 % Q
    Z2Z2 = Z2 * Z2 % Q
    U1 = X1 * Z2Z2 % Q
    U2 = X2 * Z1Z1 % Q
    S1 = Y1 * Z2 * Z2Z2 % Q
    S2 = Y2 * Z1 * Z1Z1 % Q
    if U1 == U2:
        if S1 == S2:
            return g1_double(P)
        return G1_IDENTITY
    H = (U2 - U1) % Q
    Rr = (S2 - S1) % Q
    HH = H * H % Q
    HHH = H * HH % Q
    V = U1 * HH % Q
    X3 = (Rr * Rr - HHH - 2 * V) % Q
    Y3 = (Rr * (V - X3) - S1 * HHH) % Q
    Z3 = Z1 * Z2 * H % Q
    return (X3, Y3, Z3)


def g1_neg(P):
    return (P[0], (-P[1]) % Q, P[2])


def g1_sub(P, S):
    return g1_add(P, g1_neg(S))


def g1_mul(P, k):
    """`G1 * Fr`: the scalar is taken out of Montgomery form and used as an integer
    (common/kzg.rs:34,56,83).  Plain double-and-add, most significant bit first."""
    k %= R
    acc = G1_IDENTITY
    for bit in bin(k)[2:] if k else "":
        acc = g1_double(acc)
        if bit == "1":
            acc = g1_add(acc, P)
    return acc


def g1_to_affine(P):
    """Returns (x, y) or None for the identity."""
    X, Y, Z = P
    if Z % Q == 0:
        return None
    zi = pow(Z, -1, Q)
    zi2 = zi * zi % Q
    return (X * zi2 % Q, Y * zi2 * zi % Q)


def g1_eq(P, S):
    """Projective equality, as ark's PartialEq for G1Projective."""
    a, b = g1_to_affine(P), g1_to_affine(S)
    return a == b


def g1_on_curve(P):
    a = g1_to_affine(P)
    if a is None:
        return True
    x, y = a
    return (y * y - x * x * x - CURVE_B) % Q == 0


def g1_compress(P):
    """ark-bls12-381's `serialize_compressed` for G1: 48 bytes, big-endian x, flag bits in
    the most significant byte -- bit 7 compressed, bit 6 infinity, bit 5 set when y is the
    lexicographically larger of {y, -y} (common/utils.rs:70-72, common/proof.rs:65...).
    Pinned by plonk/data/srs.json:8 (the standard generator encoding 97f1d3a7...)."""
    a = g1_to_affine(P)
    if a is None:
        out = bytearray(48)
        out[0] = 0xC0
        return bytes(out)
    x, y = a
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= 0x80
    if y > (Q - 1) // 2:
        out[0] |= 0x20
    return bytes(out)


def g1_decompress(b):
    """Inverse of g1_compress (setup_global_params/json.rs:22-48 via
    `G1::deserialize_compressed`): y = sqrt(x^3 + 4), q = 3 mod 4."""
    b = bytes(b)
    assert len(b) == 48
    flags = b[0] >> 5
    assert flags & 4, "uncompressed encodings are not used by the reference"
    if flags & 2:
        return G1_IDENTITY
    x = int.from_bytes(b, "big") & ((1 << 381) - 1)
    assert x < Q
    rhs = (x * x * x + CURVE_B) % Q
    y = pow(rhs, (Q + 1) // 4, Q)
    assert y * y % Q == rhs, "x is not on the curve"
    if (y > (Q - 1) // 2) != bool(flags & 1):
        y = Q - y
    return (x, y, 1)


def g1_hex(P):
    return g1_compress(P).hex()


# ---------------------------------------------------------------------------
# Dense univariate polynomials over Fr: list of ints, lowest degree first, with the
# ark-poly DensePolynomial invariant "no trailing zero coefficient" (zero poly = []).
# ---------------------------------------------------------------------------
def poly_trim(c):
    c = [x % R for x in c]
    while c and c[-1] == 0:
        c.pop()
    return c


def poly_degree(c):
    """ark-poly `Polynomial::degree`: 0 for the zero polynomial."""
    return 0 if not c else len(c) - 1


def poly_add(a, b):
    n = max(len(a), len(b))
    return poly_trim([(a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0) for i in range(n)])


def poly_sub(a, b):
    n = max(len(a), len(b))
    return poly_trim([(a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0) for i in range(n)])


def poly_neg(a):
    return poly_trim([-x for x in a])


def poly_scale(a, s):
    return poly_trim([x * s for x in a])


def poly_mul(a, b):
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] += x * y
    return poly_trim(out)


def poly_divmod(a, b):
    """`DenseOrSparsePolynomial::divide_with_q_and_r` (common/kzg.rs:74-76,
    common/protocols.rs:162-163,177-178): schoolbook long division, quotient and remainder."""
    a = poly_trim(a)
    b = poly_trim(b)
    if not b:
        raise ZeroDivisionError("division by the zero polynomial")
    if len(a) < len(b):
        return [], a
    q = [0] * (len(a) - len(b) + 1)
    rem = a[:]
    lead_inv = fr_inv(b[-1])
    while rem and len(rem) >= len(b):
        coef = rem[-1] * lead_inv % R
        shift = len(rem) - len(b)
        q[shift] = coef
        for i, y in enumerate(b):
            rem[shift + i] = (rem[shift + i] - coef * y) % R
        rem = poly_trim(rem)
    return poly_trim(q), rem


def poly_eval(c, x):
    """`Polynomial::evaluate` (Horner)."""
    acc = 0
    for coef in reversed(c):
        acc = (acc * x + coef) % R
    return acc


# ---------------------------------------------------------------------------
# common/utils.rs
# ---------------------------------------------------------------------------
def construct_Omega(k):
    """common/utils.rs:11-38: omega = GENERATOR^((r-1)/k); Omega = [omega^0 .. omega^(k-1)]."""
    assert k > 1, "k must be at least 2"
    assert (R - 1) % k == 0, "k must divide r - 1 (multiplicative group order)"
    omega = pow(FR_GENERATOR, (R - 1) // k, R)
    return [pow(omega, i, R) for i in range(k)]


def construct_vanishing_polynomial(k):
    """common/utils.rs:41-48: X^k - 1."""
    return [R - 1] + [0] * (k - 1) + [1]


def construct_vanishing_polynomial_from_roots(roots):
    """common/utils.rs:51-62: prod (X - root)."""
    z = [1]
    for root in roots:
        z = poly_mul(z, [(-root) % R, 1])
    return z


def derive_challenge_from_commitments(commitments):
    """common/utils.rs:65-82: Blake2s-256 over the compressed encodings, digest read as a
    little-endian integer mod r."""
    h = hashlib.blake2s()
    for c in commitments:
        h.update(g1_compress(c))
    return fr_from_le_bytes_mod_order(h.digest())


def derive_multiple_challenges_from_commitments(commitments, num_challenges):
    """common/utils.rs:85-117: base hash, then Blake2s(base || [i]) per challenge."""
    h = hashlib.blake2s()
    for c in commitments:
        h.update(g1_compress(c))
    base = h.digest()
    out = []
    for i in range(num_challenges):
        sub = hashlib.blake2s()
        sub.update(base)
        sub.update(bytes([i & 0xFF]))
        out.append(fr_from_le_bytes_mod_order(sub.digest()))
    return out


# ---------------------------------------------------------------------------
# common/kzg.rs
# ---------------------------------------------------------------------------
class CommitError(Exception):
    """common/kzg.rs:11-15 CommitError::CommitFailed."""


class GlobalParameters:
    """common/kzg.rs:17-21.  tau_g2 is carried as its opaque 96-byte compressed encoding:
    the G2 element is only consumed by the (out of scope) pairing checks."""

    def __init__(self, tau_powers_g1, tau_g2=None):
        self.tau_powers_g1 = list(tau_powers_g1)
        self.tau_g2 = tau_g2


def kzg_setup_with_tau(degree, tau):
    """common/kzg.rs:24-45 from line 29 on, with tau supplied by the caller (the reference's own draw,
    tau = Fr::rand(test_rng()), is oracle/arkrng.py: kzg_setup).  Sequential accumulator = accumulator * tau."""
    acc = G1_GENERATOR
    out = [acc]
    for _ in range(1, degree + 1):
        acc = g1_mul(acc, tau)
        out.append(acc)
    return GlobalParameters(out)


def kzg_commit(gp, f):
    """common/kzg.rs:48-60."""
    f = poly_trim(f)
    if len(gp.tau_powers_g1) < poly_degree(f) + 1:
        raise CommitError("Length of tau_powers_g1 must be at least equal to degree of polynomial + 1")
    if not f:
        # .reduce() on an empty iterator -> None.unwrap() panics (kzg.rs:57-58)
        raise RuntimeError("kzg_commit of the zero polynomial panics in the reference")
    acc = None
    for i, fi in enumerate(f):
        term = g1_mul(gp.tau_powers_g1[i], fi)
        acc = term if acc is None else g1_add(acc, term)
    return acc


def kzg_evaluate(gp, f, u):
    """common/kzg.rs:63-88: v = f(u); q = (f - v) / (X - u); proof = sum q_i [tau^i]G."""
    f = poly_trim(f)
    v = poly_eval(f, u)
    f_minus_v = poly_sub(f, poly_trim([v]))
    q, _ = poly_divmod(f_minus_v, [(-u) % R, 1])
    if not q:
        raise RuntimeError("kzg_evaluate of a constant polynomial panics in the reference (kzg.rs:85)")
    if len(q) > len(gp.tau_powers_g1):
        raise IndexError("SRS too short: index out of bounds panic in the reference (kzg.rs:83)")
    acc = None
    for i, qi in enumerate(q):
        term = g1_mul(gp.tau_powers_g1[i], qi)
        acc = term if acc is None else g1_add(acc, term)
    return v, acc


# ---------------------------------------------------------------------------
# common/polynomials.rs
# ---------------------------------------------------------------------------
def calculate_lagrange_polynomials(x_points):
    """common/polynomials.rs:14-47."""
    n = len(x_points)
    product = [1]
    for x in x_points:
        product = poly_mul(product, [(-x) % R, 1])
    out = []
    for i in range(n):
        numerator, _ = poly_divmod(product, [(-x_points[i]) % R, 1])
        den = 1
        for j in range(n):
            if j != i:
                den = den * (x_points[i] - x_points[j]) % R
        out.append(poly_scale(numerator, fr_inv(den)))
    return out


def interpolate_polynomial(x_points, y_points):
    """common/polynomials.rs:50-58."""
    lag = calculate_lagrange_polynomials(x_points)
    acc = []
    for i in range(len(x_points)):
        acc = poly_add(acc, poly_scale(lag[i], y_points[i]))
    return acc


def poly_pow(base, exp):
    """common/polynomials.rs:61-77."""
    result = [1]
    base = poly_trim(base)
    while exp > 0:
        if exp & 1:
            result = poly_mul(result, base)
        base = poly_mul(base, base)
        exp >>= 1
    return result


def compose_polynomials(f, g):
    """common/polynomials.rs:80-113: f(g(X)) by summing f_i * g^i."""
    f_coeffs = list(f)
    n = len(f_coeffs)
    if n < 2:
        # f_coeffs[0] index panic (n == 0) / empty reduce unwrap panic (n == 1)
        raise RuntimeError("compose_polynomials panics in the reference for len(f) < 2")
    g_power = [1]
    acc = []
    for i in range(n):
        if i >= 1:
            acc = poly_add(acc, poly_scale(g_power, f_coeffs[i]))
        g_power = poly_mul(g_power, g)
    return poly_add(poly_trim([f_coeffs[0]]), acc)


# ---------------------------------------------------------------------------
# common/protocols.rs (prover halves)
# ---------------------------------------------------------------------------
def compute_q_zero_test(k, f):
    """common/protocols.rs:157-166: quotient of f by X^k - 1, remainder dropped."""
    q, _ = poly_divmod(f, construct_vanishing_polynomial(k))
    return q


def compute_q_zero_test_from_roots(roots, f):
    """common/protocols.rs:169-181."""
    q, _ = poly_divmod(f, construct_vanishing_polynomial_from_roots(roots))
    return q


def prove_equality(gp, f, g, r):
    """common/protocols.rs:126-141 -> (y_f, proof_f, y_g, proof_g)."""
    y_f, proof_f = kzg_evaluate(gp, f, r)
    y_g, proof_g = kzg_evaluate(gp, g, r)
    return dict(y_f=y_f, proof_f=proof_f, y_g=y_g, proof_g=proof_g)


def prove_zero_test(gp, f, q, r):
    """common/protocols.rs:184-204."""
    com_q = kzg_commit(gp, q)
    f_r, proof_f_r = kzg_evaluate(gp, f, r)
    q_r, proof_q_r = kzg_evaluate(gp, q, r)
    return dict(com_q=com_q, f_r=f_r, proof_f_r=proof_f_r, q_r=q_r, proof_q_r=proof_q_r)


def _extra_points(f, k):
    """The f.degree() + 1 - k extra interpolation points of the running-product / running-sum polynomial t
    (protocols.rs:251-253, 354-356, 469-471, 589-591): drawn from a FRESH ark_std::test_rng(), all the values
    first, then all the abscissae.  Returns (xs, ys)."""
    extra = poly_degree(f) + 1 - k
    if extra < 0:
        raise OverflowError("usize underflow panic in the reference")
    if extra == 0:
        return [], []
    from . import arkrng
    rng = arkrng.test_rng()
    ys = [arkrng.fr_rand(rng) for _ in range(extra)]
    xs = [arkrng.fr_rand(rng) for _ in range(extra)]
    return xs, ys


def _shift(f, w):
    return compose_polynomials(f, [0, w])


def compute_t_and_t1_product_check(Omega, f):
    """common/protocols.rs:235-275."""
    k = len(Omega)
    evals = [poly_eval(f, x) for x in Omega]
    t_y, acc = [], 1
    for e in evals:
        acc = acc * e % R
        t_y.append(acc)
    xs, ys = _extra_points(f, k)
    t = interpolate_polynomial(list(Omega) + xs, t_y + ys)
    t1 = poly_sub(_shift(t, Omega[1]), poly_mul(t, _shift(f, Omega[1])))
    return t, t1


def compute_t_and_t1_product_check_rational_functions(Omega, f, g):
    """common/protocols.rs:336-385."""
    k = len(Omega)
    t_y, acc = [], 1
    for x in Omega:
        acc = acc * poly_eval(f, x) % R * fr_inv(poly_eval(g, x)) % R
        t_y.append(acc)
    xs, ys = _extra_points(f, k)
    t = interpolate_polynomial(list(Omega) + xs, t_y + ys)
    w = Omega[1]
    t1 = poly_sub(poly_mul(_shift(t, w), _shift(g, w)), poly_mul(t, _shift(f, w)))
    return t, t1


def compute_t_and_t1_sum_check(Omega, f):
    """common/protocols.rs:453-493."""
    k = len(Omega)
    t_y, acc = [], 0
    for x in Omega:
        acc = (acc + poly_eval(f, x)) % R
        t_y.append(acc)
    xs, ys = _extra_points(f, k)
    t = interpolate_polynomial(list(Omega) + xs, t_y + ys)
    w = Omega[1]
    t1 = poly_sub(_shift(t, w), poly_add(t, _shift(f, w)))
    return t, t1


def compute_t_and_t1_prescribed_permutation_check(Omega, f, g, W, r, s):
    """common/protocols.rs:559-638."""
    k = len(Omega)
    t_y, acc = [], 1
    for x in Omega:
        num = (r - s * poly_eval(W, x) - poly_eval(f, x)) % R
        den = (r - s * x - poly_eval(g, x)) % R
        acc = acc * num % R * fr_inv(den) % R
        t_y.append(acc)
    xs, ys = _extra_points(f, k)
    t = interpolate_polynomial(list(Omega) + xs, t_y + ys)
    w = Omega[1]
    t_w_x = _shift(t, w)
    r_s_w_x = poly_trim([r, (-s * w) % R])
    g_w_x = _shift(g, w)
    W_w_x = _shift(W, w)
    r_s_W_w_x = compose_polynomials([r, (-s) % R], W_w_x)
    f_w_x = _shift(f, w)
    t1 = poly_sub(poly_mul(t_w_x, poly_sub(r_s_w_x, g_w_x)),
                  poly_mul(t, poly_sub(r_s_W_w_x, f_w_x)))
    return t, t1


def _prove_t_q_common(gp, w, k, t, q, r):
    com_t = kzg_commit(gp, t)
    com_q = kzg_commit(gp, q)
    t_w_k_minus_1, proof_t_w_k_minus_1 = kzg_evaluate(gp, t, pow(w, k - 1, R))
    t_r, proof_t_r = kzg_evaluate(gp, t, r)
    t_w_r, proof_t_w_r = kzg_evaluate(gp, t, r * w % R)
    q_r, proof_q_r = kzg_evaluate(gp, q, r)
    return dict(com_t=com_t, com_q=com_q, t_w_k_minus_1=t_w_k_minus_1,
                proof_t_w_k_minus_1=proof_t_w_k_minus_1, t_r=t_r, proof_t_r=proof_t_r,
                t_w_r=t_w_r, proof_t_w_r=proof_t_w_r, q_r=q_r, proof_q_r=proof_q_r)


def prove_product_check(gp, w, k, t, q, f, r):
    """common/protocols.rs:278-309."""
    p = _prove_t_q_common(gp, w, k, t, q, r)
    p["f_w_r"], p["proof_f_w_r"] = kzg_evaluate(gp, f, r * w % R)
    return p


def prove_sum_check(gp, w, k, t, q, f, r):
    """common/protocols.rs:496-532 (same openings as the product check)."""
    return prove_product_check(gp, w, k, t, q, f, r)


def prove_product_check_rational_functions(gp, w, k, t, q, f, g, r):
    """common/protocols.rs:388-423."""
    p = prove_product_check(gp, w, k, t, q, f, r)
    p["g_w_r"], p["proof_g_w_r"] = kzg_evaluate(gp, g, r * w % R)
    return p


def prove_prescribed_permutation_check(gp, w, k, t, q, f, g, W, rp):
    """common/protocols.rs:641-687."""
    c = _prove_t_q_common(gp, w, k, t, q, rp)
    f_w_rp, proof_f_w_rp = kzg_evaluate(gp, f, rp * w % R)
    g_w_rp, proof_g_w_rp = kzg_evaluate(gp, g, rp * w % R)
    W_w_rp, proof_W_w_rp = kzg_evaluate(gp, W, rp * w % R)
    return dict(com_t=c["com_t"], com_q=c["com_q"],
                t_w_k_minus_1=c["t_w_k_minus_1"], proof_t_w_k_minus_1=c["proof_t_w_k_minus_1"],
                t_rp=c["t_r"], proof_t_rp=c["proof_t_r"],
                t_w_rp=c["t_w_r"], proof_t_w_rp=c["proof_t_w_r"],
                q_rp=c["q_r"], proof_q_rp=c["proof_q_r"],
                f_w_rp=f_w_rp, proof_f_w_rp=proof_f_w_rp,
                g_w_rp=g_w_rp, proof_g_w_rp=proof_g_w_rp,
                W_w_rp=W_w_rp, proof_W_w_rp=proof_W_w_rp)


def compute_t1_T_S_zero_test(w, T, S):
    """common/protocols.rs:722-750."""
    T_w_y = _shift(T, w)
    T_w2_y = _shift(T, w * w % R)
    one_minus_S = compose_polynomials([1, R - 1], S)
    return poly_sub(poly_add(poly_mul(S, poly_add(T, T_w_y)),
                             poly_mul(poly_mul(one_minus_S, T), T_w_y)), T_w2_y)


def prove_T_S_zero_test(gp, w, q, T, S, r):
    """common/protocols.rs:753-786."""
    com_q = kzg_commit(gp, q)
    T_r, proof_T_r = kzg_evaluate(gp, T, r)
    T_w_r, proof_T_w_r = kzg_evaluate(gp, T, w * r % R)
    T_w2_r, proof_T_w2_r = kzg_evaluate(gp, T, w * w * r % R)
    S_r, proof_S_r = kzg_evaluate(gp, S, r)
    q_r, proof_q_r = kzg_evaluate(gp, q, r)
    return dict(com_q=com_q, T_r=T_r, proof_T_r=proof_T_r, T_w_r=T_w_r, proof_T_w_r=proof_T_w_r,
                T_w2_r=T_w2_r, proof_T_w2_r=proof_T_w2_r, S_r=S_r, proof_S_r=proof_S_r,
                q_r=q_r, proof_q_r=proof_q_r)


# ---------------------------------------------------------------------------
# common/protocols.rs (verifier halves).  kzg_verify(com, u, v, proof) -> bool is passed in: the pairing check of
# oracle/pairing.py, or, for a test SRS whose trapdoor is known, the same equation checked in G1.
# ---------------------------------------------------------------------------
def kzg_verify_with_tau(tau):
    """com - [v]G == [tau - u] proof: what e(com - vG, H) == e(proof, tau H - u H) (kzg.rs:91-98) states, for
    a known trapdoor."""
    def check(com, u, v, proof):
        return g1_eq(g1_sub(com, g1_mul(G1_GENERATOR, v)), g1_mul(proof, (tau - u) % R))
    return check


def verify_equality(kzg_verify, com_f, com_g, r, p):
    """common/protocols.rs:144-154."""
    return (p["y_f"] == p["y_g"] and kzg_verify(com_f, r, p["y_f"], p["proof_f"])
            and kzg_verify(com_g, r, p["y_g"], p["proof_g"]))


def verify_zero_test(kzg_verify, k, com_f, r, p):
    """common/protocols.rs:207-217."""
    return (p["f_r"] == p["q_r"] * (pow(r, k, R) - 1) % R
            and kzg_verify(p["com_q"], r, p["q_r"], p["proof_q_r"])
            and kzg_verify(com_f, r, p["f_r"], p["proof_f_r"]))


def verify_zero_on_roots_test(kzg_verify, roots, com_f, r, p):
    """common/protocols.rs:220-232."""
    z_r = poly_eval(construct_vanishing_polynomial_from_roots(roots), r)
    return (p["f_r"] == p["q_r"] * z_r % R
            and kzg_verify(p["com_q"], r, p["q_r"], p["proof_q_r"])
            and kzg_verify(com_f, r, p["f_r"], p["proof_f_r"]))


def _verify_t_q_common(kzg_verify, w, k, r, p, last, identity):
    return (p["t_w_k_minus_1"] == last and identity
            and kzg_verify(p["com_t"], pow(w, k - 1, R), p["t_w_k_minus_1"], p["proof_t_w_k_minus_1"])
            and kzg_verify(p["com_t"], r, p["t_r"], p["proof_t_r"])
            and kzg_verify(p["com_t"], r * w % R, p["t_w_r"], p["proof_t_w_r"])
            and kzg_verify(p["com_q"], r, p["q_r"], p["proof_q_r"]))


def verify_product_check(kzg_verify, w, k, com_f, r, p):
    """common/protocols.rs:312-333."""
    z = (pow(r, k, R) - 1) % R
    ok = (p["t_w_r"] - p["t_r"] * p["f_w_r"]) % R == p["q_r"] * z % R
    return (_verify_t_q_common(kzg_verify, w, k, r, p, 1, ok)
            and kzg_verify(com_f, r * w % R, p["f_w_r"], p["proof_f_w_r"]))


def verify_product_check_rational_functions(kzg_verify, w, k, com_f, com_g, r, p):
    """common/protocols.rs:426-450."""
    z = (pow(r, k, R) - 1) % R
    ok = (p["t_w_r"] * p["g_w_r"] - p["t_r"] * p["f_w_r"]) % R == p["q_r"] * z % R
    return (_verify_t_q_common(kzg_verify, w, k, r, p, 1, ok)
            and kzg_verify(com_f, r * w % R, p["f_w_r"], p["proof_f_w_r"])
            and kzg_verify(com_g, r * w % R, p["g_w_r"], p["proof_g_w_r"]))


def verify_sum_check(kzg_verify, w, k, com_f, r, p):
    """common/protocols.rs:535-556."""
    z = (pow(r, k, R) - 1) % R
    ok = (p["t_w_r"] - (p["t_r"] + p["f_w_r"])) % R == p["q_r"] * z % R
    return (_verify_t_q_common(kzg_verify, w, k, r, p, 0, ok)
            and kzg_verify(com_f, r * w % R, p["f_w_r"], p["proof_f_w_r"]))


def verify_prescribed_permutation_check(kzg_verify, w, k, com_f, com_g, com_W, r, s, rp, p):
    """common/protocols.rs:690-719."""
    z = (pow(rp, k, R) - 1) % R
    wrp = rp * w % R
    lhs = (p["t_w_rp"] * (r - s * w * rp - p["g_w_rp"]) - p["t_rp"] * (r - s * p["W_w_rp"] - p["f_w_rp"])) % R
    return (p["t_w_k_minus_1"] == 1 and lhs == p["q_rp"] * z % R
            and kzg_verify(p["com_t"], pow(w, k - 1, R), p["t_w_k_minus_1"], p["proof_t_w_k_minus_1"])
            and kzg_verify(p["com_t"], rp, p["t_rp"], p["proof_t_rp"])
            and kzg_verify(p["com_t"], wrp, p["t_w_rp"], p["proof_t_w_rp"])
            and kzg_verify(p["com_q"], rp, p["q_rp"], p["proof_q_rp"])
            and kzg_verify(com_f, wrp, p["f_w_rp"], p["proof_f_w_rp"])
            and kzg_verify(com_g, wrp, p["g_w_rp"], p["proof_g_w_rp"])
            and kzg_verify(com_W, wrp, p["W_w_rp"], p["proof_W_w_rp"]))


def verify_T_S_zero_test(kzg_verify, w, Omega_gates, com_T, com_S, r, p):
    """common/protocols.rs:789-809."""
    z_r = poly_eval(construct_vanishing_polynomial_from_roots(Omega_gates), r)
    lhs = (p["S_r"] * (p["T_r"] + p["T_w_r"]) + (1 - p["S_r"]) * p["T_r"] * p["T_w_r"] - p["T_w2_r"]) % R
    return (lhs == p["q_r"] * z_r % R
            and kzg_verify(com_T, r, p["T_r"], p["proof_T_r"])
            and kzg_verify(com_T, w * r % R, p["T_w_r"], p["proof_T_w_r"])
            and kzg_verify(com_T, w * w * r % R, p["T_w2_r"], p["proof_T_w2_r"])
            and kzg_verify(com_S, r, p["S_r"], p["proof_S_r"])
            and kzg_verify(p["com_q"], r, p["q_r"], p["proof_q_r"]))


# ---------------------------------------------------------------------------
# setup_proving_key / setup_verification_key / prover (the example circuit)
# ---------------------------------------------------------------------------
class Setup:
    """setup_global_params/mod.rs:8-15."""

    def __init__(self, number_gates, number_public_inputs, number_witnesses, d, gp):
        self.number_gates = number_gates
        self.number_public_inputs = number_public_inputs
        self.number_witnesses = number_witnesses
        self.d = d
        self.gp = gp


def load_setup_json(obj):
    """setup_global_params/json.rs:22-48: hex -> deserialize_compressed."""
    pts = [g1_decompress(bytes.fromhex(h)) for h in obj["gp"]["tau_powers_g1"]]
    gp = GlobalParameters(pts, bytes.fromhex(obj["gp"]["tau_g2"]))
    return Setup(obj["number_gates"], obj["number_public_inputs"], obj["number_witnesses"], obj["d"], gp)


class Circuit:
    """A circuit in the trace layout the reference hard-codes for its example (prover/part1.rs:16-59,
    setup_proving_key/mod.rs:32-103): gate l occupies Omega[3l], Omega[3l+1], Omega[3l+2] (left,
    right, output), public input i sits at Omega[d-1-i], the witnesses below them.
        gates[l]   1 = addition, 0 = multiplication                       (the selector S)
        trace[j]   the value at Omega[j]                                  (the polynomial T)
        wiring[j]  W(Omega[j]) = Omega[wiring[j]]: cycles through the positions that share a value"""

    def __init__(self, gates, trace, wiring, pub_inputs, output):
        self.gates, self.trace, self.wiring = list(gates), list(trace), list(wiring)
        self.pub_inputs, self.output = list(pub_inputs), output


def example_circuit(d=12):
    """The reference's example: inputs 5, 6, witness 1; 5 + 6 = 11, 6 + 1 = 7, 11 * 7 = 77."""
    trace = [5, 6, 11, 6, 1, 7, 11, 7, 77, 1, 6, 5]
    src = [d - 2, 1, 3, d - 1, 0, 2, 6, d - 3, 4, 5, 7, 8]
    dst = [1, 3, d - 2, 0, d - 1, 6, 2, 4, d - 3, 7, 5, 8]
    wiring = [0] * d
    for a, b in zip(src, dst):
        wiring[a] = b
    return Circuit([1, 1, 0], trace, wiring, [5, 6], 77)


def synthetic_circuit(log_gates, seed, number_gates=None):
    """A random circuit of the same family at another size: 2^log_gates - 1 gates, 2 public inputs,
    1 witness, d = 3 * 2^log_gates (log_gates = 2 gives the example's shape).  Every gate takes two
    earlier values (inputs, the witness or gate outputs) and adds or multiplies them; the last
    gate's output is the circuit's output.  number_gates: fewer gates on the same d rows (the rows between
    the last gate and the witness stay empty: value 0, wired to themselves); the gate rows are then a
    proper prefix of the subgroup generated by omega^3."""
    import random as _random
    rnd = _random.Random(seed)
    ng = (1 << log_gates) - 1
    d = 3 * ng + 3
    if number_gates is not None:
        assert 1 <= number_gates <= ng
        ng = number_gates
    trace = [0] * d
    pool = []  # [value, positions]
    for pos in (d - 1, d - 2, d - 3):
        v = rnd.randrange(1, R)
        trace[pos] = v
        pool.append([v, [pos]])
    gates = []
    for l in range(ng):
        # prefer recent values so that the circuit is a chain, not a star
        a = pool[max(0, len(pool) - 1 - rnd.randrange(4))] if l else pool[0]
        b = pool[rnd.randrange(len(pool))] if l else pool[1]
        kind = rnd.randrange(2)
        out = (a[0] + b[0]) % R if kind else a[0] * b[0] % R
        gates.append(kind)
        for val, pos in ((a, 3 * l), (b, 3 * l + 1)):
            trace[pos] = val[0]
            val[1].append(pos)
        trace[3 * l + 2] = out
        pool.append([out, [3 * l + 2]])
    wiring = list(range(d))
    for _, positions in pool:
        for i, pos in enumerate(positions):
            wiring[pos] = positions[(i + 1) % len(positions)]
    return Circuit(gates, trace, wiring, [trace[d - 1], trace[d - 2]], trace[3 * ng - 1])


def _selector_and_wiring(setup, circuit=None):
    """setup_proving_key/mod.rs:32-103 == setup_verification_key/mod.rs:37-105.  Without a circuit:
    the example, with the reference's literal point lists."""
    d, ng = setup.d, setup.number_gates
    Omega = construct_Omega(d)
    Omega_gates = [Omega[3 * l] for l in range(ng)]
    if circuit is not None:
        S = interpolate_polynomial(Omega_gates, circuit.gates)
        W = interpolate_polynomial(Omega, [Omega[j] for j in circuit.wiring])
        assert poly_degree(S) == ng - 1 and poly_degree(W) == d - 1
        return S, W
    gates = [1, 1, 0]
    S = interpolate_polynomial(Omega_gates, gates)
    assert poly_degree(S) == ng - 1
    Wx = [Omega[d - 2], Omega[1], Omega[3], Omega[d - 1], Omega[0], Omega[2], Omega[6],
          Omega[d - 3], Omega[4], Omega[5], Omega[7], Omega[8]]
    Wy = [Omega[1], Omega[3], Omega[d - 2], Omega[0], Omega[d - 1], Omega[6], Omega[2],
          Omega[4], Omega[d - 3], Omega[7], Omega[5], Omega[8]]
    W = interpolate_polynomial(Wx, Wy)
    assert poly_degree(W) == d - 1
    return S, W


def reproduce_proving_key(setup, circuit=None):
    """setup_proving_key/mod.rs:23-30,32-103 -> the JSON object of data/proving_key.json."""
    S, W = _selector_and_wiring(setup, circuit)
    return {"S": [fr_to_string(c) for c in S], "W": [fr_to_string(c) for c in W]}


def reproduce_verification_key(setup, circuit=None):
    """setup_verification_key/mod.rs:23-35,37-113 -> the JSON object of
    data/verification_key.json."""
    S, W = _selector_and_wiring(setup, circuit)
    return {"com_S": g1_hex(kzg_commit(setup.gp, S)), "com_W": g1_hex(kzg_commit(setup.gp, W))}


def prover_part1(setup, Omega, circuit=None):
    """prover/part1.rs:9-69."""
    d = setup.d
    if circuit is not None:
        T = interpolate_polynomial(Omega, circuit.trace)
        assert poly_degree(T) == d - 1
        return T, kzg_commit(setup.gp, T)
    xs = [Omega[d - 1], Omega[d - 2], Omega[d - 3]] + [Omega[i] for i in range(9)]
    ys = [5, 6, 1, 5, 6, 11, 6, 1, 7, 11, 7, 77]
    T = interpolate_polynomial(xs, ys)
    assert poly_degree(T) == d - 1
    return T, kzg_commit(setup.gp, T)


def prover_part2(setup, pub_inputs, Omega, T, com_T):
    """prover/part2.rs:14-61."""
    d, gp = setup.d, setup.gp
    Omega_inputs = [Omega[d - 1 - i] for i in range(setup.number_public_inputs)]
    v = interpolate_polynomial(Omega_inputs, pub_inputs)
    assert poly_degree(v) == setup.number_public_inputs - 1
    T_minus_v = poly_sub(T, v)
    assert poly_degree(T_minus_v) == d - 1  # 11 in the example (part2.rs)
    com_v = kzg_commit(gp, v)
    com_T_minus_v = g1_sub(com_T, com_v)
    q = compute_q_zero_test_from_roots(Omega_inputs, T_minus_v)
    r = derive_challenge_from_commitments([com_T_minus_v])
    return prove_zero_test(gp, T_minus_v, q, r)


def prover_part3(setup, S, com_S, Omega, T, com_T):
    """prover/part3.rs:16-75."""
    gp = setup.gp
    Omega_gates = [Omega[3 * l] for l in range(setup.number_gates)]
    w = Omega[1]
    T_w_y = _shift(T, w)
    T_w2_y = _shift(T, w * w % R)
    one_minus_S = poly_sub([1], S)
    S_T = poly_sub(poly_add(poly_mul(S, poly_add(T, T_w_y)),
                            poly_mul(one_minus_S, poly_mul(T, T_w_y))), T_w2_y)
    for y in Omega_gates:
        assert poly_eval(S_T, y) == 0, "S_T should cancel on y"
    q = compute_q_zero_test_from_roots(Omega_gates, S_T)
    r = derive_challenge_from_commitments([com_T, com_S])
    return prove_T_S_zero_test(gp, w, q, T, S, r)


def prover_part4(setup, W, com_W, Omega, T, com_T):
    """prover/part4.rs:17-66."""
    d, gp = setup.d, setup.gp
    for y in Omega:
        assert poly_eval(T, y) == poly_eval(T, poly_eval(W, y)), "T_W should be equal to T on Omega"
    Z_Omega = construct_vanishing_polynomial(d)
    r, s, rp = derive_multiple_challenges_from_commitments([com_T, com_W], 3)
    t, t1 = compute_t_and_t1_prescribed_permutation_check(Omega, T, T, W, r, s)
    q = compute_q_zero_test(d, t1)
    assert poly_mul(q, Z_Omega) == t1, "t1 must be divisible by Z_Omega"
    return prove_prescribed_permutation_check(gp, Omega[1], d, t, q, T, T, W, rp)


def prover_part5(setup, Omega, T, output):
    """prover/part5.rs:6-27."""
    v, proof = kzg_evaluate(setup.gp, T, Omega[3 * setup.number_gates - 1])
    assert v == output, "Evaluation does not match with output"
    return proof


def proof_to_json(pub_inputs, output, com_T, p2, p3, p4, p5):
    """common/proof.rs:58-268 (field order :220-266)."""
    return {
        "pub_inputs": [fr_to_string(x) for x in pub_inputs],
        "output": fr_to_string(output),
        "com_T": g1_hex(com_T),
        "proof_T_minus_v_zero": [
            g1_hex(p2["com_q"]), fr_to_string(p2["f_r"]), g1_hex(p2["proof_f_r"]),
            fr_to_string(p2["q_r"]), g1_hex(p2["proof_q_r"])],
        "proof_T_S_zero": [
            g1_hex(p3["com_q"]),
            fr_to_string(p3["T_r"]), g1_hex(p3["proof_T_r"]),
            fr_to_string(p3["T_w_r"]), g1_hex(p3["proof_T_w_r"]),
            fr_to_string(p3["T_w2_r"]), g1_hex(p3["proof_T_w2_r"]),
            fr_to_string(p3["S_r"]), g1_hex(p3["proof_S_r"]),
            fr_to_string(p3["q_r"]), g1_hex(p3["proof_q_r"])],
        "proof_T_W_prescribed_permutation": [
            g1_hex(p4["com_t"]), g1_hex(p4["com_q"]),
            fr_to_string(p4["t_w_k_minus_1"]), g1_hex(p4["proof_t_w_k_minus_1"]),
            fr_to_string(p4["t_rp"]), g1_hex(p4["proof_t_rp"]),
            fr_to_string(p4["t_w_rp"]), g1_hex(p4["proof_t_w_rp"]),
            fr_to_string(p4["q_rp"]), g1_hex(p4["proof_q_rp"]),
            fr_to_string(p4["f_w_rp"]), g1_hex(p4["proof_f_w_rp"]),
            fr_to_string(p4["g_w_rp"]), g1_hex(p4["proof_g_w_rp"]),
            fr_to_string(p4["W_w_rp"]), g1_hex(p4["proof_W_w_rp"])],
        "proof_last_gate_KZG": g1_hex(p5),
    }


def reproduce_proof(setup, proving_key_json, verification_key_json, circuit=None):
    """prover/mod.rs:18-74 driven from the three input JSON objects read by
    bin/prover.rs:10-38 -> the JSON object of data/proof.json.  Without a circuit: the example
    (inputs 5, 6 and output 77 as hard-coded at prover/mod.rs:25-26)."""
    S = poly_trim([int(x) for x in proving_key_json["S"]])
    W = poly_trim([int(x) for x in proving_key_json["W"]])
    com_S = g1_decompress(bytes.fromhex(verification_key_json["com_S"]))
    com_W = g1_decompress(bytes.fromhex(verification_key_json["com_W"]))
    pub_inputs = [5, 6] if circuit is None else circuit.pub_inputs
    output = 77 if circuit is None else circuit.output
    Omega = construct_Omega(setup.d)
    T, com_T = prover_part1(setup, Omega, circuit)
    p2 = prover_part2(setup, pub_inputs, Omega, T, com_T)
    p3 = prover_part3(setup, S, com_S, Omega, T, com_T)
    p4 = prover_part4(setup, W, com_W, Omega, T, com_T)
    p5 = prover_part5(setup, Omega, T, output)
    return proof_to_json(pub_inputs, output, com_T, p2, p3, p4, p5)


# ---------------------------------------------------------------------------
# Transforms over Omega (the O(n log n) algorithms the product uses in place of the
# reference's Lagrange interpolation / repeated Horner evaluation).  Both directions are
# DEFINED by the reference: forward = [f(Omega[j]) for j] (protocols.rs:243-250 etc.),
# inverse = interpolate_polynomial(Omega, ys) (polynomials.rs:50-58).
# ---------------------------------------------------------------------------
def root_of_unity(n):
    assert (R - 1) % n == 0
    return pow(FR_GENERATOR, (R - 1) // n, R)


def dft_naive(coeffs, n):
    """[f(omega^j) for j in range(n)] by Horner; f may have any number of coefficients."""
    w = root_of_unity(n)
    return [poly_eval(coeffs, pow(w, j, R)) for j in range(n)]


def ntt(a, invert=False):
    """Radix-2 transform of a length-2^k list, natural order in and out, omega =
    7^((r-1)/n): forward gives evaluations over construct_Omega(n), inverse gives the
    coefficients of the interpolant (padded with zeros to length n)."""
    n = len(a)
    assert n & (n - 1) == 0
    a = [x % R for x in a]
    logn = n.bit_length() - 1
    for i in range(n):
        j = int(format(i, "0%db" % logn)[::-1], 2) if logn else 0
        if i < j:
            a[i], a[j] = a[j], a[i]
    w_n = root_of_unity(n) if n > 1 else 1
    if invert and n > 1:
        w_n = pow(w_n, -1, R)
    length = 2
    while length <= n:
        wl = pow(w_n, n // length, R)
        for start in range(0, n, length):
            w = 1
            for k in range(length // 2):
                u = a[start + k]
                v = a[start + k + length // 2] * w % R
                a[start + k] = (u + v) % R
                a[start + k + length // 2] = (u - v) % R
                w = w * wl % R
        length <<= 1
    if invert:
        ninv = pow(n, -1, R)
        a = [x * ninv % R for x in a]
    return a
