"""TEST INFRASTRUCTURE ONLY (the checker of the verifier-side tests; nothing in the product imports it).

CPU restatement of the pairing check behind `kzg_verify` (plonk/src/common/kzg.rs:91-98):

    e(com_f - v G, H) == e(proof, tau H - u H)

The reference calls `Bls12_381::pairing` of the third-party crate ark-bls12-381 0.5.0 (Cargo.toml:16), which
is absent from /root/reference; what is restated here is the published algorithm it implements: the optimal
ate pairing of BLS12-381, f_{|x|,Q}(P) conjugated (x < 0) and raised to (p^12 - 1) / r.  Only equalities of
pairings are ever consumed (kzg.rs:97), and every non-degenerate bilinear pairing G1 x G2 -> GT gives the same
truth values, so parity is anchored on the reference's own data: the golden SRS satisfies
e(tau G, H) = e(G, tau H) under this restatement (tests/test_oracle_pairing.py), all openings of the golden
proof verify, and tampered ones do not.

Deliberately plain: big integers, Fq12 as polynomials of degree < 6 in w over Fq2 with w^6 = xi = 1 + u, the
point T of the Miller loop in affine coordinates on the twist, the final exponentiation as one
square-and-multiply with the full exponent.  A product of two pairings takes about a second.

Serialisation of G2 (ark-serialize compressed form of ark-bls12-381, the same convention as the G1 points of
plonk/data/srs.json): 96 bytes, x.c1 then x.c0 big-endian, top three bits of the first byte = compressed,
infinity, y lexicographically largest (c1 compared first)."""

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
X_ABS = 0xd201000000010000  # the curve parameter is -X_ABS
XI = (1, 1)

# generator of G2 (checked in the tests: on the twist, of order r, and consistent with the golden SRS)
G2_GEN = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
           0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
          (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
           0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))
G1_GEN = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
          0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)


# ------------------------------------------------------------------------------------------ Fq2
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return (-a[0] % P, -a[1] % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_scale(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, -a[1] * n % P)


def f2_pow(a, e):
    out = (1, 0)
    for bit in bin(e)[2:]:
        out = f2_mul(out, out)
        if bit == "1":
            out = f2_mul(out, a)
    return out


def f2_sqrt(a):
    """A square root of a in Fq2, or None (p = 3 mod 4: the complex method)."""
    if a == (0, 0):
        return (0, 0)
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = f2_mul(a1, f2_mul(a1, a))
    norm = f2_mul((alpha[0], -alpha[1] % P), alpha)  # alpha^p * alpha
    if norm == (P - 1, 0):
        return None
    x0 = f2_mul(a1, a)
    if alpha == (P - 1, 0):
        root = f2_mul((0, 1), x0)
    else:
        root = f2_mul(f2_pow(f2_add((1, 0), alpha), (P - 1) // 2), x0)
    return root if f2_mul(root, root) == a else None


# ------------------------------------------------------------------------------------------ Fq12
F12_ONE = [(1, 0)] + [(0, 0)] * 5


def f12_mul(a, b):
    c = [(0, 0)] * 11
    for i in range(6):
        if a[i] == (0, 0):
            continue
        for j in range(6):
            if b[j] == (0, 0):
                continue
            c[i + j] = f2_add(c[i + j], f2_mul(a[i], b[j]))
    for k in range(6, 11):
        c[k - 6] = f2_add(c[k - 6], f2_mul(c[k], XI))
    return c[:6]


def f12_conj(a):
    """a^(p^6): w -> -w."""
    return [a[i] if i % 2 == 0 else f2_neg(a[i]) for i in range(6)]


def f12_pow(a, e):
    out = list(F12_ONE)
    for bit in bin(e)[2:]:
        out = f12_mul(out, out)
        if bit == "1":
            out = f12_mul(out, a)
    return out


# ------------------------------------------------------------------------------------------ G2 (affine, on the twist)
B2 = (4, 4)  # y^2 = x^3 + 4 (1 + u)


def g2_on_curve(Q):
    if Q is None:
        return True
    x, y = Q
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), B2)


def g2_neg(Q):
    return None if Q is None else (Q[0], f2_neg(Q[1]))


def g2_double(Q):
    if Q is None or Q[1] == (0, 0):
        return None
    x, y = Q
    lam = f2_mul(f2_scale(f2_mul(x, x), 3), f2_inv(f2_scale(y, 2)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x), x)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x, x3)), y))


def g2_add(Q, S):
    if Q is None:
        return S
    if S is None:
        return Q
    if Q[0] == S[0]:
        return g2_double(Q) if Q[1] == S[1] else None
    lam = f2_mul(f2_sub(S[1], Q[1]), f2_inv(f2_sub(S[0], Q[0])))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), Q[0]), S[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(Q[0], x3)), Q[1]))


def g2_mul(Q, k):
    k %= R
    out = None
    for bit in bin(k)[2:]:
        out = g2_double(out)
        if bit == "1":
            out = g2_add(out, Q)
    return out


def g2_decompress(b):
    if len(b) != 96 or not b[0] & 0x80:
        raise ValueError("not a compressed G2 encoding")
    if b[0] & 0x40:
        if any(b[1:]) or b[0] & 0x3f:
            raise ValueError("malformed encoding of the identity")
        return None
    largest = bool(b[0] & 0x20)
    c1 = int.from_bytes(bytes([b[0] & 0x1f]) + b[1:48], "big")
    c0 = int.from_bytes(b[48:], "big")
    if c0 >= P or c1 >= P:
        raise ValueError("coordinate not reduced")
    x = (c0, c1)
    y = f2_sqrt(f2_add(f2_mul(f2_mul(x, x), x), B2))
    if y is None:
        raise ValueError("x is not on the curve")
    ny = f2_neg(y)
    if ((y[1], y[0]) > (ny[1], ny[0])) != largest:
        y = ny
    return (x, y)


def g2_compress(Q):
    if Q is None:
        return bytes([0xc0]) + bytes(95)
    (x, y) = Q
    ny = f2_neg(y)
    out = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
    out[0] |= 0x80 | (0x20 if (y[1], y[0]) > (ny[1], ny[0]) else 0)
    return bytes(out)


# ------------------------------------------------------------------------------------------ pairing
def _line(lam, T, Pt):
    """The line of slope lam through T (on the twist), evaluated at Pt in G1 and scaled by w^3 (an element of a
    proper subfield, which the final exponentiation sends to 1): (lam x_T - y_T) - lam x_P w^2 + y_P w^3."""
    xp, yp = Pt
    out = [(0, 0)] * 6
    out[0] = f2_sub(f2_mul(lam, T[0]), T[1])
    out[2] = f2_neg(f2_scale(lam, xp))
    out[3] = (yp % P, 0)
    return out


def miller_loop(Pt, Q):
    """f_{x,Q}(Pt) up to factors the final exponentiation removes; Pt = (x, y) in G1 affine or None, Q in G2
    affine on the twist or None."""
    if Pt is None or Q is None:
        return list(F12_ONE)
    f = list(F12_ONE)
    T = Q
    for bit in bin(X_ABS)[3:]:
        lam = f2_mul(f2_scale(f2_mul(T[0], T[0]), 3), f2_inv(f2_scale(T[1], 2)))
        f = f12_mul(f12_mul(f, f), _line(lam, T, Pt))
        T = g2_double(T)
        if bit == "1":
            lam = f2_mul(f2_sub(Q[1], T[1]), f2_inv(f2_sub(Q[0], T[0])))
            f = f12_mul(f, _line(lam, T, Pt))
            T = g2_add(T, Q)
    return f12_conj(f)  # x is negative


def final_exponentiation(f):
    return f12_pow(f, (P ** 12 - 1) // R)


def pairing(Pt, Q):
    return final_exponentiation(miller_loop(Pt, Q))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1, with one final exponentiation."""
    f = list(F12_ONE)
    for Pt, Q in pairs:
        f = f12_mul(f, miller_loop(Pt, Q))
    return final_exponentiation(f) == F12_ONE


# ------------------------------------------------------------------------------------------ kzg_verify
def kzg_verify(tau_g2, com_f, u, v, proof):
    """plonk/src/common/kzg.rs:91-98.  G1 arguments in the Jacobian form of oracle/pyref.py, tau_g2 affine on the
    twist, u and v integers mod r.  e(com_f - v G, H) == e(proof, tau H - u H), taken as
    e(com_f - v G, H) * e(-proof, tau H - u H) == 1."""
    from . import pyref
    lhs = pyref.g1_sub(com_f, pyref.g1_mul(pyref.G1_GENERATOR, v))
    q = g2_add(tau_g2, g2_neg(g2_mul(G2_GEN, u)))
    return pairing_product_is_one([(pyref.g1_to_affine(lhs), G2_GEN),
                                   (pyref.g1_to_affine(pyref.g1_neg(proof)), q)])


def kzg_verify_batch(tau_g2, checks, rho):
    """All of `checks` = [(com_f, u, v, proof), ...] at once: with weights rho_i (rho_0 = 1 is fine, the rest
    must be unpredictable to whoever made the proofs),
        L = sum rho_i (com_i - v_i G + u_i proof_i),  R = sum rho_i proof_i,  e(L, H) == e(R, tau H).
    Equivalent to every kzg_verify holding, except with probability about len(checks) / r."""
    from . import pyref
    L, Rs = pyref.G1_IDENTITY, pyref.G1_IDENTITY
    for (com, u, v, proof), w in zip(checks, rho):
        term = pyref.g1_add(pyref.g1_sub(com, pyref.g1_mul(pyref.G1_GENERATOR, v)), pyref.g1_mul(proof, u))
        L = pyref.g1_add(L, pyref.g1_mul(term, w))
        Rs = pyref.g1_add(Rs, pyref.g1_mul(proof, w))
    return pairing_product_is_one([(pyref.g1_to_affine(L), G2_GEN), (pyref.g1_to_affine(pyref.g1_neg(Rs)), tau_g2)])
