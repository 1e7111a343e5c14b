"""Writes the gfx950 instruction sequences of the Montgomery products on unsaturated limbs as bodies of
inline-assembly statements, and checks every sequence against big-integer arithmetic with a small interpreter
of the instructions it uses:

  plonk_amd/csrc/mont29_asm.inc       Fr, 9 x 29-bit limbs (radix 2^261):  a * b
  plonk_amd/csrc/mont28_mul_asm.inc   Fq, 14 x 28-bit limbs (radix 2^392): a * b
  plonk_amd/csrc/mont28_sqr_asm.inc   Fq: a * a (off-diagonal products once, against the doubled operand)
  plonk_amd/csrc/mont28_mul2_asm.inc  Fq: a * b + c * d with one reduction

A sequence is one dependent chain on a 64-bit accumulator (two waves per SIMD hide the latency of the chain):
  columns 0..N-1 :  acc += sum a_i b_(k-i) - sum m_i p_(k-i)   (the second sum as signed multiply-adds with -p_j),
                    m_k = acc / p mod 2^BITS (the low limb of the running column times p^-1), acc >>= BITS (arithmetic:
                    subtracting m_k p has cleared the low bits);
                    the last digit is taken 2^BITS lower, which adds p to the result and keeps it positive
  columns N..2N-1:  the same sums, limb k - N of the result = acc mod 2^BITS, acc >>= BITS.
For Fr, p = 1 mod 2^29: m_k is the low limb itself and the product with p_0 is never formed (the last column
then needs an increment).  The result registers hold the digits m_k until limb k of the result is written.
The compiler's code for the same product splits the column sums into independent chains and joins them with
64-bit additions, and forms m_k for an additive reduction: about 20 % more instructions.
usage: python scripts/gen_mont_asm.py [--check-only]"""
import os
import random
import sys

FR = dict(name="fr", N=9, BITS=29,
          P=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001)
FQ = dict(name="fq", N=14, BITS=28,
          P=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab)


def limbs(v, f):
    return [(v >> (f["BITS"] * i)) & ((1 << f["BITS"]) - 1) for i in range(f["N"])]


def program(f, kind):
    """List of (opcode, dst, src...) with symbolic operands.  kind: mul | sqr | mul2.
    sqr takes the operand as ("a", i) and its double as ("b", i)."""
    N = f["N"]
    p0_is_one = limbs(f["P"], f)[0] == 1
    p = []
    for k in range(2 * N):
        first = k == 0
        terms = []
        if kind == "sqr":
            for i in range(N):
                j = k - i
                if i < j < N:
                    terms.append((("b", i), ("a", j)))  # 2 a_i a_j
            if k % 2 == 0 and k // 2 < N:
                terms.append((("a", k // 2), ("a", k // 2)))
        else:
            for i in range(N):
                j = k - i
                if 0 <= j < N:
                    terms.append((("a", i), ("b", j)))
                    if kind == "mul2":
                        terms.append((("c", i), ("d", j)))
        for x, y in terms:
            p.append(("mad_u", x, y, "zero" if first else "acc"))
            first = False
        for i in range(N):
            j = k - i
            if 0 <= j < N and i < min(k, N) and not (p0_is_one and j == 0):
                p.append(("mad_i", ("m", i), ("np", j)))
        if k < N:
            # the digit of this column
            if p0_is_one:
                p.append(("and" if k < N - 1 else "or", ("m", k), "acc_lo"))
            else:
                p.append(("mul_lo", ("m", k), "acc_lo"))
                p.append(("and" if k < N - 1 else "or", ("m", k), ("m", k)))
                p.append(("mad_i", ("m", k), ("np", 0)))
        else:
            p.append(("and", ("out", k - N), "acc_lo"))
        if k < 2 * N - 1:
            p.append(("ashr",))
        if k == N - 1 and p0_is_one:
            p.append(("inc",))
    return p


def interpret(f, kind, ops):
    """Runs the program on limb lists (dict a, b, c, d) with 64-bit wrap-around semantics."""
    N, BITS = f["N"], f["BITS"]
    MASK = (1 << BITS) - 1
    M64 = (1 << 64) - 1
    pl = limbs(f["P"], f)
    pinv = pow(f["P"], -1, 1 << BITS)

    def s32(x):
        x &= 0xffffffff
        return x - (1 << 32) if x >> 31 else x

    def s64(x):
        x &= M64
        return x - (1 << 64) if x >> 63 else x

    acc = 0
    m = [0] * N
    out = [0] * N
    np_ = [(-pl[j]) & 0xffffffff for j in range(N)]

    def val(x):
        kind_, i = x
        if kind_ == "m":
            return m[i]
        return ops[kind_][i]

    for ins in program(f, kind):
        op = ins[0]
        if op == "mad_u":
            base = 0 if ins[3] == "zero" else acc
            x, y = val(ins[1]), val(ins[2])
            assert 0 <= x < (1 << 32) and 0 <= y < (1 << 32)
            acc = (base + x * y) & M64
        elif op == "mad_i":
            acc = (acc + s32(m[ins[1][1]]) * s32(np_[ins[2][1]])) & M64
        elif op == "mul_lo":
            m[ins[1][1]] = ((acc & 0xffffffff) * pinv) & 0xffffffff
        elif op == "and":
            src = (acc & 0xffffffff) if ins[2] == "acc_lo" else m[ins[2][1]]
            if ins[1][0] == "m":
                m[ins[1][1]] = src & MASK
            else:
                out[ins[1][1]] = src & MASK
        elif op == "or":
            src = (acc & 0xffffffff) if ins[2] == "acc_lo" else m[ins[2][1]]
            m[ins[1][1]] = (src | (0xffffffff ^ MASK)) & 0xffffffff
        elif op == "ashr":
            assert -(1 << 63) <= s64(acc) < (1 << 63)
            acc = (s64(acc) >> BITS) & M64
        elif op == "inc":
            acc = (acc + 1) & M64
    return out


def check(f, kind, trials=600):
    N, BITS, P = f["N"], f["BITS"], f["P"]
    cap = 1 << (N * BITS)
    rng = random.Random(7)
    for trial in range(trials):
        if trial == 0:
            vals = [0, 0, 0, 0]
        elif f is FR:
            # data below the capacity, the factor below r
            vals = [cap - 1, P - 1, 0, 0] if trial == 1 else [rng.getrandbits(N * BITS), rng.randrange(P), 0, 0]
        else:
            # lazy values: small multiples of q (the accumulation loop stays below 8 q)
            vals = [30 * P - 1] * 4 if trial == 1 else [rng.randrange(30 * P) for _ in range(4)]
        a, b, c, d = (limbs(v, f) for v in vals)
        if f is FR and trial % 3 == 2:
            # data limbs may be slightly unnormalised: up to 2^30 per limb stays within the signed range
            a = [x + rng.getrandbits(BITS) for x in a]
            vals[0] = sum(x << (BITS * i) for i, x in enumerate(a))
        if kind == "sqr":
            ops = dict(a=a, b=[2 * x for x in a])
            want = vals[0] * vals[0]
        elif kind == "mul2":
            ops = dict(a=a, b=b, c=c, d=d)
            want = vals[0] * vals[1] + vals[2] * vals[3]
        else:
            ops = dict(a=a, b=b)
            want = vals[0] * vals[1]
        out = interpret(f, kind, ops)
        got = sum(x << (BITS * i) for i, x in enumerate(out))
        assert (got * cap) % P == want % P, (f["name"], kind, trial)
        assert got <= want // cap + P, (f["name"], kind, trial)
    return True


def operand_numbers(f, kind):
    """Operand layout of the asm statement: results (= digits), inputs, -p_j, mask, ~mask, p^-1 mod 2^BITS."""
    N = f["N"]
    p0_is_one = limbs(f["P"], f)[0] == 1
    groups = {"out": 0, "a": N, "b": 2 * N}
    nxt = 3 * N
    if kind == "mul2":
        groups["c"] = nxt
        groups["d"] = nxt + N
        nxt += 2 * N
    first_np = 1 if p0_is_one else 0
    groups["np"] = nxt - first_np
    nxt += N - first_np
    groups["mask"] = nxt
    groups["high"] = nxt + 1
    groups["pinv"] = nxt + 2
    return groups


def text(f, kind, acc="v[2:3]", acc_lo="v2"):
    g = operand_numbers(f, kind)

    def opnd(x):
        if x == "acc":
            return acc
        if x == "acc_lo":
            return acc_lo
        if x == "zero":
            return "0"
        k, i = x
        if k == "m":
            k = "out"
        return "%%%d" % (g[k] + i)

    lines = []
    for ins in program(f, kind):
        op = ins[0]
        if op == "mad_u":
            lines.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (acc, opnd(ins[1]), opnd(ins[2]), opnd(ins[3])))
        elif op == "mad_i":
            lines.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (acc, opnd(ins[1]), opnd(ins[2]), acc))
        elif op == "mul_lo":
            lines.append("v_mul_lo_u32 %s, %s, %%%d" % (opnd(ins[1]), acc_lo, g["pinv"]))
        elif op == "and":
            lines.append("v_and_b32 %s, %%%d, %s" % (opnd(ins[1]), g["mask"], opnd(ins[2])))
        elif op == "or":
            lines.append("v_or_b32 %s, %%%d, %s" % (opnd(ins[1]), g["high"], opnd(ins[2])))
        elif op == "ashr":
            lines.append("v_ashrrev_i64 %s, %d, %s" % (acc, f["BITS"], acc))
        elif op == "inc":
            lines.append("v_lshl_add_u64 %s, %s, 0, 1" % (acc, acc))
    return lines


FILES = [("mont29_asm.inc", FR, "mul"), ("mont28_mul_asm.inc", FQ, "mul"), ("mont28_sqr_asm.inc", FQ, "sqr"),
         ("mont28_mul2_asm.inc", FQ, "mul2")]


def header(f, kind):
    g = operand_numbers(f, kind)
    N = f["N"]
    names = dict(mul="a * b", sqr="a * a (b = 2 a)", mul2="a * b + c * d")
    ins = "%%%d.. a, %%%d.. b" % (g["a"], g["b"])
    if kind == "mul2":
        ins += ", %%%d.. c, %%%d.. d" % (g["c"], g["d"])
    first_np = 1 if limbs(f["P"], f)[0] == 1 else 0
    tail = "" if first_np else ", %%%d p^-1 mod 2^%d" % (g["pinv"], f["BITS"])
    return ["Generated by scripts/gen_mont_asm.py: Montgomery %s on %d x %d-bit limbs for gfx950." % (names[kind], N, f["BITS"]),
            "Operands: %%0.. result, %s, %%%d.. -p_%d..-p_%d, %%%d 2^%d - 1, %%%d its complement%s;"
            % (ins, g["np"] + first_np, first_np, N - 1, g["mask"], f["BITS"], g["high"], tail),
            "accumulator v[2:3], vcc clobbered."]


def render(f, kind):
    return ["// %s\n" % h for h in header(f, kind)] + ['"%s\\n"\n' % ln for ln in text(f, kind)]


if __name__ == "__main__":
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plonk_amd", "csrc")
    for name, f, kind in FILES:
        assert check(f, kind)
        print("%s: %d instructions, checked against big-integer arithmetic" % (name, len(text(f, kind))), file=sys.stderr)
        if "--check-only" not in sys.argv:
            with open(os.path.join(root, name), "w") as out:
                out.writelines(render(f, kind))
