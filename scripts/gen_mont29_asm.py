"""Writes plonk_amd/csrc/mont29_asm.inc: the gfx950 instruction sequence of one Montgomery product on 9 x 29-bit
limbs (Fr, radix 2^261) as the body of an inline-assembly statement, and checks the sequence against big-integer
arithmetic with a small interpreter of the five instructions it uses.

The sequence is one dependent chain on a 64-bit accumulator (two waves per SIMD hide the latency of the chain):
  columns 0..8 :  acc += sum a_i b_(k-i) - sum m_i r_(k-i)   (the second sum as signed products with -r_j),
                  m_k = acc mod 2^29  (r = 1 mod 2^29, so subtracting m_k r clears the low 29 bits without
                  the product with r_0 ever being formed),  acc >>= 29 (arithmetic);
                  the last digit is taken as m_8 - 2^29, which adds r to the result and keeps it positive
  columns 9..17:  the same sums, limb k - 9 of the result = acc mod 2^29, acc >>= 29.
190 instructions: 153 multiply-adds, 18 shifts, 18 and/or, 1 increment.
usage: python scripts/gen_mont29_asm.py [--check-only]"""
import os
import random
import sys

N, BITS = 9, 29
MASK = (1 << BITS) - 1
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
RL = [(R >> (BITS * i)) & MASK for i in range(N)]

# the result registers also hold the digits m_k: limb j of the result is written after the last use of m_j


def program():
    """List of (opcode, dst, src...) with symbolic operands."""
    p = []
    for k in range(2 * N):
        first = True
        for i in range(N):
            j = k - i
            if 0 <= j < N:
                p.append(("mad_u", "acc", ("a", i), ("b", j), "zero" if (k == 0 and first) else "acc"))
                first = False
        for i in range(N):
            j = k - i
            if 1 <= j < N and i < min(k, N):
                p.append(("mad_i", "acc", ("m", i), ("nr", j), "acc"))
        if k < N - 1:
            p.append(("and", ("m", k), "mask", "acc_lo"))
        elif k == N - 1:
            p.append(("or", ("m", k), "high", "acc_lo"))
        else:
            p.append(("and", ("out", k - N), "mask", "acc_lo"))
        if k < 2 * N - 1:
            p.append(("ashr", "acc"))
        if k == N - 1:
            p.append(("inc", "acc"))
    return p


def interpret(a, b):
    """Runs the program on limb lists a, b with 64-bit wrap-around semantics."""
    M64 = (1 << 64) - 1

    def s32(x):
        x &= 0xffffffff
        return x - (1 << 32) if x >> 31 else x

    def s64(x):
        x &= M64
        return x - (1 << 64) if x >> 63 else x

    acc = 0
    m = [0] * N
    out = [0] * N
    nr = [(-RL[j]) & 0xffffffff for j in range(N)]
    for ins in program():
        op = ins[0]
        if op == "mad_u":
            x = a[ins[2][1]]
            y = b[ins[3][1]]
            base = 0 if ins[4] == "zero" else acc
            acc = (base + x * y) & M64
        elif op == "mad_i":
            acc = (acc + s32(m[ins[2][1]]) * s32(nr[ins[3][1]])) & M64
        elif op == "and":
            v = acc & MASK
            if ins[1][0] == "m":
                m[ins[1][1]] = v
            else:
                out[ins[1][1]] = v
        elif op == "or":
            m[ins[1][1]] = (acc & 0xffffffff) | 0xe0000000
        elif op == "ashr":
            acc = (s64(acc) >> BITS) & M64
        elif op == "inc":
            acc = (acc + 1) & M64
    return out


def check():
    rng = random.Random(7)
    for trial in range(2000):
        if trial == 0:
            av, bv = 0, 0
        elif trial == 1:
            av, bv = (1 << 261) - 1, R - 1
        else:
            av, bv = rng.getrandbits(261), rng.randrange(R)
        # data limbs may be slightly unnormalised: up to 2^30 per limb is within the signed range of the columns
        a = [(av >> (BITS * i)) & MASK for i in range(N)]
        if trial % 3 == 2:
            a = [x + rng.getrandbits(29) for x in a]
            av = sum(x << (BITS * i) for i, x in enumerate(a))
        b = [(bv >> (BITS * i)) & MASK for i in range(N)]
        out = interpret(a, b)
        got = sum(x << (BITS * i) for i, x in enumerate(out))
        assert (got << 261) % R == (av * bv) % R, trial
        assert got <= (av * bv >> 261) + R, trial
    return True


def text(out0, a0, b0, nr0, smask, shigh, acc, acc_lo):
    def opnd(x):
        if x == "acc":
            return acc
        if x == "acc_lo":
            return acc_lo
        if x == "zero":
            return "0"
        if x == "mask":
            return "%%%d" % smask
        if x == "high":
            return "%%%d" % shigh
        kind, i = x
        if kind in ("m", "out"):
            return "%%%d" % (out0 + i)
        if kind == "a":
            return "%%%d" % (a0 + i)
        if kind == "b":
            return "%%%d" % (b0 + i)
        if kind == "nr":
            return "%%%d" % (nr0 + i - 1)
        raise ValueError(x)

    lines = []
    for ins in program():
        op = ins[0]
        if op == "mad_u":
            lines.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (acc, opnd(ins[2]), opnd(ins[3]), opnd(ins[4])))
        elif op == "mad_i":
            lines.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (acc, opnd(ins[2]), opnd(ins[3]), acc))
        elif op == "and":
            lines.append("v_and_b32 %s, %s, %s" % (opnd(ins[1]), opnd(ins[2]), acc_lo))
        elif op == "or":
            lines.append("v_or_b32 %s, %s, %s" % (opnd(ins[1]), opnd(ins[2]), acc_lo))
        elif op == "ashr":
            lines.append("v_ashrrev_i64 %s, 29, %s" % (acc, acc))
        elif op == "inc":
            lines.append("v_lshl_add_u64 %s, %s, 0, 1" % (acc, acc))
    return lines


def write(name, header, lines):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plonk_amd", "csrc", name)
    with open(path, "w") as f:
        for h in header:
            f.write("// %s\n" % h)
        for ln in lines:
            f.write('"%s\\n"\n' % ln)


if __name__ == "__main__":
    assert check()
    single = text(0, 9, 18, 27, 35, 36, "v[2:3]", "v2")
    print("%d instructions per product, checked against big-integer arithmetic" % len(single), file=sys.stderr)
    if "--check-only" not in sys.argv:
        write("mont29_asm.inc",
              ["Generated by scripts/gen_mont29_asm.py: one Montgomery product on 9 x 29-bit limbs for gfx950.",
               "Operands: %0-%8 result, %9-%17 a, %18-%26 b, %27-%34 -r_1..-r_8, %35 2^29 - 1, %36 ~(2^29 - 1);",
               "accumulator v[2:3], vcc clobbered."], single)
