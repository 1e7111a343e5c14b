"""The verifier (`verifier::run`, plonk/src/verifier/mod.rs:13-42, parts 1-4) on top of the C ABI.

What the reference checks with 15 separate `kzg_verify` calls (30 pairings) is checked here with the same field
equations and ONE folded pairing-product check (`plonk_hip_kzg_verify_batch`): the (commitment, point, value,
proof) tuples are collected in the order the reference asserts them,

    part 1  verify_zero_on_roots_test        protocols.rs:220-232   2 tuples   (verifier/part1.rs:13-64)
    part 2  verify_T_S_zero_test             protocols.rs:789-809   5 tuples   (verifier/part2.rs:13-51)
    part 3  verify_prescribed_permutation_check  protocols.rs:690-719  7 tuples  (verifier/part3.rs:13-50)
    part 4  kzg_verify of the last gate's output  verifier/part4.rs:8-31      1 tuple

and verified at once.  The only device work is what the reference's verifier does too: one commitment of the
degree-1 polynomial through the public inputs (verifier/part1.rs:46), plus decoding / encoding of group elements;
it needs a context with the SRS (two points are enough)."""
import hashlib

import numpy as np

from .backend import kzg_verify_batch
from .prover import Q_MOD, R_MOD, from_mont, root_of_unity, to_mont, trim


class Verifier:
    def __init__(self, backend, d, number_gates, number_public_inputs, tau_g2):
        """backend: a context whose SRS holds at least number_public_inputs points (gp.tau_powers_g1);
        tau_g2: gp.tau_g2, 96 bytes compressed."""
        self.be = backend
        self.d = d
        self.number_gates = number_gates
        self.number_public_inputs = number_public_inputs
        self.tau_g2 = bytes(tau_g2)
        self.w = root_of_unity(d)

    # ------------------------------------------------------------------ helpers
    def _points(self, hexes):
        return self.be.g1_decompress([bytes.fromhex(h) for h in hexes], check_subgroup=True)

    def _encode(self, points):
        return self.be.g1_compress(np.stack([np.asarray(p, dtype=np.uint64).reshape(18) for p in points]))

    def _subtract(self, p, q):
        neg = np.array(q, dtype=np.uint64).reshape(18)
        y = int.from_bytes(neg[6:12].tobytes(), "little")
        neg[6:12] = np.frombuffer(((Q_MOD - y) % Q_MOD).to_bytes(48, "little"), dtype=np.uint64)
        d_in = self.be.to_device(np.stack([np.asarray(p, dtype=np.uint64).reshape(18), neg]))
        d_out = self.be.malloc(144)
        try:
            self.be.g1_sum_dev(d_in, 2, d_out)
            self.be.sync()
            return self.be.from_device(d_out, (18,))
        finally:
            self.be.free(d_in)
            self.be.free(d_out)

    @staticmethod
    def _challenges(encodings, count=None):
        """common/utils.rs:65-82 and :85-117."""
        h = hashlib.blake2s()
        for enc in encodings:
            h.update(bytes(enc))
        if count is None:
            return int.from_bytes(h.digest(), "little") % R_MOD
        base = h.digest()
        return [int.from_bytes(hashlib.blake2s(base + bytes([i & 0xFF])).digest(), "little") % R_MOD for i in range(count)]

    def _vanishing_at(self, roots_step, count, x):
        """prod_{l < count} (x - w^(roots_step l)), in closed form when the roots are a whole subgroup less its
        last element (the circuits of the example's family), else by the product."""
        P, w, d = R_MOD, self.w, self.d
        n = count + 1
        if roots_step == 3 and d == 3 * n:
            return (pow(x, n, P) - 1) * pow(x - pow(w, 3 * count, P), -1, P) % P
        z = 1
        for l in range(count):
            z = z * (x - pow(w, roots_step * l, P)) % P
        return z

    # ------------------------------------------------------------------ verifier::run
    def checks(self, verification_key, proof):
        """(field equations hold, [(com, u, v, proof_point), ...]) for a proof in the JSON form of
        common/proof.rs:220-266 and a verification key {"com_S", "com_W"} (hex)."""
        P, w, d, ng = R_MOD, self.w, self.d, self.number_gates
        if len(proof["pub_inputs"]) < self.number_public_inputs:
            return False, []
        p1, p2, p3 = proof["proof_T_minus_v_zero"], proof["proof_T_S_zero"], proof["proof_T_W_prescribed_permutation"]
        hexes = [proof["com_T"], verification_key["com_S"], verification_key["com_W"], p1[0], p1[2], p1[4],
                 p2[0], p2[2], p2[4], p2[6], p2[8], p2[10],
                 p3[0], p3[1], p3[3], p3[5], p3[7], p3[9], p3[11], p3[13], p3[15], proof["proof_last_gate_KZG"]]
        pts = self._points(hexes)
        (com_T, com_S, com_W, q1, pf_f_r, pf_q_r, q2, pf_T_r, pf_T_w_r, pf_T_w2_r, pf_S_r, pf_q2_r,
         com_t, q3, pf_t_last, pf_t_rp, pf_t_w_rp, pf_q_rp, pf_f_w_rp, pf_g_w_rp, pf_W_w_rp, pf_out) = pts
        tuples = []
        ok = True

        # part 1 (verifier/part1.rs:13-64): T - v vanishes on the public-input rows
        pub_inputs = [int(x) % P for x in proof["pub_inputs"][: self.number_public_inputs]]
        rows_in = [pow(w, d - 1 - i, P) for i in range(self.number_public_inputs)]
        v = trim(self.be.interpolate_polynomial(to_mont(rows_in), to_mont(pub_inputs)))
        if v.shape[0] != self.number_public_inputs:  # "v must be of degree 1" (verifier/part1.rs:38-42)
            return False, []
        com_v = self.be.kzg_commit(v)
        com_T_minus_v = self._subtract(com_T, com_v)
        r1 = self._challenges(self._encode([com_T_minus_v]))
        f_r, q_r = int(p1[1]) % P, int(p1[3]) % P
        z = 1
        for x in rows_in:
            z = z * (r1 - x) % P
        ok = ok and f_r == q_r * z % P
        tuples += [(q1, r1, q_r, pf_q_r), (com_T_minus_v, r1, f_r, pf_f_r)]

        # part 2 (verifier/part2.rs:13-51): the gate constraint on Omega_gates = {w^(3l)}
        r2 = self._challenges([bytes.fromhex(proof["com_T"]), bytes.fromhex(verification_key["com_S"])])
        T_r, T_w_r, T_w2_r, S_r, q2_r = (int(p2[i]) % P for i in (1, 3, 5, 7, 9))
        ok = ok and (S_r * (T_r + T_w_r) + (1 - S_r) * T_r * T_w_r - T_w2_r) % P == \
            q2_r * self._vanishing_at(3, ng, r2) % P
        tuples += [(com_T, r2, T_r, pf_T_r), (com_T, w * r2 % P, T_w_r, pf_T_w_r),
                   (com_T, w * w % P * r2 % P, T_w2_r, pf_T_w2_r), (com_S, r2, S_r, pf_S_r), (q2, r2, q2_r, pf_q2_r)]

        # part 3 (verifier/part3.rs:13-50): W is the prescribed permutation of T over Omega
        r, s_, rp = self._challenges([bytes.fromhex(proof["com_T"]), bytes.fromhex(verification_key["com_W"])], 3)
        t_last, t_rp, t_w_rp, q_rp, f_w_rp, g_w_rp, W_w_rp = (int(p3[i]) % P for i in (2, 4, 6, 8, 10, 12, 14))
        ok = ok and t_last == 1
        ok = ok and (t_w_rp * (r - s_ * w * rp - g_w_rp) - t_rp * (r - s_ * W_w_rp - f_w_rp)) % P == \
            q_rp * (pow(rp, d, P) - 1) % P
        w_rp = rp * w % P
        tuples += [(com_t, pow(w, d - 1, P), t_last, pf_t_last), (com_t, rp, t_rp, pf_t_rp), (com_t, w_rp, t_w_rp, pf_t_w_rp),
                   (q3, rp, q_rp, pf_q_rp), (com_T, w_rp, f_w_rp, pf_f_w_rp), (com_T, w_rp, g_w_rp, pf_g_w_rp),
                   (com_W, w_rp, W_w_rp, pf_W_w_rp)]

        # part 4 (verifier/part4.rs:8-31): the output is the value of T on the last gate's output row
        tuples.append((com_T, pow(w, 3 * ng - 1, P), int(proof["output"]) % P, pf_out))
        return ok, tuples

    def run(self, verification_key, proof, rho=None):
        """True when the reference's `verifier::run` would pass all its assertions."""
        try:
            ok, tuples = self.checks(verification_key, proof)
        except Exception as e:  # an element that does not decode is a failed deserialisation in the reference
            from .backend import ReferencePanic
            if isinstance(e, (ReferencePanic, ValueError, KeyError, IndexError)):
                return False
            raise
        if not ok:
            return False
        coms = np.stack([np.asarray(t[0], dtype=np.uint64).reshape(18) for t in tuples])
        proofs = np.stack([np.asarray(t[3], dtype=np.uint64).reshape(18) for t in tuples])
        return kzg_verify_batch(self.tau_g2, coms, to_mont([t[1] for t in tuples]), to_mont([t[2] for t in tuples]),
                                proofs, rho)


def run(backend, setup, verification_key, proof):
    """`verifier::run(setup, verification_key, proof)` (plonk/src/verifier/mod.rs:13-42) for the JSON objects of
    plonk/data/{srs,verification_key,proof}.json; `backend` must hold setup's tau_powers_g1."""
    v = Verifier(backend, setup["d"], setup["number_gates"], setup["number_public_inputs"],
                 bytes.fromhex(setup["gp"]["tau_g2"]))
    return v.run(verification_key, proof)
