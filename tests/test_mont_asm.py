"""The hand-scheduled Montgomery products (plonk_amd/csrc/mont29_asm.inc for the transforms, mont28_*_asm.inc for
the bucket accumulation) are generated: the generator's instruction-level model is checked against big-integer
arithmetic, and the committed files must be what the generator writes."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator():
    spec = importlib.util.spec_from_file_location("gen_mont_asm", os.path.join(ROOT, "scripts", "gen_mont_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


GEN = _generator()


@pytest.mark.parametrize("name,field,kind", [(n, f["name"], k) for n, f, k in GEN.FILES])
def test_instruction_sequence_computes_the_montgomery_product(name, field, kind):
    f = GEN.FR if field == "fr" else GEN.FQ
    assert GEN.check(f, kind, trials=200)


@pytest.mark.parametrize("name,field,kind", [(n, f["name"], k) for n, f, k in GEN.FILES])
def test_committed_sequence_is_the_generated_one(name, field, kind):
    f = GEN.FR if field == "fr" else GEN.FQ
    with open(os.path.join(ROOT, "plonk_amd", "csrc", name)) as fh:
        assert fh.readlines() == GEN.render(f, kind)


def test_instruction_counts():
    counts = {name: len(GEN.text(f, kind)) for name, f, kind in GEN.FILES}
    assert counts == {"mont29_asm.inc": 189, "mont28_mul_asm.inc": 461, "mont28_sqr_asm.inc": 370,
                      "mont28_mul2_asm.inc": 657}
