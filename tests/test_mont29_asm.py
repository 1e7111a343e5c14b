"""The hand-scheduled Montgomery product of the transforms (plonk_amd/csrc/mont29_asm.inc) is generated: the
generator's instruction-level model is checked against big-integer arithmetic, and the committed file must be
what the generator writes."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator():
    spec = importlib.util.spec_from_file_location("gen_mont29_asm", os.path.join(ROOT, "scripts", "gen_mont29_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_instruction_sequence_computes_the_montgomery_product():
    assert _generator().check()


def test_committed_sequence_is_the_generated_one():
    gen = _generator()
    lines = gen.text(0, 9, 18, 27, 35, 36, "v[2:3]", "v2")
    with open(os.path.join(ROOT, "plonk_amd", "csrc", "mont29_asm.inc")) as f:
        body = [ln.strip() for ln in f if ln.startswith('"')]
    assert body == ['"%s\\n"' % ln for ln in lines]
    assert len(lines) == 189
