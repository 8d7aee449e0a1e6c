"""The build recipe of the C-ABI library: flags that are there for correctness, not speed."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_amortised_translation_unit_is_built_without_packed_f32_instructions():
    """`amort_kernel.hip` runs its kernels beside each other on two streams; packed-f32 VALU instructions returned wrong values
    beside a bf16-MFMA kernel on the same CUs (profiles/r4/x6_notes.txt section 4).  The GPU suite checks the effect itself
    (test_narrow_weight_gradient_is_bit_exact_beside_matrix_core_products_on_a_second_stream); this one keeps the flag in the
    recipe `__graft_entry__.build()` runs."""
    text = open(os.path.join(ROOT, "brancher_amd", "csrc", "Makefile")).read()
    rule = re.search(r"^build/amort_kernel\.o: FLAGS \+= (.*)$", text, re.M)
    assert rule, "no per-object flags for amort_kernel.o"
    assert "-target-feature" in rule.group(1) and "-packed-fp32-ops" in rule.group(1)
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert '"make"' in entry and "csrc" in entry      # build() drives this Makefile
