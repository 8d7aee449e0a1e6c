"""Host logic of the program shares (lowering.Program.shares) on CPU: every share keeps all sampling
records, the model's log-prob work is partitioned — by records and, inside a multi-element record, by elements — and only
share 0 carries the posterior's own entropy / log q terms."""
import numpy as np
import pytest

from brancher_amd import lowering, workloads as W

OP_NAFF, OP_NODE, OP_REC_BEGIN, OP_REC_END = 1, 2, 5, 6
F_SAMPLE = 1


def _units(code, records):
    """(sink units, non-sink instruction count, sum |imm| of sampling instructions)"""
    sink_units, plain, own = [], 0, 0.0
    for r in records:
        body = code[r["code_begin"]:r["code_end"]]
        if r["flags"]:
            # a sink unit is identified by its instructions' opcodes and first-operand start, one per element
            key = tuple(int(w[0]) & 0xFFFFFF for w in body) + tuple(int(w[1]) for w in body)
            sink_units.append((key, int(r["n_elems"])))
        else:
            plain += len(body)
            for w in body:
                op, flags = int(w[0]) & 0xFF, (int(w[0]) >> 8) & 0xFF
                if op in (OP_NAFF, OP_NODE) and flags & F_SAMPLE:
                    own += abs(np.array([w[6], w[7]], dtype=np.uint32).view(np.float32)).sum()
    return sink_units, plain, own


@pytest.mark.parametrize("builder,kwargs", [("build_readme_ar", dict(T=20)), ("build_beta_binomial", dict(n_obs=30)),
                                            ("build_multivariate_regression", dict(n=100)), ("build_vector_latent", dict(n_obs=9, dim=4))])
def test_shares_partition_the_model_records(builder, kwargs):
    prog = lowering.lower(getattr(W, builder)(W.native_api(), **kwargs), None, "pathwise")
    full_sinks, full_plain, full_own = _units(prog.code, prog.records)
    total = sum(n for _, n in full_sinks)
    assert prog.shares, "a program with this many model terms has shares"
    for V, parts in prog.shares.items():
        assert len(parts) == V
        elems = 0
        for v, (code, records) in enumerate(parts):
            sinks, plain, own = _units(code, records)
            assert plain == full_plain                      # every share samples the whole posterior
            assert (own == pytest.approx(full_own)) if v == 0 else (own == 0.0)
            elems += sum(n for _, n in sinks)
            assert len(code) <= len(prog.code)
        assert elems == total                               # every element of every model record exactly once


def test_blackbox_programs_have_no_shares():
    prog = lowering.lower(W.build_readme_ar(W.native_api(), T=20), None, "blackbox")
    assert prog.shares == {}
