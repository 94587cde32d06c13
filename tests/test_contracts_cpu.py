"""The parity contract of the full-size GPU tests, tested on its own (CPU, synthetic probabilities): tests/contracts.py decides whether a
device result passes, so each of its exits is exercised here with data built to take exactly that exit - and the cases it must REJECT are
checked to be rejected (a numeric regression must not be able to pass as a "re-decided query")."""
import numpy as np
import pytest

from contracts import MAX_REDECIDED, TAU_PROB, category_map, class_probability_contract

Q, K = 100, 20


def _reference(seed=0, confident=True):
    """[Q, K+1] log-probabilities: most queries confident (top-2 margin far above the bounds), a few near-ties."""
    rng = np.random.default_rng(seed)
    logits = rng.normal(0.0, 1.0, size=(Q, K + 1))
    top = rng.integers(0, K + 1, size=Q)
    logits[np.arange(Q), top] += 9.0 if confident else 0.5
    logits[:5] = rng.normal(0.0, 0.05, size=(5, K + 1))       # five undecided queries: every label is a near-tie
    logits -= np.log(np.exp(logits).sum(-1, keepdims=True))
    return logits


def _perturb(logp, q, amount):
    """Move `amount` of probability mass of query q from its top class to its runner-up (stays a distribution)."""
    p = np.exp(logp)
    order = np.argsort(p[q])
    a, b = order[-1], order[-2]
    p[q, a] -= amount
    p[q, b] += amount
    return np.log(np.clip(p, 1e-30, None))


def test_clean_result_passes_and_reports_per_query_errors():
    ref = _reference()
    got = _perturb(ref, 7, 0.5 * TAU_PROB)
    e = class_probability_contract(got, ref, K)
    assert e.shape == (Q,) and abs(e[7] - 0.5 * TAU_PROB) < 1e-9 and np.delete(e, 7).max() < 1e-12


def test_query_beyond_the_bound_needs_the_attribution_reference():
    ref = _reference()
    got = _perturb(ref, 11, 2.0 * TAU_PROB)
    with pytest.raises(AssertionError, match="no attribution reference"):
        class_probability_contract(got, ref, K)


def test_query_reproduced_by_the_ideal_head_passes():
    """The fp32 head on the device's own features moves the same query by the same amount: the backbone error met a hard decision of the reference."""
    ref = _reference()
    got = _perturb(ref, 11, 2.0 * TAU_PROB)
    ideal = _perturb(ref, 11, 2.0 * TAU_PROB + 0.3 * TAU_PROB)
    calls = []

    def lazy():
        calls.append(1)
        return ideal

    class_probability_contract(got, ref, K, ideal=lazy)
    assert calls == [1]                                    # evaluated once, and only because a query exceeded the bound
    calls.clear()
    class_probability_contract(_perturb(ref, 11, 0.2 * TAU_PROB), ref, K, ideal=lazy)
    assert calls == []


def test_device_head_error_on_a_stable_query_is_rejected():
    """The device differs from the ideal head on the SAME features and the reference itself does not move that query: a defect, not a re-decision."""
    ref = _reference()
    got = _perturb(ref, 11, 2.0 * TAU_PROB)
    ideal = ref.copy()                                     # an exact head makes nothing of the device's features: the error is the device head's
    with pytest.raises(AssertionError, match="no instability probe"):
        class_probability_contract(got, ref, K, ideal=ideal)
    stable = np.full(Q, 1e-3)
    with pytest.raises(AssertionError, match="decides STABLY"):
        class_probability_contract(got, ref, K, ideal=ideal, instability=stable)
    unstable = stable.copy()
    unstable[11] = 0.8 * TAU_PROB                          # the reference moves this query by more than TAU_PROB / 2 under device-sized perturbations
    class_probability_contract(got, ref, K, ideal=ideal, instability=unstable)
    unstable[11], unstable[12] = 1e-3, 0.9                 # another query's instability does not excuse this one
    with pytest.raises(AssertionError, match="decides STABLY"):
        class_probability_contract(got, ref, K, ideal=ideal, instability=unstable)


def test_too_many_or_too_large_redecisions_are_rejected_even_with_proofs():
    ref = _reference()
    got = ref
    for q in range(20, 20 + MAX_REDECIDED + 1):
        got = _perturb(got, q, 2.0 * TAU_PROB)
    with pytest.raises(AssertionError):
        class_probability_contract(got, ref, K, ideal=got)            # every query "explained", but more of them than a picture may have
    big = _perturb(ref, 30, 0.3)
    with pytest.raises(AssertionError):
        class_probability_contract(big, ref, K, ideal=big)            # one query, but moved by more than TAU_REDECIDED


def test_label_flip_on_a_decided_query_is_rejected_and_on_a_near_tie_is_not():
    ref = _reference()
    p = np.exp(ref)
    q = 40
    a, b = np.argsort(p[q])[-1], np.argsort(p[q])[-2]
    flipped = p.copy()
    flipped[q, a], flipped[q, b] = p[q, b], p[q, a]                    # swaps the two top probabilities of a confident query: error ~ its margin
    got = np.log(flipped)
    with pytest.raises(AssertionError):
        class_probability_contract(got, ref, K, ideal=got)
    tie = np.exp(ref)
    a, b = np.argsort(tie[2])[-1], np.argsort(tie[2])[-2]              # query 2 is one of the near-ties: its label may go either way
    tie[2, a], tie[2, b] = tie[2, b], tie[2, a]
    e = class_probability_contract(np.log(tie), ref, K)
    assert e[2] < TAU_PROB


def test_too_few_decided_queries_is_rejected():
    """A reference whose queries are all near-ties tests nothing: the contract insists on a minimum of margin-decided queries."""
    ref = _reference(confident=False)
    with pytest.raises(AssertionError):
        class_probability_contract(ref, ref, K)


def test_category_map_ignores_segment_numbering():
    pan_a = np.array([[0, 1, 1], [2, 2, 0]])
    info_a = [dict(id=1, category_id=17, isthing=True), dict(id=2, category_id=3, isthing=False)]
    pan_b = np.array([[0, 2, 2], [1, 1, 0]])                           # the same picture with the two segments numbered the other way round
    info_b = [dict(id=1, category_id=3, isthing=False), dict(id=2, category_id=17, isthing=True)]
    ca, cb = category_map(pan_a, info_a), category_map(pan_b, info_b)
    assert np.array_equal(ca, cb) and ca[0, 0] == -1 and ca[0, 1] == 17 and ca[1, 0] == 3
    assert np.array_equal(category_map(np.zeros((2, 2), np.int64), []), -np.ones((2, 2), np.int64))
