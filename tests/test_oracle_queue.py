"""Indexed binary heap (src/priorityqueue.jl) -- deterministic checks in the style of test/priority.jl:11-32."""
import numpy as np
import pytest

import oracle_lib as O


def test_enqueue_order_peek_and_change_key():
    rng = np.random.default_rng(3)
    n = 257
    vals = rng.random(n)
    q = O.PQ(n)
    for k in range(n):
        q.enqueue(k, vals[k])
        assert q.check()
    assert len(q) == n
    model = vals.copy()
    for _ in range(5000):
        k, v = q.peek()
        assert v == model.min() and model[k] == v
        j = int(rng.integers(n))
        nv = float(rng.random() * 2 if rng.random() < 0.7 else np.inf)
        q[j] = nv
        model[j] = nv
        assert q[j] == nv
    assert q.check()


def test_pop_sequence_is_sorted():
    rng = np.random.default_rng(4)
    n = 100
    vals = rng.random(n)
    q = O.PQ(n)
    for k in range(n):
        q.enqueue(k, vals[k])
    out = []
    for _ in range(n):
        k, v = q.peek()
        out.append(v)
        q[k] = np.inf
    assert out == sorted(vals.tolist())


def test_ties_are_resolved_deterministically():
    q = O.PQ(4)
    for k in range(4):
        q.enqueue(k, 1.0)
    k0, v0 = q.peek()
    assert v0 == 1.0 and k0 == 0
    q[0] = 2.0  # percolate_down: on equal children the RIGHT child is taken (src/priorityqueue.jl:50)
    assert q.peek() == (2, 1.0)
