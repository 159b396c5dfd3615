"""The measurement harness's own logic (bench.py), on the CPU: which batch a step takes, what `parity` compares with what,
when two passes count as the same results.  No GPU, no kernels."""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_logic_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("depth,n_batches", [(3, 3), (3, 2), (3, 4), (1, 3), (2, 4), (4, 2), (3, 6)])
def test_rotation_changes_every_lanes_batch(bench, depth, n_batches):
    """Consecutive steps of one lane (lane = step % depth: one context with its own mask) take different batches, so every
    context's mask sees different frames from one of its batches to the next; every batch is used."""
    seq = [bench.batch_index(i, depth, n_batches) for i in range(240)]
    assert set(seq) == set(range(n_batches))
    for lane in range(depth):
        mine = seq[lane::depth]
        assert all(a != b for a, b in zip(mine, mine[1:])), (depth, n_batches, lane)


def fake_step(base, n_cam, max_points, P=4):
    """records / correspondence output of base time step `base`: (lists per camera, oracle-style dict)"""
    rng = np.random.default_rng(base)
    lists = [rng.integers(0, 1000, (int(rng.integers(0, 4)), 2)).tolist() for _ in range(n_cam)]
    k = int(rng.integers(0, P))
    ref = {"root": np.arange(k, dtype=np.int32), "groups": rng.integers(0, 1000, (k, n_cam, 2)).astype(float),
           "xyz": rng.normal(size=(k, 3)), "order": rng.permutation(k).astype(np.int32)}
    return lists, ref


def test_parity_report_follows_the_batches_time_shift(bench):
    """Slot s of a batch with time shift d holds base time step (s + d) % T: the report must compare it with THAT step's
    oracle results -- and notice a wrong centroid, a wrong group, a 3-D point off by more than 1e-7."""
    T, C, MP, P = 12, 3, 8, 4
    results = [(b,) + fake_step(b, C, MP, P) for b in range(T)]

    def batch(shift, spoil=None):
        rec = np.zeros((T * C, 2 + 2 * MP), np.int32)
        out = {"n": np.zeros(T, np.int32), "grp": np.zeros((T, P, C, 2)), "xyz": np.zeros((T, P, 3)), "order": np.zeros((T, P), np.int32)}
        for s in range(T):
            lists, ref = fake_step((s + shift) % T, C, MP, P)
            for c in range(C):
                rec[s * C + c, 0] = len(lists[c])
                rec[s * C + c, 2:2 + 2 * len(lists[c])] = np.array(lists[c], np.int32).reshape(-1)
                rec[s * C + c, 2 + 2 * len(lists[c]):] = 77  # whatever an earlier batch left beyond the count
            k = len(ref["root"])
            out["n"][s] = k
            out["grp"][s, :k], out["xyz"][s, :k], out["order"][s, :k] = ref["groups"], ref["xyz"], ref["order"]
        if spoil == "centroid":
            i = int(np.argmax(rec[:, 0] > 0))
            rec[i, 2] += 1
        if spoil == "xyz":
            s = int(np.argmax(out["n"] > 0))
            out["xyz"][s, 0, 0] += 1e-6
        if spoil == "group":
            s = int(np.argmax(out["n"] > 0))
            out["grp"][s, 0, 0, 0] += 1
        return shift, rec, out

    ok = bench.parity_report(results, [batch(0), batch(5), batch(10)], C, MP, T)
    assert ok["ok"] and ok["time_steps_compared"] == 3 * T and ok["centroid_mismatches"] == 0 and ok["rmse_3d_vs_oracle"] == 0.0
    assert not bench.parity_report(results, [batch(5)[:1] + batch(4)[1:]], C, MP, T)["ok"]  # the wrong shift is a mismatch
    bad = bench.parity_report(results, [batch(0), batch(5, "centroid")], C, MP, T)
    assert not bad["ok"] and bad["centroid_mismatches"] == 1
    bad = bench.parity_report(results, [batch(3, "xyz")], C, MP, T)
    assert not bad["ok"] and bad["max_abs_3d"] > 5e-7 and bad["correspondence_mismatches"] == 0
    bad = bench.parity_report(results, [batch(7, "group")], C, MP, T)
    assert not bad["ok"] and bad["correspondence_mismatches"] == 1


def test_same_results_ignores_what_lies_beyond_the_counts(bench):
    T, P, C, R = 4, 3, 2, 10
    rec = torch.zeros((T * C, R), dtype=torch.int32)
    rec[:, 0] = torch.tensor([1, 0, 2, 1, 0, 0, 3, 1])
    rec[:, 2:] = torch.arange(T * C * (R - 2), dtype=torch.int32).reshape(T * C, R - 2)
    out = {"n": torch.tensor([2, 0, -2, 1], dtype=torch.int32), "xyz": torch.randn(T, P, 3, dtype=torch.float64),
           "grp": torch.randn(T, P, C, 2, dtype=torch.float64), "order": torch.zeros((T, P), dtype=torch.int32)}
    rec2 = rec.clone()
    out2 = {k: v.clone() for k, v in out.items()}
    rec2[1, 2:] = -5            # image 1 holds no point: stale words
    rec2[0, 4:] = -7            # image 0 holds one point: words beyond it
    out2["xyz"][1] = 9.0        # time step 1 has no root; step 2 failed (-2): neither is compared
    out2["xyz"][2] = 9.0
    out2["xyz"][3, 1:] = 9.0
    assert bench.same_results(rec, out, rec2, out2)
    rec3 = rec2.clone()
    rec3[0, 2] += 1             # a live centroid word
    assert not bench.same_results(rec, out, rec3, out2)
    out3 = {k: v.clone() for k, v in out2.items()}
    out3["xyz"][0, 1, 2] += 1e-9
    assert not bench.same_results(rec, out, rec2, out3)
    out4 = {k: v.clone() for k, v in out2.items()}
    out4["n"][3] = 2
    assert not bench.same_results(rec, out, rec2, out4)
