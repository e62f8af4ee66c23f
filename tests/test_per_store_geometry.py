"""CPU: the closed-form geometry of the sum tree's N-row vector store (gymrl_amd/csrc/per_store_device.hpp, the functions
per.hip's kernels call) against the oracle's member-list restatement of the same definition, and the definition itself
against the reference's one-row SumTree.update at N = 1 (rainbow_dqn_cartpole.py:122-128)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("per_store") / "libper_store_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                           os.path.join(ROOT, "tests", "per_store_host.cpp")])
    L = C.CDLL(so)
    L.host_tree_store.restype = None
    return L


def _host_store(L, tree, cap, start, B, prio=None, scalar=0.0):
    pr = None if prio is None else np.ascontiguousarray(prio, np.float64)
    L.host_tree_store(tree.ctypes.data_as(C.c_void_p), C.c_int64(cap), C.c_int64(start),
                      None if pr is None else pr.ctypes.data_as(C.c_void_p), C.c_double(scalar), C.c_int(B))


@pytest.mark.parametrize("cap", [1, 2, 3, 5, 6, 7, 8, 20, 33, 64, 100, 1000, 1024, 20000, 1 << 15])
def test_vector_store_geometry_matches_member_lists(host, oracle, cap):
    rng = np.random.default_rng(cap)
    ref = oracle.SumTree(cap)
    tree = np.zeros(2 * cap - 1, np.float64)
    rounds = 60 if cap <= 1024 else 12
    for rnd in range(rounds):
        B = int(rng.integers(1, cap + 1)) if rnd % 3 else min(cap, int(rng.integers(1, 9)))
        if rnd == 1:
            B = cap                                                    # the whole ring, from a random position
        start = int(rng.integers(0, 4 * cap))                          # cursors beyond cap wrap like the kernel's % cap
        if rnd % 2:
            pr = rng.random(B) * 4 + 1e-3
            _host_store(host, tree, cap, start, B, prio=pr)
            ref.update_many(idx_start=start, prio=pr)
        else:
            p = float(rng.random() * 3 + 0.1)
            _host_store(host, tree, cap, start, B, scalar=p)
            ref.update_many(idx_start=start, prio_scalar=p, B=B)
        assert np.array_equal(tree, ref.tree), (cap, rnd, B, start)
    # the tree is still a sum tree (every internal node ~ the sum of its children; incremental drift only)
    for t in range(cap - 1):
        assert abs(tree[t] - (tree[2 * t + 1] + tree[2 * t + 2])) <= 1e-9 * max(1.0, tree[t])


def test_vector_store_chunks_of_8192(host, oracle):
    cap = 20000
    rng = np.random.default_rng(3)
    ref = oracle.SumTree(cap)
    tree = np.zeros(2 * cap - 1, np.float64)
    for B, start in ((20000, 7), (8193, 19990), (16384, 12345)):
        pr = rng.random(B) + 0.5
        _host_store(host, tree, cap, start, B, prio=pr)
        ref.update_many(idx_start=start, prio=pr)
        assert np.array_equal(tree, ref.tree)


@pytest.mark.parametrize("cap", [5, 16, 20, 1000])
def test_vector_store_at_one_row_is_the_reference_update(oracle, cap):
    """B = 1: node := node + change, the reference's SumTree.update — the vector-store definition changes nothing the
    reference defines (its fixtures in tests/golden/sumtree.npz are one-row stores)."""
    rng = np.random.default_rng(cap)
    a, b = oracle.SumTree(cap), oracle.SumTree(cap)
    for _ in range(300):
        i, p = int(rng.integers(0, cap)), float(rng.random() * 5)
        a.update(i, p)                                                 # orc_tree_update: :122-128 verbatim
        b.update_many(idx_start=i, prio_scalar=p, B=1)                 # the vector store with one row
        assert np.array_equal(a.tree, b.tree)
