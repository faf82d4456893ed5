"""The C-ABI library loads without a GPU and exports every symbol include/nmpc.h declares.
No compute calls here (they need a device); argument validation that is decided on the host is checked."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from iterative_learning_nmpc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.load()


def test_header_symbols_are_exported_and_bound(lib):
    from iterative_learning_nmpc_amd import _lib
    header = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))))
    declared = set(re.findall(r"\b(nmpc_[a-z_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None


def test_model_dims(lib):
    v = [ctypes.c_int() for _ in range(4)]
    assert lib.nmpc_model_dims(1, *[ctypes.byref(x) for x in v]) == 0
    assert [x.value for x in v] == [12, 12, 16, 16]
    assert lib.nmpc_model_dims(0, *[ctypes.byref(x) for x in v]) == 0
    assert [x.value for x in v] == [4, 2, 0, 4]
    assert lib.nmpc_model_dims(9, None, None, None, None) == -1
    assert lib.nmpc_model_dims(1, None, None, None, None) == 0       # out pointers are optional


def test_create_rejects_bad_dims_on_the_host(lib):
    from iterative_learning_nmpc_amd import _lib
    h = ctypes.c_void_p()
    for dims in (_lib.NmpcDims(5, 50, 8, 0), _lib.NmpcDims(1, 0, 8, 0), _lib.NmpcDims(1, 50, 0, 0),
                 _lib.NmpcDims(1, 50, 8, 3)):
        assert lib.nmpc_create(ctypes.byref(dims), 0, ctypes.byref(h)) == -1
        assert not h.value and lib.nmpc_last_error(None)
    assert lib.nmpc_create(None, 0, ctypes.byref(h)) == -1


def test_dataset_calls_reject_bad_arguments_on_the_host(lib):
    """include/nmpc_dataset.h: argument checks come before any launch (no GPU needed)."""
    one = ctypes.c_void_p(8)                     # a non-null placeholder, never dereferenced on these paths
    assert lib.nmpc_ring_append(None, 4, 1, one, 8, 0, None) == -1
    assert lib.nmpc_ring_append(one, 4, 1, one, 8, 8, None) == -1            # first_slot outside the ring
    assert lib.nmpc_ring_append(one, 0, 1, one, 8, 0, None) == -1
    assert lib.nmpc_ring_append(None, 4, 0, None, 8, 0, None) == 0           # nothing to append
    assert lib.nmpc_column_stats(one, 10, 65, one, one, one, None) == -1     # cols <= 64
    assert lib.nmpc_column_stats(one, 0, 4, one, one, one, None) == -1
    assert b"cols" in lib.nmpc_dataset_last_error()
    assert lib.nmpc_column_stats_scratch(44) >= 44 and lib.nmpc_column_stats_scratch(0) == 0
    assert lib.nmpc_assemble_batch(one, 44, one, None, 1, one, 3, None, None, one, 12, 10, one, 4, one, one, None) == -1
    assert lib.nmpc_assemble_batch(one, 44, None, None, 1, None, 3, None, None, one, 12, 10, one, 4, one, one, None) == -1
    assert lib.nmpc_assemble_batch(None, 44, None, None, 1, None, 0, None, None, None, 0, 10, None, 0, None, None, None) == 0


def test_torque_model_is_validated_on_the_host(lib):
    """include/nmpc_torque.h: a malformed tree is rejected before any device call."""
    from iterative_learning_nmpc_amd import _lib
    import numpy as np
    n = 3
    arr = dict(parent=np.array([-1, 0, 1], np.int32), type=np.zeros(n, np.int32), axis=np.tile(np.float32([0, 0, 1]), (n, 1)),
               placement=np.tile(np.float32([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]), (n, 1)), mass=np.ones(n, np.float32),
               com=np.zeros((n, 3), np.float32), inertia=np.tile(np.float32([1, 0, 0, 1, 0, 1]), (n, 1)),
               foot_joint=np.array([2], np.int32), foot_offset=np.zeros((1, 3), np.float32))

    def create(n_act=2, **over):
        a = {k: np.ascontiguousarray(over.get(k, v)) for k, v in arr.items()}
        m = _lib.NmpcTreeModel()
        m.n_joints, m.n_actuated, m.n_feet = n, n_act, 1
        for k, v in a.items():
            setattr(m, k, v.ctypes.data_as(ctypes.POINTER(ctypes.c_int if v.dtype == np.int32 else ctypes.c_float)))
        h = ctypes.c_void_p()
        return lib.nmpc_torque_create(ctypes.byref(m), 0, ctypes.byref(h)), lib.nmpc_torque_last_error(None)

    assert create(parent=np.array([-1, 2, 1], np.int32)) == (-1, b"parents must come before their children")
    assert create(type=np.array([0, 2, 0], np.int32))[0] == -1
    assert create(axis=np.tile(np.float32([0, 0, 2]), (n, 1)))[0] == -1
    assert create(foot_joint=np.array([3], np.int32))[0] == -1
    assert create(n_act=4)[0] == -1
    assert lib.nmpc_torque_create(None, 0, None) == -1
    assert lib.nmpc_id_torques_batch(None, 1, None, None, None, None, None, None) == -1


def test_python_layer_refuses_to_run_without_a_device():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    with pytest.raises(RuntimeError, match="no CPU path"):
        BatchedNmpcSolver(1, 50, 4)


def test_product_never_imports_the_oracle():
    """No import, include, link or path reference to oracle/ anywhere in the shipped package."""
    pkg = os.path.join(ROOT, "iterative_learning_nmpc_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|#\s*include[^\n]*oracle|liboracle|oracle[/\\.]", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".h", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), os.path.join(dirpath, f)
