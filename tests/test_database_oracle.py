"""The database oracle (oracle/database_oracle.py) against vectors recorded from the reference's own
Database class (tests/golden/make_golden_database.py): ring placement, statistics, normalised items,
and the npz file the reference itself wrote."""
import os

import numpy as np
import pytest

from oracle.database_oracle import DatabaseOracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def replay(g):
    limit, n_state, n_action, n_chunks = [int(v) for v in g["dims"]]
    db = DatabaseOracle(limit, norm_input=True, goal_type=str(g["goal_type"]))
    for c in range(n_chunks):
        db.append(g[f"chunk{c}.states"].astype(np.float64), g[f"chunk{c}.actions"].astype(np.float64),
                  vc_goals=g[f"chunk{c}.vc_goals"].astype(np.float64), cc_goals=g[f"chunk{c}.cc_goals"].astype(np.float64))
        yield c, db


@pytest.mark.parametrize("name", ["vc", "cc", "tiny"])
def test_database_oracle_matches_reference_database(name):
    g = np.load(os.path.join(GOLD, f"database_{name}.npz"))
    for c, db in replay(g):
        assert [db.start, db.length] == list(g[f"after{c}.start_length"])            # ring bookkeeping: exact
        np.testing.assert_allclose(db.states_mean, g[f"after{c}.states_mean"], rtol=0, atol=0)
        np.testing.assert_allclose(db.states_std, g[f"after{c}.states_std"], rtol=0, atol=0)
    L = len(db)
    for f in DatabaseOracle.FIELDS:                                                   # physical row order: exact
        assert np.array_equal(db.rows[f][:L], g[f"final.{f}"])
    assert np.array_equal(db.states_norm(), g["final.states_norm"])
    assert np.array_equal(db.cc_goals_mean, g["final.cc_goals_mean"]) and np.array_equal(db.cc_goals_std, g["final.cc_goals_std"])
    ms = db.get_database_mean_std()
    assert np.array_equal(np.asarray(ms[2]), g["final.goal_mean"]) and np.array_equal(np.asarray(ms[3]), g["final.goal_std"])
    x, y = db.batch(g["batch.idx"])
    assert x.dtype == np.float32 and np.array_equal(x, g["batch.x"]) and np.array_equal(y, g["batch.y"])
    db.norm_input = False
    assert np.array_equal(db.batch(g["batch.idx"])[0], g["batch.x_raw"])
    assert db.get_database_mean_std() is None


def test_database_oracle_reads_the_file_the_reference_saved(tmp_path):
    ref = np.load(os.path.join(GOLD, "database_loaded_by_reference.npz"))
    db = DatabaseOracle(limit=96)
    db.load_from_npz(os.path.join(GOLD, "database_saved_by_reference.npz"))
    assert len(db) == int(ref["length"])
    assert np.array_equal(db.states_mean, ref["states_mean"]) and np.array_equal(db.states_std, ref["states_std"])
    assert np.array_equal(db.states_norm(), ref["states_norm"])
    # and writes the same schema back
    out = str(tmp_path / "again.npz")
    db.save_as_npz(out)
    a, b = np.load(out), np.load(os.path.join(GOLD, "database_saved_by_reference.npz"))
    assert sorted(a.files) == sorted(b.files) == sorted(DatabaseOracle.FIELDS)
    assert all(np.array_equal(a[f], b[f]) and a[f].dtype == b[f].dtype for f in a.files)


def test_append_needs_a_goal():
    with pytest.raises(ValueError):
        DatabaseOracle(4).append(np.zeros((2, 3)), np.zeros((2, 1)))
