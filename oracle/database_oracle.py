"""CPU restatement of the reference's training database (aggregation + normalisation)
-- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (iterative_learning_nmpc_amd/) never does.

What it restates, in numpy float64 (DAgger/utils/database.py; Behavior_Cloning/utils/database.py is the
same file):
  * `Database.append` :105-154 -- a ring of `limit` rows: while there is room the length grows, afterwards
    the start moves; a row lands at (start + length - 1) % limit after that update,
  * `Database.calc_input_mean_std` :208-255 -- np.mean / np.std (population) over the PHYSICAL rows
    [0, length); states normalised from column 1 on (the phase stays), 'vc' goals left alone (mean 0,
    std 1), 'cc' goals normalised in every column,
  * `Database.__getitem__` :54-84 -- x = hstack(state_norm, goal), y = action, by physical index; the
    training loop casts to fp32 (train_locosafedagger.py:95),
  * `save_as_npz` / `load_from_npz` :273-315 -- keys states, vc_goals, cc_goals, actions.
PARITY PINNED: tests/test_database_oracle.py checks it against vectors recorded from the reference's
own class (tests/golden/make_golden_database.py, database_*.npz)."""
import numpy as np


class DatabaseOracle:
    FIELDS = ("states", "vc_goals", "cc_goals", "actions")

    def __init__(self, limit, norm_input=True, goal_type="vc"):
        assert goal_type in ("vc", "cc")
        self.limit, self.length, self.start = int(limit), 0, 0
        self.norm_input, self.goal_type = bool(norm_input), goal_type
        self.rows = {f: None for f in self.FIELDS}
        self.states_mean = self.states_std = self.cc_goals_mean = self.cc_goals_std = None
        self.vc_goals_mean, self.vc_goals_std = 0.0, 1.0

    def __len__(self):
        return self.length

    def append(self, states, actions, vc_goals=None, cc_goals=None):
        if vc_goals is None and cc_goals is None:
            raise ValueError("both vc_goals and cc_goals cant be empty!")
        given = {"states": states, "actions": actions, "vc_goals": vc_goals, "cc_goals": cc_goals}
        for f, a in given.items():
            if a is not None and self.rows[f] is None:
                self.rows[f] = np.zeros((self.limit, np.shape(a)[1]))
        for i in range(len(states)):
            if self.length < self.limit:
                self.length += 1
            else:
                self.start = (self.start + 1) % self.limit
            slot = (self.start + self.length - 1) % self.limit
            for f, a in given.items():
                if a is not None:
                    self.rows[f][slot] = a[i]
        self.calc_input_mean_std()

    def calc_input_mean_std(self):
        s = self.rows["states"][:self.length]
        self.states_mean, self.states_std = np.mean(s, axis=0), np.std(s, axis=0)
        if self.rows["cc_goals"] is not None:
            c = self.rows["cc_goals"][:self.length]
            self.cc_goals_mean, self.cc_goals_std = np.mean(c, axis=0), np.std(c, axis=0)

    def states_norm(self):
        s = self.rows["states"][:self.length].copy()
        with np.errstate(divide="ignore", invalid="ignore"):
            s[:, 1:] = (s[:, 1:] - self.states_mean[1:]) / self.states_std[1:]
        return s

    def batch(self, idx):
        """(x, y) of __getitem__ over idx, cast to fp32 as the training loop does."""
        idx = np.asarray(idx)
        state = (self.states_norm() if self.norm_input else self.rows["states"])[idx]
        if self.goal_type == "vc":
            goal = self.rows["vc_goals"][idx]
        else:
            goal = self.rows["cc_goals"][idx]
            if self.norm_input:
                with np.errstate(divide="ignore", invalid="ignore"):
                    goal = (goal - self.cc_goals_mean) / self.cc_goals_std
        return np.hstack((state, goal)).astype(np.float32), self.rows["actions"][idx].astype(np.float32)

    def get_database_mean_std(self):
        if not self.norm_input:
            return None
        if self.goal_type == "vc":
            return [self.states_mean, self.states_std, self.vc_goals_mean, self.vc_goals_std]
        return [self.states_mean, self.states_std, self.cc_goals_mean, self.cc_goals_std]

    def save_as_npz(self, filename):
        np.savez(filename, **{f: self.rows[f][:self.length] for f in self.FIELDS})

    def load_from_npz(self, filename):
        data = np.load(filename)
        for f in self.FIELDS:
            if f not in data:
                raise ValueError(f"Missing field '{f}' in NPZ file.")
            self.rows[f] = np.asarray(data[f], dtype=np.float64)
        self.length, self.start = len(self.rows["states"]), 0
        self.calc_input_mean_std()
