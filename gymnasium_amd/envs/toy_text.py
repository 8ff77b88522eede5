"""ToyText tabular environments (FrozenLake, CliffWalking, Taxi) as bit-exact integer kernels.

Every env here is a finite MDP: the reference precomputes ``P[s][a] = [(prob, next_state, reward, terminated), ...]`` in its
constructor and ``step`` is ``i = categorical_sample([t[0] for t in P[s][a]], np_random)`` -- ``np.argmax(np.cumsum(p) >
np_random.random())`` (gymnasium/envs/toy_text/utils.py:4-8) -- followed by a table lookup; ``reset`` draws the start state the
same way from ``initial_state_distrib``.  The classes below rebuild those tables from the env definitions (grid maps, wall
strings, slip rules) and hand them to the engine (``mi_tabular_load``); the kernels then only need the per-env PCG64 stream
and integer lookups, so trajectories equal the reference's bit for bit (tests/golden/toytext_*.npz are generated from it).

  FrozenLakeVectorEnv    gymnasium/envs/toy_text/frozen_lake.py:226-360  (FrozenLake-v1, FrozenLake8x8-v1)
  CliffWalkingVectorEnv  gymnasium/envs/toy_text/cliffwalking.py:103-207 (CliffWalking-v1, CliffWalkingSlippery-v1)
  TaxiVectorEnv          gymnasium/envs/toy_text/taxi.py:163-472         (Taxi-v4 incl. is_rainy -- another table -- and fickle_passenger -- one
                                                                          flag per sub-env, one reset draw, one Generator.choice in the kernel)
  generate_random_map    gymnasium/envs/toy_text/frozen_lake.py:34-83    (FrozenLake's random maps; desc=None, map_name=None)
  BlackjackVectorEnv     gymnasium/envs/toy_text/blackjack.py:56-232     (Blackjack-v1; integer card game, not a table)
"""
from __future__ import annotations

import numpy as np

from ..gym_api import error, logger, spaces
from ..vector.hip_vector_env import HipVectorEnv


class TabularVectorEnv(HipVectorEnv):
    KIND = "tabular"
    INFO_KEYS = ("prob",)
    N_RESET_INFO_KEYS = 1
    RESET_PROB_IS_INT = True  # FrozenLake / CliffWalking reset() returns {"prob": 1} (an int); Taxi returns 1.0
    HOST_INFOS = True         # "prob" dtype quirk, Taxi's action_mask table: assembled on the host (one read-back per step with output="torch")

    def _build(self):
        """Return (P, initial_state_distrib): P[s][a] = list of (prob, next_state, reward, terminated)."""
        raise NotImplementedError

    def _build_tables(self, num_envs: int):
        """Environments whose sub-environments do NOT share one MDP (FrozenLake with a map per sub-environment) return the engine's arrays
        directly: dict(csprob, prob, next_state, reward, terminated, count [num_tables, nS, nA(, K)], isd_csprob [num_tables, nS], env_table [num_envs]);
        None (default): one table for all, from _build()."""
        return None

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, render_mode=None, **kwargs):
        many = self._build_tables(int(num_envs))
        if many is not None:
            self._tab = many
            self.P = self.initial_state_distrib = None  # one per sub-environment: see `desc`
            _, self.nS, self.nA, _ = many["csprob"].shape
            super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, render_mode=render_mode, **kwargs)
            self._engine.load_table(**self._tab)
            return
        P, isd = self._build()
        self.P, self.initial_state_distrib = P, np.asarray(isd, dtype=np.float64)
        self.nS, self.nA = len(P), len(P[0])
        K = max(len(P[s][a]) for s in range(self.nS) for a in range(self.nA))
        self._tab = dict(
            csprob=np.ones((self.nS, self.nA, K), dtype=np.float64), prob=np.zeros((self.nS, self.nA, K), dtype=np.float64),
            next_state=np.zeros((self.nS, self.nA, K), dtype=np.int32), reward=np.zeros((self.nS, self.nA, K), dtype=np.float64),
            terminated=np.zeros((self.nS, self.nA, K), dtype=np.uint8), count=np.zeros((self.nS, self.nA), dtype=np.int32),
            isd_csprob=np.cumsum(self.initial_state_distrib))
        for s in range(self.nS):
            for a in range(self.nA):
                tr = P[s][a]
                n = len(tr)
                self._tab["count"][s, a] = n
                self._tab["csprob"][s, a, :n] = np.cumsum(np.asarray([t[0] for t in tr]))  # exactly what categorical_sample forms
                for k, (p, ns, r, te) in enumerate(tr):
                    self._tab["prob"][s, a, k], self._tab["next_state"][s, a, k] = p, ns
                    self._tab["reward"][s, a, k], self._tab["terminated"][s, a, k] = r, te
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, render_mode=render_mode, **kwargs)
        self._engine.load_table(**self._tab)

    def _single_spaces(self):
        return spaces.Discrete(self.nS), spaces.Discrete(self.nA)

    def _engine_params(self):
        return (float(self.nS), float(self.nA))

    def _parse_reset_options(self, options):
        return None

    def get_state(self):
        """(state[N, state_dim], elapsed_steps, flags); state columns: state index, probability of the last transition (and, for Taxi with
        fickle_passenger, the engine's word holding fickle_step and the generator's buffered 32-bit half)."""
        return super().get_state()

    def _reset_infos(self, mask):
        sel = np.ones(self.num_envs, dtype=np.bool_) if mask is None else mask.view(np.bool_).copy()
        prob = np.where(sel, 1.0, 0.0)  # frozen_lake.py:348 / taxi.py:452: {"prob": 1}
        return {"prob": prob.astype(np.int64) if self.RESET_PROB_IS_INT else prob, "_prob": sel}

    def _build_infos(self):
        # Reference quirk, mirrored: VectorEnv._add_info (vector_env.py:277-338) types a key's array after the FIRST sub-env
        # that supplies it in a step.  When sub-env 0 is in its autoreset step its info is the reset info {"prob": 1} -- an
        # int -- so the whole "prob" array becomes int64 and the other sub-envs' probabilities are truncated (1/3 -> 0).
        # The same happens under SAME_STEP when sub-env 0 finishes: its top-level entry is then the reset info (sync_vector_env.py:319).
        pending0 = bool(self._was_done[0])
        infos = super()._build_infos()
        mode = self.autoreset_mode.value
        env0_resetting = pending0 if mode == "NextStep" else (bool(infos["_final_info"][0]) if (mode == "SameStep" and "_final_info" in infos) else False)
        if self.RESET_PROB_IS_INT and env0_resetting:
            infos["prob"] = infos["prob"].astype(np.int64)
        return infos


LEFT, DOWN, RIGHT, UP = 0, 1, 2, 3  # frozen_lake.py:15-18
FROZEN_LAKE_MAPS = {  # frozen_lake.py:20-32
    "4x4": ["SFFF", "FHFH", "FFFH", "HFFG"],
    "8x8": ["SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF", "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"],
}


def _board(desc) -> list:
    """A board as a list of row strings, whatever the caller passed (list of str, bytes, or a NumPy 'c' / str array like FrozenLakeEnv.desc)."""
    rows = []
    for r in desc:
        if isinstance(r, (bytes, np.bytes_)):
            rows.append(r.decode())
        elif isinstance(r, (str, np.str_)):
            rows.append(str(r))
        else:
            rows.append("".join(x.decode() if isinstance(x, (bytes, np.bytes_)) else str(x) for x in r))
    return rows


def _is_tile(x) -> bool:
    return isinstance(x, (str, bytes, np.str_, np.bytes_)) and len(x) == 1


def _is_list_of_boards(desc) -> bool:
    """One board is 2-D in tiles: a sequence of rows, each a string or a sequence of single characters (FrozenLakeEnv.desc is an (nrow, ncol) array of
    bytes).  A list of boards has one more level: its elements are themselves boards.  Decided by dimensionality, not by how the rows are spelled:
    `[env.desc for env in envs]`, lists of lists of characters and lists of lists of row strings are all lists of boards."""
    if isinstance(desc, np.ndarray):
        return desc.ndim == 3 or (desc.ndim == 2 and desc.size > 0 and not _is_tile(desc.flat[0]))  # (a 2-D array of ROW STRINGS: one board per row of the array)
    if len(desc) == 0:
        return False
    first = desc[0]
    if isinstance(first, (str, bytes, np.str_, np.bytes_)):
        return False  # rows spelled as strings: one board
    if isinstance(first, np.ndarray):
        return first.ndim == 2 or (first.ndim == 1 and first.size > 0 and not _is_tile(first.flat[0]))
    if len(first) == 0:
        return False
    return not _is_tile(first[0])  # a row of single tiles: one board; anything longer / nested: a board of its own


def _check_board(rows):
    """Rows of equal length made of the four tile letters (frozen_lake.py:98-106): anything else would silently build a different MDP."""
    if not rows or any(len(r) != len(rows[0]) for r in rows) or len(rows[0]) == 0:
        raise ValueError(f"a FrozenLake board needs rows of equal, non-zero length, got {rows!r}")
    bad = sorted({c for r in rows for c in r} - set("SFHG"))
    if bad:
        raise ValueError(f"a FrozenLake board is made of the tiles S, F, H, G; found {bad!r} in {rows!r}")
    if not any("S" in r for r in rows):
        raise ValueError(f"a FrozenLake board needs a start tile S: {rows!r}")


def _goal_reachable(board, size: int) -> bool:
    """frozen_lake.py:34-53 is_valid: is "G" reachable from (0, 0) over non-hole tiles (4-neighbourhood)?  (Any graph search gives the same answer.)"""
    seen, stack = {(0, 0)}, [(0, 0)]
    while stack:
        r, c = stack.pop()
        for r2, c2 in ((r + 1, c), (r, c + 1), (r - 1, c), (r, c - 1)):
            if 0 <= r2 < size and 0 <= c2 < size and (r2, c2) not in seen:
                if board[r2][c2] == "G":
                    return True
                if board[r2][c2] != "H":
                    seen.add((r2, c2)), stack.append((r2, c2))
    return False


def generate_random_map(size: int = 8, p: float = 0.8, seed: int | None = None) -> list[str]:
    """gymnasium.envs.toy_text.frozen_lake.generate_random_map (frozen_lake.py:56-83): boards of i.i.d. frozen ("F", probability p) / hole tiles from
    `seeding.np_random(seed)`, start top-left, goal bottom-right, re-drawn until the goal is reachable.  Same generator calls as the reference
    (Generator.choice with p: one `random((size, size))` block per attempt), so the same seed gives the same map (tests/golden/frozenlake_random_maps.npz)."""
    from ..gym_api import seeding

    rng, _ = seeding.np_random(seed)
    while True:
        p = min(1, p)
        board = rng.choice(["F", "H"], (size, size), p=[p, 1 - p])
        board[0][0], board[-1][-1] = "S", "G"
        if _goal_reachable(board, size):
            return ["".join(row) for row in board]


class FrozenLakeVectorEnv(TabularVectorEnv):
    DEFAULT_MAX_EPISODE_STEPS = 100

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, render_mode=None, desc=None, map_name: str = "4x4",
                 is_slippery: bool = True, success_rate: float = 1.0 / 3.0, reward_schedule=(1, 0, 0), **kwargs):
        # frozen_lake.py:241-242: `desc = generate_random_map()` -- unseeded (OS entropy).  The reference's SyncVectorEnv constructs num_envs scalar
        # envs, i.e. draws one map PER SUB-ENVIRONMENT: so does this class (one transition table per sub-environment, mi_tabular_table.env_table).
        # `desc` may also be a list of num_envs boards -- one map per sub-environment, reproducibly: [generate_random_map(8, 0.8, seed + i) for i in ...].
        self.descs = None
        if desc is None and map_name is None:
            if int(num_envs) > 4096:
                logger.warn(f"FrozenLake with map_name=None draws one random map per sub-environment like SyncVectorEnv's scalar envs do: {num_envs} maps are "
                            "generated in a Python loop and every distinct board gets its own transition table on the device (8x8: ~22 KB each, no LDS "
                            "staging in rollouts); pass `desc=` (one board, or a list of a few boards repeated) to avoid that cost")
            self.descs = [generate_random_map() for _ in range(int(num_envs))]
        elif desc is not None and _is_list_of_boards(desc):
            self.descs = [_board(b) for b in desc]
            if len(self.descs) != int(num_envs):
                raise ValueError(f"a list of boards must have one board per sub-environment: got {len(self.descs)} for num_envs={num_envs}")
            if len({(len(b), len(b[0])) for b in self.descs}) != 1:
                raise ValueError("the boards of one vector environment must have the same shape (one observation space)")
        for b in (self.descs if self.descs is not None else ([_board(desc)] if desc is not None else [])):
            _check_board(b)
        self.desc = self.descs[0] if self.descs is not None else _board(desc if desc is not None else FROZEN_LAKE_MAPS[map_name])
        self.is_slippery, self.success_rate, self.reward_schedule = bool(is_slippery), success_rate, tuple(reward_schedule)
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, render_mode=render_mode, **kwargs)

    def _build_tables(self, num_envs):
        """One transition table per DISTINCT board, built for all of them at once (the same rules as _build(), as array arithmetic over the boards)."""
        if self.descs is None:
            return None
        letters = np.array([[list(row) for row in b] for b in self.descs])  # [N, nrow, ncol] of 1-character strings
        boards, env_table = np.unique(letters.reshape(len(self.descs), -1), axis=0, return_inverse=True)
        env_table = env_table.reshape(-1)
        M, (nrow, ncol) = len(boards), letters.shape[1:]
        if M == 1:  # all sub-environments share the board after all
            self.descs = None
            return None
        boards = boards.reshape(M, nrow, ncol)
        nS = nrow * ncol
        row, col = np.divmod(np.arange(nS), ncol)
        r2 = np.stack([row, np.minimum(row + 1, nrow - 1), row, np.maximum(row - 1, 0)], 1)  # [nS, 4 directions]: LEFT, DOWN, RIGHT, UP
        c2 = np.stack([np.maximum(col - 1, 0), col, np.minimum(col + 1, ncol - 1), col], 1)
        landed = boards[:, r2, c2]  # [M, nS, 4] letter of the tile a move in direction b ends on
        sched = np.asarray(self.reward_schedule, dtype=np.float64)
        rew_b = np.where(landed == "G", sched[0], np.where(landed == "H", sched[1], sched[2]))
        term_b = (landed == "G") | (landed == "H")
        next_b = np.broadcast_to(r2 * ncol + c2, (M, nS, 4))
        here_terminal = np.isin(boards.reshape(M, nS), ("G", "H"))  # [M, nS]
        K = 3 if self.is_slippery else 1
        fail_rate = (1.0 - self.success_rate) / 2.0
        a = np.arange(4)
        b_of = np.stack([(a - 1) % 4, a, (a + 1) % 4], 1) if self.is_slippery else a[:, None]  # [4 actions, K] direction of outcome k
        p_of = np.where(b_of == a[:, None], self.success_rate, fail_rate) if self.is_slippery else np.ones((4, 1))
        prob = np.broadcast_to(p_of, (M, nS, 4, K)).copy()
        nxt, rew, term = next_b[:, :, b_of].copy(), rew_b[:, :, b_of].copy(), term_b[:, :, b_of].copy()
        count = np.full((M, nS, 4), K, dtype=np.int32)
        # a sub-environment standing on G / H has one outcome: (1.0, s, 0, True)
        t = here_terminal
        prob[t], nxt[t], rew[t], term[t] = 0.0, 0, 0.0, False
        prob[t, :, 0], term[t, :, 0], count[t] = 1.0, True, 1
        nxt[t, :, 0] = np.broadcast_to(np.arange(nS), (M, nS))[t][:, None]
        csprob = np.ones((M, nS, 4, K))
        for k in range(K):  # np.cumsum of the valid outcomes (categorical_sample, utils.py:4-8); entries beyond `count` stay 1.0
            csprob[..., k] = np.where(k < count, np.cumsum(prob, axis=-1)[..., k], 1.0)
        isd = (boards.reshape(M, nS) == "S").astype(np.float64)
        isd /= isd.sum(axis=1, keepdims=True)
        return dict(csprob=csprob, prob=prob, next_state=nxt.astype(np.int32), reward=rew, terminated=term.astype(np.uint8), count=count,
                    isd_csprob=np.cumsum(isd, axis=1), env_table=env_table.astype(np.int32))

    def _build(self):
        desc, nrow, ncol = self.desc, len(self.desc), len(self.desc[0])
        fail_rate = (1.0 - self.success_rate) / 2.0
        isd = np.array([[c == "S" for c in row] for row in desc]).astype("float64").ravel()
        isd /= isd.sum()

        def move(row, col, a):
            if a == LEFT:
                col = max(col - 1, 0)
            elif a == DOWN:
                row = min(row + 1, nrow - 1)
            elif a == RIGHT:
                col = min(col + 1, ncol - 1)
            else:
                row = max(row - 1, 0)
            return row, col

        def outcome(row, col, a):
            r2, c2 = move(row, col, a)
            letter = desc[r2][c2]
            reward = self.reward_schedule["GHF".index(letter if letter in "GHF" else "F")]
            return r2 * ncol + c2, reward, letter in "GH"

        P = {}
        for row in range(nrow):
            for col in range(ncol):
                s = row * ncol + col
                P[s] = {}
                for a in range(4):
                    if desc[row][col] in "GH":
                        P[s][a] = [(1.0, s, 0, True)]
                    elif self.is_slippery:
                        P[s][a] = [((self.success_rate if b == a else fail_rate), *outcome(row, col, b)) for b in ((a - 1) % 4, a, (a + 1) % 4)]
                    else:
                        P[s][a] = [(1.0, *outcome(row, col, a))]
        return P, isd


class CliffWalkingVectorEnv(TabularVectorEnv):
    DEFAULT_MAX_EPISODE_STEPS = None

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, render_mode=None, is_slippery: bool = False, **kwargs):
        self.is_slippery = bool(is_slippery)
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, render_mode=render_mode, **kwargs)

    def _build(self):
        nrow, ncol = 4, 12
        start = 3 * ncol
        deltas_of = {0: (-1, 0), 1: (0, 1), 2: (1, 0), 3: (0, -1)}  # UP, RIGHT, DOWN, LEFT (cliffwalking.py:13-18)
        P = {}
        for s in range(nrow * ncol):
            row, col = divmod(s, ncol)
            P[s] = {}
            for a in range(4):
                acts = [(a - 1) % 4, a, (a + 1) % 4] if self.is_slippery else [a]
                out = []
                for b in acts:
                    r2 = max(min(row + deltas_of[b][0], nrow - 1), 0)
                    c2 = max(min(col + deltas_of[b][1], ncol - 1), 0)
                    if r2 == 3 and 1 <= c2 <= ncol - 2:  # the cliff
                        out.append((1 / len(acts), start, -100, False))
                    else:
                        out.append((1 / len(acts), r2 * ncol + c2, -1, (r2, c2) == (nrow - 1, ncol - 1)))
                P[s][a] = out
        isd = np.zeros(nrow * ncol)
        isd[start] = 1.0
        return P, isd


TAXI_MAP = ["+---------+", "|R: | : :G|", "| : | : : |", "| : : : : |", "| | : | : |", "|Y| : |B: |", "+---------+"]  # taxi.py:15-23
TAXI_LOCS = [(0, 0), (0, 4), (4, 0), (4, 3)]


class TaxiVectorEnv(TabularVectorEnv):
    DEFAULT_MAX_EPISODE_STEPS = 200
    RESET_PROB_IS_INT = False

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, render_mode=None, is_rainy: bool = False,
                 fickle_passenger: bool = False, rainy_probability: float = 0.8, fickle_probability: float = 0.3, **kwargs):
        self.is_rainy, self.fickle_passenger = bool(is_rainy), bool(fickle_passenger)
        self.rainy_probability, self.fickle_probability = rainy_probability, fickle_probability
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, render_mode=render_mode, **kwargs)
        self._action_mask = np.stack([self.action_mask(s) for s in range(self.nS)])

    def _engine_params(self):
        # params[2] != 0: the fickle-passenger rule of taxi.py:436-451 / :462-464 runs in the kernel (tab_fickle, engine.hip), params[3] = its probability
        return (float(self.nS), float(self.nA), 1.0 if self.fickle_passenger else 0.0, float(self.fickle_probability))

    @staticmethod
    def encode(row, col, pass_loc, dest):
        return ((row * 5 + col) * 5 + pass_loc) * 4 + dest

    @staticmethod
    def decode(i):
        dest, i = i % 4, i // 4
        pass_loc, i = i % 5, i // 5
        col, row = i % 5, i // 5
        return row, col, pass_loc, dest

    def action_mask(self, state):  # taxi.py:397-417
        mask = np.zeros(6, dtype=np.int8)
        row, col, pass_loc, dest = self.decode(state)
        mask[0] = row < 4
        mask[1] = row > 0
        mask[2] = col < 4 and TAXI_MAP[row + 1][2 * col + 2] == ":"
        mask[3] = col > 0 and TAXI_MAP[row + 1][2 * col] == ":"
        mask[4] = pass_loc < 4 and (row, col) == TAXI_LOCS[pass_loc]
        mask[5] = pass_loc == 4 and (row, col) in TAXI_LOCS
        return mask

    @staticmethod
    def _pickup_dropoff(a, row, col, pass_idx, dest):
        """taxi.py:173-199 _pickup / _dropoff: (new passenger index, reward, terminated) of action 4 / 5."""
        p2, reward, term = pass_idx, -1, False
        if a == 4:
            if pass_idx < 4 and (row, col) == TAXI_LOCS[pass_idx]:
                p2 = 4
            else:
                reward = -10
        elif (row, col) == TAXI_LOCS[dest] and pass_idx == 4:
            p2, term, reward = dest, True, 20
        elif (row, col) in TAXI_LOCS and pass_idx == 4:
            p2 = TAXI_LOCS.index((row, col))
        else:
            reward = -10
        return p2, reward, term

    @staticmethod
    def _can_move(a, row, col):
        """Is the primary move of action a (0 south, 1 north, 2 east, 3 west) possible from (row, col)?  (taxi.py:271-276)"""
        return ((a == 0 and row < 4) or (a == 1 and row > 0) or (a == 2 and TAXI_MAP[1 + row][2 * col + 2] == ":") or (a == 3 and TAXI_MAP[1 + row][2 * col] == ":"))

    @staticmethod
    def _lateral(row, col, dr, dc):
        """taxi.py:230-245 _calc_new_position: where a sideways drift ends (an interior wall or the boundary leaves the taxi in place)."""
        r2, c2 = max(0, min(row + dr, 4)), max(0, min(col + dc, 4))
        if dc == 1 and TAXI_MAP[1 + r2][2 * c2] != ":":
            return row, col
        if dc == -1 and TAXI_MAP[1 + r2][2 * c2 + 2] != ":":
            return row, col
        return r2, c2

    def _build(self):
        # (forward, left, right) of each heading (taxi.py:262-267)
        moves = {0: ((1, 0), (0, 1), (0, -1)), 1: ((-1, 0), (0, -1), (0, 1)), 2: ((0, 1), (-1, 0), (1, 0)), 3: ((0, -1), (1, 0), (-1, 0))}
        lateral_probability = (1.0 - self.rainy_probability) / 2.0
        P, isd = {}, np.zeros(500)
        for row in range(5):
            for col in range(5):
                for pass_idx in range(5):
                    for dest in range(4):
                        s = self.encode(row, col, pass_idx, dest)
                        if pass_idx < 4 and pass_idx != dest:
                            isd[s] += 1
                        P[s] = {}
                        for a in range(6):
                            if a >= 4:
                                p2, reward, term = self._pickup_dropoff(a, row, col, pass_idx, dest)
                                P[s][a] = [(1.0, self.encode(row, col, p2, dest), reward, term)]
                            elif not self.is_rainy:  # taxi.py:201-228: a blocked move leaves the taxi where it is
                                r2, c2 = row, col
                                if self._can_move(a, row, col):
                                    r2, c2 = max(0, min(row + moves[a][0][0], 4)), max(0, min(col + moves[a][0][1], 4))
                                P[s][a] = [(1.0, self.encode(r2, c2, pass_idx, dest), -1, False)]
                            else:  # taxi.py:247-308: intended move with rainy_probability, a drift to either side with the rest; a blocked move drifts nowhere
                                fwd = left = right = (row, col)
                                if self._can_move(a, row, col):
                                    fwd = (max(0, min(row + moves[a][0][0], 4)), max(0, min(col + moves[a][0][1], 4)))
                                    left, right = self._lateral(row, col, *moves[a][1]), self._lateral(row, col, *moves[a][2])
                                P[s][a] = [(self.rainy_probability, self.encode(*fwd, pass_idx, dest), -1, False),
                                           (lateral_probability, self.encode(*left, pass_idx, dest), -1, False),
                                           (lateral_probability, self.encode(*right, pass_idx, dest), -1, False)]
        isd /= isd.sum()
        return P, isd

    def _with_action_mask(self, infos):
        obs = self._obs.cpu().numpy() if self.output == "torch" else self._obs
        infos["action_mask"] = self._action_mask[np.asarray(obs).reshape(-1)]
        infos["_action_mask"] = np.ones(self.num_envs, dtype=np.bool_)
        return infos

    def _build_infos(self):
        infos = self._with_action_mask(TabularVectorEnv._build_infos(self))
        if "final_info" in infos:  # SAME_STEP: the finishing step's info carries the mask of the final state (taxi.py:470)
            dones = infos["_final_info"]
            final = self._host(self._final).reshape(-1)
            infos["final_info"]["action_mask"] = np.where(dones[:, None], self._action_mask[np.where(dones, final, 0)], 0).astype(np.int8)
            infos["final_info"]["_action_mask"] = dones.copy()
        return infos

    def _reset_infos(self, mask):
        return self._with_action_mask(super()._reset_infos(mask))


# id -> (creator, max_episode_steps, reward_threshold, kwargs): gymnasium/envs/__init__.py:139-171
class BlackjackVectorEnv(HipVectorEnv):
    """Blackjack-v1 (gymnasium/envs/toy_text/blackjack.py:56-232): not a table lookup -- cards are drawn with
    ``np_random.choice(deck)`` (Lemire-bounded 32-bit halves of the PCG64 stream), the dealer plays out on `stick` -- but every
    quantity is an integer, so the kernel is bit-exact.  Observations are the reference's Tuple(Discrete(32), Discrete(11),
    Discrete(2)) batched by SyncVectorEnv: a tuple of three int64 arrays (player sum, dealer's showing card, usable ace)."""

    KIND = "blackjack"
    HOST_INFOS = True  # SAME_STEP final_obs is an object array of tuples

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, render_mode=None, natural: bool = False, sab: bool = False, **kwargs):
        self.natural, self.sab = bool(natural), bool(sab)
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, render_mode=render_mode, **kwargs)

    def _single_spaces(self):
        return spaces.Tuple((spaces.Discrete(32), spaces.Discrete(11), spaces.Discrete(2))), spaces.Discrete(2)

    def _engine_params(self):
        return (float(self.natural), float(self.sab))

    def _parse_reset_options(self, options):
        return None

    def _tuple(self, obs):  # (N, 3) -> the batched Tuple observation (vector/utils/space_utils.py concatenate for Tuple spaces)
        return (obs[:, 0], obs[:, 1], obs[:, 2])

    def reset(self, *, seed=None, options=None):
        obs, info = super().reset(seed=seed, options=options)
        return self._tuple(obs), info

    def step(self, actions):
        obs, r, te, tr, info = super().step(actions)
        if "final_obs" in info:  # a 1-D object array of tuples, also when every sub-env finished (np.array would build (N, 3))
            out = np.empty(self.num_envs, dtype=object)
            for i, f in enumerate(info["final_obs"]):
                out[i] = None if f is None else tuple(int(x) for x in f)
            info["final_obs"] = out
        return self._tuple(obs), r, te, tr, info


ENV_TABLE = {
    "Blackjack-v1": (BlackjackVectorEnv, None, None, {"sab": True, "natural": False}),   # envs/__init__.py:134-138
    "FrozenLake-v1": (FrozenLakeVectorEnv, 100, 0.70, {"map_name": "4x4"}),
    "FrozenLake8x8-v1": (FrozenLakeVectorEnv, 200, 0.85, {"map_name": "8x8"}),
    "CliffWalking-v1": (CliffWalkingVectorEnv, None, None, {}),
    "CliffWalkingSlippery-v1": (CliffWalkingVectorEnv, None, None, {"is_slippery": True}),
    "Taxi-v4": (TaxiVectorEnv, 200, 8, {}),
}
