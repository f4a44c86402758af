"""CPU: the oracle's MT19937 / randint / sample_by_key_ids restatement (oracle/sampler.py) against numpy's own legacy
generator (the third-party arithmetic the reference calls, sampler.py:240-241) and against golden vectors produced
by running the reference's Sampler (tests/golden/gen_sampler_golden.py).  Bit-exact: integer work."""
import glob
import os

import numpy as np
import pytest

from oracle import sampler as OS

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "sampler_*.npz"))
               if not p.endswith("sampler_popularity.npz"))   # the oracle restates the uniform sampler; popularity IS numpy (GPU test)


@pytest.mark.parametrize("seed", [0, 1, 5, 2020, 2 ** 32 - 1])
def test_mt19937_state_and_stream_match_numpy(seed):
    rs = OS.MT19937(seed)
    np.random.seed(seed)
    st = np.random.get_state()
    np.testing.assert_array_equal(rs.key, st[1])
    assert rs.pos == st[2] == 624
    raw = np.concatenate([rs.raw_block(), (rs.advance(624), rs.raw_block())[1]])     # two twists
    ref = np.random.randint(0, 2 ** 32, len(raw), dtype=np.uint32)
    np.testing.assert_array_equal(raw, ref)


@pytest.mark.parametrize("high,n", [(3, 10), (5, 1000), (1026, 5000), (1683, 2048), (100001, 8192), (2 ** 31, 700),
                                    (2 ** 32 - 5, 1300), (2, 7)])
def test_randint_matches_numpy_across_calls(high, n):
    rs = OS.MT19937(2020)
    np.random.seed(2020)
    for k in range(4):                         # one continuing stream, as in training
        a = rs.randint(1, high, n + k)
        b = np.random.randint(1, high, n + k)
        np.testing.assert_array_equal(a, b)
    st = np.random.get_state()
    # numpy twists lazily as well: after the same draws both sit at the same position of the same block
    np.testing.assert_array_equal(rs.key, st[1])
    assert rs.pos == st[2]


def test_state_hand_over_to_numpy_mid_stream():
    rs = OS.MT19937(7)
    rs.randint(1, 1000, 333)
    np.random.set_state(rs.get_state())
    a = np.random.randint(1, 50, 100)
    np.random.seed(7)
    np.random.randint(1, 1000, 333)
    np.testing.assert_array_equal(a, np.random.randint(1, 50, 100))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_sample_by_key_ids_matches_reference_golden(path):
    z = np.load(path)
    item_num, user_num = int(z["item_num"]), int(z["user_num"])
    used = [set() for _ in range(user_num)]
    for u, i in zip(z["train_user"], z["train_item"]):
        used[u].add(int(i))
    rs = OS.MT19937(int(z["seed"]))
    for c in range(int(z["n_calls"])):
        neg = OS.sample_by_key_ids(rs, z[f"users{c}"], int(z[f"num{c}"]), used, item_num)
        np.testing.assert_array_equal(neg, z[f"neg{c}"])
        users = np.tile(z[f"users{c}"], int(z[f"num{c}"]))
        assert all(int(v) not in used[u] for u, v in zip(users, neg))
    np.testing.assert_array_equal(rs.key, z["final_key"])
    assert rs.pos == int(z["final_pos"])
    np.testing.assert_array_equal(OS.MT19937(int(z["seed"])).randint(1, item_num, 3000), z["randint_stream"])
