"""GPU: the device negative sampler (fr_sample_negatives through fairrec.sampler) is BIT-EXACT with the reference's
numpy-based sampler: against numpy's own generator (available on the GPU box: it is the third-party arithmetic the
reference calls), against the oracle, and against golden vectors produced by running the reference's Sampler."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "sampler_*.npz")))


def _rs(seed=None):
    from fairrec.sampler import DeviceRandomState
    return DeviceRandomState("cuda", seed)


@pytest.mark.parametrize("seed", [0, 1, 2020, 2 ** 32 - 1])
def test_seed_matches_numpy_state(seed):
    rs = _rs(seed)
    np.random.seed(seed)
    st, ref = rs.get_state(), np.random.get_state()
    np.testing.assert_array_equal(st[1], ref[1])
    assert st[2] == ref[2] == 624


@pytest.mark.parametrize("high,n", [(3, 10), (5, 1000), (1026, 5000), (1683, 2048), (100001, 8192), (1000001, 8192),
                                    (2 ** 31, 700), (2 ** 32 - 5, 1300), (2, 7), (1683, 1), (1683, 623), (1683, 625)])
def test_randint_stream_is_numpy_bit_for_bit(high, n):
    rs = _rs(2020)
    np.random.seed(2020)
    for k in range(4):                                   # one continuing stream across calls
        a = rs.randint(1, high, n + 3 * k).cpu().numpy()
        b = np.random.randint(1, high, n + 3 * k)
        np.testing.assert_array_equal(a, b)
    st, ref = rs.get_state(), np.random.get_state()
    np.testing.assert_array_equal(st[1], ref[1])
    assert st[2] == ref[2]


def test_stream_hand_over_numpy_device_numpy():
    rs = _rs(7)
    np.random.seed(7)
    np.testing.assert_array_equal(rs.randint(1, 1000, 333).cpu().numpy(), np.random.randint(1, 1000, 333))
    np.random.set_state(rs.get_state())                  # host takes the stream over (e.g. the trainer's mask draws)
    host = np.random.choice([0, 1], 5)
    rs.set_state(np.random.get_state())                  # and hands it back
    a = rs.randint(1, 77, 1000).cpu().numpy()
    np.random.seed(7)
    np.random.randint(1, 1000, 333)
    np.testing.assert_array_equal(host, np.random.choice([0, 1], 5))
    np.testing.assert_array_equal(a, np.random.randint(1, 77, 1000))


class _DS:
    uid_field, iid_field = "user_id", "item_id"

    def __init__(self, user_num, item_num, u, i):
        self.user_num, self.item_num = user_num, item_num
        self.inter_feat = {"user_id": torch.from_numpy(u), "item_id": torch.from_numpy(i)}


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_sampler_matches_reference_golden(path):
    from fairrec.sampler import Sampler
    z = np.load(path)
    item_num, user_num = int(z["item_num"]), int(z["user_num"])
    rs = _rs(int(z["seed"]))
    dist = str(z["distribution"]) if "distribution" in z.files else "uniform"      # "popularity": numpy draws on the shared stream
    sampler = Sampler("train", _DS(user_num, item_num, z["train_user"], z["train_item"]), dist, device="cuda",
                      random_state=rs).set_phase("train")
    for c in range(int(z["n_calls"])):
        users = z[f"users{c}"]
        neg = sampler.sample_by_user_ids(torch.from_numpy(users).cuda(), None, int(z[f"num{c}"]))
        np.testing.assert_array_equal(neg.cpu().numpy(), z[f"neg{c}"])
    st = rs.get_state()
    np.testing.assert_array_equal(st[1], z["final_key"])
    assert st[2] == int(z["final_pos"])
    assert int(rs.err_flag.item()) == 0


def test_sampler_matches_oracle_on_adversarial_used_sets():
    """Users whose used-set is everything but one or two items (long rejection chains), interleaved with empty users,
    B not a multiple of the wave or block size, num > 1."""
    from fairrec.sampler import Sampler
    from oracle import sampler as OS
    rng = np.random.default_rng(3)
    user_num, item_num = 40, 67
    u, i = [], []
    for usr in range(1, 12):                              # users 1..11: all items but (usr % 3 + 1) of them
        keep = rng.choice(np.arange(1, item_num), size=usr % 3 + 1, replace=False)
        its = np.setdiff1d(np.arange(1, item_num), keep)
        u += [usr] * len(its)
        i += list(its)
    u, i = np.array(u, dtype=np.int64), np.array(i, dtype=np.int64)
    used = [set() for _ in range(user_num)]
    for a, b in zip(u, i):
        used[a].add(int(b))
    rs = _rs(99)
    ors = OS.MT19937(99)
    sampler = Sampler("train", _DS(user_num, item_num, u, i), device="cuda", random_state=rs).set_phase("train")
    rounds = torch.zeros(1, dtype=torch.int32, device="cuda")
    for n, num in ((1, 1), (63, 1), (65, 2), (1025, 1), (3000, 3)):
        users = rng.integers(1, 25, size=n).astype(np.int64)
        indptr, items, _ = sampler.used_ids
        neg = rs.sample_excluding(1, item_num, torch.from_numpy(users).cuda(), num, indptr, items, rounds)
        ref = OS.sample_by_key_ids(ors, users, num, used, item_num)
        np.testing.assert_array_equal(neg.cpu().numpy(), ref)
        assert int(rounds.item()) >= 1
    st = rs.get_state()
    np.testing.assert_array_equal(st[1], ors.key)
    assert st[2] == ors.pos


def test_full_size_batch_properties():
    """BASELINE sizes (B = 8192, 1,000,001 items, 1,000,001 users x 20 interactions): every negative is in range and
    outside its user's used-set; the stream position matches numpy when no rejection by used-set occurs."""
    from fairrec.sampler import Sampler
    g = torch.Generator().manual_seed(0)
    user_num = item_num = 1_000_001
    u = torch.arange(1, user_num).repeat_interleave(20)
    i = torch.randint(1, item_num, (u.numel(),), generator=g)
    rs = _rs(2020)
    sampler = Sampler("train", _DS(user_num, item_num, u.numpy(), i.numpy()), device="cuda", random_state=rs).set_phase("train")
    users = torch.randint(1, user_num, (8192,), generator=g)
    neg = sampler.sample_by_user_ids(users.cuda(), None, 1).cpu()
    assert int(neg.min()) >= 1 and int(neg.max()) < item_num
    key = u * item_num + i
    assert not torch.isin(users * item_num + neg, key).any()
    np.random.seed(2020)
    ref = np.random.randint(1, item_num, 8192)
    same = neg.numpy() == ref
    assert same.mean() > 0.99          # the rare used-set hits are re-drawn from later in the stream


@pytest.mark.parametrize("model,pairwise", [("PFCN_PMF", True), ("NFCF", False)])
def test_train_dataloader_negative_sampling_on_device(model, pairwise):
    """TrainDataLoader with `neg_sampling: {uniform: k}`: epoch shuffle by torch.randperm (interaction.py:293-297), one
    continuing numpy-compatible stream across batches and epochs, pair-wise / point-wise assembly
    (abstract_dataloader.py:182-198); everything stays on the GPU."""
    from fairrec.config import Config
    from fairrec.data.dataloader import TrainDataLoader
    from fairrec.data.dataset import synthetic_dataset
    from fairrec.sampler import DeviceRandomState, Sampler
    from oracle import sampler as OS
    k = 1 if pairwise else 2
    cfg = Config(model=model, config_dict={"train_batch_size": 96, "neg_sampling": {"uniform": k}, "device": "cuda",
                                           "LABEL_FIELD": "label"})
    ds = synthetic_dataset(cfg, n_users=60, n_items=50, n_inter=500, seed=3)
    u_all, i_all = ds.inter_feat["user_id"].numpy().copy(), ds.inter_feat["item_id"].numpy().copy()
    used = [set() for _ in range(60)]
    for a, b in zip(u_all, i_all):
        used[a].add(int(b))
    rs = DeviceRandomState("cuda", 2020)
    sampler = Sampler("train", ds, device="cuda", random_state=rs).set_phase("train")
    ds.to("cuda")
    dl = TrainDataLoader(cfg, ds, sampler=sampler, shuffle=True)
    step = 96 // (k if pairwise else 1 + k)
    assert dl.step == step
    ors = OS.MT19937(2020)
    torch.manual_seed(5)
    order = np.arange(500)
    for epoch in range(2):
        batches = list(dl)
        gen = torch.Generator().manual_seed(5)
        for _ in range(epoch + 1):
            perm = torch.randperm(500, generator=gen).numpy()
        order = order[perm] if epoch else perm          # the dataset is re-shuffled in place every epoch
        u_ep, i_ep = u_all[order], i_all[order]
        assert len(batches) == -(-500 // step)
        for b, inter in enumerate(batches):
            us, its = u_ep[b * step:(b + 1) * step], i_ep[b * step:(b + 1) * step]
            neg = OS.sample_by_key_ids(ors, us, k, used, 50)
            assert inter["user_id"].is_cuda
            if pairwise:
                np.testing.assert_array_equal(inter["user_id"].cpu().numpy(), np.tile(us, k))
                np.testing.assert_array_equal(inter["item_id"].cpu().numpy(), np.tile(its, k))
                np.testing.assert_array_equal(inter["neg_item_id"].cpu().numpy(), neg)
            else:
                np.testing.assert_array_equal(inter["user_id"].cpu().numpy(), np.tile(us, 1 + k))
                np.testing.assert_array_equal(inter["item_id"].cpu().numpy(), np.concatenate([its, neg]))
                lab = np.zeros(len(us) * (1 + k), dtype=np.float32)
                lab[:len(us)] = 1.0
                np.testing.assert_array_equal(inter["label"].cpu().numpy(), lab)
            assert "gender" in inter            # user features joined on the device


def test_host_draws_interleave_with_device_draws_like_one_numpy_stream():
    """The trainers' per-epoch attribute-mask draw (trainer.py:879-882) and the batches' negative draws consume ONE
    numpy stream in the reference.  With the device feed the stream is handed to numpy for the host draw and back."""
    from fairrec.sampler import global_random_state, host_numpy_stream
    from fairrec.utils import init_seed
    rs = global_random_state("cuda")
    init_seed(123)
    a1 = rs.randint(1, 500, 700).cpu().numpy()
    with host_numpy_stream():
        m = np.random.choice([0, 1], 3)
    a2 = rs.randint(1, 90, 1300).cpu().numpy()
    np.random.seed(123)
    np.testing.assert_array_equal(a1, np.random.randint(1, 500, 700))
    np.testing.assert_array_equal(m, np.random.choice([0, 1], 3))
    np.testing.assert_array_equal(a2, np.random.randint(1, 90, 1300))


def test_quick_start_pfcn_with_device_negative_sampling(tmp_path):
    """run_recbole for a pairwise model with the reference's default `neg_sampling: {uniform: 1}`: dataset resident on
    the GPU, negatives from the device sampler (never a positive of train / valid / test), training end to end."""
    from fairrec.quick_start import run_recbole
    out = run_recbole(model="PFCN_PMF", config_dict={
        "embedding_size": 16, "sst_attr_list": ["gender"], "filter_mode": "none", "epochs": 2, "train_batch_size": 256,
        "synthetic_users": 200, "synthetic_items": 150, "synthetic_interactions": 4000, "device": "cuda",
        "checkpoint_dir": str(tmp_path), "eval_step": 0}, saved=False)
    assert out["test_result"] is None or isinstance(out["test_result"], dict)


def test_multi_call_sampling_equals_consecutive_single_key_calls():
    """fr_sample_negatives_calls = the evaluation loader's user-by-user draws (general_dataloader.py:141-146: one
    sample_by_user_ids call per user with all keys equal, 100 negatives per positive) in ONE launch, bit-exact with the
    consecutive calls on one numpy stream (oracle) and with the single-call kernel."""
    from fairrec.sampler import Sampler
    from oracle import sampler as OS
    rng = np.random.default_rng(5)
    user_num, item_num = 60, 300
    u = rng.integers(1, user_num, 2500).astype(np.int64)
    i = rng.integers(1, item_num, 2500).astype(np.int64)
    heavy = np.setdiff1d(np.arange(1, item_num), rng.choice(np.arange(1, item_num), 4, replace=False))
    u, i = np.concatenate([u, np.full(len(heavy), 7)]), np.concatenate([i, heavy])        # user 7: 4 items left
    used = [set() for _ in range(user_num)]
    for a, b in zip(u, i):
        used[a].add(int(b))
    rs, rs1, ors = _rs(31), _rs(31), OS.MT19937(31)
    smp = Sampler("test", _DS(user_num, item_num, u, i), device="cuda", random_state=rs).set_phase("test")
    indptr, items, _ = smp.used_ids
    keys = np.array([3, 7, 12, 7, 40, 41, 59], dtype=np.int64)
    counts = np.array([200, 100, 300, 700, 0, 100, 1100], dtype=np.int64)                   # positives x 100, one empty
    got = rs.sample_calls(1, item_num, torch.from_numpy(keys), torch.from_numpy(counts), indptr, items).cpu().numpy()
    ref, single = [], []
    for k, c in zip(keys, counts):
        if c:
            ref.append(OS.sample_by_key_ids(ors, np.full(c // 100, k), 100, used, item_num))
            single.append(rs1.sample_excluding(1, item_num, torch.full((c // 100,), int(k)).cuda(), 100, indptr, items).cpu().numpy())
    np.testing.assert_array_equal(got, np.concatenate(ref))
    np.testing.assert_array_equal(got, np.concatenate(single))
    st = rs.get_state()
    np.testing.assert_array_equal(st[1], ors.key)
    assert st[2] == ors.pos


@pytest.mark.parametrize("heavy_share", [0.0, 0.05, 0.6], ids=["sparse", "some_heavy", "mostly_heavy"])
def test_long_call_sequences_resolved_speculatively_equal_consecutive_calls(heavy_share):
    """A sequence of hundreds of single-key calls (an evaluation batch: general_dataloader.py:141-146) takes the speculative
    form of csrc/sampler.hip -- accepted values generated once, every call laid out as if none before it had collided, the
    colliding ones resolved round by round and everything behind them shifted.  Against consecutive calls on one numpy stream
    (oracle): the same values AND the same generator state afterwards -- with no collisions at all, with a few users whose
    used-set leaves 4 items (every call of theirs collides for rounds on end), and with so many of those that the slack of
    accepted values runs out and the kernel falls back to call-by-call consumption from the incoming state."""
    from fairrec.sampler import Sampler
    from oracle import sampler as OS
    rng = np.random.default_rng(11)
    user_num, item_num, n_calls = 400, 300, 600
    u = rng.integers(1, user_num, 1500).astype(np.int64)
    i = rng.integers(1, item_num, 1500).astype(np.int64)
    heavy_users = rng.choice(np.arange(1, user_num), int(heavy_share * (user_num - 1)), replace=False)
    for hu in heavy_users:
        left = rng.choice(np.arange(1, item_num), 4, replace=False)
        hv = np.setdiff1d(np.arange(1, item_num), left)
        u, i = np.concatenate([u, np.full(len(hv), hu)]), np.concatenate([i, hv])
    used = [set() for _ in range(user_num)]
    for a, b in zip(u, i):
        used[a].add(int(b))
    rs, ors = _rs(77), OS.MT19937(77)
    smp = Sampler("test", _DS(user_num, item_num, u, i), device="cuda", random_state=rs).set_phase("test")
    indptr, items, _ = smp.used_ids
    keys = rng.integers(1, user_num, n_calls).astype(np.int64)
    counts = (rng.integers(0, 4, n_calls) * 100).astype(np.int64)            # 0-3 positives x 100 negatives
    for rep in range(2):                                                     # (twice: the stream continues across launches)
        got = rs.sample_calls(1, item_num, torch.from_numpy(keys), torch.from_numpy(counts), indptr, items).cpu().numpy()
        ref = [OS.sample_by_key_ids(ors, np.full(c // 100, k), 100, used, item_num) for k, c in zip(keys, counts) if c]
        np.testing.assert_array_equal(got, np.concatenate(ref))
        st = rs.get_state()
        np.testing.assert_array_equal(st[1], ors.key)
        assert st[2] == ors.pos
    rs.check_device_errors() if hasattr(rs, "check_device_errors") else None


@pytest.mark.parametrize("n", [2, 3, 10, 623, 624, 625, 1000, 624 * 3 + 5, 70_001, 8192 * 1024 + 77])
def test_device_randperm_is_torch_randperm(n):
    """fr_randperm (the epoch shuffle of interaction.py:293-297 computed on the device) against torch.randperm itself, from
    generator positions inside, at the end of and right behind a block of 624 words: the same permutation bit for bit, the
    same generator state afterwards (what the next consumer of torch's CPU stream draws is unchanged), twice in a row."""
    from fairrec.sampler.torch_stream import randperm
    for skip in (0, 1, 300, 623, 624):
        torch.manual_seed(1234 + n)
        if skip:
            torch.rand(skip)                      # (one 32-bit word per float draw... whatever it takes: both sides do it)
        st = torch.get_rng_state()
        want = [torch.randperm(n), torch.randperm(n)]
        after = torch.rand(5)
        torch.set_rng_state(st)
        got = [randperm(n, "cuda"), randperm(n, "cuda")]
        assert got[0].dtype == torch.int64 and got[0].is_cuda
        for g, w in zip(got, want):
            assert torch.equal(g.cpu(), w), (n, skip)
        assert torch.equal(torch.rand(5), after), (n, skip)
        if n > 100_000:
            break


def test_shuffle_of_a_device_resident_interaction_equals_the_host_shuffle():
    from fairrec.data.interaction import Interaction
    g = torch.Generator().manual_seed(3)
    cols = {"user_id": torch.randint(1, 1000, (5000,), generator=g), "rating": torch.rand(5000, generator=g)}
    host, dev = Interaction({k: v.clone() for k, v in cols.items()}), Interaction(cols).to("cuda")
    for _ in range(3):
        torch.manual_seed(7)
        host.shuffle()
        a = torch.get_rng_state()
        torch.manual_seed(7)
        dev.shuffle()
        assert torch.equal(torch.get_rng_state(), a)
        for k in cols:
            assert torch.equal(dev[k].cpu(), host[k])


def test_device_randperm_speculation_is_dropped_when_somebody_else_draws():
    """randperm() computes the NEXT call's permutation ahead on a side stream and hands it out only if torch's CPU generator
    is found in exactly the state the speculation started from.  A sequence of shuffles with and without foreign draws in
    between (torch.rand, a randperm of another size, a re-seed) must equal the host's sequence call for call."""
    from fairrec.sampler import torch_stream as ts
    n = 50_000
    def sequence(perm):
        torch.manual_seed(99)
        out = [perm(n), perm(n)]                  # second call: the speculation's hit
        torch.rand(3)                             # a foreign draw: the speculation is stale
        out.append(perm(n))
        out.append(perm(777))                     # another size in between
        out.append(perm(n))
        torch.manual_seed(5)                      # re-seeded
        out.append(perm(n))
        out.append(torch.rand(4))
        return out
    want = sequence(torch.randperm)
    ts._AHEAD.clear()
    got = sequence(lambda k: ts.randperm(k, "cuda"))
    assert len(ts._AHEAD) >= 1                    # something IS computed ahead
    for g, w in zip(got, want):
        assert torch.equal(g.cpu(), w)


def test_call_sequence_of_an_evaluation_batch_at_baseline_sizes():
    """The evaluation loader's draws at the sizes of BASELINE configs[1] -- 100 001 items, 1 000 001 users with 20 interactions
    each, a batch of 2 800 users x 100 negatives per positive (280 k values, ~290 calls with a collision among them): the
    speculative form against the SAME calls issued one by one on a twin generator (the single-call kernel, itself held to
    numpy by the tests above): every id, and the generator state afterwards."""
    from fairrec.sampler import Sampler
    g = torch.Generator().manual_seed(1)
    user_num, item_num, n_calls = 1_000_001, 100_001, 2800
    u = torch.arange(1, user_num).repeat_interleave(20)
    i = torch.randint(1, item_num, (u.numel(),), generator=g)
    rs, rs1 = _rs(404), _rs(404)
    smp = Sampler("test", _DS(user_num, item_num, u.numpy(), i.numpy()), device="cuda", random_state=rs).set_phase("test")
    indptr, items, _ = smp.used_ids
    keys = torch.randint(1, user_num, (n_calls,), generator=g)
    counts = torch.full((n_calls,), 100, dtype=torch.int64)
    counts[::97] = 300
    got = rs.sample_calls(1, item_num, keys, counts, indptr, items)
    single = torch.cat([rs1.sample_excluding(1, item_num, torch.full((int(c) // 100,), int(k)).cuda(), 100, indptr, items)
                        for k, c in zip(keys.tolist(), counts.tolist())])
    assert torch.equal(got, single)
    a, b = rs.get_state(), rs1.get_state()
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2] == b[2]
    # ... and no id is one its user has seen
    owner = torch.repeat_interleave(keys, counts).cuda()
    key_used = (u * item_num + i).cuda()
    assert not torch.isin(owner * item_num + got, key_used).any()
