"""GPU: the HIP kernels of the row-sharded step (bucket / padded gather / score / owner fairness / grads / apply)
against the reference's golden vectors, run as a 1-rank "world" over RCCL (world_size 1 exercises every kernel and
every collective call; the multi-rank exchange schedule itself is covered by tests/test_sharded_gloo.py)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pg(rccl_world1):
    yield


@pytest.mark.parametrize("lookahead", [False, True], ids=["inline", "lookahead"])
@pytest.mark.parametrize("case", ["focf_none", "focf_value", "focf_absolute", "focf_under", "focf_over",
                                  "focf_value_grouped", "focf_value_d128", "focf_value_pad", "focf_value_long",
                                  "focf_nonparity"])
def test_sharded_hip_matches_reference_golden(pg, case, lookahead):
    from fairrec.sharded import ShardedFocfEngine
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    lr, wd, fw = (float(x) for x in z["hyper"][:3])
    eng = ShardedFocfEngine(torch.tensor(z["U0"], device="cuda"), torch.tensor(z["I0"], device="cuda"),
                            str(z["objective"]), fw, lr, wd, capacity_factor=1.0)
    snaps = set(int(s) for s in z["snaps"])
    losses = []
    T = z["user_id"].shape[0]
    batches = [[torch.tensor(z[k][t], device="cuda") for k in ("user_id", "item_id", "rating", "sst")] for t in range(T)]
    for t in range(T):
        nxt = None
        if lookahead and t + 1 < T and t % 4 != 3:   # the next step's bucket / id exchange / sort runs on a side stream
            nxt = (batches[t + 1][0], batches[t + 1][1], batches[t + 1][3])
        loss, pred = eng.forward(*batches[t], next_batch=nxt)
        losses.append(loss.reshape(1).clone())
        if t == 0:
            np.testing.assert_allclose(pred.cpu().numpy(), z["pred_step1"], rtol=1e-4, atol=1e-6)
        eng.backward_adam()
        if (t + 1) in snaps:
            eng.flush()
            for tag, tab in (("U", eng.U), ("I", eng.I)):
                a, b = tab.weight.cpu().numpy(), z[f"{tag}_after{t + 1}"]
                assert (np.abs(a - b) <= 1e-4 * np.abs(b) + 1e-6).all(), (tag, t + 1, np.abs(a - b).max())
    np.testing.assert_allclose(torch.cat(losses).cpu().numpy(), z["loss"], rtol=1e-4)
    eng.check_device_errors()


def test_bucket_by_owner_bit_exact():
    """Dense [G, cap] layout and the shared [G, 2*cap+1] layout (second list + the (min, max) pair of an aux column in
    the trailing slot of every chunk)."""
    from fairrec.sharded import HipOps
    ops = HipOps("cuda")
    g = torch.Generator().manual_seed(5)
    for M, G, cap, shared in ((1, 1, 1, False), (100, 2, 80, True), (8192, 8, 1200, True), (8192, 4, 8192, False),
                              (5000, 3, 100, True), (8192, 8, 2048, True)):
        idx = torch.randint(0, 10 ** 6, (M,), generator=g, dtype=torch.int64)
        aux = torch.randn(M, generator=g) if shared else None
        stride, offset, aux_slot = (2 * cap + 1, cap, 2 * cap) if shared else (cap, 0, 0)
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        send = torch.full((G * stride,), -7, dtype=torch.int64, device="cuda")
        slot = torch.empty(M, dtype=torch.int32, device="cuda")
        counts = torch.empty(G, dtype=torch.int32, device="cuda")
        ops.bucket_by_owner(idx.cuda(), G, cap, stride, offset, send, slot, counts, aux.cuda() if shared else None,
                            aux_slot, err)
        send, slot, counts = send.cpu().numpy(), slot.cpu().numpy(), counts.cpu().numpy()
        exp_send = np.full(G * stride, -7, dtype=np.int64)          # slots of the other list stay untouched
        for o in range(G):
            exp_send[o * stride + offset:o * stride + offset + cap] = -1
        exp_slot = np.full(M, -1, dtype=np.int32)
        fill = np.zeros(G, dtype=np.int64)
        overflow = False
        for j, r in enumerate(idx.numpy()):
            o = r % G
            if fill[o] < cap:
                exp_send[o * stride + offset + fill[o]] = r // G
                exp_slot[j] = o * stride + offset + fill[o]
                fill[o] += 1
            else:
                overflow = True
        if shared:
            pair = np.array([aux.min().item(), aux.max().item()], dtype=np.float32).view(np.int64)[0]
            for o in range(G):
                exp_send[o * stride + aux_slot] = pair
        np.testing.assert_array_equal(send, exp_send)
        np.testing.assert_array_equal(slot, exp_slot)
        np.testing.assert_array_equal(counts, fill)
        assert bool(int(err.item()) & 4) == overflow


def test_sort_ignores_padding_slots():
    from tests_helpers import sort_segments
    idx = torch.tensor([5, -1, 3, 5, -1, 0, 3], dtype=torch.int64).cuda()
    perm, seg_start, seg_row, nseg, err = sort_segments(idx, 10)
    assert err == 0 and nseg == 3
    assert seg_row[:3].tolist() == [0, 3, 5]
    assert seg_start[:4].tolist() == [0, 1, 3, 5]        # 5 real ids; the two -1 slots belong to no segment
    assert perm[:5].tolist() == [5, 2, 6, 0, 3]
