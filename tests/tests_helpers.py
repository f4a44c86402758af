"""Small helpers shared by the GPU tests."""
import torch


def sort_segments(idx, n_rows):
    from fairrec import _C
    M = idx.numel()
    dev = idx.device
    perm = torch.full((M + 1,), -1, dtype=torch.int32, device=dev)
    seg_start = torch.full((M + 1,), -1, dtype=torch.int32, device=dev)
    seg_row = torch.full((M + 1,), -1, dtype=torch.int32, device=dev)
    nseg = torch.zeros(1, dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    _C.check(_C.lib().fr_sort_segments(idx.data_ptr(), M, n_rows, perm.data_ptr(), seg_start.data_ptr(),
                                       seg_row.data_ptr(), None, nseg.data_ptr(), err.data_ptr(),
                                       _C.current_stream()), "sort")
    torch.cuda.synchronize()
    return perm.cpu().numpy(), seg_start.cpu().numpy(), seg_row.cpu().numpy(), int(nseg), int(err)
