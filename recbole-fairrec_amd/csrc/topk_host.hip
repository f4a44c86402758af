// torch.topk(scores, k, dim=-1, largest=True, sorted=True) of the CPU backend, TIES INCLUDED, for the rows an evaluation cannot
// rank without them (host code; no device work).
//
// Reference call site: trainer.py:441-456 scatters the candidates' scores of a user batch into a dense [users, n_items] matrix
// of -inf and collector.py:149 takes `torch.topk(scores_tensor, max(self.topk), dim=-1)` of it, on the CPU.  Which of several
// EQUALLY scored candidates enter a list, and in which order, is then decided by ATen's CPU kernel
// (aten/src/ATen/native/cpu/TopKImpl.h, topk_impl_loop): per row a vector of (value, index) pairs in index order and
//     k * 64 <= n :  std::partial_sort(begin, begin + k, end, gt)
//     otherwise   :  std::nth_element(begin, begin + k - 1, end, gt);  std::sort(begin, begin + k - 1, gt)
// with gt(x, y) = (isnan(x) && !isnan(y)) || x.value > y.value -- no index in the comparison, so libstdc++'s heap select /
// introselect / introsort decide.  Those are header templates: the same calls on the same pairs in the same order give the
// same lists (pinned against torch.topk itself on tie-heavy rows by tests/test_host_logic.py).  With trained scores no list
// hangs on a tie and the device ranking (evaluator/collector.py) stands; an untrained scorer that clamps -- FairGo's predict
// at 0, a saturated sigmoid -- puts most candidates on one value, and then every ranking metric, the validation score, which
// epoch saves and which pretrain checkpoint enters FairGo's finetune stage are functions of that order.
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

extern "C" int fr_topk_like_torch_cpu(const float* rows, int64_t n_rows, int64_t n, int32_t k, int64_t* idx_out, float* val_out) {
    FR_CHECK_ARG(rows && idx_out && n_rows >= 0 && n >= 1 && k >= 1 && k <= n, "fr_topk_like_torch_cpu: bad argument (k %d, n %lld)",
                 k, (long long)n);
    typedef std::pair<float, int64_t> elem_t;
    const auto gt = [](const elem_t& x, const elem_t& y) -> bool {
        return (std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first);
    };
    std::vector<elem_t> queue((size_t)n);
    const bool use_partial_sort = (int64_t)k * 64 <= n;
    for (int64_t r = 0; r < n_rows; ++r) {
        const float* row = rows + (size_t)r * (size_t)n;
        for (int64_t j = 0; j < n; ++j) {
            queue[(size_t)j].first = row[j];
            queue[(size_t)j].second = j;
        }
        if (use_partial_sort) {
            std::partial_sort(queue.begin(), queue.begin() + k, queue.end(), gt);
        } else {
            std::nth_element(queue.begin(), queue.begin() + k - 1, queue.end(), gt);
            std::sort(queue.begin(), queue.begin() + k - 1, gt);
        }
        for (int32_t j = 0; j < k; ++j) {
            idx_out[(size_t)r * k + j] = queue[(size_t)j].second;
            if (val_out) val_out[(size_t)r * k + j] = queue[(size_t)j].first;
        }
    }
    return FR_OK;
}
