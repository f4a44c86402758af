// The batch COMPOSITION of the reference's FOCFDataLoader (focf_dataloader.py:37-51) for a whole epoch, in one host call.
//
// Replaces, per batch: `while cnt < step: iid = np.random.choice(select_item[is_select], 1, False); ...` -- host logic in the
// reference too, and kept on the host here: every pick consumes numpy's global MT19937 stream, and which rows form a batch
// must be the reference's.  What a pick costs in numpy: `choice(cands, 1, replace=False)` is `permutation(len(cands))[:1]`,
// i.e. an arange, a full Fisher-Yates shuffle of it (legacy `_shuffle_raw`: for i = n-1 .. 1: j = random_interval(i); swap) and
// one element read -- 70-250 us per pick at 5 000 candidates, 82 picks per 8192-row batch, against 46 us for the batch's whole
// training step on the GPU.  Only element 0 of the permutation is used, and it can be had without the array: the draws j_i
// depend on the generator only, and the value that ends at position 0 is found by walking the swaps backwards in time
// (= upwards in i): p = 0; for i = 1 .. n-1: if j_i == p then p = i.  (A swap(i, j_i) with i > p moves position p's final value
// only when j_i == p, and then that value came from position i; p < i always, so `p == i` never happens.)  The draws
// themselves are numpy's: `random_interval(max)` = masked rejection on single 32-bit outputs (`mt19937_next32 & mask > max`:
// draw again), the generator regenerating its 624 words when `pos` reaches 624 -- restated from numpy/random/src
// (distributions.c: random_interval; mt19937.c: mt19937_gen), pinned against numpy itself by tests/test_host_logic.py.
// 2.2-2.7 ns per candidate and pick on this container's host (numpy: 14 for the permutation alone, 50 with the boolean gather
// and choice()'s argument handling of the reference's loop at 5 000 candidates).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace {

struct Mt {
    uint32_t* key;
    int pos;
    uint32_t tw[624];                        // the tempered outputs of the current 624 words (temper(key[k]) for all k)
    inline void temper_block() {
        for (int k = 0; k < 624; ++k) {      // independent per word: the compiler vectorises it
            uint32_t y = key[k];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            tw[k] = y;
        }
    }
    inline void gen() {
        constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, A = 0x9908b0dfu;
        int kk;
        uint32_t y;
        for (kk = 0; kk < 624 - 397; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
        }
        for (; kk < 623; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + (397 - 624)] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
        }
        y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
        pos = 0;
        temper_block();
    }
};

// np.random.permutation(n)[0] of the legacy generator: the same draws in the same order.  The rejection loop
// `while ((v = next32() & mask) > i);` per draw is run over the words instead: every word is tested against the CURRENT i
// (j[i] = v; i -= (v <= i)), branch-free, with the mask constant while i stays between two powers of two -- the loop-carried
// chain is a compare and a subtract per generator word.
inline int64_t perm_first(Mt& mt, int64_t n, std::vector<uint32_t>& jv) {
    if (n <= 1) return 0;                    // (n == 1: no draw at all, as `for i in reversed(range(1, 1))`)
    jv.resize((size_t)n);
    uint32_t* __restrict__ j = jv.data();
    uint32_t i = (uint32_t)(n - 1);
    while (i >= 1) {
        uint32_t mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const uint32_t lo = (mask >> 1) + 1;                       // i in [lo, mask]: this mask
        while (i >= lo) {
            if (mt.pos == 624) mt.gen();
            const uint32_t* w = mt.tw + mt.pos;
            const int avail = 624 - mt.pos;
            int t = 0;
            for (; t < avail && i >= lo; ++t) {
                const uint32_t v = w[t] & mask;
                j[i] = v;                                          // a rejected word's store is overwritten by the next
                i -= (uint32_t)(v <= i);
            }
            mt.pos += t;
        }
    }
    // element 0 of the permutation: p = 0; the smallest i > p with j[i] == p moves p to i (see the header).  p moves about
    // ln n times: blocks of 64 are tested without an exit (vectorised), and walked only when one of them holds a hit.
    uint32_t p = 0;
    int64_t k = 1;
    while (k < n) {
        const int64_t end = (k + 64 <= n) ? k + 64 : n;
        uint32_t hit = 0;
        for (int64_t q = k; q < end; ++q) hit |= (uint32_t)(j[q] == p);
        if (hit)
            for (int64_t q = k; q < end; ++q)
                if (j[q] == p) p = (uint32_t)q;
        k = end;
    }
    return (int64_t)p;
}

}  // namespace

// state: numpy's legacy layout, uint32 key[624] + pos (625 words, HOST memory), advanced in place.
// item_uniques [n_uniq]: the ascending distinct item ids of the item-sorted interaction table; indptr [n_items + 1]: its CSR.
// A batch: items picked (each at most once per batch) until >= step rows; the epoch: batches until `pr` (advanced by `step`
// per batch, focf_dataloader.py:49) reaches pr_end.  picks_out [cap]: the picked item ids, batch after batch;
// batch_end_out [max_batches]: the number of picks up to and including batch b.  Returns FR_EINVAL when cap / max_batches are
// too small (nothing of the state is advanced then: the caller sizes by the bound n_batches * ceil(step / min degree) + ...).
extern "C" int fr_focf_compose_epoch(uint32_t* state, const int64_t* item_uniques, int64_t n_uniq, const int64_t* indptr,
                                     int64_t step, int64_t pr, int64_t pr_end, int64_t* picks_out, int64_t cap,
                                     int64_t* batch_end_out, int64_t max_batches, int64_t* n_batches_out) {
    FR_CHECK_ARG(state && item_uniques && indptr && picks_out && batch_end_out && n_batches_out && step >= 1 && n_uniq >= 1,
                 "fr_focf_compose_epoch: bad argument");
    FR_CHECK_ARG(state[624] <= 624, "fr_focf_compose_epoch: generator position %u", state[624]);
    uint32_t key[624];
    memcpy(key, state, sizeof(key));
    Mt mt;
    mt.key = key;
    mt.pos = (int)state[624];
    mt.temper_block();
    std::vector<int64_t> cands;
    std::vector<uint32_t> j;
    int64_t n_picks = 0, n_batches = 0;
    while (pr < pr_end) {
        FR_CHECK_ARG(n_batches < max_batches, "fr_focf_compose_epoch: more than %lld batches", (long long)max_batches);
        cands.assign(item_uniques, item_uniques + n_uniq);
        int64_t cnt = 0;
        while (cnt < step && !cands.empty()) {
            const int64_t k = perm_first(mt, (int64_t)cands.size(), j);
            const int64_t iid = cands[(size_t)k];
            cnt += indptr[iid + 1] - indptr[iid];
            cands.erase(cands.begin() + k);
            FR_CHECK_ARG(n_picks < cap, "fr_focf_compose_epoch: more than %lld picks", (long long)cap);
            picks_out[n_picks++] = iid;
        }
        batch_end_out[n_batches++] = n_picks;
        pr += step;
    }
    memcpy(state, key, sizeof(key));
    state[624] = (uint32_t)mt.pos;
    *n_batches_out = n_batches;
    return FR_OK;
}
