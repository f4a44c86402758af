// The body of the batch-index sort (sort_segments.hip) as a device function, so that a launch of another kernel can carry the
// sort as its first workgroups (table.hip: the id sort of a lookup rides in the gather's launch -- one latency-bound
// workgroup beside hundreds of gather workgroups instead of a launch of its own in front of them).
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace fr {

static constexpr int SORT_THREADS = 1024;

// 8-bit digits; 7 bits when 16K 64-bit keys leave less LDS for the per-wave tables
__host__ __device__ constexpr int sort_digit_bits(int kpt) { return kpt >= 16 ? 7 : 8; }

#ifdef FR_SORT_STAMPS   // diagnostic build only: phase time stamps of block 0 / thread 0
static __device__ unsigned long long g_sort_stamps[16];   // (one copy per translation unit that carries the body)
#define SORT_STAMP(i) do { if (bid == 0 && threadIdx.x == 0) g_sort_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SORT_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ int block_exclusive_scan_1024(int x, int* scratch /*>=17 ints*/, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(inc, o, 64);
        if (lane >= o) inc += y;
    }
    if (lane == 63) scratch[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int w = lane < 16 ? scratch[lane] : 0;
        int winc = w;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            int y = __shfl_up(winc, o, 64);
            if (lane >= o) winc += y;
        }
        if (lane < 16) scratch[lane] = winc - w;  // exclusive wave offsets
        if (lane == 15) scratch[16] = winc;
    }
    __syncthreads();
    total = scratch[16];
    return scratch[wid] + inc - x;
}

// Stable LSD radix sort of the row ids (DB bits per pass) with wave-level multisplit ranking:
// wave w owns the contiguous chunk [w*P/16, (w+1)*P/16) and walks it 64 keys at a time, so
// (wave, round, lane) order == batch order and ties keep ascending batch position.
// The lanes of a round that share a digit find each other through a 64-bit lane mask OR-ed into LDS
// (order-independent, hence deterministic): rank = popcount(mask & lower lanes).
template <int KPT>
__device__ __forceinline__ void sort_segments_body(const SortJobList& jobs, const int npass_, uint32_t* err, const unsigned bid,
                                                   unsigned char* smem) {
    constexpr int P = KPT * SORT_THREADS;
    constexpr int DB = sort_digit_bits(KPT);
    constexpr int NB = 1 << DB;                    // bins per pass
    constexpr int HPT = 16 * NB / SORT_THREADS;    // (wave, digit) counters per thread in the scan
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);          // [P]
    unsigned long long* masks = keys + P;                                           // [16 waves][NB]
    int* hist = reinterpret_cast<int*>(masks + 16 * NB);                            // [16 waves][NB]
    int* scratch = hist + 16 * NB;                                                  // [32]
    float* fscratch = reinterpret_cast<float*>(scratch + 32);                       // [32]

    SORT_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    if (bid >= jobs.n) {
        // Stamp workgroups (fused FOCF step only): stamp[row] = max(stamp[row], stamp_val) and last_out[j] = last[row]
        // for every id of every list, 1024 ids per workgroup.  Random accesses that miss every cache: issued by ONE
        // workgroup per list they took 17 us each (a CU has a bounded number of misses in flight); spread over M / 1024
        // CUs they are not seen.
        int x = bid - jobs.n;
        for (int q = 0; q < jobs.n; ++q) {
            const int nb = jobs.j[q].stamp ? (jobs.M[q] + SORT_THREADS - 1) / SORT_THREADS : 0;
            if (x < nb) {
                const SortJob& sj = jobs.j[q];
                const int j = x * SORT_THREADS + tid;
                if (j < jobs.M[q]) {
                    const long long r = sj.idx[sj.lay.at(j)];
                    const bool ok = r >= 0 && r < sj.n_rows;   // a bad id is the sorter's to report
                    if (ok) atomicMax(&sj.stamp[r], sj.stamp_val);
                    if (sj.last_out) sj.last_out[j] = ok ? sj.last[r] : 0;
                    if (sj.rec) {   // (id of the other list, own id, rec_f0, aux); ids clamped like the sort keys
                        long long o = sj.rec_idx[j];
                        if (o < 0 || o >= sj.rec_rows) o = 0;
                        sj.rec[j] = make_int4((int)o, ok ? (int)r : 0, __float_as_int(sj.rec_f0[j]),
                                              sj.aux ? __float_as_int(sj.aux[j]) : 0);
                    }
                }
                return;
            }
            x -= nb;
        }
        return;
    }

    const SortJob& job = jobs.j[bid];
    const int M = jobs.M[bid];
    bool bad = false;
    float lo = INFINITY, hi = -INFINITY;
    if (M > 0) {
        // all the loads of the list in flight at once (positions clamped instead of branched around)
        long long r_[KPT];
        float ax_[KPT];
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const int j = q * SORT_THREADS + tid, jc = j < M ? j : M - 1;
            r_[q] = job.idx[job.lay.at(jc)];
            if (job.aux) ax_[q] = job.aux[jc];
        }
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const int j = q * SORT_THREADS + tid;
            unsigned long long k = ~0ull;
            if (j < M) {
                long long r = r_[q];
                if (r == -1 && !job.rec) {
                    r = 0xFFFFFFFFll;   // padding slot ("hole"): sorts behind every real row, belongs to no segment
                } else if (r < 0 || r >= job.n_rows) {
                    bad = true;
                    r = 0;
                }
                k = ((unsigned long long)r << 32) | (unsigned)j;
                if (job.aux) {
                    lo = fminf(lo, ax_[q]);
                    hi = fmaxf(hi, ax_[q]);
                }
            }
            keys[j] = k;
        }
    } else {
#pragma unroll
        for (int q = 0; q < KPT; ++q) keys[q * SORT_THREADS + tid] = ~0ull;
    }
    if (bad && err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
    for (int j = tid; j < 16 * NB; j += SORT_THREADS) masks[j] = 0ull;

    // optional min/max of a float column (the sensitive attribute): the group of a row is its rank
    // among the values present in the batch (focf.py:77)
    if (job.aux) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, o, 64));
            hi = fmaxf(hi, __shfl_xor(hi, o, 64));
        }
        if (lane == 0) {
            fscratch[wid] = lo;
            fscratch[16 + wid] = hi;
        }
    }
    __syncthreads();
    if (job.aux && tid == 0) {
        float l2 = fscratch[0], h2 = fscratch[16];
        for (int w = 1; w < 16; ++w) {
            l2 = fminf(l2, fscratch[w]);
            h2 = fmaxf(h2, fscratch[16 + w]);
        }
        job.aux_minmax[0] = l2;
        job.aux_minmax[1] = h2;
    }

    SORT_STAMP(1);
    const int chunk = wid * (P / 16);
    unsigned long long* mymask = masks + wid * NB;
    int* myhist = hist + wid * NB;
    unsigned long long key[KPT];
    // `npass_` = passes | (significant bits of the LAST pass) << 8.  A last pass of one or two bits (17- or 18-bit row ids: the
    // 100 001-row item tables) puts every lane of a round into at most four bins (+ the padding bin): 64 LDS atomics on one
    // word, in turn -- 5.5 us for that pass against 2.9 for a full one.  Such a pass ranks by ballots instead: same ranks.
    const int npass = npass_ & 0xff, last_bits = npass_ >> 8;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = 32 + DB * pass;
        const bool few = pass == npass - 1 && last_bits >= 1 && last_bits <= 2;
#pragma unroll
        for (int q = 0; q < HPT; ++q) hist[tid * HPT + q] = 0;
        int lrank[KPT], dig[KPT];
#pragma unroll
        for (int r = 0; r < KPT; ++r) key[r] = keys[chunk + r * 64 + lane];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            const int d = (int)(key[r] >> shift) & (NB - 1);
            if (few) {
                unsigned long long peers = 0ull;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const unsigned long long mv = __ballot(d == v);
                    if (d == v) peers = mv;
                }
                const unsigned long long mp = __ballot(d == NB - 1);      // padding slots: all ones in every digit
                if (d == NB - 1) peers = mp;
                const int before = *(volatile int*)&myhist[d];
                const int leader = __ffsll((long long)peers) - 1;
                if (lane == leader) *(volatile int*)&myhist[d] = before + __popcll(peers);
                lrank[r] = before + __popcll(peers & lt_mask);
                dig[r] = d;
                continue;
            }
            // one LDS round trip per round: OR my lane bit in, then read the mask and the running count back
            // (measured: taking the bit back with a second atomic and counting with a third, so that no round waits for
            // the previous one, is not faster -- the phase is bound by the LDS atomics, not by their latency)
            __hip_atomic_fetch_or(&mymask[d], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned long long peers = *(volatile unsigned long long*)&mymask[d];
            const int before = *(volatile int*)&myhist[d];   // same-digit keys of this wave's earlier rounds
            const int leader = __ffsll((long long)peers) - 1;
            if (lane == leader) {   // after every peer's reads (same wave, LDS is in order)
                *(volatile int*)&myhist[d] = before + __popcll(peers);
                *(volatile unsigned long long*)&mymask[d] = 0ull;
            }
            lrank[r] = before + __popcll(peers & lt_mask);
            dig[r] = d;
        }
        SORT_STAMP(2 + 3 * pass);
        __syncthreads();
        {   // exclusive scan of the 16*NB counters in (digit, wave) order
            // thread t owns digit t / (16 / HPT), waves HPT*(t % (16 / HPT)) .. +HPT-1
            constexpr int TPD = 16 / HPT;   // threads per digit
            const int d = tid / TPD, w0 = (tid % TPD) * HPT;
            int* hp = hist + w0 * NB + d;
            int h[HPT], sum = 0;
#pragma unroll
            for (int q = 0; q < HPT; ++q) {
                h[q] = hp[q * NB];
                sum += h[q];
            }
            int total;
            int ex = block_exclusive_scan_1024(sum, scratch, total);
#pragma unroll
            for (int q = 0; q < HPT; ++q) {
                hp[q * NB] = ex;
                ex += h[q];
            }
        }
        SORT_STAMP(3 + 3 * pass);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; ++r) keys[myhist[dig[r]] + lrank[r]] = key[r];
        __syncthreads();
        SORT_STAMP(4 + 3 * pass);
    }

    // segment heads: element j = q*1024 + tid (conflict-free LDS reads, coalesced global writes);
    // a 64-bit ballot per (q, wave) word, then one small scan over the 16*KPT word counts
    int* wcnt = hist;   // reuse: [KPT][16]
    unsigned long long bal[KPT];
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        const int j = q * SORT_THREADS + tid;
        const unsigned long long k = keys[j];
        const unsigned long long kp = j > 0 ? keys[j - 1] : ~k;
        key[q] = k;
        const bool valid = (unsigned)(k >> 32) != 0xFFFFFFFFu;
        const bool head = j < M && valid && (j == 0 || (unsigned)(k >> 32) != (unsigned)(kp >> 32));
        // exactly one thread sees the end of the real keys (first hole / sentinel, or the end of the array)
        if (!valid && (j == 0 || (unsigned)(kp >> 32) != 0xFFFFFFFFu)) scratch[17] = j;
        if (valid && j == P - 1) scratch[17] = P;
        bal[q] = __ballot(head);
        if (lane == 0) wcnt[q * 16 + wid] = __popcll(bal[q]);
    }
    SORT_STAMP(12);
    __syncthreads();
    if (wid == 0) {   // exclusive scan of 16*KPT <= 256 counts by one wave, 4 per lane
        constexpr int NW = 16 * KPT;
        constexpr int PL = (NW + 63) / 64;
        int c[PL], sum = 0;
#pragma unroll
        for (int q = 0; q < PL; ++q) {
            const int idx = lane * PL + q;
            c[q] = idx < NW ? wcnt[idx] : 0;
            sum += c[q];
        }
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(inc, o, 64);
            if (lane >= o) inc += y;
        }
        int ex = inc - sum;
#pragma unroll
        for (int q = 0; q < PL; ++q) {
            const int idx = lane * PL + q;
            if (idx < NW) wcnt[idx] = ex;
            ex += c[q];
        }
        if (lane == 63) scratch[16] = inc;
    }
    __syncthreads();
    SORT_STAMP(13);
    const int total = scratch[16];
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        const int j = q * SORT_THREADS + tid;
        const unsigned long long k = key[q];
        if (j < M && (unsigned)(k >> 32) != 0xFFFFFFFFu) {
            const int seg = wcnt[q * 16 + wid] + __popcll(bal[q] & (lt_mask | (1ull << lane))) - 1;
            const int b = (int)(unsigned)k;
            job.perm[j] = b;
            if (job.pos_of) job.pos_of[b] = j;
            if ((bal[q] >> lane) & 1ull) {
                job.seg_start[seg] = j;
                job.seg_row[seg] = (int)(unsigned)(k >> 32);
                if (job.seg_first) job.seg_first[seg] = b;
            }
            if (job.seg_of) job.seg_of[b] = seg;
        }
    }
    if (tid == 0) {
        job.seg_start[total] = scratch[17];   // number of real (non-padding) ids
        job.n_seg[0] = total;
    }
    if (job.info) {
        // per batch position: where its segment starts and how many members it has (the sorted keys in LDS are not needed
        // any more: their space holds the segment starts)
        int* ss = reinterpret_cast<int*>(keys);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const int j = q * SORT_THREADS + tid;
            if ((bal[q] >> lane) & 1ull) ss[wcnt[q * 16 + wid] + __popcll(bal[q] & lt_mask)] = j;
        }
        if (tid == 0) ss[total] = scratch[17];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KPT; ++q) {
            const int j = q * SORT_THREADS + tid;
            const unsigned long long k = key[q];
            if (j < M && (unsigned)(k >> 32) != 0xFFFFFFFFu) {
                const int seg = wcnt[q * 16 + wid] + __popcll(bal[q] & (lt_mask | (1ull << lane))) - 1;
                const int j0 = ss[seg];
                job.info[(size_t)(unsigned)k * job.info_stride] = make_int2(j0 | ((ss[seg + 1] - j0) << 16), seg);
            }
        }
        if (job.cnt)
            for (int q = tid; q < total; q += SORT_THREADS) job.cnt[q] = 0u;
    }
    SORT_STAMP(14);
}

// dynamic LDS a launch that carries sort_segments_body<KPT> needs
template <int KPT>
constexpr size_t sort_lds_bytes() {
    return (size_t)KPT * SORT_THREADS * 8 + (size_t)16 * (1 << sort_digit_bits(KPT)) * (8 + 4) + 64 * sizeof(int);
}

}  // namespace fr
