// Negative sampler on the device, bit-exact with the reference's host sampler.
//
// Replaces: recbole/sampler/sampler.py:240-241 Sampler._uni_sampling = np.random.randint(1, item_num, n), and the
// rejection loop of AbstractSampler.sample_by_key_ids (:145-197).  The third-party arithmetic underneath is numpy's
// legacy RandomState: MT19937 (mt19937_seed / mt19937_gen) and, for the int64 default dtype on a range below 2^32,
// masked rejection on single 32-bit outputs (buffered_bounded_masked_uint32).  The generator state lives in device
// memory in numpy's own layout (key[624], pos), so the stream can be handed to and taken back from
// np.random.set_state / get_state at any point.
//
// One workgroup does a whole call: the MT recurrence is sequential in blocks of 624 words (each block = four
// barrier-separated parallel phases), the accept/reject filter and the "which positions collide with the user's
// used-set" compaction are block-wide ordered scans, and the re-draw rounds loop inside the kernel until no position
// is left -- the number of rounds is data dependent, and the stream position after the call must be exact before the
// next batch draws from it, so the loop can not be cut short on the host side without a sync.
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"

namespace fr {

static constexpr int MT_N = 624, MT_M = 397;
static constexpr int SAMPLER_THREADS = 1024;

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// new state block from the old one (mt19937_gen's three loops as parallel phases: [0,227) reads old words only,
// [227,454) and [454,623) read words the previous phase produced, word 623 reads new[0] and new[396])
__device__ __forceinline__ void mt_twist(const uint32_t* __restrict__ o, uint32_t* __restrict__ n, int t) {
    if (t < MT_N - MT_M) n[t] = o[t + MT_M] ^ mt_mix(o[t], o[t + 1]);
    __syncthreads();
    if (t >= MT_N - MT_M && t < 2 * (MT_N - MT_M)) n[t] = n[t - (MT_N - MT_M)] ^ mt_mix(o[t], o[t + 1]);
    __syncthreads();
    if (t >= 2 * (MT_N - MT_M) && t < MT_N - 1) n[t] = n[t - (MT_N - MT_M)] ^ mt_mix(o[t], o[t + 1]);
    __syncthreads();
    if (t == MT_N - 1) n[t] = n[MT_M - 1] ^ mt_mix(o[t], n[0]);
    __syncthreads();
}

// ordered block-wide exclusive scan of one flag per thread; returns the rank, `total` = number of set flags
__device__ __forceinline__ int flag_scan(bool flag, int* wave_cnt, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();   // wave_cnt may still be read by the previous call
    if (lane == 0) wave_cnt[wid] = __popcll(b);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SAMPLER_THREADS / 64; ++w) {
        const int c = wave_cnt[w];
        off += w < wid ? c : 0;
        tot += c;
    }
    total = tot;
    return off + in_wave;
}

__device__ __forceinline__ bool used_contains(const int32_t* __restrict__ items, long long lo, long long hi, int v) {
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        const int x = items[mid];
        if (x == v) return true;
        if (x < v) lo = mid + 1;
        else hi = mid;
    }
    return false;
}

// Two calling modes: one call over `total` positions keyed by key_ids[i % n_keys]  (call_offsets == nullptr), or a
// SEQUENCE of calls c = 0..n_calls-1, call c drawing positions [call_offsets[c], call_offsets[c+1]) for the single key
// call_keys[c] -- each call completes its re-draw rounds before the next one draws, exactly like consecutive
// sample_by_user_ids calls on one numpy stream (the evaluation loader samples user by user, general_dataloader.py:141-146).
__global__ __launch_bounds__(SAMPLER_THREADS) void sample_negatives_kernel(
    uint32_t* __restrict__ state, long long low, uint32_t span, uint32_t mask, const int64_t* __restrict__ key_ids,
    long long n_keys, long long total_all, const int64_t* __restrict__ call_keys,
    const int64_t* __restrict__ call_offsets, long long n_calls, const int64_t* __restrict__ used_indptr,
    const int32_t* __restrict__ used_items, long long n_users, int64_t* __restrict__ out_all,
    int32_t* __restrict__ list_a, int32_t* __restrict__ list_b, int32_t* __restrict__ rounds_out, uint32_t* err) {
    __shared__ uint32_t mt[2][MT_N];
    __shared__ int wave_cnt[SAMPLER_THREADS / 64];
    __shared__ int s_last;
    const int t = threadIdx.x;
    int cur = 0;
    if (t < MT_N) mt[0][t] = state[t];
    int pos = (int)state[MT_N];
    __syncthreads();

    if (span == 0) {   // numpy: a one-value range consumes nothing
        const long long n_out = call_offsets ? call_offsets[n_calls] : total_all;
        for (long long e = t; e < n_out; e += SAMPLER_THREADS) out_all[e] = low;
        if (t == 0 && rounds_out) rounds_out[0] = 1;
        return;
    }

    int rounds = 0;
    const long long calls = call_offsets ? n_calls : 1;
    for (long long call = 0; call < calls; ++call) {
    const long long o0 = call_offsets ? call_offsets[call] : 0;
    const long long total = call_offsets ? call_offsets[call + 1] - o0 : total_all;
    int64_t* __restrict__ out = out_all + o0;
    const long long call_key = call_keys ? call_keys[call] : -1;
    const int32_t* list = nullptr;      // positions to (re)draw, ascending; nullptr = all of [0, total)
    int32_t* next = list_a;
    long long need = total;
    while (need > 0) {
        // ---- draw `need` values, in position order, from the continuing stream ----
        long long produced = 0;
        while (produced < need) {
            if (pos == MT_N) {
                mt_twist(mt[cur], mt[cur ^ 1], t);
                cur ^= 1;
                pos = 0;
            }
            uint32_t v = 0;
            bool acc = false;
            if (t >= pos && t < MT_N) {
                v = mt_temper(mt[cur][t]) & mask;
                acc = v <= span;
            }
            int cnt;
            const int k = flag_scan(acc, wave_cnt, cnt);
            const long long remaining = need - produced;
            if (acc && k < remaining) {
                const long long e = produced + k;
                out[list ? (long long)list[e] : e] = low + (long long)v;
                if (k == remaining - 1) s_last = t;     // the draw that yields the last value ends the consumption
            }
            __syncthreads();
            if (cnt >= remaining) {
                pos = s_last + 1;
                produced = need;
            } else {
                pos = MT_N;
                produced += cnt;
            }
            __syncthreads();   // s_last is rewritten in the next iteration
        }
        ++rounds;
        if (!used_indptr) break;
        __threadfence_block();
        // ---- which of the positions just drawn hit their key's used-set?  (ordered compaction -> next round) ----
        long long kept = 0;
        for (long long base = 0; base < need; base += SAMPLER_THREADS) {
            const long long e = base + t;
            bool hit = false;
            long long i = 0;
            if (e < need) {
                i = list ? (long long)list[e] : e;
                const long long u = call_keys ? call_key : key_ids[i % n_keys];
                if (u < 0 || u >= n_users) {
                    if (err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
                } else {
                    hit = used_contains(used_items, used_indptr[u], used_indptr[u + 1], (int)out[i]);
                }
            }
            int cnt;
            const int k = flag_scan(hit, wave_cnt, cnt);
            if (hit) next[kept + k] = (int32_t)i;
            kept += cnt;
        }
        __syncthreads();
        __threadfence_block();
        list = next;
        next = (next == list_a) ? list_b : list_a;
        need = kept;
    }
    __syncthreads();   // the next call reuses the collision lists
    }
    __syncthreads();
    if (t < MT_N) state[t] = mt[cur][t];
    if (t == 0) {
        state[MT_N] = (uint32_t)pos;
        if (rounds_out) rounds_out[0] = rounds;
    }
}

// ---- A long SEQUENCE of single-key calls, resolved speculatively ------------------------------------------------------
// The evaluation loader draws user by user (general_dataloader.py:141-146): thousands of calls of 101-303 values per batch of
// users, each of which must finish its re-draw rounds before the next one draws.  Call by call that is ~11 us of barriers per
// call in sample_negatives_kernel -- 30 ms per evaluation batch, 200 times the scoring and ranking of the batch.  But the
// stream's ACCEPTED values (tempered word & mask <= span) are a function of the generator alone, a call consumes them in order
// -- its positions first, then one more per collision with the user's used-set, round by round -- and collisions are rare
// (|used-set| / n_items per draw).  So: (A) the accepted values of the whole sequence, plus slack, are generated once (the
// twist is the only sequential part: ~1.5 us per 624 words), each with its raw index in the stream and the generator block it
// came from kept; (B) every call is laid out as if NO call before it had collided, shifted by D = the extra values consumed so
// far; the first call with a collision is found in parallel, everything before it is final, that one call is resolved round
// by round exactly as the sequential kernel does, D grows by what it consumed beyond its size, and the search resumes behind
// it.  The generator state handed back is the block of the last consumed raw word and the position behind it: the same
// values, the same stream position as call-by-call consumption (tests/test_sampler_hip.py holds both forms against numpy).
// If the slack does not suffice (a pathological collision rate), nothing has been published yet and the sequence runs call by
// call from the start.
static constexpr int CALLS_WINDOW = 1 << 16;      // positions examined per search for the next colliding call

__device__ __forceinline__ void calls_sequential(uint32_t (&mt)[2][MT_N], int& cur, int& pos, int* wave_cnt, int& s_last,
                                                 long long low, uint32_t span, uint32_t mask, const int64_t* __restrict__ call_keys,
                                                 const int64_t* __restrict__ call_offsets, long long n_calls,
                                                 const int64_t* __restrict__ used_indptr, const int32_t* __restrict__ used_items,
                                                 long long n_users, int64_t* __restrict__ out_all, int32_t* list_a, int32_t* list_b,
                                                 uint32_t* err) {
    const int t = threadIdx.x;
    for (long long call = 0; call < n_calls; ++call) {
        const long long o0 = call_offsets[call], total = call_offsets[call + 1] - o0;
        int64_t* __restrict__ out = out_all + o0;
        const long long call_key = call_keys[call];
        const int32_t* list = nullptr;
        int32_t* next = list_a;
        long long need = total;
        while (need > 0) {
            long long produced = 0;
            while (produced < need) {
                if (pos == MT_N) {
                    mt_twist(mt[cur], mt[cur ^ 1], t);
                    cur ^= 1;
                    pos = 0;
                }
                uint32_t v = 0;
                bool acc = false;
                if (t >= pos && t < MT_N) {
                    v = mt_temper(mt[cur][t]) & mask;
                    acc = v <= span;
                }
                int cnt;
                const int k = flag_scan(acc, wave_cnt, cnt);
                const long long remaining = need - produced;
                if (acc && k < remaining) {
                    const long long e = produced + k;
                    out[list ? (long long)list[e] : e] = low + (long long)v;
                    if (k == remaining - 1) s_last = t;
                }
                __syncthreads();
                if (cnt >= remaining) {
                    pos = s_last + 1;
                    produced = need;
                } else {
                    pos = MT_N;
                    produced += cnt;
                }
                __syncthreads();
            }
            __threadfence_block();
            long long kept = 0;
            for (long long base = 0; base < need; base += SAMPLER_THREADS) {
                const long long e = base + t;
                bool hit = false;
                long long i = 0;
                if (e < need) {
                    i = list ? (long long)list[e] : e;
                    if (call_key < 0 || call_key >= n_users) {
                        if (err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
                    } else {
                        hit = used_contains(used_items, used_indptr[call_key], used_indptr[call_key + 1], (int)out[i]);
                    }
                }
                int cnt;
                const int k = flag_scan(hit, wave_cnt, cnt);
                if (hit) next[kept + k] = (int32_t)i;
                kept += cnt;
            }
            __syncthreads();
            __threadfence_block();
            list = next;
            next = (next == list_a) ? list_b : list_a;
            need = kept;
        }
        __syncthreads();
    }
}

// (A) The accepted values.  A1, one small workgroup that does nothing but twist the generator's blocks in LDS, every block kept
// (sample_calls_twist_kernel; the call-by-call kernel's 1024 threads spend four barriers per block AND temper, mask, scan).
// The block count is what the wanted number of accepted values needs at the mask's acceptance rate plus a margin; should
// chance leave fewer, (B) sees it and falls back.  A2, one wave per block over the chip: temper, mask, count the accepted
// words.  A3, the same waves again: each sums the counts of the blocks before its own and writes its accepted words, with
// their raw index in the stream, at their places.
__device__ __forceinline__ long long calls_want(long long total) { return total + (total / 32 > 1024 ? total / 32 : 1024); }

__global__ __launch_bounds__(256) void sample_calls_twist_kernel(const uint32_t* __restrict__ state, uint32_t span, uint32_t mask,
                                                                 const int64_t* __restrict__ call_offsets, long long n_calls,
                                                                 long long cap, uint32_t* __restrict__ snap, long long nblk_max,
                                                                 long long* __restrict__ hdr) {
    // Word n of the stream needs words n - 227, n - 623 and n - 624: a block of 624 is three dependent phases of <= 227 words.
    // Four waves, two copies of the block (old -> new, so that no phase overwrites what a neighbour still reads), one barrier
    // per phase: ~0.3 us per block (one wave walking the block 64 words at a time in place: 1.2 us).
    __shared__ uint32_t buf[2][MT_N];
    const int t = threadIdx.x;
    for (int k = t; k < MT_N; k += 256) {
        const uint32_t w = state[k];
        buf[0][k] = w;
        snap[k] = w;
    }
    const int pos0 = (int)state[MT_N];
    long long want = calls_want(call_offsets[n_calls]);
    if (want > cap) want = cap;
    // blocks beyond the incoming one: accepted words per block = 624 (span + 1) / (mask + 1) on average, 6 % and two blocks spare
    const double rate = ((double)span + 1.0) / ((double)mask + 1.0);
    long long need = want - (long long)((MT_N - pos0) * rate * 0.9);
    long long nblk = need > 0 ? (long long)((double)need / (MT_N * rate * 0.94)) + 2 : 0;
    if (nblk > nblk_max - 1) nblk = nblk_max - 1;
    __syncthreads();
    constexpr int H = MT_N - MT_M;      // 227
    // (the phases hand LDS words to each other: the barrier waits for this wave's LDS traffic only, not for the block copies
    // on their way to global memory, which __syncthreads' fence would drain three times per block)
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    int cur = 0;
    for (long long blk = 1; blk <= nblk; ++blk) {
        const uint32_t* __restrict__ o = buf[cur];
        uint32_t* __restrict__ n = buf[cur ^ 1];
        if (t < H) n[t] = o[t + MT_M] ^ mt_mix(o[t], o[t + 1]);
        lds_barrier();
        if (t < H) n[H + t] = n[t] ^ mt_mix(o[H + t], o[H + t + 1]);
        lds_barrier();
        {
            const int k = 2 * H + t;
            if (k < MT_N - 1) n[k] = n[k - H] ^ mt_mix(o[k], o[k + 1]);
            else if (k == MT_N - 1) n[k] = n[MT_M - 1] ^ mt_mix(o[k], n[0]);
        }
        lds_barrier();
        for (int k = t; k < MT_N; k += 256) snap[blk * MT_N + k] = n[k];
        cur ^= 1;
    }
    if (t == 0) {
        hdr[1] = nblk + 1;      // blocks kept, the incoming one included
        hdr[2] = pos0;
        hdr[3] = want;
    }
}

// PASS 0: cnt[blk] = accepted words of block blk; PASS 1: the words themselves, behind those of the blocks before
template <int PASS>
__global__ __launch_bounds__(64) void sample_calls_temper_kernel(uint32_t span, uint32_t mask, const uint32_t* __restrict__ snap,
                                                                 long long* __restrict__ hdr, uint32_t* __restrict__ cnt,
                                                                 uint32_t* __restrict__ acc_val, uint32_t* __restrict__ acc_raw) {
    const long long blk = blockIdx.x, nblk = hdr[1];
    if (blk >= nblk) return;
    const int l = threadIdx.x, pos = blk == 0 ? (int)hdr[2] : 0;
    const long long want = hdr[3];
    long long before = 0;
    if (PASS == 1) {
        for (long long b0 = 0; b0 < blk; b0 += 64) before += b0 + l < blk ? cnt[b0 + l] : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor((int)before, o, 64);      // (below 2^31: the buffers hold < 2^30 values)
    }
    long long produced = before;
    for (int k0 = pos & ~63; k0 < MT_N; k0 += 64) {
        const int k = k0 + l;
        uint32_t v = 0;
        bool acc = false;
        if (k >= pos && k < MT_N) {
            v = mt_temper(snap[blk * MT_N + k]) & mask;
            acc = v <= span;
        }
        const unsigned long long bal = __ballot(acc);
        if (PASS == 1) {
            const long long e = produced + __popcll(bal & ((1ull << l) - 1ull));
            if (acc && e < want) {
                acc_val[e] = v;
                acc_raw[e] = (uint32_t)(blk * MT_N + k);
            }
        }
        produced += __popcll(bal);
    }
    if (PASS == 0 && l == 0) cnt[blk] = (uint32_t)(produced - before);
    if (PASS == 1 && l == 0 && blk == nblk - 1) hdr[0] = produced < want ? produced : want;      // accepted values available
}

// (P) every position of the sequence, in parallel over the chip: its call, and for each shift d < CALLS_SHIFTS whether the
// accepted value it would take under that shift (acc_val[p + d]) lies in its user's used-set -- one bit per shift.  The
// sequential part (B) then finds "the first position that collides under the current shift" by reading one word per position.
static constexpr int CALLS_SHIFTS = 32;
__global__ __launch_bounds__(256) void sample_calls_hits_kernel(long long low, const int64_t* __restrict__ call_keys,
                                                                const int64_t* __restrict__ call_offsets, long long n_calls,
                                                                const int64_t* __restrict__ used_indptr,
                                                                const int32_t* __restrict__ used_items, long long n_users,
                                                                const uint32_t* __restrict__ acc_val,
                                                                const long long* __restrict__ n_acc_in, uint32_t* __restrict__ hmask,
                                                                int32_t* __restrict__ pos_call, uint32_t* err) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = call_offsets[n_calls];
    if (p >= total) return;
    long long a = 0, b = n_calls - 1;      // the call of position p: the largest c with call_offsets[c] <= p
    while (a < b) {
        const long long mid = (a + b + 1) >> 1;
        if (call_offsets[mid] <= p) a = mid;
        else b = mid - 1;
    }
    pos_call[p] = (int32_t)a;
    const long long u = call_keys[a];
    uint32_t bits = 0;
    if (u < 0 || u >= n_users) {
        if (err) atomicOr(err, FR_DEV_ERR_INDEX_RANGE);
    } else {
        const long long ulo = used_indptr[u], uhi = used_indptr[u + 1];
        const long long n_acc = n_acc_in[0];
        if (uhi > ulo) {
#pragma unroll 4
            for (int d = 0; d < CALLS_SHIFTS; ++d)
                if (p + d < n_acc && used_contains(used_items, ulo, uhi, (int)(low + (long long)acc_val[p + d]))) bits |= 1u << d;
        }
    }
    hmask[p] = bits;
}

// (B) ONE workgroup: the calls in order, everything before the next colliding position final
__global__ __launch_bounds__(SAMPLER_THREADS) void sample_calls_fast_kernel(
    uint32_t* __restrict__ state, long long low, uint32_t span, uint32_t mask, const int64_t* __restrict__ call_keys,
    const int64_t* __restrict__ call_offsets, long long n_calls, const int64_t* __restrict__ used_indptr,
    const int32_t* __restrict__ used_items, long long n_users, int64_t* __restrict__ out, const uint32_t* __restrict__ acc_val,
    const uint32_t* __restrict__ acc_raw, const uint32_t* __restrict__ snap, const long long* __restrict__ n_acc_in,
    const uint32_t* __restrict__ hmask, const int32_t* __restrict__ pos_call, int32_t* list_a, int32_t* list_b, uint32_t* err) {
    __shared__ uint32_t mt[2][MT_N];
    __shared__ int wave_cnt[SAMPLER_THREADS / 64];
    __shared__ int s_last;
    __shared__ long long s_first;
    const int t = threadIdx.x;
    int cur = 0;
    int pos = (int)state[MT_N];
    const int pos0 = pos;
    const long long total = call_offsets[n_calls];
    const long long n_acc = n_acc_in[0];
    __syncthreads();
    long long D = 0, p0 = 0;      // positions before p0 are final; D = accepted values consumed beyond the positions so far
    bool over = total + D > n_acc;
    while (p0 < total && !over) {
        const long long pe = p0 + CALLS_WINDOW < total ? p0 + CALLS_WINDOW : total;
        if (t == 0) s_first = pe;
        __syncthreads();
        long long mine = pe;
        for (long long p = p0 + t; p < pe; p += SAMPLER_THREADS)
            if ((hmask[p] >> D) & 1u) {
                mine = p;
                break;
            }
        if (mine < pe) atomicMin((unsigned long long*)&s_first, (unsigned long long)mine);
        __syncthreads();
        const long long pf0 = s_first;                       // the first colliding position of the window (pe: none)
        const long long cf = pf0 < pe ? pos_call[pf0] : -1;
        const long long pf = cf >= 0 ? call_offsets[cf] : pe;      // the positions before the colliding call are final
        if (pf + D > n_acc) {
            over = true;
            break;
        }
        for (long long p = p0 + t; p < pf; p += SAMPLER_THREADS) out[p] = low + (long long)acc_val[p + D];
        __syncthreads();
        if (cf < 0) {
            p0 = pe;
            continue;
        }
        // the colliding call, round by round: its positions take the next accepted values in position order, the positions that
        // hit the used-set take the values behind those, and so on
        const long long n = call_offsets[cf + 1] - pf, base = pf + D;
        const long long u = call_keys[cf];
        const long long ulo = used_indptr[u], uhi = used_indptr[u + 1];
        const int32_t* list = nullptr;
        int32_t* next = list_a;
        long long need = n, consumed = 0;
        while (need > 0) {
            if (base + consumed + need > n_acc) {
                over = true;
                break;
            }
            for (long long e = t; e < need; e += SAMPLER_THREADS)
                out[pf + (list ? (long long)list[e] : e)] = low + (long long)acc_val[base + consumed + e];
            __threadfence_block();
            __syncthreads();
            long long kept = 0;
            for (long long b0 = 0; b0 < need; b0 += SAMPLER_THREADS) {
                const long long e = b0 + t;
                bool hit = false;
                long long i = 0;
                if (e < need) {
                    i = list ? (long long)list[e] : e;
                    hit = used_contains(used_items, ulo, uhi, (int)out[pf + i]);
                }
                int cnt;
                const int k = flag_scan(hit, wave_cnt, cnt);
                if (hit) next[kept + k] = (int32_t)i;
                kept += cnt;
            }
            __syncthreads();
            __threadfence_block();
            consumed += need;
            list = next;
            next = (next == list_a) ? list_b : list_a;
            need = kept;
        }
        if (over) break;
        D += consumed - n;
        p0 = pf + n;
        if (D >= CALLS_SHIFTS) over = true;      // (the hit bits cover CALLS_SHIFTS shifts)
    }
    __syncthreads();
    if (over) {
        // too many collisions for the slack / the shifts prepared: call by call from the incoming state (nothing of the stream
        // was published; the outputs written so far are overwritten)
        if (t < MT_N) mt[0][t] = state[t];
        cur = 0;
        pos = pos0;
        __syncthreads();
        calls_sequential(mt, cur, pos, wave_cnt, s_last, low, span, mask, call_keys, call_offsets, n_calls, used_indptr, used_items,
                         n_users, out, list_a, list_b, err);
        __syncthreads();
        if (t < MT_N) state[t] = mt[cur][t];
        if (t == 0) state[MT_N] = (uint32_t)pos;
        return;
    }
    // the generator after the last consumed word
    const long long E = total + D;
    if (E > 0) {
        const uint32_t R = acc_raw[E - 1];
        const long long bf = R / MT_N;
        if (t < MT_N) state[t] = snap[bf * MT_N + t];
        if (t == 0) state[MT_N] = (uint32_t)(R % MT_N) + 1u;
    }
}

// mt19937_seed(state, seed): Knuth's LCG over the 624 words, pos = 624 (np.random.seed(int))
__global__ void mt19937_seed_kernel(uint32_t* __restrict__ state, uint32_t seed) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t s = seed;
    for (int i = 0; i < MT_N; ++i) {
        state[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    state[MT_N] = MT_N;
}

}  // namespace fr

using namespace fr;

extern "C" int fr_mt19937_seed(uint32_t* state, uint32_t seed, void* stream_) {
    FR_CHECK_ARG(state, "fr_mt19937_seed: null state");
    hipLaunchKernelGGL(mt19937_seed_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, state, seed);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" size_t fr_sample_negatives_workspace_bytes(int64_t total) {
    return total < 1 ? 0 : 2 * align_up((size_t)total * sizeof(int32_t), 256);
}

static int sample_launch(uint32_t* state, int64_t low, int64_t high, const int64_t* key_ids, int64_t n_keys, int64_t total,
                         const int64_t* call_keys, const int64_t* call_offsets, int64_t n_calls, int64_t max_call,
                         const int64_t* used_indptr, const int32_t* used_items, int64_t n_users, int64_t* out,
                         int32_t* rounds_out, void* ws, size_t ws_bytes, uint32_t* err_flag, hipStream_t stream) {
    int32_t *la = nullptr, *lb = nullptr;
    if (used_indptr) {
        FR_CHECK_ARG(ws && ws_bytes >= fr_sample_negatives_workspace_bytes(max_call), "fr_sample_negatives: workspace too small");
        la = (int32_t*)ws;
        lb = (int32_t*)((char*)ws + align_up((size_t)max_call * sizeof(int32_t), 256));
    }
    const uint32_t span = (uint32_t)(high - 1 - low);
    uint32_t mask = span;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    ProfScope prof(K_SAMPLE_NEG, stream);
    FR_LAUNCH(prof, sample_negatives_kernel, dim3(1), dim3(SAMPLER_THREADS), 0, stream, state, (long long)low, span, mask,
              key_ids, (long long)n_keys, (long long)total, call_keys, call_offsets, (long long)n_calls, used_indptr,
              used_items, (long long)n_users, out, la, lb, rounds_out, err_flag);
    FR_CHECK_LAUNCH();
    return FR_OK;
}

extern "C" int fr_sample_negatives(uint32_t* state, int64_t low, int64_t high, const int64_t* key_ids, int64_t n_keys,
                                   int32_t num, const int64_t* used_indptr, const int32_t* used_items, int64_t n_users,
                                   int64_t* out, int32_t* rounds_out, void* ws, size_t ws_bytes, uint32_t* err_flag,
                                   void* stream_) {
    FR_CHECK_ARG(state && out && n_keys >= 1 && num >= 1 && n_keys * (int64_t)num <= (1ll << 30),
                 "fr_sample_negatives: bad size");
    FR_CHECK_ARG(high > low && high - 1 - low < 0xffffffffll, "fr_sample_negatives: range [%lld, %lld) not below 2^32",
                 (long long)low, (long long)high);
    FR_CHECK_ARG(!used_indptr || (used_items && key_ids && n_users >= 1 && ws), "fr_sample_negatives: used-set arguments");
    const int64_t total = n_keys * (int64_t)num;
    return sample_launch(state, low, high, key_ids, n_keys, total, nullptr, nullptr, 1, total, used_indptr, used_items,
                         n_users, out, rounds_out, ws, ws_bytes, err_flag, (hipStream_t)stream_);
}

// A sequence of single-key calls on one stream: call c fills out[call_offsets[c] .. call_offsets[c+1]) for key
// call_keys[c]; `max_call` >= the longest call sizes the workspace (fr_sample_negatives_workspace_bytes(max_call)).
// workspace of the speculative form of a call sequence of `total` values in all (0 values: the call-by-call form's)
static void calls_fast_layout(int64_t total, int64_t max_call, size_t& cap, size_t& nblk, size_t off[9], size_t& bytes) {
    cap = (size_t)total + std::max<size_t>(1024, (size_t)total / 32);
    nblk = 2 * cap / MT_N + 4;      // the mask keeps more than every second word
    size_t o = 0;
    auto take = [&](size_t b) { const size_t at = o; o += align_up(b, 256); return at; };
    off[0] = take(cap * 4);                 // accepted values
    off[1] = take(cap * 4);                 // ... their raw indices
    off[2] = take(nblk * MT_N * 4);         // the generator's blocks
    off[3] = take((size_t)max_call * 4);    // collision lists of the call being resolved
    off[4] = take((size_t)max_call * 4);
    off[5] = take(32);                      // (accepted values available, blocks kept, incoming position, values wanted)
    off[6] = take((size_t)total * 4);       // per position: for which shifts it collides
    off[7] = take((size_t)total * 4);       // ... and its call
    off[8] = take(nblk * 4);                // accepted words per block
    bytes = o;
}
extern "C" size_t fr_sample_negatives_calls_workspace_bytes(int64_t total, int64_t max_call) {
    if (total < 1 || max_call < 1) return 0;
    size_t cap, nblk, off[9], bytes;
    calls_fast_layout(total, max_call, cap, nblk, off, bytes);
    return bytes;
}

extern "C" int fr_sample_negatives_calls(uint32_t* state, int64_t low, int64_t high, const int64_t* call_keys,
                                         const int64_t* call_offsets, int64_t n_calls, int64_t max_call,
                                         const int64_t* used_indptr, const int32_t* used_items, int64_t n_users,
                                         int64_t* out, void* ws, size_t ws_bytes, uint32_t* err_flag, void* stream_) {
    FR_CHECK_ARG(state && out && call_keys && call_offsets && n_calls >= 1 && max_call >= 1 && max_call <= (1ll << 30),
                 "fr_sample_negatives_calls: bad argument");
    FR_CHECK_ARG(high > low && high - 1 - low < 0xffffffffll, "fr_sample_negatives_calls: range not below 2^32");
    FR_CHECK_ARG(used_indptr && used_items && n_users >= 1 && ws, "fr_sample_negatives_calls: used-set arguments");
    // With a workspace of fr_sample_negatives_calls_workspace_bytes(total, max_call) a sequence of many calls is resolved
    // speculatively (sample_calls_fast_kernel); `total` is not known to the host without a synchronisation, so the caller says
    // how much room there is and the size is read back from it: the layout for `total_hint` = the largest total the workspace
    // holds is what the kernel uses (it stops generating at its capacity, and falls back call by call if that is too little).
    static const bool fast_env = !(getenv("FAIRREC_SAMPLER_CALLS_FAST") && atoi(getenv("FAIRREC_SAMPLER_CALLS_FAST")) == 0);
    if (fast_env && n_calls >= 16 && high - 1 - low > 0) {
        // the largest `total` whose layout fits the workspace (bisection on the monotone size function)
        int64_t lo_t = 0, hi_t = (int64_t)1 << 30;
        while (lo_t < hi_t) {
            const int64_t mid = (lo_t + hi_t + 1) >> 1;
            if (fr_sample_negatives_calls_workspace_bytes(mid, max_call) <= ws_bytes) lo_t = mid;
            else hi_t = mid - 1;
        }
        if (lo_t >= n_calls) {      // (a workspace sized for the call-by-call form only: no room)
            size_t cap, nblk, off[9], bytes;
            calls_fast_layout(lo_t, max_call, cap, nblk, off, bytes);
            const uint32_t span = (uint32_t)(high - 1 - low);
            uint32_t mask = span;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
            char* w = (char*)ws;
            ProfScope prof(K_SAMPLE_NEG, (hipStream_t)stream_);
            FR_LAUNCH(prof, sample_calls_twist_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, (const uint32_t*)state, span, mask,
                      call_offsets, (long long)n_calls, (long long)cap, (uint32_t*)(w + off[2]), (long long)nblk,
                      (long long*)(w + off[5]));
            FR_LAUNCH(prof, sample_calls_temper_kernel<0>, dim3((unsigned)nblk), dim3(64), 0, (hipStream_t)stream_, span, mask,
                      (const uint32_t*)(w + off[2]), (long long*)(w + off[5]), (uint32_t*)(w + off[8]), (uint32_t*)(w + off[0]),
                      (uint32_t*)(w + off[1]));
            FR_LAUNCH(prof, sample_calls_temper_kernel<1>, dim3((unsigned)nblk), dim3(64), 0, (hipStream_t)stream_, span, mask,
                      (const uint32_t*)(w + off[2]), (long long*)(w + off[5]), (uint32_t*)(w + off[8]), (uint32_t*)(w + off[0]),
                      (uint32_t*)(w + off[1]));
            FR_LAUNCH(prof, sample_calls_hits_kernel, dim3((unsigned)((lo_t + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                      (long long)low, call_keys, call_offsets, (long long)n_calls, used_indptr, used_items, (long long)n_users,
                      (const uint32_t*)(w + off[0]), (const long long*)(w + off[5]), (uint32_t*)(w + off[6]), (int32_t*)(w + off[7]),
                      err_flag);
            FR_LAUNCH(prof, sample_calls_fast_kernel, dim3(1), dim3(SAMPLER_THREADS), 0, (hipStream_t)stream_, state, (long long)low,
                      span, mask, call_keys, call_offsets, (long long)n_calls, used_indptr, used_items, (long long)n_users, out,
                      (const uint32_t*)(w + off[0]), (const uint32_t*)(w + off[1]), (const uint32_t*)(w + off[2]),
                      (const long long*)(w + off[5]), (const uint32_t*)(w + off[6]), (const int32_t*)(w + off[7]),
                      (int32_t*)(w + off[3]), (int32_t*)(w + off[4]), err_flag);
            FR_CHECK_LAUNCH();
            return FR_OK;
        }
    }
    return sample_launch(state, low, high, nullptr, 1, 0, call_keys, call_offsets, n_calls, max_call, used_indptr, used_items,
                         n_users, out, nullptr, ws, ws_bytes, err_flag, (hipStream_t)stream_);
}
