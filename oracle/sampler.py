"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the reference's negative sampler and of the third-party arithmetic underneath it:

  * recbole/sampler/sampler.py:240-241   Sampler._uni_sampling  = np.random.randint(1, item_num, n)
  * recbole/sampler/sampler.py:145-197   AbstractSampler.sample_by_key_ids (draw, then re-draw the positions whose
                                         value is in the key's used-set, until none is left)
  * numpy 2.2.6 legacy RandomState (third party, pinned by this image): MT19937 seeded by an int
    (numpy/random/src/mt19937/mt19937.c mt19937_seed), `randint` on a 32-bit range = masked rejection on single
    32-bit outputs (numpy/random/src/distributions/distributions.c buffered_bounded_masked_uint32 with the 32-bit
    generator, called from random_bounded_uint64_fill because the default dtype is int64 and the range fits 32 bits).

Pinned by tests/test_oracle_sampler.py against numpy itself (same interpreter, here and on the GPU box) and against
golden vectors produced by running the reference's own Sampler (tests/golden/gen_sampler_golden.py).
"""
import numpy as np

N, M = 624, 397
UPPER, LOWER, MATRIX_A = 0x80000000, 0x7FFFFFFF, 0x9908B0DF


class MT19937:
    """State layout of numpy's legacy generator: key[624] + pos (pos == 624: the next draw twists first)."""

    def __init__(self, seed=None):
        self.key = np.zeros(N, dtype=np.uint32)
        self.pos = N
        if seed is not None:
            self.seed(seed)

    def seed(self, seed):
        s = int(seed) & 0xFFFFFFFF
        key = [0] * N
        for i in range(N):                      # mt19937_seed: Knuth's LCG, pos = 624
            key[i] = s
            s = (1812433253 * (s ^ (s >> 30)) + i + 1) & 0xFFFFFFFF
        self.key = np.array(key, dtype=np.uint32)
        self.pos = N

    # numpy interchange: np.random.get_state() / set_state()
    def get_state(self):
        return ("MT19937", self.key.copy(), int(self.pos), 0, 0.0)

    def set_state(self, state):
        self.key = np.asarray(state[1], dtype=np.uint32).copy()
        self.pos = int(state[2])

    def _twist(self):
        k = self.key.astype(np.uint64)
        new = np.empty(N, dtype=np.uint64)

        def f(a, b):
            y = (a & UPPER) | (b & LOWER)
            return (y >> 1) ^ np.where(y & 1, MATRIX_A, 0).astype(np.uint64)

        new[:N - M] = k[M:] ^ f(k[:N - M], k[1:N - M + 1])                       # i in [0, 227): old values only
        a, b = N - M, 2 * (N - M)
        new[a:b] = new[:N - M] ^ f(k[a:b], k[a + 1:b + 1])                        # [227, 454): new[i - 227]
        new[b:N - 1] = new[a:a + (N - 1 - b)] ^ f(k[b:N - 1], k[b + 1:N])         # [454, 623)
        y = (int(k[N - 1]) & UPPER) | (int(new[0]) & LOWER)
        new[N - 1] = int(new[M - 1]) ^ (y >> 1) ^ (MATRIX_A if y & 1 else 0)
        self.key = new.astype(np.uint32)
        self.pos = 0

    @staticmethod
    def temper(y):
        y = y.astype(np.uint64)
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return (y & 0xFFFFFFFF).astype(np.uint32)

    def raw_block(self):
        """The tempered outputs still unread in the current state block (twisting first if it is exhausted);
        the caller consumes a prefix of them with `advance`."""
        if self.pos >= N:
            self._twist()
        return self.temper(self.key[self.pos:])

    def advance(self, k):
        self.pos += int(k)

    def randint(self, low, high, n):
        """np.random.randint(low, high, n) for high - 1 - low < 2**32 - 1 (int64 output)."""
        rng = int(high) - 1 - int(low)
        assert 0 <= rng < 0xFFFFFFFF
        out = np.empty(n, dtype=np.int64)
        if rng == 0:
            out[:] = low
            return out
        mask = rng
        for s in (1, 2, 4, 8, 16):
            mask |= mask >> s
        got = 0
        while got < n:
            raw = self.raw_block() & np.uint32(mask)
            ok = np.nonzero(raw <= rng)[0]
            if len(ok) >= n - got:                  # the draw that yields the last value ends the consumption
                last = ok[n - got - 1]
                out[got:] = low + raw[ok[:n - got]].astype(np.int64)
                self.advance(last + 1)
                got = n
            else:
                out[got:got + len(ok)] = low + raw[ok].astype(np.int64)
                got += len(ok)
                self.advance(len(raw))
        return out


def sample_by_key_ids(rs: MT19937, key_ids, num, used_ids, item_num):
    """sampler.py:145-197 with distribution 'uniform' and no group labels; used_ids[key] is a set of item ids.
    Both branches of the reference (all keys equal / mixed keys) consume the stream identically: draw total_num
    values, then re-draw exactly the positions still colliding, in ascending position order."""
    key_ids = np.tile(np.asarray(key_ids, dtype=np.int64), num)
    total = len(key_ids)
    value_ids = np.zeros(total, dtype=np.int64)
    check = np.arange(total)
    while len(check) > 0:
        value_ids[check] = rs.randint(1, item_num, len(check))
        check = np.array([i for i in check if int(value_ids[i]) in used_ids[key_ids[i]]], dtype=np.int64)
    return value_ids
