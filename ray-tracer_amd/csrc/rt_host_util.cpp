/*
 * rt_host_util.cpp -- integer helpers of the sampling path that run on the host at scene set-up:
 * the radical-inverse permutation (BuildQuasiIndex, Patterns.cpp:1697-1736) and the Marsaglia
 * multiply-with-carry generator (MonteCarlo.h:17-27).  Both are bit-exact integer work.
 */
#include <stdint.h>
#include <vector>
#include <algorithm>
#include "../../include/rt_amd.h"

/* The reference sorts 0..n-1 by digit 0 (least significant, base `prime`), then recursively by digit
   1 inside every bucket, and so on: the result lists the indices in ascending order of their
   digit-reversed value.  Here that key is computed once per index and the indices are sorted by it. */
extern "C" void rt_build_quasi_index(int *out, int prime, int n)
{
    if (n <= 0 || prime < 2) return;
    int ndigits = 0;
    for (int64_t m = (int64_t)n - 1; m > 0; m /= prime) ndigits++;
    if (ndigits == 0) ndigits = 1;
    std::vector<std::pair<int64_t, int> > key((size_t)n);
    for (int i = 0; i < n; i++) {
        int64_t k = 0;
        int v = i;
        for (int d = 0; d < ndigits; d++) { k = k * prime + (v % prime); v /= prime; }
        key[(size_t)i] = std::make_pair(k, i);
    }
    std::sort(key.begin(), key.end());
    for (int i = 0; i < n; i++) out[i] = key[(size_t)i].second;
}

extern "C" void rt_random_unsigned(uint32_t st[2], int n, uint32_t *out)
{
    uint32_t x = st[0], c = st[1];
    for (int i = 0; i < n; i++) {
        uint64_t v = (uint64_t)1965537969u * x + c;
        x = (uint32_t)v;
        c = (uint32_t)(v >> 32);
        out[i] = x;
    }
    st[0] = x; st[1] = c;
}


/* The multiply-with-carry generator of MonteCarlo.h:17-27, x' = (a x + c) mod 2^32, c' = (a x + c) >> 32 with
   a = 1965537969, is multiplication by a modulo m = a 2^32 - 1 of the combined state s = c 2^32 + x
   (a 2^32 = 1 mod m, so a s = a x + c 2^32 a = a x + c).  n draws ahead is therefore s a^n mod m. */
static const unsigned __int128 kMwcA = 1965537969u;
static inline uint64_t mwc_modulus() { return (uint64_t)((kMwcA << 32) - 1); }
static inline uint64_t mwc_mulmod(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % mwc_modulus()); }
uint64_t rt_mwc_power(uint64_t n)
{
    uint64_t r = 1, b = (uint64_t)kMwcA;
    for (; n; n >>= 1) { if (n & 1) r = mwc_mulmod(r, b); b = mwc_mulmod(b, b); }
    return r;
}
uint64_t rt_mwc_jump(uint64_t state, uint64_t power) { return mwc_mulmod(state, power); }

extern "C" void rt_random_skip(uint32_t state_xc[2], uint64_t ndraws)
{
    uint64_t s = ((uint64_t)state_xc[1] << 32) | state_xc[0];
    s = rt_mwc_jump(s, rt_mwc_power(ndraws));
    state_xc[0] = (uint32_t)s; state_xc[1] = (uint32_t)(s >> 32);
}
