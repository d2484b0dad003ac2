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
