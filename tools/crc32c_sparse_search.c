// Search for sparse polynomials T(x), deg T < 64, with x^D == T(x) (mod P) where
// P is the CRC-32C generator 0x1_1EDC6F41.  Any such T lets a D-bit wide
// accumulator row be folded onto the next row with popcount(T) shift/XOR terms
// instead of a table lookup per byte.  The candidates are r ^ clmul(P, g) for all
// 2^32 multipliers g, r = x^D mod P, walked in Gray-code order so each step is one
// XOR.  Usage: crc32c_sparse_search D [max_weight]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static const uint64_t P = 0x11EDC6F41ull;

static uint32_t xpow_mod(uint64_t n) {          // x^n mod P, normal (non-reflected) bit order
    uint64_t r = 1;                              // polynomial "1"
    for (uint64_t i = 0; i < n; i++) {
        r <<= 1;
        if (r & (1ull << 32)) r ^= P;
    }
    return (uint32_t)r;
}

int main(int argc, char **argv) {
    uint64_t D = argc > 1 ? strtoull(argv[1], 0, 0) : 8192;
    int maxw = argc > 2 ? atoi(argv[2]) : 9;
    uint64_t t = xpow_mod(D);
    int best = 99;
    // Gray code walk over g: g_i = i ^ (i >> 1); flipping bit ctz(i) at step i.
    for (uint64_t i = 1;; i++) {
        int w = __builtin_popcountll(t);
        if (w <= maxw && w <= best + 1) {
            if (w < best) best = w;
            int deg = 63 - __builtin_clzll(t | 1);
            printf("D=%llu weight=%d deg=%d T=0x%016llx terms:", (unsigned long long)D, w, deg,
                   (unsigned long long)t);
            for (int b = 63; b >= 0; b--) if (t >> b & 1) printf(" %d", b);
            printf("\n");
            fflush(stdout);
        }
        if (i >> 32) break;
        t ^= P << __builtin_ctzll(i);
    }
    return 0;
}
