// standalone check of hop_loop on synthetic record data in an LDS ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
namespace nisshi {
typedef __attribute__((address_space(3))) uint32_t lds_u32;
#define HOP_ONLY
#include "hop_only.inc"
}
__global__ void k(const uint8_t *data, uint32_t len, uint32_t want, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[12 * 256 + 16];
    for (uint32_t i = threadIdx.x; i < 12 * 256 + 16; i += 64) ring[i] = ((const uint32_t *)data)[i];
    __syncthreads();
    uint32_t dbg[8];
    uint32_t ro = 3, left = len - 3, rem = want - 1, my_left = 0xdeadbeef;
    ro = __builtin_amdgcn_readfirstlane(ro); left = __builtin_amdgcn_readfirstlane(left); rem = __builtin_amdgcn_readfirstlane(rem);
    nisshi::hop_loop<2, 12288>((uint32_t)(uintptr_t)(nisshi::lds_u32 *)ring, 0, ro, left, rem, my_left, dbg);
    if (threadIdx.x == 0) for (int i = 0; i < 8; i++) out[67 + i] = dbg[i];
    out[threadIdx.x] = my_left;
    if (threadIdx.x == 0) { out[64] = ro; out[65] = left; out[66] = rem; }
}
int main() {
    std::vector<uint8_t> h(12 * 1024 + 64, 0xEE);
    uint32_t pos = 3, n = 0;
    std::vector<uint32_t> starts;
    uint32_t x = 12345;
    while (true) {
        x = x * 1664525u + 1013904223u;
        uint32_t body = 70 + (x >> 16) % 130;
        if (pos + 2 + body > 10240) break;
        uint32_t z = body * 2;
        h[pos] = (z & 0x7f) | 0x80; h[pos + 1] = z >> 7;
        starts.push_back(pos);
        pos += 2 + body; n++;
    }
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, h.size()); hipMalloc(&o, 75 * 4);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, 10240, 64, o);
    uint32_t r[75];
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    printf("records in data %u; ro=%u left=%u rem=%d\n", n, r[64], r[65], (int)r[66]);
    printf("lo=%08x t=%08x a=%u d0=%08x d1=%08x vlo=%08x va=%u vsh=%u; data %02x %02x %02x\n", r[67],r[68],r[69],r[70],r[71],r[72],r[73],r[74],h[3],h[4],h[5]);
    int bad = 0;
    for (int i = 0; i < 64 && i < (int)n; i++) {
        uint32_t got = 10240 - r[i];
        if (got != starts[i]) { if (bad < 5) printf("lane %d start %u want %u\n", i, got, starts[i]); bad++; }
    }
    printf("bad=%d\n", bad);
    return bad != 0;
}
