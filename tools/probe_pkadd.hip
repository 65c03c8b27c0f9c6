// probe: semantics of ds_pk_add_bf16 on gfx950 (two bf16 adds per dword in LDS; which rounding?)
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_pkadd.hip -o tools/_bin/probe_pkadd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
__global__ void k(const unsigned* a, const unsigned* b, unsigned* out, int n) {
    __shared__ unsigned buf[64];
    if ((int)threadIdx.x < n) buf[threadIdx.x] = a[threadIdx.x];
    __syncthreads();
    if ((int)threadIdx.x < n) {
        unsigned addr = (unsigned)(uintptr_t)(&buf[threadIdx.x]);
        unsigned v = b[threadIdx.x];
        asm volatile("ds_pk_add_bf16 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory");
    }
    __syncthreads();
    if ((int)threadIdx.x < n) out[threadIdx.x] = buf[threadIdx.x];
}
static float bf(unsigned h) { unsigned u = h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    // (lo, hi) pairs: 1 + 2; 1 + 2^-8 (tie -> even = 1); (1 + 2^-7) + 2^-8 (tie -> even = 1 + 2^-6); -1.5 + 0.25; 3e38 + 3e38 (overflow); 1e-40-ish denormals
    unsigned ha[8] = {0x3f803f80u, 0x3f813f80u, 0xbfc03f80u, 0x7f617f61u, 0x00010001u, 0x3f803f81u, 0x42c842c8u, 0x3f803f80u};
    unsigned hb[8] = {0x40004000u, 0x3b803b80u, 0x3e803b00u, 0x7f617f61u, 0x00010001u, 0x3b813b7fu, 0x3f003f00u, 0xbf80bf80u};
    unsigned *a, *b, *o, ho[8];
    hipMalloc(&a, 32); hipMalloc(&b, 32); hipMalloc(&o, 32);
    hipMemcpy(a, ha, 32, hipMemcpyHostToDevice); hipMemcpy(b, hb, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o, 8);
    hipMemcpy(ho, o, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i)
        printf("lo: %04x + %04x = %04x (%g + %g = %g, exact %.9g) | hi: %04x + %04x = %04x (%g + %g = %g, exact %.9g)\n",
               ha[i] & 0xffff, hb[i] & 0xffff, ho[i] & 0xffff, bf(ha[i] & 0xffff), bf(hb[i] & 0xffff), bf(ho[i] & 0xffff), (double)bf(ha[i] & 0xffff) + bf(hb[i] & 0xffff),
               ha[i] >> 16, hb[i] >> 16, ho[i] >> 16, bf(ha[i] >> 16), bf(hb[i] >> 16), bf(ho[i] >> 16), (double)bf(ha[i] >> 16) + bf(hb[i] >> 16));
    return 0;
}
