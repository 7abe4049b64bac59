// cumask_probe.hip -- which XCDs / CUs does a stream created with hipExtStreamCreateWithCUMask run on?
// Build (on the GPU box or here): hipcc --offload-arch=gfx950 -O2 tools/ubench/cumask_probe.hip -o tools/ubench/cumask_probe
// Prints, per mask variant, the histogram of HW_REG_XCC_ID over 2048 workgroups and the number of distinct (xcc, se, cu).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <set>
#include <vector>

__global__ void census(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
        const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        out[blockIdx.x] = (xcc << 24) | (hwid & 0xffffff);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)spin) {}
    }
}

static void run(const char* name, const std::vector<uint32_t>& mask, bool use_mask) {
    hipStream_t s;
    if (use_mask) {
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
    } else {
        hipStreamCreate(&s);
    }
    const int n = 2048;
    unsigned* d;
    hipMalloc(&d, n * sizeof(unsigned));
    hipMemset(d, 0xff, n * sizeof(unsigned));
    hipLaunchKernelGGL(census, dim3(n), dim3(256), 0, s, d, 2000 /* 20 us */);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    int hist[16] = {0};
    std::set<unsigned> cus;
    for (int i = 0; i < n; ++i) {
        hist[(h[i] >> 24) & 15]++;
        const unsigned hw = h[i] & 0xffffff;
        cus.insert(((h[i] >> 24) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15));
    }
    printf("%-28s xcc hist:", name);
    for (int x = 0; x < 8; ++x) printf(" %4d", hist[x]);
    printf("   distinct (xcc,se,sh,cu): %zu\n", cus.size());
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    run("no mask", {}, false);
    std::vector<uint32_t> all(8, 0xffffffffu);
    run("all 256 bits", all, true);
    std::vector<uint32_t> hi(8, 0), lo(8, 0), il_hi(8, 0), il_lo(8, 0), one(8, 0);
    for (int b = 0; b < 256; ++b) {
        if (b >= 128) hi[b / 32] |= 1u << (b % 32); else lo[b / 32] |= 1u << (b % 32);
        if ((b % 8) >= 4) il_hi[b / 32] |= 1u << (b % 32); else il_lo[b / 32] |= 1u << (b % 32);
        if ((b % 8) == 5) one[b / 32] |= 1u << (b % 32);
    }
    run("bits 128..255", hi, true);
    run("bits 0..127", lo, true);
    run("bits b%8 >= 4", il_hi, true);
    run("bits b%8 < 4", il_lo, true);
    run("bits b%8 == 5", one, true);
    return 0;
}
