// xcd_gather_layout.hip -- the backward GRU kernel's per-step exchange, modelled closely: 32 blocks per XCD; thread
// (row = tid / 16, unit = tid % 16) of block r publishes three values (gates 0..2 of unit 16 r + unit, batch row `row`);
// then wave w, lane (i = lane % 16, g = lane / 16) gathers 24 x 16 bytes: its MFMA A-operand fragments of row i.
//   LAYOUT 0: the exchange buffer is the row-major (16, 1536) matrix the GEMMs want: a wave's load instruction touches
//             16 rows x 64 contiguous bytes, a wave's store instruction 4 rows x 64 bytes.
//   LAYOUT 1: tiled [k / 16][row][k % 16]: every producer store instruction and every consumer load instruction covers
//             one contiguous 1 KB tile.
//   WT 1: 4-byte agent-scope (write-through) stores, as the kernel; WT 0: plain stores.
// Between gather and publish the block idles DELAY ticks of 10 ns (the MFMAs and the gate math of the real step), so
// that a gather normally finds its data present.  Reported: time per step minus the delay, polling trips per step.
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_gather_layout xcd_gather_layout.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kXcds = 8;
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }
__device__ __forceinline__ bool sent(f4 v) {
    return (__builtin_bit_cast(unsigned, v.x) == 0x7fc00001u) | (__builtin_bit_cast(unsigned, v.y) == 0x7fc00001u) |
           (__builtin_bit_cast(unsigned, v.z) == 0x7fc00001u) | (__builtin_bit_cast(unsigned, v.w) == 0x7fc00001u);
}

template <int LAYOUT, int WT>
__global__ __launch_bounds__(256) void k(int* reg, float* buf, int steps, int delay, long long* out) {
    __shared__ float pad[24 * 1024];  // one block per CU
    __shared__ int s_x, s_r;
    pad[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) { s_x = xcc_id(); s_r = atomicAdd(&reg[s_x], 1); }
    __syncthreads();
    const int x = s_x, r = s_r, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
    if (r >= 32) { if (tid == 0) out[blockIdx.x] = -1; return; }
    const long per_step = (long)kXcds * 16 * 1536;
    long long errs = 0, trips = 0;
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {
        float* base = buf + (long)s * per_step + (long)x * 16 * 1536;
        const float val = (float)(s + 1);
        {
            const int row = tid >> 4, uj = tid & 15;
#pragma unroll
            for (int G = 0; G < 3; ++G) {
                float* p = LAYOUT == 0 ? base + row * 1536 + G * 512 + 16 * r + uj : base + (G * 32 + r) * 256 + row * 16 + uj;
                if (WT) __hip_atomic_store(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *p = val;
            }
        }
        __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        f4 a[24];
        for (int spins = 0; spins < (1 << 12); ++spins) {
            bool st = false;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                const int off = LAYOUT == 0 ? i * 1536 + w * 384 + 16 * u + 4 * g
                                            : ((u / 8) * 32 + w * 8 + (u % 8)) * 256 + i * 16 + 4 * g;
                a[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(res, off * 4, 0, 16));
            }
#pragma unroll
            for (int u = 0; u < 24; ++u) st |= sent(a[u]);
            ++trips;
            if (__builtin_amdgcn_ballot_w64(st) == 0) break;
        }
#pragma unroll
        for (int u = 0; u < 24; ++u) errs += a[u].x != val;
        const long long td = wall_clock64();
        while (wall_clock64() - td < delay) {}
        __syncthreads();  // the kernel's one barrier per step
    }
    const long long t1 = wall_clock64();
    __shared__ unsigned long long s_err;
    if (tid == 0) s_err = 0;
    __syncthreads();
    atomicAdd(&s_err, (unsigned long long)errs);
    __syncthreads();
    if (tid == 0) { out[blockIdx.x] = (t1 - t0); out[256 + blockIdx.x] = (long long)s_err; out[512 + blockIdx.x] = trips; }
}

template <int LAYOUT, int WT>
int run(const char* name, int steps, int delay, int* reg, float* buf, long long* out, size_t bytes) {
    CHECK(hipMemset(reg, 0, 64));
    CHECK(hipMemset(out, 0, 768 * 8));
    CHECK(hipMemsetD32((hipDeviceptr_t)buf, 0x7fc00001, bytes / 4));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<LAYOUT, WT>), dim3(256), dim3(256), 0, 0, reg, buf, steps, delay, out);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> h(768);
    CHECK(hipMemcpy(h.data(), out, 768 * 8, hipMemcpyDeviceToHost));
    long long mx = 0, errs = 0, trips = 0; int gone = 0;
    for (int b = 0; b < 256; ++b) { if (h[b] < 0) ++gone; else if (h[b] > mx) mx = h[b]; errs += h[256 + b]; trips += h[512 + b]; }
    printf("%-44s delay %4d ns: %6.2f us per step, %6.2f without the delay; %.2f trips per step (%d unplaced, %lld wrong)\n", name,
           delay * 10, mx * 0.01 / steps, mx * 0.01 / steps - delay * 0.01, trips / 256.0 / steps, gone, errs);
    return 0;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 300;
    int* reg; float* buf; long long* out;
    const size_t bytes = (size_t)steps * kXcds * 16 * 1536 * 4;
    CHECK(hipMalloc(&reg, 64)); CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 768 * 8));
    for (int delay : {0, 180}) {
        run<0, 1>("row-major, 4-byte write-through stores", steps, delay, reg, buf, out, bytes);
        run<0, 0>("row-major, plain 4-byte stores", steps, delay, reg, buf, out, bytes);
        run<1, 1>("tiled, 4-byte write-through stores", steps, delay, reg, buf, out, bytes);
        run<1, 0>("tiled, plain 4-byte stores", steps, delay, reg, buf, out, bytes);
    }
    return 0;
}
