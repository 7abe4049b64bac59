// xcd_split_exchange.hip -- which all-to-all is cheaper inside an XCD, per recurrence step of the backward GRU kernel?
//   A (what gru_bwd_persist_kernel does): every block publishes its 16 x 48 slice (3 KB); every block then reads the
//     WHOLE 16 x 1536 row block (98 KB) -- 3.1 MB per XCD and step through L2 -> L1.
//   B (input-stationary split): every block publishes a 16 x 512 partial (32 KB); every block then reads, from each of
//     the 32 partials, the 16 x 16 piece of its own units (32 x 1 KB) -- 1 MB written + 1 MB read per XCD and step.
// Flag-less as the kernel: buffers pre-filled with a NaN sentinel, fresh per step (no reuse), write-through stores
// (WT=1) or plain stores (WT=0), sc1 16-byte loads, a wave re-reads its share until it holds no sentinel.
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_split_exchange xcd_split_exchange.hip ; run: ./xcd_split_exchange [steps]
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

// buf layout per step: A: [xcd][16 rows][1536]            (block r writes columns 48 r .. 48 r + 47 of every row)
//                      B: [xcd][32 producers][16 rows][512] (consumer c reads columns 16 c .. 16 c + 15 of every producer row)
template <int MODE, int WT, int LD>
__global__ __launch_bounds__(256) void k(int* reg, float* buf, int steps, long long* out) {
    __shared__ float pad[24 * 1024];  // one block per CU
    __shared__ int s_x, s_r;
    pad[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) { s_x = xcc_id(); s_r = atomicAdd(&reg[s_x], 1); }
    __syncthreads();
    const int x = s_x, r = s_r, tid = threadIdx.x;
    if (r >= 32) { if (tid == 0) out[blockIdx.x] = -1; return; }
    // (a block beyond the 32nd of its XCD would break the pattern: reported as unplaced)
    const long per_step = (long)kXcds * (MODE == 0 ? 16 * 1536 : 32 * 16 * 512);
    long long errs = 0;
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {
        float* base = buf + (long)s * per_step + (long)x * (MODE == 0 ? 16 * 1536 : 32 * 16 * 512);
        const float val = (float)(s + 1);
        if (MODE == 0) {  // 16 rows x 48 floats: 192 float4, one per thread (threads 0..191)
            if (tid < 192) {
                const int row = tid / 12, q = tid % 12;
                float* p = base + row * 1536 + 48 * r + 4 * q;
                const f4 v = {val, val, val, val};
                if (WT == 1) { __hip_atomic_store(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 1, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                          __hip_atomic_store(p + 2, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 3, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else if (WT == 2) { volatile float* q2 = p; q2[0] = val; q2[1] = val; q2[2] = val; q2[3] = val; }  // plain 4-byte stores
                else if (WT == 3) {  // one 16-byte store, sc1 (write-through)
                    __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 16, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), wres, 0, 0, 16);
                }
                else if (WT == 4) {  // one 16-byte store, sc0 sc1
                    __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 16, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), wres, 0, 0, 17);
                }
                else *reinterpret_cast<f4*>(p) = v;
            }
        } else {  // 16 rows x 512 floats = 2048 float4: 8 per thread
            float* mine = base + (long)r * 16 * 512;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float* p = mine + (tid + 256 * u) * 4;
                if (WT) { __hip_atomic_store(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 1, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                          __hip_atomic_store(p + 2, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 3, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else { const f4 v = {val, val, val, val}; *reinterpret_cast<f4*>(p) = v; }
            }
        }
        __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        if (MODE == 0) {  // 16 x 1536 floats = 6144 float4: 24 per thread
            f4 a[24];
            for (int spins = 0; spins < (1 << 12); ++spins) {
                bool st = false;
                asm volatile("" ::: "memory");  // the loads must be re-issued every trip
                if (LD == 2) asm volatile("buffer_inv sc1" ::: "memory");       // drop the CU's L1 copies, then plain loads
#pragma unroll
                for (int u = 0; u < 24; ++u)
                    a[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                     res, (tid + 256 * (LD >= 4 ? (u + (LD == 5 ? 3 * r : r)) % 24 : u)) * 16, 0,
                                     LD == 0 || LD >= 4 ? 16 : LD == 1 ? 1 : LD == 3 ? 2 : 0));  // LD 4 / 5: start rotated by the block's rank
#pragma unroll
                for (int u = 0; u < 24; ++u) st |= sent(a[u]);
                if (__builtin_amdgcn_ballot_w64(st) == 0) break;
            }
#pragma unroll
            for (int u = 0; u < 24; ++u) errs += a[u].x != val;
        } else {  // 32 producers x 16 rows x 16 floats: 2048 float4... no: 32 x 16 x 4 float4 = 2048 float4: 8 per thread
            f4 a[8];
            for (int spins = 0; spins < (1 << 16); ++spins) {
                bool st = false;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = tid + 256 * u;           // (producer, row, quad) = (i / 64, (i / 4) % 16, i % 4)
                    const int off = ((i >> 6) * 16 * 512 + ((i >> 2) & 15) * 512 + 16 * r + 4 * (i & 3)) * 4;
                    a[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(res, off, 0, 16));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) st |= sent(a[u]);
                if (__builtin_amdgcn_ballot_w64(st) == 0) break;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) errs += a[u].x != val;
        }
        __syncthreads();  // the kernel's one barrier per step
    }
    const long long t1 = wall_clock64();
    __shared__ unsigned long long s_err;
    if (tid == 0) s_err = 0;
    __syncthreads();
    atomicAdd(&s_err, (unsigned long long)errs);
    __syncthreads();
    if (tid == 0) { out[blockIdx.x] = (t1 - t0); out[256 + blockIdx.x] = (long long)s_err; }
}

template <int MODE, int WT, int LD = 0>
int run(const char* name, int steps, int* reg, float* buf, long long* out, size_t bytes) {
    CHECK(hipMemset(reg, 0, 64));
    CHECK(hipMemset(out, 0, 512 * 8));
    CHECK(hipMemsetD32((hipDeviceptr_t)buf, 0x7fc00001, bytes / 4));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<MODE, WT, LD>), dim3(256), dim3(256), 0, 0, reg, buf, steps, out);
    CHECK(hipDeviceSynchronize());
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) printf("launch error: %s\n", hipGetErrorString(le));
    std::vector<long long> h(512);
    CHECK(hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost));
    long long mx = 0, errs = 0; int gone = 0;
    for (int i = 0; i < 256; ++i) { if (h[i] < 0) ++gone; else if (h[i] > mx) mx = h[i]; errs += h[256 + i]; }
    printf("%-50s %7.2f us per step (%d steps; slowest block; %d blocks unplaced; %lld wrong values)\n", name, mx * 0.01 / steps, steps, gone, errs);
    return 0;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 300;
    int* reg; float* buf; long long* out;
    const size_t bytes = (size_t)steps * kXcds * 32 * 16 * 512 * 4;
    CHECK(hipMalloc(&reg, 64)); CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 512 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0, 4>("A, plain 16-byte stores, sc1 loads, order rotated by rank", steps, reg, buf, out, bytes);
        run<0, 0, 5>("A, plain 16-byte stores, sc1 loads, order rotated by 3 rank", steps, reg, buf, out, bytes);
        run<0, 1, 4>("A, 4-byte write-through stores, sc1 loads, rotated by rank", steps, reg, buf, out, bytes);
        run<0, 0, 1>("A, plain 16-byte stores, sc0 loads", steps, reg, buf, out, bytes);
        run<0, 0, 3>("A, plain 16-byte stores, nt loads", steps, reg, buf, out, bytes);
        run<0, 1, 1>("A, 4-byte write-through stores, sc0 loads", steps, reg, buf, out, bytes);
        run<0, 1>("A gather 98 KB / block, write-through", steps, reg, buf, out, bytes);
        run<0, 0>("A gather 98 KB / block, plain 16-byte stores", steps, reg, buf, out, bytes);
        run<1, 0>("B partials 32 KB out + 32 KB in, plain stores", steps, reg, buf, out, bytes);
    }
    return 0;
}
