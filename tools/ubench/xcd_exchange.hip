// xcd_exchange.hip -- feasibility probe for an XCD-local persistent recurrence (DESIGN.md 3.3 "next lever").
//
// Question: what does one step of "publish my slice of h, wait for the other CUs of MY XCD, read the whole h" cost when
// the exchange never leaves the XCD's shared L2?  (The chip-wide hand-off with sc1 write-through stores / sc1 loads
// measured ~10 us per step: as much as a kernel boundary.)
//
// 256 workgroups (one per CU, forced by 100 KB of LDS).  Each reads its XCC id from the hardware register and joins that
// XCD's group with an L2-executed (workgroup-scope) atomic; nothing below assumes WHICH workgroups share an XCD.
// Per step a workgroup stores 64 floats per row x 16 rows of "h" with plain stores (write-through L1 -> the XCD's L2),
// waits for them, bumps the group's counter (workgroup-scope RMW: executes in the local L2), polls the counter with an
// agent-scope (L1-bypassing) load, invalidates its L1 (the acquire half of an agent fence: buffer_inv sc1) and reads all members' slices with
// plain loads (L2 hits).  Every value read is checked against the expected (step, producer) pattern.
//
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_exchange xcd_exchange.hip ; run: ./xcd_exchange [steps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kXcds = 8, kMaxMembers = 64, kSlice = 1024;  // floats published per workgroup per step (4 KB)

struct Shared {
    int members[kXcds];              // registration counters
    int ready;                       // workgroups registered (chip-wide, agent scope, once)
    int step_ctr[kXcds][16];         // one counter per XCD, padded to its own 64-byte line
    int rank_block[kXcds][kMaxMembers];
};

__device__ __forceinline__ int xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
}

// mode 0: full step; 1: publish + barrier only; 2: + L1 invalidate; 3: + read ONE member's slice instead of all
__global__ __launch_bounds__(256) void exchange_kernel(Shared* S, float* buf /* [2][kXcds][kMaxMembers][kSlice] */,
                                                       int steps, long long* out /* per block: ticks, errors, xcc, n */,
                                                       int mode) {
    __shared__ float lds_pad[24 * 1024];  // 96 KB: one workgroup per CU
    __shared__ int s_rank, s_xcc, s_n, s_timeout;
    lds_pad[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) {
        s_timeout = 0;
        const int x = xcc_id();
        s_xcc = x;
        s_rank = __hip_atomic_fetch_add(&S->members[x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&S->ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&S->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x &&
               ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
        s_n = __hip_atomic_load(&S->members[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int x = s_xcc, rank = s_rank, n = s_n;
    long long errors = 0;
    const long long t0 = wall_clock64();
    bool timeout = false;
    for (int s = 0; s < steps && !timeout; ++s) {
        float* mine = buf + (((long)(s & 1) * kXcds + x) * kMaxMembers + rank) * kSlice;
        for (int i = threadIdx.x; i < kSlice; i += 256) mine[i] = (float)(s * 131 + rank * 7 + (i & 3));
        __builtin_amdgcn_s_waitcnt(0);  // stores have left for L2
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&S->step_ctr[x][0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int spins = 0;
            // poll with an agent-scope load (sc1: bypasses this CU's L1, served by the XCD's L2).  A workgroup-scope
            // load (sc0) -- which is also what hipcc turns a workgroup-scope fetch_add(0) into -- keeps hitting the
            // stale L1 line and never sees the other CUs' increments (measured: every workgroup timed out).
            while (__hip_atomic_load(&S->step_ctr[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n * (s + 1)) {
                if (++spins > (1 << 20)) { s_timeout = 1; break; }
            }
        }
        __syncthreads();
        timeout = s_timeout != 0;
        if (mode == 1) continue;
        if (mode == 4) {
            // no L1 invalidate: read the slices with sc1 (L1-bypassing) 16-byte buffer loads, 8 in flight
            typedef float f4 __attribute__((ext_vector_type(4)));
            __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(buf + ((long)(s & 1) * kXcds + x) * kMaxMembers * kSlice), 0, 0x7fffffff, 0x00020000);
            for (int m0 = 0; m0 < n; m0 += 8) {
                f4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = min(m0 + u, n - 1);
                    v[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      res, (m * kSlice + threadIdx.x * 4) * 4, 0, 16));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = min(m0 + u, n - 1);
                    const float want = (float)(s * 131 + m * 7);
                    if (v[u].x != want || v[u].y != want + 1 || v[u].z != want + 2 || v[u].w != want + 3) ++errors;
                }
            }
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1: drop this CU's L1 lines
        if (mode == 2) continue;
        // read every member's slice (n x 4 KB), 16-byte loads
        const float4* all = reinterpret_cast<const float4*>(buf + ((long)(s & 1) * kXcds + x) * kMaxMembers * kSlice);
        for (int m = 0; m < (mode == 3 ? 1 : n); ++m) {
            const float4 v = all[(long)m * (kSlice / 4) + threadIdx.x];  // 256 threads x 16 B = the 4 KB slice
            const float want = (float)(s * 131 + m * 7);
            if (v.x != want || v.y != want + 1 || v.z != want + 2 || v.w != want + 3) ++errors;
        }
    }
    const long long t1 = wall_clock64();
    // reduce errors over the block
    __shared__ long long s_err;
    if (threadIdx.x == 0) s_err = 0;
    __syncthreads();
    atomicAdd((unsigned long long*)&s_err, (unsigned long long)errors);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0;
        out[blockIdx.x * 4 + 1] = s_err + (timeout ? (1LL << 40) : 0);
        out[blockIdx.x * 4 + 2] = x;
        out[blockIdx.x * 4 + 3] = n;
    }
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount;
    Shared* S;
    float* buf;
    long long* out;
    CHECK(hipMalloc(&S, sizeof(Shared)));
    CHECK(hipMemset(S, 0, sizeof(Shared)));
    CHECK(hipMalloc(&buf, sizeof(float) * 2 * kXcds * kMaxMembers * kSlice));
    CHECK(hipMalloc(&out, sizeof(long long) * 4 * blocks));
  for (int mode = 0; mode < 5; ++mode) {
    CHECK(hipMemset(S, 0, sizeof(Shared)));
    hipLaunchKernelGGL(exchange_kernel, dim3(blocks), dim3(256), 0, 0, S, buf, steps, out, mode);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> h(4 * blocks);
    CHECK(hipMemcpy(h.data(), out, sizeof(long long) * 4 * blocks, hipMemcpyDeviceToHost));
    int per_xcd[16] = {0};
    long long errs = 0, max_ticks = 0;
    for (int b = 0; b < blocks; ++b) {
        per_xcd[h[4 * b + 2] & 15]++;
        errs += h[4 * b + 1];
        if (h[4 * b] > max_ticks) max_ticks = h[4 * b];
    }
    printf("CUs %d, steps %d\nworkgroups per XCC id:", blocks, steps);
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcd[x]);
    printf("\nblock i -> xcc:");
    for (int b = 0; b < 24; ++b) printf(" %lld", h[4 * b + 2]);
    printf(" ...\nerrors (wrong values read; 2^40 = timeout): %lld\n", errs);
    printf("mode %d per-step latency (0: publish 4 KB, XCD barrier, L1 invalidate, read n x 4 KB; 1: barrier only; "
           "2: + invalidate; 3: + read one slice; 4: barrier + sc1 loads of all slices, no invalidate): %.3f us (100 MHz wall clock)\n", mode, max_ticks / 100.0 / steps);
    if (errs) return 2;
  }
    return 0;
}
