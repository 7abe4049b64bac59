// Instruction cost probe for gfx950: ns per instruction for dependent and independent chains, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define N_ITERS 4096
#define DPP(old, src, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), ctrl, 0xf, 0xf, false))

template <int OP, int ILP>
__global__ void probe(float* out, float seed) {
    float v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < N_ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (OP == 0) v[i] = v[i] + 1.0001f;
                if (OP == 1) v[i] = __builtin_amdgcn_exp2f(v[i]) ;
                if (OP == 2) v[i] = __builtin_amdgcn_logf(v[i]);
                if (OP == 3) v[i] = DPP(v[i], v[i], 0x138);   // wave_shr:1
                if (OP == 4) v[i] = DPP(v[i], v[i], 0x111);   // row_shr:1
                if (OP == 5) v[i] = __builtin_fmaxf(__builtin_fmaxf(v[i], seed), 0.5f);  // max3
                if (OP == 6) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
                if (OP == 7) v[i] = __shfl_up(v[i], 1, 64);
                if (OP == 8) v[i] = __builtin_amdgcn_fmed3f(v[i], seed, 0.5f);
                if (OP == 9) v[i] = (v[i] > seed) ? v[i] : 0.25f * v[i];   // cmp + cndmask + mul
                if (OP == 10) v[i] = __builtin_amdgcn_rcpf(v[i]);
                if (OP == 11) v[i] = DPP(v[i], v[i], 0x130);   // wave_shl:1
                if (OP == 12) v[i] = DPP(v[i], v[i], 0x142);   // row_bcast15
                if (OP == 13) v[i] = __builtin_amdgcn_ldexpf(v[i], (int)threadIdx.x & 1);   // v_ldexp_f32
                if (OP == 14) v[i] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v[i]), (int)threadIdx.x) - 1);  // v_max_i32 + v_sub
                if (OP == 15) v[i] = (float)__builtin_amdgcn_frexp_expf(v[i]) + 3.0f;   // frexp_exp + cvt + add
                if (OP == 16) { out[(blockIdx.x * blockDim.x + threadIdx.x) + (it & 7) * 65536] = v[i]; v[i] += 1.0f; }   // store + add
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int ILP>
void run(const char* name, float* d, int blocks, int threads) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<OP, ILP><<<blocks, threads>>>(d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<OP, ILP><<<blocks, threads>>>(d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)N_ITERS * 8 * ILP;
    printf("%-14s ILP=%d blocks=%d thr=%d: %.2f ns/instr/wave  (%.3f ms)\n", name, ILP, blocks, threads, ms * 1e6 / n, ms);
}

int main() {
    float* d; hipMalloc(&d, 1 << 24);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("clockRate %d kHz, CUs %d\n", p.clockRate, p.multiProcessorCount);
#define ALL(OP, NAME) run<OP,1>(NAME, d, 256, 256); run<OP,4>(NAME, d, 256, 256); run<OP,4>(NAME, d, 256, 1024);
    ALL(0, "v_add") ALL(6, "v_fma") ALL(1, "v_exp") ALL(2, "v_log") ALL(10, "v_rcp") ALL(3, "dpp wave_shr") ALL(11, "dpp wave_shl") ALL(4, "dpp row_shr") ALL(12, "dpp row_bcast15")
    ALL(13, "v_ldexp") ALL(14, "imax+sub") ALL(15, "frexp+cvt+add") ALL(16, "store+add")
    ALL(5, "v_max3") ALL(8, "v_med3") ALL(9, "cmp+cndmask+mul") ALL(7, "shfl_up")
    // warm clock check: same probe after everything
    run<0,1>("v_add (again)", d, 256, 256);
    run<0,1>("v_add 32blk", d, 32, 64);
    run<1,1>("v_exp 32blk", d, 32, 64);
    return 0;
}
