// internal.h -- C++-side interfaces shared between translation units of libspeech_amd (not part of the C ABI).
#pragma once
#include "common.h"

struct SaGemmEpilogue {
    // addr(m, n) = (m / (m_inner * m_mid)) * s_outer + ((m / m_inner) % m_mid) * s_mid + (m % m_inner) + n * col_stride
    int m_inner, m_mid;   // m_inner == 0: plain ldc layout
    long s_outer, s_mid, col_stride;
    int relu;         // clamp the result at zero
};

ctcStatus_t sa_gemm_f32_impl(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, long lda,
                             const float* B, long ldb, float beta, float* C, long ldc, const float* bias,
                             const SaGemmEpilogue* ep, void* workspace, size_t workspace_bytes, hipStream_t stream);
extern "C" size_t sa_gemm_workspace_bytes(int M, int N, int K);
// nprob (<= 8) problems of identical shape in one launch; array arguments are host arrays of device pointers
struct SaGemmOpts {
    int no_split;           // never split K (no workspace, no reduce launch)
    int pad_lds;            // one block of this launch per CU (side-stream launches beside a persistent kernel)
    float* const* colsum;   // trans_a products: per problem, also write the column sums of A (K x M) -> [M]; or null
    unsigned xcc_mask;      // != 0: run only on the XCDs in the mask (persistent tile loop); needs tile_counter
    unsigned* tile_counter; // device word, zero before the launch
};
ctcStatus_t sa_gemm_f32_group_impl(int nprob, int trans_a, int trans_b, int M, int N, int K, float alpha,
                                   const float* const* A, long lda, const float* const* B, long ldb, float beta,
                                   float* const* C, long ldc, const float* const* bias, const SaGemmEpilogue* ep,
                                   void* workspace, size_t workspace_bytes, hipStream_t stream,
                                   const SaGemmOpts* opts = nullptr);
size_t sa_gemm_group_workspace_bytes(int nprob, int M, int N, int K);
extern "C" size_t sa_colsum_workspace_bytes(int M, int N);
