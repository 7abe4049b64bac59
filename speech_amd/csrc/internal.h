// internal.h -- C++-side interfaces shared between translation units of libspeech_amd (not part of the C ABI).
#pragma once
#include "common.h"
#include "dropout.h"

// options.hip: the library's run-time options (include/speech_amd.h: sa_set_option).  The order is the table's.
enum SaOpt {
    SA_OPT_CTC_PROB, SA_OPT_CTC_WIDE, SA_OPT_CTC_DIRECT, SA_OPT_CTC_DBG, SA_OPT_GEMM_EXACT, SA_OPT_GEMM_THIN,
    SA_OPT_GRU_PERSIST, SA_OPT_GRU_SPIN_LIMIT, SA_OPT_GRU_FAULT, SA_OPT_GRU_FWD_CHUNKS, SA_OPT_GRU_FUSE_DX, SA_OPT_GRU_TILED,
    SA_OPT_GRU_FUSED, SA_OPT_GRU_TIMING, SA_OPT_GRU_SHARED_PACK, SA_OPT_GRU_PACK_IN_KERNEL, SA_OPT_GRU_BWD_ONE,
    SA_OPT_GRU_FWD_REPORT, SA_OPT_GRU_FWD_PLANES, SA_OPT_GRU_EXP, SA_OPT_S2S_KERNELS, SA_OPT_CONV_DX_PHASES, SA_OPT_COUNT
};
long sa_opt(SaOpt id);

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
    int exact = 0;          // != 0: the f32-input MFMA kernel whatever the shape (no packed operands, no workspace for them)
    float* const* colsum;   // trans_a products: per problem, also write the column sums of A (K x M) -> [M]; or null
    unsigned xcc_mask;      // != 0: run only on the XCDs in the mask (persistent tile loop); needs tile_counter
    unsigned* tile_counter; // device word, zero before the launch
    // XCD-filtered launches assume the dispatcher spreads their blocks over the XCDs (block b -> XCD b % 8) closely enough
    // that the allowed XCDs receive one block per tile.  That is an observation, not a contract: behind every filtered
    // launch a one-thread kernel on the same stream ORs code 4 into this word (the library's sticky health word, see
    // sa_gru_health_flag) when fewer than all tiles were drawn -- the step's optimiser update is then skipped and the
    // caller replays it without the filtered path, exactly as for a failed persistent-kernel hand-off.
    unsigned* err_word = nullptr;
    // sa_gemm_pk_group only: dropout on the output -- the element written at address c is multiplied by the mask factor of
    // index c - drop_base of mask stream drop_stream (the output is a window of the masked tensor).  No split-K then.
    const SaDrop* drop = nullptr;
    unsigned drop_stream = 0;
    const float* drop_base = nullptr;
    int b_kb_stride = 0;    // sa_gemm_pk_group only: k-tiles between B's packed row blocks (0: ceil(K / 16))
};
ctcStatus_t sa_gemm_f32_group_impl(int nprob, int trans_a, int trans_b, int M, int N, int K, float alpha,
                                   const float* const* A, long lda, const float* const* B, long ldb, float beta,
                                   float* const* C, long ldc, const float* const* bias, const SaGemmEpilogue* ep,
                                   void* workspace, size_t workspace_bytes, hipStream_t stream,
                                   const SaGemmOpts* opts = nullptr);
size_t sa_gemm_group_workspace_bytes(int nprob, int M, int N, int K);
extern "C" size_t sa_colsum_workspace_bytes(int M, int N);

// dropout.hip: the stand-alone forms of the mask (dropout.h)
ctcStatus_t sa_dropout_nchw_strided_impl(float* y, int B, int O, int To, int Fo, long ys_b, long ys_c, long ys_t,
                                         const SaDrop& d, unsigned stream_id, hipStream_t stream);
// out[i] = (in ? in[i] : 1) * factor(idx0 + i); in == out allowed
ctcStatus_t sa_dropout_apply_impl(const float* in, float* out, size_t n, size_t idx0, const SaDrop& d,
                                  unsigned stream_id, hipStream_t stream);

// Packed split-bf16 operands as objects (gemm_f32.hip, "split-bf16 GEMM on packed operands"): a caller that multiplies the
// same matrix several times packs it once.  Layout: tiles of 128 rows x 16 k x 3 bf16 planes; R rows, reduction length K.
size_t sa_pk_operand_bytes(int R, int K);
int sa_pk_rowsum_parts(int K);                    // partial row sums per row written by sa_pk_pack(cs_part)
bool sa_pk_enabled(int M, int N, int K, int nprob);  // the size / SA_GEMM_EXACT rule of sa_gemm_f32_group_impl
// nprob matrices of R logical rows x K.  kcontig: src[p][r * ld + k]; else src[p][k * ld + r], and rows r >= R_lo are taken
// from src_hi[p][k * ld + (r - R_lo)] when src_hi != NULL.  cs_part (m-contiguous form only, or NULL):
// [nprob][sa_pk_rowsum_parts(K)][ceil128(R)] partial row sums (fold them with sa_pk_rowsum_fold).
ctcStatus_t sa_pk_pack(int nprob, const float* const* src, const float* const* src_hi, int R_lo, long ld, int R, int K,
                       int kcontig, char* dst, size_t dst_stride, float* cs_part, hipStream_t stream, int kb_stride = 0,
                       int kb_lead = 0, int zero_lead = 0);
// kb_lead > 0 (m-contiguous form): the matrix starts kb_lead tiles into each row block (kb_stride >= ceil(K / 16) + kb_lead);
// zero_lead: the tiles in front of it are written as zeros.  A product that reads the operand at its base then sees the
// matrix shifted by kb_lead k-tiles behind zeros, one that reads it at base + kb_lead tiles sees the matrix itself.
// kb_stride > 0: the packed row blocks are kb_stride k-tiles apart (default ceil(K / 16)) -- with dst_stride = a whole number
// of tiles, the nprob matrices land side by side along k inside ONE operand whose reduction length is the sum of theirs.
ctcStatus_t sa_pk_rowsum_fold(int nprob, const float* cs_part, int nparts, int Rpad, int M, int split, int jump,
                              float* const* out, float beta, hipStream_t stream, int split2 = 0, int jump2 = 0,
                              float* const* out2 = nullptr);  // out2: a second row map of the same M rows, same launch
size_t sa_gemm_pk_group_workspace_bytes(int nprob, int M, int N, int K);
ctcStatus_t sa_gemm_pk_group(int nprob, int M, int N, int K, const char* const* Apk, int a_split, int a_jump,
                             unsigned a_jump_probs, const char* const* Bpk, float beta, float* const* C, long ldc, void* workspace,
                             size_t workspace_bytes, hipStream_t stream, const SaGemmOpts* opts = nullptr);
// (opts: only xcc_mask / tile_counter / err_word -- the XCD-filtered form of the launch --, drop* and b_kb_stride are read)
