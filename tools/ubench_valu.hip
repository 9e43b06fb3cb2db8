// ubench_valu.hip -- wave64 integer VALU / DPP / LDS-atomic issue rates on gfx950, to price the tile kernels'
// instruction streams (is a wave64 integer VALU op 2 or 4 SIMD cycles?).
//
// Build + run on the MI355X box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_valu tools/ubench_valu.hip && /tmp/ubench_valu
// Every kernel runs `iters` iterations of an unrolled block of 64 independent-enough ops per lane (8 chains of 8),
// grid = 256 CUs x `waves_per_simd` x 4 SIMDs, so the SIMDs are saturated and latency is hidden by the other chains.
// Reported: cycles per wave-instruction per SIMD = (time x clock x 1024 SIMDs) / (waves x instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { OP_OR, OP_SUB, OP_LSHL, OP_LSHR, OP_CND_VCC, OP_CND_SGPR, OP_MAX, OP_CMP, OP_AND_OR, OP_LSHL_OR, OP_BCNT, OP_MOV, OP_MAD24, OP_DPP_NONOP, OP_MOV_DPP, OP_SDWA, OP_ADD, OP_AND, OP_MIN, OP_BFE, OP_XOR, OP_CNDMASK, OP_LSHL_ADD, OP_ADD3, OP_DPP_ADD, OP_MAX_DPP, OP_MBCNT, OP_FMA, OP_CMP_ADDC,
       OP_SUB_U16PK, OP_LDS_ADD_SPREAD, OP_LDS_ADD_SAME4, OP_LDS_ADD_SAME16, OP_LDS_READ, OP_LDS_WRITE, OP_READLANE, OP_N };
static const char *op_name[OP_N] = {"v_or_b32", "v_sub_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_cndmask_b32 (vcc)", "v_cndmask_b32 e64 (sgpr mask)", "v_max_u32", "v_cmp_lt_u32 (to vcc)",
                                     "v_and_or_b32", "v_lshl_or_b32", "v_bcnt_u32_b32", "v_mov_b32", "v_mad_u32_u24", "v_add_u32 dpp row_shr:1 (8 chains, no nop)", "v_mov_b32 dpp row_shr:1 (8 chains)", "v_add_u32 sdwa (WORD_1)",
                                     "v_add_u32", "v_and_b32", "v_min_u32", "v_bfe_u32", "v_xor_b32", "v_cndmask_b32", "v_lshl_add_u32", "v_add3_u32",
                                     "v_add_u32 dpp row_shr:1", "v_max_u32 dpp row_shr:1", "v_mbcnt lo+hi (2 ops)", "v_fma_f32", "v_cmp_lt+v_addc (2 ops)",
                                     "v_pk_sub_u16", "ds_add_u32 64 distinct banks/addresses", "ds_add_u32 4 lanes per address", "ds_add_u32 16 lanes per address",
                                     "ds_read_b32", "ds_write_b32", "v_readlane_b32 (to SGPR)"};

template <int OP>
__global__ __launch_bounds__(256) void k_ubench(uint32_t *out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[4096];
    const uint32_t tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = 0;
    __syncthreads();
    uint32_t a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = seed * (c + 1) + tid;
    float f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = (float)(seed + c);
    const uint32_t b = seed | 1u;
    uint32_t sacc = 0;
    const unsigned long long mask64 = 0x5555aaaa3333ccccull ^ seed;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(tid), "v"(b) : "vcc");   // a defined VCC for the cndmask rows
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if constexpr (OP == OP_OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_SUB) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[c]));
                else if constexpr (OP == OP_LSHR) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[c]));
                else if constexpr (OP == OP_CND_VCC) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_CND_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "s"(mask64));
                else if constexpr (OP == OP_MAX) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_CMP) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[c]), "v"(b) : "vcc");
                else if constexpr (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_BCNT) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_DPP_NONOP) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[c]));
                else if constexpr (OP == OP_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[c]));
                else if constexpr (OP == OP_SDWA) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_MIN) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a[c]));
                else if constexpr (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b) : );
                else if constexpr (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_DPP_ADD) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(a[c]));
                else if constexpr (OP == OP_MAX_DPP) asm volatile("v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(a[c]));
                else if constexpr (OP == OP_MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0\n v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(a[c]) : "s"(b));
                else if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[c]));
                else if constexpr (OP == OP_CMP_ADDC) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[c]) : "v"(b) : "vcc");
                else if constexpr (OP == OP_SUB_U16PK) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if constexpr (OP == OP_LDS_ADD_SPREAD) atomicAdd(&lds[(tid + 64u * c) & 4095u], 1u);
                else if constexpr (OP == OP_LDS_ADD_SAME4) atomicAdd(&lds[((tid >> 2) + 64u * c) & 4095u], 1u);
                else if constexpr (OP == OP_LDS_ADD_SAME16) atomicAdd(&lds[((tid >> 4) + 64u * c) & 4095u], 1u);
                else if constexpr (OP == OP_LDS_READ) { uint32_t x; asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"((tid * 4u + 256u * c) & 16383u)); a[c] = x; }
                else if constexpr (OP == OP_LDS_WRITE) asm volatile("ds_write_b32 %0, %1" : : "v"((tid * 4u + 256u * c) & 16383u), "v"(a[c]) : "memory");
                else if constexpr (OP == OP_READLANE) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(a[c])); sacc += s; }
            }
        }
        if constexpr (OP == OP_LDS_READ) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    uint32_t r = sacc;
#pragma unroll
    for (int c = 0; c < 8; ++c) r ^= a[c] ^ (uint32_t)f[c];
    if (r == 0x12345u) out[tid] = r + lds[tid];
}

template <int OP>
static void run(int waves_per_simd, uint32_t *d_out, double clock_ghz) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (waves_per_simd x 4 SIMDs / 4 waves per block)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_ubench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 10, 12345u);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_ubench<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double per_block_ops = (OP == OP_MBCNT || OP == OP_CMP_ADDC) ? 128.0 : 64.0;
    const double wave_instr_per_simd = (double)waves_per_simd * iters * per_block_ops;   // per SIMD
    const double cycles = best * 1e-3 * clock_ghz * 1e9;
    printf("%-40s waves/SIMD %d  %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", op_name[OP], waves_per_simd, best,
           cycles / wave_instr_per_simd, clock_ghz);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int OP> static void run_all(uint32_t *d, double ghz) { run<OP>(1, d, ghz); run<OP>(2, d, ghz); run<OP>(4, d, ghz); }

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, clockRate %.3f GHz (cycles below assume that clock; DVFS may run lower)\n", p.gcnArchName, p.multiProcessorCount, ghz);
    uint32_t *d; CK(hipMalloc(&d, 4096));
    run_all<OP_OR>(d, ghz); run_all<OP_SUB>(d, ghz); run_all<OP_LSHL>(d, ghz); run_all<OP_LSHR>(d, ghz); run_all<OP_CND_VCC>(d, ghz); run_all<OP_CND_SGPR>(d, ghz);
    run_all<OP_MAX>(d, ghz); run_all<OP_CMP>(d, ghz); run_all<OP_AND_OR>(d, ghz); run_all<OP_LSHL_OR>(d, ghz); run_all<OP_BCNT>(d, ghz); run_all<OP_MOV>(d, ghz);
    run_all<OP_MAD24>(d, ghz); run_all<OP_DPP_NONOP>(d, ghz); run_all<OP_MOV_DPP>(d, ghz); run_all<OP_SDWA>(d, ghz);
    run_all<OP_ADD>(d, ghz); run_all<OP_AND>(d, ghz); run_all<OP_MIN>(d, ghz); run_all<OP_BFE>(d, ghz); run_all<OP_XOR>(d, ghz);
    run_all<OP_CNDMASK>(d, ghz); run_all<OP_LSHL_ADD>(d, ghz); run_all<OP_ADD3>(d, ghz); run_all<OP_DPP_ADD>(d, ghz); run_all<OP_MAX_DPP>(d, ghz);
    run_all<OP_MBCNT>(d, ghz); run_all<OP_FMA>(d, ghz); run_all<OP_CMP_ADDC>(d, ghz); run_all<OP_SUB_U16PK>(d, ghz);
    run_all<OP_LDS_ADD_SPREAD>(d, ghz); run_all<OP_LDS_ADD_SAME4>(d, ghz); run_all<OP_LDS_ADD_SAME16>(d, ghz); run_all<OP_LDS_READ>(d, ghz);
    run_all<OP_LDS_WRITE>(d, ghz); run_all<OP_READLANE>(d, ghz);
    CK(hipFree(d));
    return 0;
}
