// Issue-rate microbenchmark for gfx950: how many cycles does one SIMD need per wave64 instruction of a given kind,
// with 1 / 2 / 4 / 8 resident waves per SIMD?  (tools/microbench; not part of the product.)
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2000;   // loop trips
constexpr int UNR = 8;        // asm blocks per trip, 8 independent instructions each

#define OP8(fmt) \
    asm volatile(fmt("%0") "\n" fmt("%1") "\n" fmt("%2") "\n" fmt("%3") "\n" fmt("%4") "\n" fmt("%5") "\n" fmt("%6") "\n" fmt("%7") \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc")

#define F_ADD(r) "v_add_u32 " r ", " r ", %8"
#define F_FMA(r) "v_fma_f32 " r ", " r ", %8, %9"
#define F_PKFMA(r) "v_fma_f32 " r ", %8, " r ", %9"
#define F_AND(r) "v_and_b32 " r ", " r ", %8"
#define F_XOR(r) "v_xor_b32 " r ", " r ", %8"
#define F_PKADD16(r) "v_pk_add_u16 " r ", " r ", %8"
#define F_PKMIN16(r) "v_pk_min_u16 " r ", " r ", %8"
#define F_PKMAX16(r) "v_pk_max_u16 " r ", " r ", %8"
#define F_PKSUB16C(r) "v_pk_sub_u16 " r ", " r ", %8 clamp"
#define F_MIN(r) "v_min_u32 " r ", " r ", %8"
#define F_DOT4(r) "v_dot4_u32_u8 " r ", " r ", %8, %9"
#define F_MULLO(r) "v_mul_lo_u32 " r ", " r ", %8"
#define F_MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, %9"
#define F_MUL24(r) "v_mul_u32_u24 " r ", " r ", %8"
#define F_ALIGN(r) "v_alignbyte_b32 " r ", " r ", %8, 1"
#define F_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 2, %8"
#define F_ADD3(r) "v_add3_u32 " r ", " r ", %8, %9"
#define F_BFE(r) "v_bfe_u32 " r ", " r ", 3, 8"
#define F_PERM(r) "v_perm_b32 " r ", " r ", %8, %9"
#define F_CMP(r) "v_cmp_gt_u32 vcc, " r ", %8"
#define F_CMPCND(r) "v_cmp_gt_u32 vcc, " r ", %8\n v_cndmask_b32 " r ", " r ", %9, vcc"
#define F_CNDMASK(r) "v_cndmask_b32 " r ", " r ", %8, vcc"
#define F_SAD(r) "v_sad_u8 " r ", " r ", %8, %9"
#define F_BCNT(r) "v_bcnt_u32_b32 " r ", " r ", %8"
#define F_MOVDPP(r) "v_mov_b32_dpp " r ", " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define F_ADDDPP(r) "v_add_u32_dpp " r ", " r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"
#define F_CVT(r) "v_cvt_f32_u32 " r ", " r
#define F_LSHR(r) "v_lshrrev_b32 " r ", 3, " r
#define F_SUBREV(r) "v_sub_u32 " r ", %8, " r
#define F_MED3(r) "v_med3_i32 " r ", " r ", %8, %9"
#define F_MAX3(r) "v_max3_u32 " r ", " r ", %8, %9"
#define F_PKMAD16(r) "v_pk_mad_u16 " r ", " r ", %8, %9"
#define F_PKMUL16(r) "v_pk_mul_lo_u16 " r ", " r ", %8"
#define F_PKLSHR16(r) "v_pk_lshrrev_b16 " r ", 1, " r
#define F_PKMAX3H(r) "v_pk_maximum3_f16 " r ", " r ", %8, %9"
#define F_PKMIN3H(r) "v_pk_minimum3_f16 " r ", " r ", %8, %9"
#define F_BITOP3(r) "v_bitop3_b32 " r ", " r ", %8, %9 bitop3:0x96"
#define F_MIN3(r) "v_min3_u32 " r ", " r ", %8, %9"
#define F_MULHI(r) "v_mul_hi_u32 " r ", " r ", %8"
#define F_MADI24(r) "v_mad_i32_i24 " r ", " r ", %8, %9"
#define F_OR(r) "v_or_b32 " r ", " r ", %8"
#define F_LSHL(r) "v_lshlrev_b32 " r ", 3, " r
#define F_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 8, %8"
#define F_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9"
#define F_SUB(r) "v_sub_u32 " r ", " r ", %8"
#define F_CVTUB(r) "v_cvt_f32_ubyte0 " r ", " r
#define F_MULF(r) "v_mul_f32 " r ", " r ", %8"
#define F_ADDF(r) "v_add_f32 " r ", " r ", %8"
#define F_MBCNT(r) "v_mbcnt_lo_u32_b32 " r ", %8, " r
#define F_MOV(r) "v_mov_b32 " r ", %8"
#define F_MAXU16(r) "v_max_u16 " r ", " r ", %8"
#define F_ADDU16(r) "v_add_u16 " r ", " r ", %8"
#define F_RNDNE(r) "v_rndne_f32 " r ", " r
#define F_CVTI(r) "v_cvt_i32_f32 " r ", " r
#define F_DOT2(r) "v_dot2_u32_u16 " r ", " r ", %8, %9"
#define F_FMAF64(r) "v_add_u32 " r ", " r ", %8"

#define KERNEL(NAME, FMT, PER)                                                          \
    __global__ __launch_bounds__(64) void k_##NAME(uint32_t *out, uint32_t b, uint32_t c) \
    {                                                                                     \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < ITERS; ++i) {                                                 \
            _Pragma("unroll") for (int u = 0; u < UNR; ++u) OP8(FMT);                      \
        }                                                                                 \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;       \
    }                                                                                     \
    static const int per_##NAME = PER;

KERNEL(add_u32, F_ADD, 1)
KERNEL(fma_f32, F_FMA, 1)
KERNEL(and_b32, F_AND, 1)
KERNEL(xor_b32, F_XOR, 1)
KERNEL(pk_add_u16, F_PKADD16, 1)
KERNEL(pk_min_u16, F_PKMIN16, 1)
KERNEL(pk_max_u16, F_PKMAX16, 1)
KERNEL(pk_sub_u16_clamp, F_PKSUB16C, 1)
KERNEL(pk_mad_u16, F_PKMAD16, 1)
KERNEL(pk_mul_lo_u16, F_PKMUL16, 1)
KERNEL(pk_lshrrev_b16, F_PKLSHR16, 1)
KERNEL(min_u32, F_MIN, 1)
KERNEL(max3_u32, F_MAX3, 1)
KERNEL(med3_i32, F_MED3, 1)
KERNEL(dot4_u32_u8, F_DOT4, 1)
KERNEL(mul_lo_u32, F_MULLO, 1)
KERNEL(mad_u32_u24, F_MAD24, 1)
KERNEL(mul_u32_u24, F_MUL24, 1)
KERNEL(alignbyte, F_ALIGN, 1)
KERNEL(lshl_add_u32, F_LSHLADD, 1)
KERNEL(add3_u32, F_ADD3, 1)
KERNEL(bfe_u32, F_BFE, 1)
KERNEL(lshrrev_b32, F_LSHR, 1)
KERNEL(perm_b32, F_PERM, 1)
KERNEL(cmp_gt_u32, F_CMP, 1)
KERNEL(cmp_cndmask, F_CMPCND, 2)
KERNEL(cndmask, F_CNDMASK, 1)
KERNEL(sad_u8, F_SAD, 1)
KERNEL(bcnt, F_BCNT, 1)
KERNEL(mov_dpp_quad, F_MOVDPP, 1)
KERNEL(add_dpp_row_shr, F_ADDDPP, 1)
KERNEL(cvt_f32_u32, F_CVT, 1)
KERNEL(pk_maximum3_f16, F_PKMAX3H, 1)
KERNEL(pk_minimum3_f16, F_PKMIN3H, 1)
KERNEL(bitop3_b32, F_BITOP3, 1)
KERNEL(min3_u32, F_MIN3, 1)
KERNEL(mul_hi_u32, F_MULHI, 1)
KERNEL(mad_i32_i24, F_MADI24, 1)
KERNEL(or_b32, F_OR, 1)
KERNEL(lshlrev_b32, F_LSHL, 1)
KERNEL(lshl_or_b32, F_LSHLOR, 1)
KERNEL(and_or_b32, F_ANDOR, 1)
KERNEL(sub_u32, F_SUB, 1)
KERNEL(cvt_f32_ubyte0, F_CVTUB, 1)
KERNEL(mul_f32, F_MULF, 1)
KERNEL(add_f32, F_ADDF, 1)
KERNEL(mbcnt_lo, F_MBCNT, 1)
KERNEL(mov_b32, F_MOV, 1)
KERNEL(max_u16, F_MAXU16, 1)
KERNEL(add_u16, F_ADDU16, 1)
KERNEL(rndne_f32, F_RNDNE, 1)
KERNEL(cvt_i32_f32, F_CVTI, 1)
KERNEL(dot2_u32_u16, F_DOT2, 1)

#define OP8D(fmt) \
    asm volatile(fmt("%0") "\n" fmt("%1") "\n" fmt("%2") "\n" fmt("%3") "\n" fmt("%4") "\n" fmt("%5") "\n" fmt("%6") "\n" fmt("%7") \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))
#define F_FMA64(r) "v_fma_f64 " r ", " r ", %8, %9"
#define F_MUL64(r) "v_mul_f64 " r ", " r ", %8"
#define F_ADD64(r) "v_add_f64 " r ", " r ", %8"
#define KERNELD(NAME, FMT)                                                              \
    __global__ __launch_bounds__(64) void k_##NAME(uint32_t *out, uint32_t bi, uint32_t ci) \
    {                                                                                     \
        double b = 1.0 + bi * 1e-9, c = ci * 1e-12;                                        \
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < ITERS; ++i) {                                                 \
            _Pragma("unroll") for (int u = 0; u < UNR; ++u) OP8D(FMT);                     \
        }                                                                                 \
        out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    }                                                                                     \
    static const int per_##NAME = 1;
KERNELD(fma_f64, F_FMA64)
KERNELD(mul_f64, F_MUL64)
KERNELD(add_f64, F_ADD64)

// LDS: conflict-free b32 reads, byte reads, and 64-lane random u16 gathers
template <int MODE>
__global__ __launch_bounds__(64) void k_lds(uint32_t *out, uint32_t b, uint32_t c)
{
    __shared__ uint32_t buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = i * b;
    __syncthreads();
    uint32_t acc = 0;
    uint32_t idx = MODE == 2 ? ((threadIdx.x * 2654435761u) >> 21) & 2046 : threadIdx.x * 4;   // byte address
    const char *base = reinterpret_cast<const char *>(buf);
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            uint32_t v;
            if (MODE == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(idx), "n"(u * 256 % 2048));
            else if (MODE == 1) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(idx), "n"(u * 256 % 2048));
            else asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(v) : "v"(idx), "n"(u * 26 % 2048));
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            acc += v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 64 + threadIdx.x] = acc + base[0];
}

struct Case { const char *name; void (*fn)(uint32_t *, uint32_t, uint32_t); int per; int n_per_trip; };

__global__ void k_check_pk3(uint32_t *bad)
{
    // all (x, y, z) byte triples of a 64 x 64 x 64 lattice: packed f16 maximum3 / minimum3 on (0x0400 | v) patterns
    // must equal the integer max / min in both halves
    const uint32_t x = (threadIdx.x * 4 + 1) & 255, y = (blockIdx.x * 4 + 2) & 255;
    for (uint32_t z = 0; z < 256; ++z) {
        const uint32_t px = x * 0xFFFF0001u + 0x04FF0400u, py = y * 0xFFFF0001u + 0x04FF0400u, pz = z * 0xFFFF0001u + 0x04FF0400u;
        uint32_t mx, mn;
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(mx) : "v"(px), "v"(py), "v"(pz));
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(mn) : "v"(px), "v"(py), "v"(pz));
        const uint32_t imx = max(x, max(y, z)), imn = min(x, min(y, z));
        const uint32_t emx = (imx | 0x400u) | (((255u - imn) | 0x400u) << 16);
        const uint32_t emn = (imn | 0x400u) | (((255u - imx) | 0x400u) << 16);
        if (mx != emx || mn != emn) atomicAdd(bad, 1u);
    }
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    printf("device %s, %d CUs, clock %.0f MHz (nominal)\n", prop.name, cus, clk_khz / 1000.0);
    uint32_t *out;
    CK(hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * 4));
    std::vector<Case> cases = {
#define C(NAME) {#NAME, k_##NAME, per_##NAME, UNR * 8},
        C(add_u32) C(fma_f32) C(and_b32) C(xor_b32) C(pk_add_u16) C(pk_min_u16) C(pk_max_u16) C(pk_sub_u16_clamp) C(pk_mad_u16)
        C(pk_mul_lo_u16) C(pk_lshrrev_b16) C(min_u32) C(max3_u32) C(med3_i32) C(dot4_u32_u8) C(mul_lo_u32) C(mad_u32_u24)
        C(mul_u32_u24) C(alignbyte) C(lshl_add_u32) C(add3_u32) C(bfe_u32) C(lshrrev_b32) C(perm_b32) C(cmp_gt_u32) C(cmp_cndmask)
        C(cndmask) C(sad_u8) C(bcnt) C(mov_dpp_quad) C(add_dpp_row_shr) C(cvt_f32_u32)
        C(pk_maximum3_f16) C(pk_minimum3_f16) C(bitop3_b32) C(min3_u32) C(mul_hi_u32) C(mad_i32_i24) C(or_b32) C(lshlrev_b32)
        C(lshl_or_b32) C(and_or_b32) C(sub_u32) C(cvt_f32_ubyte0) C(mul_f32) C(add_f32) C(mbcnt_lo) C(mov_b32) C(max_u16)
        C(add_u16) C(rndne_f32) C(cvt_i32_f32) C(dot2_u32_u16) C(fma_f64) C(mul_f64) C(add_f64)
        {"ds_read_b32 (no conflict)", k_lds<0>, 1, 64}, {"ds_read_u8 (no conflict)", k_lds<1>, 1, 64},
        {"ds_read_u16 (random gather)", k_lds<2>, 1, 64},
    };
    {
        uint32_t *bad, hbad = 0;
        CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(k_check_pk3, dim3(64), dim3(64), 0, 0, bad);
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        printf("v_pk_maximum3_f16 / v_pk_minimum3_f16 on (0x0400 | byte) patterns vs integer max / min: %u mismatches\n", hbad);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-30s %10s %10s %10s %10s   (cycles per wave64 instruction per SIMD; LDS rows: per CU)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    for (const Case &cs : cases) {
        printf("%-30s", cs.name);
        const bool lds = cs.name[0] == 'd' && cs.name[1] == 's';
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * 4 * wps;   // one-wave workgroups; the dispatcher spreads them evenly
            hipLaunchKernelGGL(cs.fn, dim3(blocks), dim3(64), 0, 0, out, 3u, 5u);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(cs.fn, dim3(blocks), dim3(64), 0, 0, out, 3u, 5u);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double cycles = ms * 1e-3 * clk_khz * 1e3;
            const double instr_per_simd = (double)ITERS * cs.n_per_trip * cs.per * wps * (lds ? 4 : 1);
            printf(" %10.2f", cycles / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
