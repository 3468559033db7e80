// Issue cost of the vector-instruction classes the encode kernels are made of, on one SIMD of gfx950, with 1, 2 and 3 waves
// resident per SIMD (the kernels run at 3).  A class's cost is what one more instruction of it adds to the SIMD's time --
// the weight tools/isa_mix.py gives it.  Output: profiles/instr_rates_r04.txt.
//   hipcc --offload-arch=gfx950 -O2 -o build/instr_rates tools/instr_rates.hip && build/instr_rates
// Method: one workgroup of 256 * W threads on one CU (W waves per SIMD), every wave runs REP x 32 copies of the instruction on
// 16 independent destination registers (no dependent chains shorter than 16 instructions), s_memtime around the loop; the
// figure is the slowest wave's ticks / (REP * 32 * W): cycles of SIMD time per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>

#define REP 256

#define BODY16(INS) \
    INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15)

// d: 16 x 64-bit destinations, a/b/c: 64-bit sources, i0/i1: 32-bit sources
#define KERNEL(NAME, ASMSTR, OUTC, ...)                                                                     \
    __global__ void __launch_bounds__(768) NAME(long long *out, double fa, double fb, int ia)               \
    {                                                                                                       \
        __shared__ double lds[2048];                                                                        \
        double a = fa + threadIdx.x, b = fb - threadIdx.x, c = fa * 3.0;                                     \
        int i0 = ia + threadIdx.x, i1 = ia * 3 + 1;                                                          \
        (void)c; (void)i0; (void)i1; (void)a; (void)b;                                                       \
        unsigned la = (threadIdx.x * 8u) & 8191u; (void)la;                                                  \
        unsigned la16 = (threadIdx.x * 16u) & 8191u; (void)la16;                                             \
        typedef double __attribute__((ext_vector_type(2))) d2_t; d2_t q[16]; for (int k = 0; k < 16; k++) { q[k].x = fa + k; q[k].y = fb; } \
        lds[threadIdx.x] = a; lds[threadIdx.x + 768] = b;                                                    \
        __syncthreads();                                                                                    \
        double d[16]; int e[16];                                                                            \
        for (int k = 0; k < 16; k++) { d[k] = a + k; e[k] = i0 + k; }                                        \
        __builtin_amdgcn_s_barrier();                                                                       \
        long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        for (int r = 0; r < REP; r++) {                                                                     \
            _Pragma("unroll") for (int u = 0; u < 2; u++) {                                                 \
                BODY16(NAME##_INS)                                                                          \
            }                                                                                               \
        }                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        double s = 0; int si = 0;                                                                           \
        for (int k = 0; k < 16; k++) { s += d[k] + q[k].x + q[k].y; si += e[k]; }                                              \
        if (s == 1.2345 && si == 77) out[1000] = 1;                                                          \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                       \
    }

// ---- 64-bit destination, two 64-bit sources
#define add_f64_INS(k) asm volatile("v_add_f64 %0, %1, %2" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(add_f64, , )
#define mul_f64_INS(k) asm volatile("v_mul_f64 %0, %1, %2" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(mul_f64, , )
#define fma_f64_INS(k) asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(d[k]) : "v"(a), "v"(b), "v"(c));
KERNEL(fma_f64, , )
#define max_f64_INS(k) asm volatile("v_max_f64 %0, %1, %2" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(max_f64, , )
#define rcp_f64_INS(k) asm volatile("v_rcp_f64 %0, %1" : "+v"(d[k]) : "v"(a));
KERNEL(rcp_f64, , )
#define sqrt_f64_INS(k) asm volatile("v_sqrt_f64 %0, %1" : "+v"(d[k]) : "v"(a));
KERNEL(sqrt_f64, , )
#define rsq_f64_INS(k) asm volatile("v_rsq_f64 %0, %1" : "+v"(d[k]) : "v"(a));
KERNEL(rsq_f64, , )
#define ldexp_f64_INS(k) asm volatile("v_ldexp_f64 %0, %1, %2" : "+v"(d[k]) : "v"(a), "v"(i0));
KERNEL(ldexp_f64, , )
#define div_scale_f64_INS(k) asm volatile("v_div_scale_f64 %0, vcc, %1, %2, %1" : "+v"(d[k]) : "v"(a), "v"(b) : "vcc");
KERNEL(div_scale_f64, , )
#define div_fixup_f64_INS(k) asm volatile("v_div_fixup_f64 %0, %1, %2, %3" : "+v"(d[k]) : "v"(a), "v"(b), "v"(c));
KERNEL(div_fixup_f64, , )
#define cvt_f64_i32_INS(k) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(d[k]) : "v"(i0));
KERNEL(cvt_f64_i32, , )
#define cvt_f64_u32_INS(k) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(d[k]) : "v"(i0));
KERNEL(cvt_f64_u32, , )
#define cvt_i32_f64_INS(k) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(e[k]) : "v"(a));
KERNEL(cvt_i32_f64, , )
#define cvt_f64_f32_INS(k) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[k]) : "v"(i0));
KERNEL(cvt_f64_f32, , )
#define cmp_lt_f64_INS(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
KERNEL(cmp_lt_f64, , )
#define cmp_class_f64_INS(k) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(a), "v"(i0) : "vcc");
KERNEL(cmp_class_f64, , )
#define lshlrev_b64_INS(k) asm volatile("v_lshlrev_b64 %0, %1, %2" : "+v"(d[k]) : "v"(i0), "v"(a));
KERNEL(lshlrev_b64, , )
#define lshl_add_u64_INS(k) asm volatile("v_lshl_add_u64 %0, %1, 3, %2" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(lshl_add_u64, , )
#define mad_u64_u32_INS(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "+v"(d[k]) : "v"(i0), "v"(i1), "v"(a) : "vcc");
KERNEL(mad_u64_u32, , )
// ---- 32-bit
#define mov_b32_INS(k) asm volatile("v_mov_b32 %0, %1" : "+v"(e[k]) : "v"(i0));
KERNEL(mov_b32, , )
#define mov_b32_dpp_INS(k) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(e[k]) : "v"(i1));
KERNEL(mov_b32_dpp, , )
#define mov_b32_dpp_row_shr_INS(k) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(e[k]) : "v"(i1));
KERNEL(mov_b32_dpp_row_shr, , )
#define cndmask_b32_INS(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(cndmask_b32, , )
#define cndmask_b32_e64_INS(k) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(cndmask_b32_e64, , )
#define cmp_cnd2_INS(k) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "+v"(e[k]) : "v"(a), "v"(b), "v"(i0), "v"(i1) : "vcc");
KERNEL(cmp_cnd2, , )
#define or_b32_INS(k) asm volatile("v_or_b32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(or_b32, , )
#define sub_u32_INS(k) asm volatile("v_sub_u32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(sub_u32, , )
#define lshrrev_b32_INS(k) asm volatile("v_lshrrev_b32 %0, 3, %1" : "+v"(e[k]) : "v"(i0));
KERNEL(lshrrev_b32, , )
#define lshlrev_b32_v_INS(k) asm volatile("v_lshlrev_b32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(lshlrev_b32_v, , )
#define max_u32_INS(k) asm volatile("v_max_u32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(max_u32, , )
#define mov_b64_INS(k) asm volatile("v_mov_b64 %0, %1" : "+v"(d[k]) : "v"(a));
KERNEL(mov_b64, , )
#define cvt_u32_f64_INS(k) asm volatile("v_cvt_u32_f64 %0, %1" : "+v"(e[k]) : "v"(a));
KERNEL(cvt_u32_f64, , )
#define perm_b32_INS(k) asm volatile("v_perm_b32 %0, %1, %2, %1" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(perm_b32, , )
#define and_or_b32_INS(k) asm volatile("v_and_or_b32 %0, %1, %2, %1" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(and_or_b32, , )
#define add_f64_lit_INS(k) asm volatile("v_add_f64 %0, %1, 1.0" : "+v"(d[k]) : "v"(a));
KERNEL(add_f64_lit, , )
#define mul_f64_sgpr_INS(k) asm volatile("v_mul_f64 %0, %1, s[20:21]" : "+v"(d[k]) : "v"(a));
KERNEL(mul_f64_sgpr, , )
#define add_f64_abs_INS(k) asm volatile("v_add_f64 %0, |%1|, -%2" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(add_f64_abs, , )
#define add_u32_INS(k) asm volatile("v_add_u32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(add_u32, , )
#define lshlrev_b32_INS(k) asm volatile("v_lshlrev_b32 %0, 3, %1" : "+v"(e[k]) : "v"(i0));
KERNEL(lshlrev_b32, , )
#define and_b32_INS(k) asm volatile("v_and_b32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(and_b32, , )
#define xor_b32_INS(k) asm volatile("v_xor_b32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(xor_b32, , )
#define bfe_u32_INS(k) asm volatile("v_bfe_u32 %0, %1, 3, 5" : "+v"(e[k]) : "v"(i0));
KERNEL(bfe_u32, , )
#define lshl_or_b32_INS(k) asm volatile("v_lshl_or_b32 %0, %1, 3, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(lshl_or_b32, , )
#define add3_u32_INS(k) asm volatile("v_add3_u32 %0, %1, %2, %1" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(add3_u32, , )
#define mul_lo_u32_INS(k) asm volatile("v_mul_lo_u32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(mul_lo_u32, , )
#define mul_u32_u24_INS(k) asm volatile("v_mul_u32_u24 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(mul_u32_u24, , )
#define cmp_lt_u32_INS(k) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(i0), "v"(i1) : "vcc");
KERNEL(cmp_lt_u32, , )
#define cmp_lt_u32_e64_INS(k) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(i0), "v"(i1) : "s20", "s21");
KERNEL(cmp_lt_u32_e64, , )
#define readlane_b32_INS(k) asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(i0) : "s20");
KERNEL(readlane_b32, , )
#define readfirstlane_b32_INS(k) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(i0) : "s20");
KERNEL(readfirstlane_b32, , )
#define writelane_b32_INS(k) asm volatile("v_writelane_b32 %0, s20, 5" : "+v"(e[k]) : : );
KERNEL(writelane_b32, , )
#define add_f32_INS(k) asm volatile("v_add_f32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(add_f32, , )
#define pk_add_f32_INS(k) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(pk_add_f32, , )
#define pk_mov_b32_INS(k) asm volatile("v_pk_mov_b32 %0, %1, %2" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(pk_mov_b32, , )
#define cvt_i32_f32_INS(k) asm volatile("v_cvt_i32_f32 %0, %1" : "+v"(e[k]) : "v"(i0));
KERNEL(cvt_i32_f32, , )
#define ffbh_u32_INS(k) asm volatile("v_ffbh_u32 %0, %1" : "+v"(e[k]) : "v"(i0));
KERNEL(ffbh_u32, , )
#define bcnt_u32_INS(k) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(bcnt_u32, , )
#define mbcnt_lo_INS(k) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %2" : "+v"(e[k]) : "v"(i0), "v"(i1));
KERNEL(mbcnt_lo, , )
// ---- LDS (cycles of the wave's issue stream; the LDS pipe itself is shared by the CU)
#define ds_read_b64_INS(k) asm volatile("ds_read_b64 %0, %1" : "+v"(d[k]) : "v"(la) : "memory");
KERNEL(ds_read_b64, , )
#define ds_read2_b64_INS(k) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "+v"(q[k]) : "v"(la) : "memory");
KERNEL(ds_read2_b64, , )
#define ds_read_b64x2_INS(k) asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "+v"(d[k]), "+v"(d[(k + 1) & 15]) : "v"(la) : "memory");
KERNEL(ds_read_b64x2, , )
#define ds_read_b128_INS(k) asm volatile("ds_read_b128 %0, %1" : "+v"(q[k]) : "v"(la16) : "memory");
KERNEL(ds_read_b128, , )
#define ds_read_b32_INS(k) asm volatile("ds_read_b32 %0, %1" : "+v"(e[k]) : "v"(la) : "memory");
KERNEL(ds_read_b32, , )
#define ds_write_b64_INS(k) asm volatile("ds_write_b64 %0, %1" : : "v"(la), "v"(a) : "memory");
KERNEL(ds_write_b64, , )
#define ds_bpermute_b32_INS(k) asm volatile("ds_bpermute_b32 %0, %1, %2" : "+v"(e[k]) : "v"(la), "v"(i0) : "memory");
KERNEL(ds_bpermute_b32, , )
#define ds_or_b32_INS(k) asm volatile("ds_or_b32 %0, %1" : : "v"(la), "v"(i0) : "memory");
KERNEL(ds_or_b32, , )
// ---- scalar
#define s_add_u32_INS(k) asm volatile("s_add_u32 s20, s21, s22" : : : "s20", "scc");
KERNEL(s_add_u32, , )
#define s_nop_INS(k) asm volatile("s_nop 0");
KERNEL(s_nop, , )
// ---- mixes: do a 32-bit op and an fp64 op of one wave pair up?  (16 x [f64, b32])
#define mix_f64_b32_INS(k) asm volatile("v_add_f64 %0, %2, %3\n\tv_and_b32 %1, %4, %5" : "+v"(d[k]), "+v"(e[k]) : "v"(a), "v"(b), "v"(i0), "v"(i1));
KERNEL(mix_f64_b32, , )
#define mix_f64_salu_INS(k) asm volatile("v_add_f64 %0, %1, %2\n\ts_add_u32 s20, s21, s22" : "+v"(d[k]) : "v"(a), "v"(b) : "s20", "scc");
KERNEL(mix_f64_salu, , )
#define mix_f64_lds_INS(k) asm volatile("v_add_f64 %0, %2, %3\n\tds_read_b64 %1, %4" : "+v"(d[k]), "=v"(a) : "v"(b), "v"(c), "v"(la) : "memory");
KERNEL(mix_f64_lds, , )


// ---- patterns: a 64-bit select as the compiler writes it (one compare, two v_cndmask on the same mask), VCC form and SGPR-pair form
#define cmp_cnd3_vcc_INS(k) asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %4, vcc" : "+v"(e[k]), "+v"(e[(k + 1) & 15]) : "v"(a), "v"(b), "v"(i0), "v"(i1) : "vcc");
KERNEL(cmp_cnd3_vcc, , )
#define cmp_cnd3_sgpr_INS(k) asm volatile("v_cmp_lt_f64 s[20:21], %2, %3\n\tv_cndmask_b32_e64 %0, %4, %5, s[20:21]\n\tv_cndmask_b32_e64 %1, %5, %4, s[20:21]" : "+v"(e[k]), "+v"(e[(k + 1) & 15]) : "v"(a), "v"(b), "v"(i0), "v"(i1) : "s20", "s21");
KERNEL(cmp_cnd3_sgpr, , )
#define cmp_cnd5_vcc_INS(k) asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %4, vcc\n\tv_cndmask_b32 %0, %5, %4, vcc\n\tv_cndmask_b32 %1, %4, %5, vcc" : "+v"(e[k]), "+v"(e[(k + 1) & 15]) : "v"(a), "v"(b), "v"(i0), "v"(i1) : "vcc");
KERNEL(cmp_cnd5_vcc, , )
#define add_cnd_vcc_INS(k) asm volatile("v_add_f64 %0, %2, %3\n\tv_cndmask_b32 %1, %4, %5, vcc" : "+v"(d[k]), "+v"(e[k]) : "v"(a), "v"(b), "v"(i0), "v"(i1));
KERNEL(add_cnd_vcc, , )
#define add3_cnd_vcc_INS(k) asm volatile("v_add_f64 %0, %2, %3\n\tv_mul_f64 %0, %2, %3\n\tv_add_f64 %0, %2, %3\n\tv_cndmask_b32 %1, %4, %5, vcc" : "+v"(d[k]), "+v"(e[k]) : "v"(a), "v"(b), "v"(i0), "v"(i1));
KERNEL(add3_cnd_vcc, , )
#define fmac_f64_INS(k) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(fmac_f64, , )
#define add_f64_nop_INS(k) asm volatile("v_add_f64 %0, %1, %2\n\ts_nop 0" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(add_f64_nop, , )
#define add_f64_nop1_INS(k) asm volatile("v_add_f64 %0, %1, %2\n\ts_nop 1" : "+v"(d[k]) : "v"(a), "v"(b));
KERNEL(add_f64_nop1, , )
#define readlane_rot_INS(k) asm volatile("v_readlane_b32 s%1, %0, 5" : : "v"(i0), "n"(20 + (k & 7)) : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31");
KERNEL(readlane_rot, , )
#define cmp_rot_INS(k) asm volatile("v_cmp_lt_f64 s[%2:%3], %0, %1" : : "v"(a), "v"(b), "n"(20 + 2 * (k & 7)), "n"(21 + 2 * (k & 7)) : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31");
KERNEL(cmp_rot, , )
#define mul_add_dep_INS(k) asm volatile("v_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %2" : "+v"(d[0]) : "v"(a), "v"(b));
KERNEL(mul_add_dep, , )

typedef void (*kern_t)(long long *, double, double, int);
struct Entry { const char *name; kern_t k; int per; };

int main()
{
    long long *d; hipMalloc(&d, 8 * 2048);
    std::vector<Entry> v = {
#define E(n) {#n, n, 1}
#define E2(n) {#n, n, 2}
        E(add_f64), E(mul_f64), E(fma_f64), E(max_f64), E(rcp_f64), E(sqrt_f64), E(rsq_f64), E(ldexp_f64), E(div_scale_f64), E(div_fixup_f64),
        E(cvt_f64_i32), E(cvt_f64_u32), E(cvt_i32_f64), E(cvt_f64_f32), E(cmp_lt_f64), E(cmp_class_f64), E(lshlrev_b64), E(lshl_add_u64), E(mad_u64_u32),
        E(mov_b32), E(mov_b32_dpp), E(mov_b32_dpp_row_shr), E(cndmask_b32), E(cndmask_b32_e64), E2(cmp_cnd2), E(or_b32), E(sub_u32), E(lshrrev_b32), E(lshlrev_b32_v), E(max_u32), E(mov_b64), E(cvt_u32_f64), E(perm_b32), E(and_or_b32), E(add_f64_lit), E(mul_f64_sgpr), E(add_f64_abs), E(add_u32), E(lshlrev_b32), E(and_b32), E(xor_b32), E(bfe_u32), E(lshl_or_b32), E(add3_u32),
        E(mul_lo_u32), E(mul_u32_u24), E(cmp_lt_u32), E(cmp_lt_u32_e64), E(readlane_b32), E(readfirstlane_b32), E(writelane_b32), E(add_f32), E(pk_add_f32), E(pk_mov_b32),
        E(cvt_i32_f32), E(ffbh_u32), E(bcnt_u32), E(mbcnt_lo),
        E(ds_read_b64), E(ds_read2_b64), E2(ds_read_b64x2), E(ds_read_b128), E(ds_read_b32), E(ds_write_b64), E(ds_bpermute_b32), E(ds_or_b32), E(s_add_u32), E(s_nop),
        {"cmp_cnd3_vcc", cmp_cnd3_vcc, 3}, {"cmp_cnd3_sgpr", cmp_cnd3_sgpr, 3}, {"cmp_cnd5_vcc", cmp_cnd5_vcc, 5}, E2(add_cnd_vcc), {"add3_cnd_vcc", add3_cnd_vcc, 4}, E(fmac_f64), E2(add_f64_nop), E2(add_f64_nop1), E(readlane_rot), E(cmp_rot), E2(mul_add_dep), E2(mix_f64_b32), E2(mix_f64_salu), E2(mix_f64_lds),
    };
    printf("# cycles of SIMD time per wave-instruction (s_memtime ticks of the slowest wave / instructions issued on its SIMD); 100 MHz-independent: ticks are shader clocks\n");
    printf("%-22s %10s %10s %10s\n", "instruction", "1 wave", "2 waves", "3 waves");
    for (auto &e : v) {
        printf("%-22s", e.name);
        for (int W = 1; W <= 3; W++) {
            long long h[12];
            double best = 1e30;
            for (int rep = 0; rep < 3; rep++) {
                hipLaunchKernelGGL(e.k, dim3(1), dim3(256 * W), 0, 0, d, 1.5, 2.5, 3);
                hipDeviceSynchronize();
                hipMemcpy(h, d, 8 * 4 * W, hipMemcpyDeviceToHost);
                long long mx = 0;
                for (int i = 0; i < 4 * W; i++) if (h[i] > mx) mx = h[i];
                double c = (double)mx / ((double)REP * 32 * e.per * W);
                if (c < best) best = c;
            }
            printf(" %10.2f", best);
        }
        printf("\n");
    }
    return 0;
}
