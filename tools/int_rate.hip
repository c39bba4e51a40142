// Micro-benchmark: issue rates on gfx950 of the integer / LDS instructions the horizontal resampling pass is built from
// (wave64 instructions per SIMD-cycle for VALU forms, LDS cycles per wave-instruction for the reads).
//   hipcc --offload-arch=gfx950 -O3 tools/int_rate.hip -o /tmp/int_rate && /tmp/int_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int w, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x + 3 * i; }
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<int *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned addr = (threadIdx.x & 63) * pitch;
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
#define X(i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "s"(w));
            R8(X)
#undef X
        } else if (MODE == 1) {
#define X(i) asm volatile("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a[i]) : "v"(b[i]), "s"(w));
            R8(X)
#undef X
        } else if (MODE == 2) {
#define X(i) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 3) {
#define X(i) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(a[i]) : "v"(b[i]));
            R8(X)
#undef X
        } else if (MODE == 4) {
#define X(i) asm volatile("v_and_b32 %0, 0xff, %1" : "=v"(a[i]) : "v"(b[i]));
            R8(X)
#undef X
        } else if (MODE == 5) {
            int s[8];
#define X(i) asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s[i]) : "v"(b[i]), "s"(w));
            R8(X)
#undef X
            asm volatile("" :: "s"(s[0]), "s"(s[1]), "s"(s[2]), "s"(s[3]), "s"(s[4]), "s"(s[5]), "s"(s[6]), "s"(s[7]));
        } else if (MODE == 6) {
#define X(i) asm volatile("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 7) {
#define X(i) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "s"(w));
            R8(X)
#undef X
        } else if (MODE == 8) {
#define X(i) asm volatile("v_alignbyte_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]), "s"(w));
            R8(X)
#undef X
        } else if (MODE == 9) {
#define X(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]), "s"(w));
            R8(X)
#undef X
        } else if (MODE == 10) {
#define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "s"(w));
            R8(X)
#undef X
        } else if (MODE == 11) {
#define X(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            R8(X)
#undef X
        } else if (MODE == 12) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 13) {
            long long q[8];
#define X(i) asm volatile("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(q[i]) : "v"((long long)b[i]), "v"((long long)b[(i + 1) & 7]));
            R8(X)
#undef X
            for (int i = 0; i < 8; ++i) a[i] ^= (int)q[i];
        } else if (MODE == 14) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 15) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
            R8(X)
#undef X
        } else if (MODE == 16) {
#define X(i) asm volatile("v_div_scale_f32 %0, vcc, %1, %2, %1" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]) : "vcc");
            R8(X)
#undef X
        } else if (MODE == 17) {
#define X(i) asm volatile("v_div_fixup_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7]));
            R8(X)
#undef X
        } else if (MODE == 18) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 19) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 30) {
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]) : "vcc");
            R8(X)
#undef X
        } else if (MODE == 31) {
            unsigned long long m = 0x5555555555555555ull;
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]), "s"(m));
            R8(X)
#undef X
        } else if (MODE == 32) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            R8(X)
#undef X
        } else if (MODE == 33) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(b[i]), "v"(b[(i + 1) & 7]) : "vcc");
            R8(X)
#undef X
        } else if (MODE == 34) {
            unsigned long long m;
#define X(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
            asm volatile("" :: "s"(m));
        } else if (MODE == 35) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 36) {
#define X(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 37) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 38) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
            R8(X)
#undef X
        } else if (MODE == 39) {
#define X(i) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            R8(X)
#undef X
        } else if (MODE == 20) {
#define X(i) asm volatile("ds_read_u8 %0, %1 offset:" #i : "=v"(a[i]) : "v"(addr));
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (MODE == 21) {
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:4*" #i : "=v"(a[i]) : "v"(addr));
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (MODE == 22) {
            long long q[8];
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:8*" #i : "=v"(q[i]) : "v"(addr));
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]));
            for (int i = 0; i < 8; ++i) a[i] ^= (int)q[i];
        } else if (MODE == 23) {
#define X(i) asm volatile("ds_read_u16 %0, %1 offset:2*" #i : "=v"(a[i]) : "v"(addr));
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (MODE == 24) {   // broadcast read: every lane the same address
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:4*" #i : "=v"(a[i]) : "v"(0u));
            R8(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        }
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int *d, int pitch)
{
    const int blocks = 256 * 4;   // 4 blocks of 4 waves per CU = 4 waves per SIMD (32 KB of LDS each)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, d, 3, pitch);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, d, 3, pitch);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double winstr = (double)blocks * 4 * ITER * 8;
    if (MODE < 20)
        printf("%-34s %.3f ms  SIMD-cycles per wave-instruction @2.4GHz: %.2f\n", name, ms, (ms * 1e-3) * 1024.0 * 2.4e9 / winstr);
    else
        printf("%-34s %.3f ms  CU-cycles per wave-instruction @2.4GHz: %.2f   (pitch %d)\n", name, ms, (ms * 1e-3) * 256.0 * 2.4e9 / winstr, pitch);
}
int main()
{
    int *d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("v_mad_i32_i24 v, v, s, v", d, 0); run<7>("v_mul_u32_u24 v, v, s", d, 0);
    run<1>("v_mul_i32_i24_sdwa BYTE_1 (sgpr)", d, 0); run<6>("v_mul_i32_i24_sdwa BYTE_1 (vgpr)", d, 0);
    run<2>("v_add3_u32", d, 0); run<3>("v_bfe_u32", d, 0); run<4>("v_and_b32", d, 0); run<5>("v_readlane_b32 (sgpr lane)", d, 0);
    run<18>("v_fma_f32", d, 0); run<11>("v_rcp_f32", d, 0); run<12>("v_mul_lo_u32", d, 0); run<13>("v_lshl_add_u64", d, 0);
    run<14>("v_add_f32_dpp row_shr:1", d, 0); run<19>("v_add_f32_dpp wave_shr:1", d, 0); run<15>("v_mov_b32_dpp wave_shr:1", d, 0);
    run<16>("v_div_scale_f32", d, 0); run<17>("v_div_fixup_f32", d, 0);
    run<35>("v_mul_f32", d, 0); run<36>("v_add_f32", d, 0); run<37>("v_fmac_f32", d, 0); run<32>("v_mov_b32", d, 0); run<39>("v_add_u32", d, 0);
    run<30>("v_cndmask_b32_e32 (vcc)", d, 0); run<31>("v_cndmask_b32_e64 (sgpr mask)", d, 0);
    run<33>("v_cmp_lt_f32 -> vcc", d, 0); run<34>("v_cmp_lt_f32_e64 -> sgpr", d, 0); run<38>("v_cvt_f32_i32", d, 0);
    run<8>("v_alignbyte_b32", d, 0); run<9>("v_perm_b32", d, 0); run<10>("v_dot4_u32_u8", d, 0);
    for (int pitch : {228, 232, 4}) {
        run<20>("ds_read_u8", d, pitch); run<23>("ds_read_u16", d, pitch); run<21>("ds_read_b32", d, pitch); run<22>("ds_read_b64", d, pitch);
    }
    run<24>("ds_read_b32 broadcast", d, 0);
    return 0;
}
