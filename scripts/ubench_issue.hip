// Issue rate of the instructions k_ydrop2's row loop is made of, on a FULL chip (gfx950): every SIMD holds W waves (1, 2, 4, 8), each
// wave runs a long unrolled stream of independent instructions of one kind; the HIP-event time of the launch gives the SIMD cycles one
// wave64 instruction occupies its pipe for.  Settles what DESIGN.md section 6 prices `roofline.valu` against (round 4: SQ counters said a
// quad-cycle per int32 VALU instruction; the guide's wave-scheduling section says two cycles).
// Build + run: hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_issue scripts/ubench_issue.hip && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)

template <int KIND>
__global__ __launch_bounds__(64) void k_issue(int *sink, int iters, int y0) {
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int y = y0 + (int)blockIdx.x;
    for (int it = 0; it < iters; it++) {
        // 8 independent chains x 8 = 64 instructions per iteration
        if (KIND == 0) { R8(asm volatile("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));) }
        if (KIND == 1) { R8(asm volatile("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));) }
        if (KIND == 2) { R8(asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));) }
        if (KIND == 3) { R8(asm volatile("v_add3_u32 %0, %0, %8, 7\n v_add3_u32 %1, %1, %8, 7\n v_add3_u32 %2, %2, %8, 7\n v_add3_u32 %3, %3, %8, 7\n v_add3_u32 %4, %4, %8, 7\n v_add3_u32 %5, %5, %8, 7\n v_add3_u32 %6, %6, %8, 7\n v_add3_u32 %7, %7, %8, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));) }
        if (KIND == 4) { R8(asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));) }
        if (KIND == 5) { R8(asm volatile("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (KIND == 6) { R8(asm volatile("v_perm_b32 %0, %0, %8, %0\n v_perm_b32 %1, %1, %8, %1\n v_perm_b32 %2, %2, %8, %2\n v_perm_b32 %3, %3, %8, %3\n v_perm_b32 %4, %4, %8, %4\n v_perm_b32 %5, %5, %8, %5\n v_perm_b32 %6, %6, %8, %6\n v_perm_b32 %7, %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0x03020100));) }
        if (KIND == 7) { R8(asm volatile("v_cmp_lt_i32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_i32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_i32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_i32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y) : "vcc");) }
        if (KIND == 8) {   // the scalar unit: 64 dependent-free s_add / s_max of one wave per iteration (8 registers)
            int s0 = y, s1 = y + 1, s2 = y + 2, s3 = y + 3, s4 = y + 4, s5 = y + 5, s6 = y + 6, s7 = y + 7;
            R8(asm volatile("s_add_i32 %0, %0, 3\n s_max_i32 %1, %1, %0\n s_add_i32 %2, %2, 3\n s_max_i32 %3, %3, %2\n s_add_i32 %4, %4, 3\n s_max_i32 %5, %5, %4\n s_add_i32 %6, %6, 3\n s_max_i32 %7, %7, %6" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7));)
            a0 += s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
        }
        if (KIND == 9) {   // half VALU, half SALU, interleaved: do the two pipes issue side by side from one wave / from several?
            int s0 = y, s1 = y + 1, s2 = y + 2, s3 = y + 3;
            R8(asm volatile("v_max_i32 %0, %0, %12\n s_add_i32 %8, %8, 3\n v_max_i32 %1, %1, %12\n s_add_i32 %9, %9, 3\n v_max_i32 %2, %2, %12\n s_add_i32 %10, %10, 3\n v_max_i32 %3, %3, %12\n s_add_i32 %11, %11, 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(y));)
            a4 += s0 + s1 + s2 + s3;
        }
    }
    sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
static void run(const char *name, int *sink, int cus, double mhz) {
    const int iters = 2000;
    for (int w : {1, 2, 4, 8}) {
        const int blocks = cus * 4 * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(64), 0, 0, sink, 10, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(64), 0, 0, sink, iters, 1);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 64.0 * w;
        const double cycles = ms * 1e-3 * mhz * 1e6;
        printf("%-28s %d waves/SIMD: %8.3f ms  -> %.2f cycles per wave64 instruction per SIMD\n", name, w, ms, cycles / instr_per_simd);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const double mhz = pr.clockRate / 1000.0;
    printf("%s: %d CUs, %.0f MHz (hipDeviceProp clockRate)\n", pr.gcnArchName, pr.multiProcessorCount, mhz);
    int *sink;
    hipMalloc((void **)&sink, (size_t)pr.multiProcessorCount * 4 * 8 * 64 * sizeof(int));
    run<0>("v_max_i32", sink, pr.multiProcessorCount, mhz);
    run<1>("v_sub_u32", sink, pr.multiProcessorCount, mhz);
    run<2>("v_alignbit_b32", sink, pr.multiProcessorCount, mhz);
    run<3>("v_add3_u32", sink, pr.multiProcessorCount, mhz);
    run<4>("v_pk_max_i16", sink, pr.multiProcessorCount, mhz);
    run<5>("v_max_i32_dpp row_shr:1", sink, pr.multiProcessorCount, mhz);
    run<6>("v_perm_b32", sink, pr.multiProcessorCount, mhz);
    run<7>("v_cmp + v_cndmask", sink, pr.multiProcessorCount, mhz);
    run<8>("s_add / s_max (SALU)", sink, pr.multiProcessorCount, mhz);
    run<9>("v_max + s_add interleaved", sink, pr.multiProcessorCount, mhz);
    return 0;
}
