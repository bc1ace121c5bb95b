// Micro-benchmarks of per-wave issue costs on gfx950 (one wave per SIMD, 4 waves per workgroup, 1 workgroup).
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench scripts/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))

__global__ void k_bench(long long *out, int *sink, int y0) {
    __shared__ int lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (i * 4 + 64) & 16383;   // pointer chain in bytes
    __syncthreads();
    int x = tid, y = y0, a = tid + 1, b = tid + 2, c = tid + 3, d = tid + 4;
    long long t0, t1;
    int k = 0;
#define BEGIN() t0 = clock64()
#define END(n) do { t1 = clock64(); if (tid == 0) out[k] = (t1 - t0); k++; } while (0)
    // 0: dependent v_max chain
    BEGIN(); R256(asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(y));) END(256);
    // 1: 4 independent chains
    BEGIN(); R64(asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y));) END(256);
    // 2: v_cmp (vcc) + v_cndmask dependent pairs (128 pairs = 256 instr)
    BEGIN(); R64(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");) END(256);
    // 3: v_cmp into SGPR pair + v_cndmask e64
    BEGIN(); R64(asm volatile("v_cmp_lt_i32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cmp_lt_i32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(y) : "s20", "s21");) END(256);
    // 4: dependent DPP max chain
    BEGIN(); R256(asm volatile("s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));) END(256);
    // 5: readfirstlane -> s_add -> v_add round trip (64 trips = 192 instr)
    BEGIN(); R64(asm volatile("v_readfirstlane_b32 s20, %0\n s_add_i32 s20, s20, 1\n v_add_u32 %0, s20, %0" : "+v"(x) : : "s20");) END(64);
    // 6: ballot -> s_ff1 -> v_readlane -> v_add (64 trips)
    BEGIN(); R64(asm volatile("v_cmp_ne_u32 s[20:21], 0, %0\n s_ff1_i32_b64 s22, s[20:21]\n s_and_b32 s22, s22, 63\n s_nop 3\n v_readlane_b32 s23, %0, s22\n v_add_u32 %0, s23, %0" : "+v"(x) : : "s20", "s21", "s22", "s23");) END(64);
    // 7: LDS dependent b32 reads (pointer chase, 64 reads)
    { int p = (tid & 63) * 4; BEGIN(); R64(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(p));) END(64); x += p; }
    // 8: LDS b128 reads dependent
    { int p = (tid & 63) * 16; int4 v; BEGIN(); R64(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %1, 0x3ff0, %1" : "=v"(v), "+v"(p));) END(64); x += p + v.x; }
    // 9: 64 barriers
    BEGIN(); R64(__syncthreads();) END(64);
    // 10: SALU dependent chain 256
    { int s = y0; BEGIN(); R256(asm volatile("s_add_i32 %0, %0, 3" : "+s"(s));) END(256); x += s; }
    // 11: LDS write + wait + barrier + read + wait (64 trips): the exchange pattern
    { int p = tid * 4; int v = x; BEGIN(); R64(asm volatile("ds_write_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n s_barrier\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(p) : "memory");) END(64); x += v; }
    // 12: v_add3 / v_max3 dependent
    BEGIN(); R256(asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(y));) END(256);
    // 13: global store + continue (64 stores dword)
    BEGIN(); R64(asm volatile("global_store_dword %0, %1, off" : : "v"((unsigned long long)(sink + 256 + tid)), "v"(x) : "memory");) END(64);
    // 14: taken branches (64)
    BEGIN(); R64(asm volatile("s_branch 1f\n s_nop 0\n1:\n" : : );) END(64);
    // 15: s_memtime cost itself
    BEGIN(); R64({ long long q = clock64(); x += (int)q; }) END(64);
    sink[tid] = x + a + b + c + d;
}

int main() {
    long long *out; int *sink;
    hipMalloc((void **)&out, 64 * sizeof(long long));
    hipMalloc((void **)&sink, 4096 * sizeof(int));
    for (int waves = 1; waves <= 8; waves *= 2) {
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(64 * waves), 0, 0, out, sink, 5);
        hipDeviceSynchronize();
        std::vector<long long> h(64);
        hipMemcpy(h.data(), out, 64 * sizeof(long long), hipMemcpyDeviceToHost);
        const char *names[] = {"dep v_max x256", "4 indep chains x256", "cmp vcc+cndmask x256", "cmp sgpr+nop+cndmask x256(+128 nop)", "dep DPP max x256 (+nop)",
                               "readfirstlane-sadd-vadd x64", "ballot-ff1-readlane-vadd x64", "LDS b32 chase x64", "LDS b128 chase x64", "barrier x64",
                               "SALU dep x256", "lds write-barrier-read x64", "dep v_max3 x256", "global store x64", "taken branch x64", "clock64 x64"};
        printf("waves per workgroup = %d\n", waves);
        for (int i = 0; i < 16; i++) printf("  %-40s %8lld clk\n", names[i], h[i]);
    }
    return 0;
}
