// developer microbenchmark: what instruction patterns cost in ONE wave per SIMD (the regime of k_map_obs).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/lone_wave tools/ubench/lone_wave.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int which>
__global__ __launch_bounds__(64) void k(unsigned long long *out, float *sink, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = (float)i;
    __syncthreads();
    float a = lane * 0.5f, b = 1.0001f, c = 0.3f, d = 2.f;
    unsigned int u = lane, u2 = lane + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma clang loop unroll(disable)
    for (int it = 0; it < iters; it++) {
        switch (which) {
        case 0:  // 32 independent v_fma_f32 (4 chains)
            asm volatile(REP8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0001f), "v"(0.5f));
            break;
        case 1:  // 32 dependent v_fma_f32 (1 chain)
            asm volatile(REP32("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(1.0001f), "v"(0.5f));
            break;
        case 2:  // 16 x (v_cmp -> vcc -> v_cndmask) dependent through vcc, with the 2 wait states
            asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %1, %0\n s_nop 1\n v_cndmask_b32 %1, %1, %0, vcc\n")
                         : "+v"(a), "+v"(b) :: "vcc");
            break;
        case 3:  // 16 x (v_cmp_e64 -> sgpr pair -> v_cndmask_e64)
            asm volatile(REP8("v_cmp_lt_f32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %0\n s_nop 1\n v_cndmask_b32 %1, %1, %0, s[22:23]\n")
                         : "+v"(a), "+v"(b) :: "s20", "s21", "s22", "s23");
            break;
        case 4:  // 16 x (v_readlane -> sgpr -> v_sub using it)
            asm volatile(REP8("v_readlane_b32 s20, %0, 3\n s_nop 0\n v_sub_f32 %1, s20, %1\n v_readlane_b32 s21, %0, 5\n s_nop 0\n v_sub_f32 %1, s21, %1\n")
                         : "+v"(a), "+v"(b) :: "s20", "s21");
            break;
        case 5:  // 16 x v_pk_fma_f32 independent (2 chains)
            {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 x = {a, b}, y = {c, d};
                asm volatile(REP8("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n") : "+v"(x), "+v"(y) : "v"(f2{1.0001f, 1.f}), "v"(f2{0.5f, 0.25f}));
                a = x.x; b = x.y; c = y.x; d = y.y;
            }
            break;
        case 6:  // 16 broadcast ds_read_b64 then wait
            {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 r;
                unsigned int addr = (unsigned int)(size_t)lds;  // LDS offset
                asm volatile(REP8("ds_read_b64 %0, %1 offset:64\n ds_read_b64 %0, %1 offset:128\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(r) : "v"(addr) : "memory");
                a += r.x;
            }
            break;
        case 7:  // 16 per-lane ds_read_b32 (conflict-free) then wait
            {
                float r;
                unsigned int addr = (unsigned int)(size_t)lds + lane * 4;
                asm volatile(REP8("ds_read_b32 %0, %1 offset:256\n ds_read_b32 %0, %1 offset:512\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(r) : "v"(addr) : "memory");
                a += r;
            }
            break;
        case 8:  // 16 ds_write_b32 + 16 ds_write_b16 then wait
            {
                unsigned int addr = (unsigned int)(size_t)lds + lane * 4;
                asm volatile(REP8("ds_write_b32 %0, %1 offset:256\n ds_write_b16 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n ds_write_b16 %0, %1 offset:1024\n") "s_waitcnt lgkmcnt(0)\n" :: "v"(addr), "v"(a) : "memory");
            }
            break;
        case 9:  // dependent LDS round trip: 8 x (ds_read_b32 -> wait -> use as address)
            {
                unsigned int addr = ((unsigned int)(size_t)lds) + lane * 4;
                unsigned int r = u & 1023u;
                asm volatile(REP8("v_lshl_add_u32 %0, %0, 2, %1\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_cvt_u32_f32 %0, %0\n v_and_b32 %0, 1023, %0\n") : "+v"(r) : "v"(addr) : "memory");
                u = r;
            }
            break;
        case 10:  // 32 x v_cndmask with sgpr mask (no cmp), independent
            asm volatile(REP8("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[22:23]\n v_cndmask_b32 %3, %3, %4, s[22:23]\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.5f) : "s20", "s21", "s22", "s23");
            break;
        case 11:  // 16 x (v_cmp_e64 -> sgpr; 3 unrelated valu; v_cndmask) : software-pipelined distance
            asm volatile(REP8("v_cmp_lt_f32 s[20:21], %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %2\n v_add_f32 %2, %2, %3\n v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cmp_lt_f32 s[22:23], %1, %0\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %2\n v_add_f32 %2, %2, %3\n v_cndmask_b32 %1, %1, %0, s[22:23]\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s20", "s21", "s22", "s23");
            break;
        case 12:  // 32 x v_med3_f32 dependent
            asm volatile(REP32("v_med3_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
            break;
        case 14:  // scan-like: 16 x (ds_read_b64 bcast, 2 sub, mul, fma, cmp, cndmask, or)
            {
                const float2 *st = reinterpret_cast<const float2 *>(lds);
                unsigned int wd = 0;
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const float2 xy = st[t + (it & 1)];
                    const float dx = xy.x - a, dy = xy.y - b;
                    const float d2 = __builtin_fmaf(dx, dx, dy * dy);
                    wd |= d2 < c ? 1u << t : 0u;
                }
                u ^= wd;
            }
            break;
        case 15:  // empty loop body
            asm volatile("" : "+v"(a));
            break;
        case 13:  // 32 x v_lshl_or_b32 / v_lshl_add_u32 independent
            asm volatile(REP8("v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0\n v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0\n") : "+v"(u), "+v"(u2));
            break;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = a + b + c + d + (float)u + (float)u2;
}

int main() {
    unsigned long long *out; float *sink;
    const int blocks = 256 * 4;  // one wave per SIMD on every CU (LDS 16 KB per block -> up to 10 per CU; grid = 4 per CU)
    hipMalloc(&out, blocks * 8); hipMalloc(&sink, blocks * 64 * 4);
    const char *names[] = {"32 v_fma indep", "32 v_fma dep", "16 cmp(vcc)->cndmask + nop", "16 cmp(sgpr)->cndmask + nop", "16 readlane->v_sub",
                           "16 v_pk_fma", "16 ds_read_b64 bcast + wait", "16 ds_read_b32 + wait", "16 ds_write_b32 + 16 b16 + wait",
                           "8 dependent LDS round trips (5 instr each)", "32 v_cndmask sgpr-mask indep", "16 cmp->3 valu->cndmask", "32 v_med3 dep", "32 v_lshl_or/add", "scan-like 16 roads (C++)", "empty loop"};
    const int iters = 2000;
    for (int grid : {1, blocks}) {
        printf("grid %d blocks of one wave\n", grid);
        for (int w = 0; w < 16; w++) {
            switch (w) {
#define L(n) case n: hipLaunchKernelGGL(k<n>, dim3(grid), dim3(64), 0, 0, out, sink, iters); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15)
            }
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(grid);
            hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            printf("  %-48s %8.1f cycles / iteration\n", names[w], s / grid / iters);
        }
    }
    return 0;
}
