// developer microbenchmark: cycles per Heap::replace_top (one wave per workgroup, the real LDS footprint, every CU loaded).
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -I gpudrive_lab_amd/csrc [-DGD_MAP_OBS_AW=..] -o tools/ubench/round tools/ubench/round.hip
#include "../../gpudrive_lab_amd/csrc/map_obs.hip"
#include <cstdio>
#include <vector>

namespace gd { namespace {
__global__ __launch_bounds__(64) void k_round(unsigned long long *out, float *sink, int iters, int mode, const float2 *rxy) {
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[RING * AW * 4 + SLOTS * AW * 6];
    float *s_keys = reinterpret_cast<float *>(s_buf + RING * AW * 4) - AW;
    unsigned short *s_idx = reinterpret_cast<unsigned short *>(s_buf + RING * AW * 4 + SLOTS * AW * 4) - AW;
    const int lane = threadIdx.x, col = lane % AW, sub = lane / AW;
    const Heap heap{s_keys + col, s_idx + idx_col(col)};
    unsigned int rng = 12345u + 977u * (blockIdx.x * 64 + col);
    auto rnd = [&]() -> float { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) * (1.0f / 16777216.0f); };
    if (sub == 0) {
        for (int g = 1; g <= K; g++) heap.set(g, 100.f + 900.f * rnd(), (unsigned int)g);
        s_keys[(K + 1) * AW + col] = -1.f;
        s_keys[(K + 2) * AW + col] = -1.f;
    }
    wave_sync();
    heap.make(sub);
    Heap::Top top;
    heap.load(top);
    wave_sync();
    unsigned int *s_ring = reinterpret_cast<unsigned int *>(s_buf);
    if (sub == 0)
        for (int c = 0; c < RING; c++) { rng = rng * 1664525u + 1013904223u; unsigned int a = rng; rng = rng * 1664525u + 1013904223u; s_ring[c * AW + col] = a & rng & 0x77777777u; }
    wave_sync();
    Drain dr;
    const unsigned long long agents = __ballot(sub == 0);
    unsigned long long pend = 0, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode < 2) {
#pragma clang loop unroll(disable)
        for (int it = 0; it < iters; it++) {
            const float y = top.tk[1] * (0.2f + 0.79f * rnd());
            if (mode == 0 ? true : (rnd() < 0.6f)) heap.replace_top(sub == 0, y, (unsigned int)(it & 0xffff), top);
        }
    } else {
        int head = 16;
#pragma clang loop unroll(disable)
        for (int it = 0; it < iters; it++) {
            // the kernel's per-round bookkeeping: scan decision, exit test
            const unsigned int slot = (unsigned int)head & (RING - 1);
            if (__ballot((dr.nz >> slot) & 1u) == 0ull && __popcll(agents & ~pend) >= 8) { dr.nz = 0xffffu; head += (it & 1); }
            drain_round(heap, top, dr, s_ring + col, rxy, head, 0.f, 0.f, 1.f, 0.f, sub == 0);
            pend = __ballot(dr.pending()) & agents;
            acc += pend;
            if (mode == 3 && top.tk[1] < 400.f) top.tk[1] = 900.f;  // keep the live test passing (timing only)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 12345ull) sink[0] = 1.f;
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = top.tk[1] + top.lk + s_keys[(1 + (lane & 63)) * AW + col];
}
}}

int main() {
    unsigned long long *out; float *sink;
    const int per_cu = 65536 * 2 / (16 * gd::AW * 4 + gd::SLOTS * gd::AW * 6) ;  // workgroups per CU by LDS (rough)
    (void)hipMalloc(&out, 8192 * 8); (void)hipMalloc(&sink, 8192 * 64 * 4);
    float2 *rxy; (void)hipMalloc(&rxy, 4096 * 8);
    { std::vector<float2> h(4096); unsigned int r = 1; for (auto &v : h) { r = r * 1664525u + 1013904223u; v.x = (float)(r >> 8) / 16777216.f * 25.f; r = r * 1664525u + 1013904223u; v.y = (float)(r >> 8) / 16777216.f * 25.f; } (void)hipMemcpy(rxy, h.data(), 4096 * 8, hipMemcpyHostToDevice); }
    const int iters = 2000;
    for (int grid : {1, 256 * 65536 * 2 / (int)(16 * gd::AW * 4 + gd::SLOTS * gd::AW * 6) / 2}) {
        for (int mode : {0, 1, 2, 3}) {
            hipLaunchKernelGGL(gd::k_round, dim3(grid), dim3(64), 0, 0, out, sink, iters, mode, rxy);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h(grid);
            (void)hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            printf("AW %d grid %5d mode %d (0,1: replace_top alone; 2,3: whole drain round): %8.1f cycles\n", gd::AW, grid, mode, s / grid / iters);
        }
    }
    (void)per_cu;
    return 0;
}
