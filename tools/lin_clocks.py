#!/usr/bin/env python3
"""developer tool: where a wave of k_map_obs_linear spends its time.  `build` (here, before gpurun) copies csrc to build/linclk_src,
inserts s_memtime stamps at the kernel's phase boundaries (lane 0 of every wave; sums in a __device__ array, read through an
extra exported function), and builds build/expt/expt_linclk.so; `run [workload ...]` (on the GPU box) steps the workload and prints
the shares.  The product sources are not touched.
    python tools/lin_clocks.py build && gpurun -- python tools/lin_clocks.py run synthetic_linear waymo_linear ppo_default"""
import ctypes, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpudrive_lab_amd", "csrc"), os.path.join(ROOT, "build", "linclk_src")
LIB = os.path.join(ROOT, "build", "expt", "expt_linclk.so")
NAMES = ("header + stamp test", "cull", "scan", "gather (request to arrival)", "row arithmetic", "staging + stores", "stamp + loop")


def sub(s, old, new):
    assert s.count(old) == 1, old
    return s.replace(old, new)


def build():
    shutil.rmtree(DST, ignore_errors=True)
    shutil.copytree(SRC, DST, ignore=shutil.ignore_patterns("*.o", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(ROOT, "build", "include"), dirs_exist_ok=True)
    p = os.path.join(DST, "map_obs_linear.hip")
    s = open(p).read()
    # (sums kept in registers, one sharded add per wave and phase at the end: 600 thousand adds to eight addresses per step took
    # milliseconds and were all the stamps then measured)
    s = sub(s, "template <int A_T, bool PACK>\n__global__", "__device__ unsigned long long g_lin_clk[512 * 8];\n#define LCLK(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); "
            "acc_[k] += t_ - t_last; t_last = t_; } while (0)\ntemplate <int A_T, bool PACK>\n__global__")
    s = sub(s, "    int skipped = 0;\n    Work nx = load_work(wave);", "    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n    int skipped = 0;\n    Work nx = load_work(wave);")
    s = sub(s, "    if (lane == 0 && skipped) atomicAdd(d.stat_skipped", "    if (lane == 0)\n        for (int k_ = 0; k_ < 8; k_++) atomicAdd(&g_lin_clk[(((blockIdx.x * 4u + wave) & 511u) << 3) + k_], acc_[k_]);\n    if (lane == 0 && skipped) atomicAdd(d.stat_skipped")
    s = sub(s, "        if (p.e < 0) continue;  // filler entry", "        unsigned long long t_last = __builtin_amdgcn_s_memtime();\n        if (p.e < 0) continue;  // filler entry")
    s = sub(s, "        int count = 0;\n        const float dxo", "        LCLK(0);\n        int count = 0;\n        const float dxo")
    s = sub(s, "            wave_sync();\n            // SCAN of the surviving blocks", "            wave_sync();\n            LCLK(1);\n            // SCAN of the surviving blocks")
    s = sub(s, "        wave_sync();\n        // ---- the agent's K rows ----", "        wave_sync();\n        LCLK(2);\n        // ---- the agent's K rows ----")
    # gathers: wait for them explicitly so that their latency is not charged to the row arithmetic
    s = sub(s, "#pragma unroll\n            for (int g = 0; g < GH; g++) {\n                const int pz = h + g;", "            asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n            LCLK(3);\n#pragma unroll\n            for (int g = 0; g < GH; g++) {\n                const int pz = h + g;")
    s = sub(s, "                const int nrows = min(64, K - pz * 64);\n", "                const int nrows = min(64, K - pz * 64);\n                LCLK(4);\n")
    s = sub(s, "        if (lane == 0)\n            d.pose_stamp[i] =", "        LCLK(5);\n        acc_[7] += 1ull;\n        if (lane == 0)\n            d.pose_stamp[i] =")
    s = s.rstrip("\n") + """

extern "C" void gd_lin_clocks_read(unsigned long long *out) {  // and zero them
    static unsigned long long h[512 * 8], z[512 * 8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gd::g_lin_clk), sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(gd::g_lin_clk), z, sizeof(z));
    for (int k = 0; k < 8; k++) out[k] = 0;
    for (int q = 0; q < 512; q++)
        for (int k = 0; k < 8; k++) out[k] += h[q * 8 + k];
}
"""
    # the array lives in the anonymous namespace of the kernel: move it out so that the reader can name it
    s = s.replace("__device__ unsigned long long g_lin_clk[512 * 8];\n", "", 1)
    s = sub(s, "namespace gd {\n", "namespace gd {\n__device__ unsigned long long g_lin_clk[512 * 8];\n") if s.count("namespace gd {\n") == 1 else s
    open(p, "w").write(s)
    srcs = subprocess.check_output(["sed", "-n", "s/^SRCS := //p", os.path.join(DST, "Makefile")], text=True).split()
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                           "-shared", "-o", LIB] + srcs, cwd=DST)
    print("built", LIB)


def run(workloads):
    os.environ.setdefault("GPUDRIVE_MAX_AGENTS", "64")
    os.environ["GPUDRIVE_DEV"] = "1"
    os.environ["GPUDRIVE_AMD_LIB"] = LIB
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import bench
    lib = ctypes.CDLL(LIB)
    dev = torch.device("cuda", 0)
    for wl in workloads or ["synthetic_linear"]:
        name, order, agents = bench.split_workload(wl)
        agents = agents or 64
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            sim = bench.make_sim(bench.scenes_for(name, 1024, 0, agents), bench.params_for(name), agents, 0, knn_order=order)
            batches = bench.action_batches(1024, agents, dev, seed=1234)
            act = sim.action_tensor().to_torch()
            buf = (ctypes.c_ulonglong * 8)()
            for k in range(40):
                act.copy_(batches[k % 8]); sim.step()
                if k == 9:
                    torch.cuda.synchronize(); lib.gd_lin_clocks_read(buf)
            torch.cuda.synchronize()
            lib.gd_lin_clocks_read(buf)
            c = np.array(list(buf), np.float64)
            n = max(c[7], 1.0)
            per = c[:6] / n
            print("%-18s %7.0f agents written per step, ticks per agent %7.0f (100 MHz: %5.2f us): %s" % (
                wl, n / 30, per.sum(), per.sum() / 100.0, ", ".join("%s %.0f (%.0f %%)" % (nm, v, 100 * v / per.sum()) for nm, v in zip(NAMES, per))))
            sim.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "run" else sys.argv[1:])
