#!/bin/bash
# developer tool: build the library with in-kernel cycle stamps (-DGD_STAMPS, build/expt/stamps.so, built HERE
# before gpurun) and print where a workgroup of the road kernel spends its cycles.
#   build:  tools/stamps.sh build      run on the GPU box:  tools/stamps.sh run [workload]
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build/expt && cd gpudrive_lab_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DGD_STAMPS --offload-arch=gfx950 -shared -o ../../build/expt/stamps.so kernels.hip map_obs.hip bev_lidar.hip pack_obs.hip episode.hip engine.cpp scene.cpp scene_cache.cpp
  exit $?
fi
GPUDRIVE_DEV=1 GPUDRIVE_AMD_LIB=$PWD/build/expt/stamps.so python3 - "$2" <<'PY'
import sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
import bench
wl = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else "synthetic"
args = type("A", (), dict(worlds=1024, agents=64, knn_order=0))()
dev = torch.device("cuda", 0)
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    sim = bench.make_sim(bench.scenes_for(wl, 1024, 0), bench.params_for(wl), 64, 0)
    batches = bench.action_batches(1024, 64, dev, 1234)
    bench.run_steps(sim, batches, np.arange(1024, dtype=np.int32), 30)
    torch.cuda.synchronize()
from gpudrive_lab_amd import _capi
lib = _capi.lib()
n = 2048
buf = (ctypes.c_ulonglong * (8 * n))()
rc = lib.gd_debug_read_stamps(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
a = a[a[:, 0] > 0]
np.save("gpurun_out/stamps_%s.npy" % wl, a)  # row = workgroup in launch order; tools/dispatch_sim.py replays the dispatch
names = ["total", "init(fill+make_heap)", "scan", "drain", "rounds", "scans", "vmcnt wait at scan", "vmcnt wait at round top"]
print("workload", wl, "workgroups", len(a), "rc", rc)
for i, nm in enumerate(names):
    print("  %-22s mean %10.0f  min %10.0f  max %10.0f" % (nm, a[:, i].mean(), a[:, i].min(), a[:, i].max()))
print("  cycles per round %.0f   cycles per scanned chunk %.0f" % (a[:, 3].sum() / a[:, 4].sum(), a[:, 2].sum() / max(a[:, 5].sum(), 1)))
PY
