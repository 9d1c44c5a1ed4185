#!/bin/bash
# developer tool: where a workgroup of k_bev spends its cycles (library built with tools/stamps.sh build).
#   run on the GPU box:  tools/stamps_bev.sh [workload]
cd "$(dirname "$0")/.."
GPUDRIVE_DEV=1 GPUDRIVE_AMD_LIB=$PWD/build/expt/stamps.so python3 - "$1" <<'PY'
import sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
import bench
wl = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else "bev"
dev = torch.device("cuda", 0)
with torch.cuda.stream(torch.cuda.Stream(device=dev)):
    sim = bench.make_sim(bench.scenes_for(wl, 1024, 0), bench.params_for(wl), 64, 0)
    sim.bev_observation_tensor()
    batches = bench.action_batches(1024, 64, dev, 1234)
    bench.run_steps(sim, batches, np.arange(1024, dtype=np.int32), 10)
    torch.cuda.synchronize()
from gpudrive_lab_amd import _capi
lib = ctypes.CDLL(_capi.lib_path())
n = 8192
buf = (ctypes.c_ulonglong * (8 * n))()
rc = lib.gd_debug_read_bev_stamps(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
a = a[a[:, 0] > 0]
names = ["total", "clear + road scan", "partners", "paint (wave 0, 4 bands)", "wait for the other waves", "write-out + barrier", "entities", "roads"]
print("workload", wl, "workgroups", len(a), "rc", rc)
for i, nm in enumerate(names):
    print("  %-26s mean %10.0f  min %10.0f  max %10.0f" % (nm, a[:, i].mean(), a[:, i].min(), a[:, i].max()))
PY
