#!/usr/bin/env python3
"""Developer tool: replay the dispatch of the road kernel's workgroups (gpurun_out/stamps_<workload>.npy from
tools/stamps.sh run: one row per workgroup in launch order, column 0 = its cycles) onto the 1024 wave slots the LDS
allows, first-free-slot like the hardware dispatcher, in launch order and in longest-first order."""
import heapq
import sys

import numpy as np


def makespan(costs, slots=1024):
    free = [0.0] * slots
    heapq.heapify(free)
    end = 0.0
    for c in costs:
        t = heapq.heappop(free) + c
        end = max(end, t)
        heapq.heappush(free, t)
    return end


if __name__ == "__main__":
    a = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps_synthetic.npy")
    c = a[:, 0]
    print("workgroups %d  mean %.0f  min %.0f  max %.0f cycles" % (len(c), c.mean(), c.min(), c.max()))
    print("perfect balance      %.0f" % (c.sum() / 1024))
    print("launch order         %.0f" % makespan(c))
    print("longest first        %.0f" % makespan(np.sort(c)[::-1]))
    # per world (two consecutive workgroups) longest first
    w = c[: len(c) // 2 * 2].reshape(-1, 2)
    order = np.argsort(-w.sum(1), kind="stable")
    print("longest world first  %.0f" % makespan(w[order].reshape(-1)))
