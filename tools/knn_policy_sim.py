#!/usr/bin/env python3
"""Developer tool (CPU, numpy): how many lock-step drain rounds a wave of the road kernel needs under a given scan /
drain policy, on the bench scene.  Per agent the running K-th distance decides which roads are true inserts; a policy
decides when chunks of 32 roads are scanned (with the agent's threshold of that moment) and how many candidates a lane
may try per round.  Used to size the candidate ring and the scan trigger.

    python tools/knn_policy_sim.py [seed] [step]
"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpudrive_lab_amd import synth  # noqa: E402

K = 200


def scene_arrays(seed, t=0):
    sc = synth.make_scene(seed)
    ag = np.array([[o["position"][t]["x"], o["position"][t]["y"]] for o in sc["objects"]])
    mids = []
    for r in sc["roads"]:
        g = np.array([[p["x"], p["y"]] for p in r["geometry"]])
        mids.append((g[:-1] + g[1:]) / 2)
    return ag, np.concatenate(mids)


def true_inserts(keys):
    h = [-k for k in keys[:K]]
    heapq.heapify(h)
    n = 0
    for k in keys[K:]:
        if k < -h[0]:
            heapq.heapreplace(h, -k)
            n += 1
    return n


def ring_rounds(keys, ring=16, idle_trigger=16, tries=1, chunk=32):
    """keys [agents, roads] -> (rounds, scans, candidates per agent, false candidates per agent)."""
    A, R = keys.shape
    heaps = [[-k for k in keys[a, :K]] for a in range(A)]
    for h in heaps:
        heapq.heapify(h)
    nch = (R - K + chunk - 1) // chunk
    queues = [[] for _ in range(A)]
    head = rounds = scans = cand = wasted = 0
    while True:
        while head < nch:
            mintail = min((q[0][0] if q else head) for q in queues)
            if head - mintail >= ring or sum(1 for q in queues if not q) < idle_trigger:
                break
            s, e = K + head * chunk, min(R, K + (head + 1) * chunk)
            for a in range(A):
                idx = np.nonzero(keys[a, s:e] < -heaps[a][0])[0]
                if len(idx):
                    queues[a].append([head, list(idx + s)])
                    cand += len(idx)
            head += 1
            scans += 1
        if not any(queues):
            if head >= nch:
                break
            continue
        rounds += 1
        for a in range(A):
            t = 0
            while queues[a] and t < tries:
                lst = queues[a][0][1]
                r = lst.pop(0)
                if not lst:
                    queues[a].pop(0)
                t += 1
                if keys[a, r] < -heaps[a][0]:
                    heapq.heapreplace(heaps[a], -keys[a, r])
                    break
                wasted += 1
    return rounds, scans, cand / A, wasted / A


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    t = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ag, rd = scene_arrays(seed, t)
    keys = ((rd[None] - ag[:, None]) ** 2).sum(-1)
    ins = [true_inserts(k) for k in keys]
    print("seed %d step %d: true inserts per agent mean %.0f max %d" % (seed, t, np.mean(ins), max(ins)))
    for group in (64, 32, 16):
        for ring in (8, 16, 32):
            for tries in (1, 2):
                rs = [ring_rounds(keys[s:s + group], ring=ring, idle_trigger=max(1, group // 2), tries=tries)[0]
                      for s in range(0, 64, group)]
                print("  %2d agents per wave, ring %2d chunks, %d tries per round: rounds per wave mean %.0f max %d"
                      % (group, ring, tries, np.mean(rs), max(rs)))
