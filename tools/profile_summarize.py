#!/usr/bin/env python3
"""Summaries of one tools/profile.sh run: python3 tools/profile_summarize.py gpurun_out/prof_<tag>"""
import csv,glob,collections,json,re
import sys,os
out=sys.argv[1]
def newest(pattern):
    """Every file of the run.  Run this on the GPU box (tools/profile.sh does): the local gpurun_out/ accumulates the files
    of every earlier call next to the latest one, and box clocks differ, so they cannot be told apart here."""
    files=glob.glob(pattern, recursive=True)
    dirs={}
    for f in files: dirs.setdefault(os.path.dirname(f), []).append(f)
    if any(len(v)>1 for v in dirs.values()):
        raise SystemExit("several runs' files under %s: summarise on the GPU box" % out)
    return files
KERNELS=("k_map_obs_set","k_map_obs","k_map_rows","k_world_step","k_lidar","k_bev","k_pack_obs","k_episode_step","k_reset_worlds")
def kname(k):
    m=re.search(r"(k_[a-z_]+(<[^>]*>)?)",k)
    return m.group(1) if m else None
rows=[]
for f in newest(out+"/trace/**/*kernel_stats.csv"):
    rows+=list(csv.DictReader(open(f)))
with open(out+"/kernel_stats_summary.csv","w") as fh:
    if rows:
        w=csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:8]:
    print({k:(v[:64] if isinstance(v,str) else v) for k,v in r.items() if k in ("Name","Calls","AverageNs","MinNs","MaxNs","Percentage")})
# occupancy facts per kernel: launch geometry and LDS from the kernel trace, registers / spills / waves per SIMD from the
# COMPILER (profiles/rNN_compiler_resources.json, written by `tools/resources.py --json-all` for this source stamp).
# rocprofv3's VGPR_Count is half the allocation on gfx950 (k_world_step: 48 reported, 95 allocated): rounds 2-4 took it at
# face value and reported eight waves per SIMD where the compiler says five.
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
comp={}
for f in sorted(glob.glob(os.path.join(ROOT,"profiles","r*_compiler_resources.json")), reverse=True):
    comp=json.load(open(f)); comp["file"]=os.path.basename(f); break
def compiler_row(n):
    norm=lambda x: re.sub(r"\s+","",x)
    for k,r in comp.get("kernels",{}).items():
        if norm(k)==norm(n): return r
    return None
occ={}
for f in newest(out+"/trace/**/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        n=kname(row.get("Kernel_Name",""))
        if not n or n in occ: continue
        g=lambda k: int(float(row.get(k,0) or 0))
        vg=g("VGPR_Count")+g("Accum_VGPR_Count"); lds=g("LDS_Block_Size")
        wg=max(1,g("Workgroup_Size_X"))*max(1,g("Workgroup_Size_Y"))*max(1,g("Workgroup_Size_Z"))
        waves_wg=max(1,(wg+63)//64)
        alloc=-(-max(vg,1)//8)*8
        by_vgpr=min(8,512//alloc)
        by_lds=(160*1024//lds)*waves_wg/4.0 if lds else 8
        c=compiler_row(n)
        if c:
            by_vgpr=c["waves_per_simd"]
            occ[n]=dict(vgpr=c["vgpr"],agpr=c["agpr"],sgpr=c["sgpr"],vgpr_spills=c["vgpr_spills"],sgpr_spills=c["sgpr_spills"],
                        registers_from="compiler (%s, source stamp %s)" % (comp.get("file"), comp.get("source_stamp")),
                        rocprofv3_vgpr_count=g("VGPR_Count"))
        else:
            occ[n]=dict(vgpr=None,agpr=None,sgpr=g("SGPR_Count"),registers_from="no compiler record for this kernel; rocprofv3's VGPR_Count "
                        "(half the allocation on gfx950) is %d" % g("VGPR_Count"),rocprofv3_vgpr_count=g("VGPR_Count"))
            by_vgpr=min(8,512//(-(-max(2*vg,1)//8)*8))
        grid=max(1,g("Grid_Size_X"))*max(1,g("Grid_Size_Y"))*max(1,g("Grid_Size_Z"))
        occ[n].update(lds_bytes_per_workgroup=lds,workgroup_size=wg,grid_size=grid,waves_per_simd_by_registers=by_vgpr,
                      waves_per_simd_by_lds=min(8,by_lds),waves_per_simd=min(8,by_vgpr,by_lds),
                      waves_per_simd_the_grid_provides=round(grid/64.0/1024.0,2))
json.dump(occ, open(out+"/occupancy.json","w"), indent=1)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in newest(out+"/pmc_*/**/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n=kname(row["Kernel_Name"])
        if n: agg[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
# launches that return at once (the gated reset passes of the learner-side loop find no finished world; k_map_obs behind the
# rank replay when no group needs the fallback) would drag the per-launch means down.  Only those two cases are treated:
# a kernel HAS such launches when it is the fallback kernel or the run is rl_loop, its shortest dispatch in the trace is
# under 12 us and its longest more than ten times that; its PMC means then cover only the dispatches with at least 5 % of
# the kernel's largest FETCH / WRITE value, its trace average only those of at least 20 % of the longest
durs=collections.defaultdict(list)
for f in newest(out+"/trace/**/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        n=kname(row.get("Kernel_Name",""))
        if n: durs[n].append(int(row["End_Timestamp"])-int(row["Start_Timestamp"]))
has_empty={n: (min(x)<12000 and max(x)>10*min(x)) and (n.startswith("k_map_obs<") or "rl_loop" in out) for n,x in durs.items()}
for n,v in agg.items():
    for c,x in list(v.items()):
        if c in ("FETCH_SIZE","WRITE_SIZE") and x and has_empty.get(n):
            top=max(x); v[c]=[y for y in x if y>=0.05*top] or x
def working(n,x): return [y for y in x if y>=0.2*max(x)] if has_empty.get(n) else x
with open(out+"/kernel_work_summary.json","w") as fh:
    json.dump({n:dict(launches=len(x), working_launches=len(working(n,x)),
                      avg_us_all=sum(x)/len(x)/1e3, avg_us_working=sum(working(n,x))/len(working(n,x))/1e3)
               for n,x in sorted(durs.items())}, fh, indent=1)
traffic={k:{c:{"mean":sum(x)/len(x),"n":len(x)} for c,x in v.items() if c in ("FETCH_SIZE","WRITE_SIZE")} for k,v in agg.items()}
sq={k:{c:sum(x)/len(x) for c,x in sorted(v.items()) if c.startswith("SQ_")} for k,v in agg.items()}
json.dump(traffic, open(out+"/pmc_traffic_summary.json","w"), indent=1)
json.dump(sq, open(out+"/pmc_sq_summary.json","w"), indent=1)
print(json.dumps({k:{c:round(x["mean"]) for c,x in v.items()} for k,v in traffic.items()}))
