#!/bin/bash
# developer tool: SQ / TCC counters of the step kernels, one rocprofv3 --pmc pass per counter group
# (run on the GPU box through gpurun):  tools/pmc.sh <tag> [bench args...]
# Output: gpurun_out/pmc_<tag>/summary.json, keyed by kernel name (template arguments kept).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
GROUPS_=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
 "GRBM_GUI_ACTIVE GRBM_COUNT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for G in "${GROUPS_[@]}"; do
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 bench.py --steps 3 --warmup 1 --roofline-steps 1 --no-cpu-baseline "$@" > $OUT/bench_g$i.log 2>$OUT/err_g$i.log
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections,json,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if not any(n in k for n in ("k_map_obs","k_map_rows","k_world_step","k_lidar","k_bev","k_pack_obs","k_episode_step")): continue
        m=re.search(r"(k_[a-z_]+(<[^>]*>)?)",k)
        name=m.group(1) if m else k
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for extra in ("VGPR_Count","Accum_VGPR_Count","SGPR_Count","LDS_Block_Size","Scratch_Size","Workgroup_Size","Grid_Size"):
            if extra in row and row[extra] not in ("",None): agg[name]["_"+extra]=[float(row[extra])]
summary={k:{c:(sum(x)/len(x)) for c,x in sorted(v.items())} for k,v in agg.items()}
for k,v in summary.items(): v["_launches_seen"]=max(len(x) for x in agg[k].values())
json.dump(summary, open("$OUT/summary.json","w"), indent=1)
for k,v in summary.items():
    print(k)
    for c,val in v.items(): print("   %-28s %.6g"%(c,val))
PY
