#!/bin/bash
# developer tool: SQ counters of the rank-path kernels, one rocprofv3 --pmc pass per counter group (run on the GPU box):
#   gpurun -- bash tools/pmc_knn.sh <tag> [bench args...]      (EXPT=<name> selects build/expt/expt_<name>.so)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
if [ -n "$EXPT" ]; then export GPUDRIVE_DEV=1 GPUDRIVE_AMD_LIB=$GRAFT_REPO_ROOT/build/expt/expt_$EXPT.so; fi
OUT=gpurun_out/pmcknn_$TAG
rm -rf $OUT; mkdir -p $OUT
GROUPS_=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
)
i=0
for G in "${GROUPS_[@]}"; do
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 bench.py --steps 4 --warmup 2 --spin-ms 0 --no-align --no-cpu-baseline "$@" > $OUT/bench_g$i.log 2>$OUT/err_g$i.log
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections,json,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        m=re.search(r"(k_[a-z_]+(<[^>]*>)?)",k)
        if not m: continue
        agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name,v in agg.items():
    c={a:sum(x)/len(x) for a,x in v.items()}
    w=max(c.get("SQ_WAVES",1),1)
    if c.get("SQ_INSTS_VALU",0)/w < 200: continue
    print("%-28s waves %6d | per wave: VALU %6.0f SALU %5.0f LDS %5.0f VMEM %4.0f SMEM %3.0f | VALU busy/wavecyc %.2f wait_any %.2f wait_inst %.2f | LDS conflict/idx %.2f lds_idx %.1fM busy %.2fM" % (
        name, w, c.get("SQ_INSTS_VALU",0)/w, c.get("SQ_INSTS_SALU",0)/w, c.get("SQ_INSTS_LDS",0)/w, c.get("SQ_INSTS_VMEM",0)/w, c.get("SQ_INSTS_SMEM",0)/w,
        c.get("SQ_ACTIVE_INST_VALU",0)/max(c.get("SQ_WAVE_CYCLES",1),1), c.get("SQ_WAIT_ANY",0)/max(c.get("SQ_WAVE_CYCLES",1),1), c.get("SQ_WAIT_INST_ANY",0)/max(c.get("SQ_WAVE_CYCLES",1),1),
        c.get("SQ_LDS_BANK_CONFLICT",0)/max(c.get("SQ_LDS_IDX_ACTIVE",1),1), c.get("SQ_LDS_IDX_ACTIVE",0)/1e6, c.get("SQ_BUSY_CYCLES",0)/1e6))
json.dump({k:{a:sum(x)/len(x) for a,x in v.items()} for k,v in agg.items()}, open("$OUT/summary.json","w"), indent=1)
PY
