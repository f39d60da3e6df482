#!/bin/bash
# Instruction-mix and stall counters of the headline kernel (one --pmc pass per counter group; kernel-trace only).
# Usage (through gpurun, repo root): bash scripts/pmc_instmix.sh [extra bench args]
OUT=$PWD/gpurun_out/instmix
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
ARGS="$REPO/bench.py --steps 2 --warmup 1 --no-cpu --no-mix $@"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python3 $ARGS > $OUT/p$i.log 2>&1 && echo "pass $i done" || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob("$OUT/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "k_sos" not in k: continue
        key=(k.split("(")[0],r["Counter_Name"])
        acc[key][0]+=1; acc[key][1]+=float(r["Counter_Value"])
with open("$OUT/summary.txt","w") as o:
    for (k,c),(n,v) in sorted(acc.items()):
        o.write(f"{k:60s} {c:34s} dispatches {n:4d} mean {v/n:.6e}\n")
print(open("$OUT/summary.txt").read())
PY
