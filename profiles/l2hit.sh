#!/bin/bash
# L2 hit/miss + EA read requests for the apply kernel under the current env knobs: bash profiles/l2hit.sh <L>
export TMPDIR=/tmp
L=${1:-30}
OUT=$PWD/gpurun_out/l2hit_$$
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT -- python3 bench.py --L $L --steps 4 --warmup 1 --no-cpu > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "apply" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur=[]
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "apply" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
h=sum(acc["TCC_HIT_sum"])/max(len(acc["TCC_HIT_sum"]),1); m=sum(acc["TCC_MISS_sum"])/max(len(acc["TCC_MISS_sum"]),1); e=sum(acc["TCC_EA0_RDREQ_sum"])/max(len(acc["TCC_EA0_RDREQ_sum"]),1)
print("L=$L LS=${SD_SUFFIX_BITS:-def} CH=${SD_XCD_CHUNK:-def}: hit=%.3g miss=%.3g hitrate=%.3f EA_rd=%.3g  kernel_us=%.0f" % (h,m,h/max(h+m,1),e,(sum(dur)/max(len(dur),1))/1e3))
PY
