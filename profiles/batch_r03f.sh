#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03f
mkdir -p $OUT
# four ranks on this one GPU (gloo), popcount cells with the two-hop relays forced on: the N > 1 control flow of bench.py incl. the relay self-check
SD_BENCH_BACKEND=gloo SD_RELAY=2 SD_RELAY_MIN=0 SD_RELAY_CHUNKS=4 timeout -k 10 500 python bench.py --gpus 4 --L 26 --steps 5 --warmup 1 > $OUT/bench_gloo4.json 2> $OUT/bench_gloo4.err; echo "bench gloo4 rc=$?"; cut -c1-1500 $OUT/bench_gloo4.json; tail -3 $OUT/bench_gloo4.err
SD_AUX_L=32 timeout -k 10 300 python profiles/aux_bench.py > $OUT/aux_L32.jsonl 2> $OUT/aux.err; cat $OUT/aux_L32.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shard_trace -- python3 profiles/shard_kernel_bench.py 32 8 3 > $OUT/shard_trace.log 2>&1; tail -2 $OUT/shard_trace.log
python3 profiles/summarize.py $OUT/shard_trace > $OUT/shard_rocprof_summary.txt 2>&1; grep -E "k_apply|k_pack" $OUT/shard_rocprof_summary.txt | head -12
