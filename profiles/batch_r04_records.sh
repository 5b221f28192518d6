#!/bin/bash
# refresh of the r04 bench records on the final build (no tests, no profiler): bash profiles/batch_r04_records.sh
set -u
OUT=$PWD/gpurun_out/r04_records
mkdir -p $OUT
python profiles/variants_bench.py 2>/dev/null | grep "^{" > $OUT/variants.jsonl; cat $OUT/variants.jsonl | cut -c1-160
python profiles/recursion_bench.py 2>/dev/null | grep "^{" > $OUT/recursion.jsonl; cut -c1-200 $OUT/recursion.jsonl
python profiles/smallL_bench.py 2>/dev/null | grep "^{" > $OUT/smallL.jsonl; python profiles/smallL_sqw_bench.py 2>/dev/null | grep "^{" >> $OUT/smallL.jsonl; cut -c1-170 $OUT/smallL.jsonl
python profiles/config_bench.py 2 3 4 2>/dev/null | grep "^{" > $OUT/configs.jsonl; cut -c1-220 $OUT/configs.jsonl
python profiles/general_bonds_bench.py 28 2>/dev/null | grep "^{" > $OUT/general_bonds.jsonl; python profiles/general_bonds_bench.py 24 2>/dev/null | grep "all pairs" >> $OUT/general_bonds.jsonl; cut -c1-170 $OUT/general_bonds.jsonl
for cfg in "36 9" "32 8" "34 12" "40 10" "44 8" "30 6"; do set -- $cfg; python profiles/dilute_sector_bench.py $1 $2 2>/dev/null | grep "^{"; done > $OUT/dilute.jsonl; cut -c1-170 $OUT/dilute.jsonl
python bench.py --steps 20 --warmup 5 --no-cpu --dtype f64 2>/dev/null > $OUT/bench_f64_L32.json; cut -c1-200 $OUT/bench_f64_L32.json
