#!/bin/bash
# round 4 final records: GPU suite, bench (c128 with CPU baseline, f64), rocprofv3 trace + PMC passes, recursion / variant / small-L /
# config benches, same-box A/B of a build variant, 4-rank gloo rehearsal of bench.py with relays
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_final
mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -6 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
python bench.py > $OUT/bench_c128.json 2> $OUT/bench.err; cut -c1-330 $OUT/bench_c128.json
python bench.py --steps 20 --warmup 5 --no-cpu --dtype f64 > $OUT/bench_f64.json 2>> $OUT/bench.err; cut -c1-330 $OUT/bench_f64.json
bash profiles/run_profile.sh r04 32 > $OUT/profile.log 2>&1
cp gpurun_out/prof_r04/summary.txt $OUT/rocprof_summary.txt; cp gpurun_out/prof_r04/traffic_latest.json $OUT/ 2>/dev/null; tail -2 $OUT/rocprof_summary.txt | cut -c1-400
python profiles/variants_bench.py > $OUT/variants.jsonl 2>&1; cat $OUT/variants.jsonl
python profiles/recursion_bench.py > $OUT/recursion.jsonl 2>&1; cut -c1-260 $OUT/recursion.jsonl
python profiles/smallL_bench.py > $OUT/smallL.jsonl 2>&1; cut -c1-200 $OUT/smallL.jsonl
python profiles/smallL_sqw_bench.py > $OUT/smallL_sqw.jsonl 2>&1; cat $OUT/smallL_sqw.jsonl
python examples/kpm_sqw.py > $OUT/example_kpm_sqw.txt 2>&1; cat $OUT/example_kpm_sqw.txt
python profiles/groundstate_bench.py 28 100 > $OUT/groundstate_L28.jsonl 2>&1; cat $OUT/groundstate_L28.jsonl
python profiles/config_bench.py 2 3 4 > $OUT/configs.jsonl 2>&1; cut -c1-300 $OUT/configs.jsonl
python profiles/general_bonds_bench.py 28 > $OUT/general_bonds.jsonl 2>&1; python profiles/general_bonds_bench.py 24 2>/dev/null | grep 'all pairs' >> $OUT/general_bonds.jsonl; cut -c1-200 $OUT/general_bonds.jsonl

env SD_BENCH_BACKEND=gloo SD_RELAY=2 SD_RELAY_MIN=0 python bench.py --gpus 4 --L 28 --steps 5 --warmup 2 --no-cpu > $OUT/bench_gloo4.json 2> $OUT/bench_gloo4.err; cut -c1-300 $OUT/bench_gloo4.json; tail -3 $OUT/bench_gloo4.err
