#!/bin/bash
# round-2 measurement session (GPU box, repository root): everything DESIGN.md section 5 quotes
set -e
ROOT=$(pwd); O=$ROOT/gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o runc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof.err
BTLBF_BENCH_NO_MISS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o runc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_f.json 2> $O/pmc_f.err
BTLBF_BENCH_NO_MISS=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o runc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_w.json 2> $O/pmc_w.err
cd $ROOT
python3 tools/pmc_summary.py --round r02 --kmers 12e9 --fetch $O/pmc_f --write $O/pmc_w --stats $O/prof > $O/pmc_summary.log
cp profiles/traffic.json $O/traffic.json; mkdir -p $O/profiles_r02; cp profiles/r02/* $O/profiles_r02/
tools/diag_sq.sh r02 39 20000000
python3 tools/c4_cost.py > $O/c4_cost.json 2> $O/c4_cost.err
python3 tools/config_bench.py > $O/config_bench.json 2> $O/config_bench.err
echo session done
