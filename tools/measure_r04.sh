#!/bin/bash
# round-4 measurement session (GPU box, repository root), in parts so that each fits one gpurun call:
#   tools/measure_r04.sh bench   -> contract bench (with side_configs), rocprofv3 kernel stats, PMC traffic
#   tools/measure_r04.sh sq      -> SQ counters of pass A: C2, spaced seeds (C5), ragged layout, 3*2^37 bits
#   tools/measure_r04.sh sq3x | sqspaced | sqbins -> one of those shapes alone / 512 against 256 level-0 bins at 2^37 bits
set -e
ROOT=$(pwd); O=$ROOT/gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
sq_pass() { # tag, quick_bench arguments...
	tag=$1; shift
	i=0
	for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
	           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
	           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
		i=$((i+1)); rm -rf $O/sq_${tag}_$i
		(cd /tmp && rocprofv3 --pmc $grp --kernel-trace -d $O/sq_${tag}_$i -o r -- python3 $ROOT/tools/quick_bench.py "$@" > $O/sq_${tag}_$i.log 2>&1)
	done
	python3 tools/sq_counters.py $O/sq_${tag}_1 $O/sq_${tag}_2 $O/sq_${tag}_3 > $O/sq_counters_${tag}.txt
	rm -rf $O/sq_${tag}_1 $O/sq_${tag}_2 $O/sq_${tag}_3
}
case "$1" in
bench)
	python3 bench.py > $O/bench.json 2> $O/bench.err
	(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof -o runc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-configs > $O/bench_under_rocprof.json 2> $O/prof.err)
	(cd /tmp && BTLBF_BENCH_NO_MISS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o runc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-configs > $O/pmc_f.json 2> $O/pmc_f.err)
	(cd /tmp && BTLBF_BENCH_NO_MISS=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o runc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-configs > $O/pmc_w.json 2> $O/pmc_w.err)
	python3 tools/pmc_summary.py --round r04 --kmers 12e9 --fetch $O/pmc_f --write $O/pmc_w --stats $O/prof > $O/pmc_summary.log
	cp profiles/traffic.json $O/traffic.json; mkdir -p $O/profiles_r04; cp profiles/r04/* $O/profiles_r04/ 2>/dev/null || true
	;;
sq)
	sq_pass passA 39 20000000 partitioned partitioned hitonly
	QB_SPACED=1 sq_pass passA_spaced 37 20000000 partitioned partitioned hitonly
	QB_RAGGED=1 sq_pass passA_ragged 39 20000000 partitioned partitioned hitonly
	QB_BITS='3*2**37' sq_pass passA_3x2p37 39 20000000 partitioned partitioned hitonly
	;;
sq3x) # one shape alone
	QB_BITS='3*2**37' sq_pass passA_3x2p37 39 20000000 partitioned partitioned hitonly
	;;
sqbins) # 512 against 256 level-0 bins on the same filter (2^37 bits, plain ntHash)
	BTLBF_SPLIT_BITS=9 QB_BITS='2**37' sq_pass passA_2p37_512bins 39 20000000 partitioned partitioned hitonly
	BTLBF_SPLIT_BITS=10 QB_BITS='2**37' sq_pass passA_2p37_256bins 39 20000000 partitioned partitioned hitonly
	;;
sqspaced) # C5's kernel alone
	QB_SPACED=1 sq_pass passA_spaced 37 20000000 partitioned partitioned hitonly
	;;
esac
echo session $1 done
