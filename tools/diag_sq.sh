#!/bin/bash
# SQ-counter passes over tools/quick_bench.py (run on the GPU box from the repo root):
#   tools/diag_sq.sh <tag> [log2_bits] [reads]
# writes gpurun_out/r02/sq_<tag>_{a,b,c}/ (rocprofv3 --pmc, one pass per counter group) and a summary
set -e
TAG=${1:-base}; LG=${2:-39}; N=${3:-20000000}
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
	i=$((i+1))
	out=$ROOT/gpurun_out/r02/sq_${TAG}_$i
	rm -rf $out
	rocprofv3 --pmc $grp --kernel-trace -d $out -o r -- python3 $ROOT/tools/quick_bench.py $LG $N partitioned partitioned hitonly > $ROOT/gpurun_out/r02/sq_${TAG}_$i.log 2>&1
done
cd $ROOT
python3 tools/sq_counters.py gpurun_out/r02/sq_${TAG}_1 gpurun_out/r02/sq_${TAG}_2 gpurun_out/r02/sq_${TAG}_3 > gpurun_out/r02/sq_${TAG}_summary.txt
