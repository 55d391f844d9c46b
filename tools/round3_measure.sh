#!/bin/bash
# One GPU-box session producing the round-3 artefacts (run through gpurun; outputs under gpurun_out/round3/):
# headline bench in three arithmetic modes, rocprofv3 kernel trace of the bench command, PMC passes of the forward kernel,
# event times + kernel trace + PMC passes of the non-headline kernels, stamps of the pipelined kernel.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round3
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || exit 2
timeout -k 10 400 python bench.py --math fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err || exit 3
timeout -k 10 400 python bench.py --math fp16x2 --no-cpu-baseline > $O/bench_fp16x2.json 2> $O/bench_fp16x2.err || exit 3
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args.json 2> $O/bench_driver_args.err || exit 3
# one GPU's shard of an N-GPU strong-scaling step + the kernel trace of the N = 8 shard
for n in 8 4 2; do timeout -k 10 200 python bench.py --shard-of $n > $O/shard_of_$n.json 2> $O/shard_of_$n.err || exit 3; done
timeout -k 10 200 python tools/shard_times.py > $O/shard_times.txt 2>&1 || exit 3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 300 --warmup 100 --no-cpu-baseline > $O/prof_bench.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shard8 -o shard8 -- python3 $R/bench.py --shard-of 8 --steps 300 --warmup 100 > $O/prof_shard8.log 2>&1 || exit 4
cd $R
timeout -k 10 600 bash tools/pmc_fwd.sh > $O/pmc.log 2>&1 || exit 5
rm -rf $O/pmc && mv $R/gpurun_out/pmc $O/pmc
timeout -k 10 300 python tools/run_secondary.py 40 > $O/run_secondary.json 2> $O/run_secondary.err || exit 6
cd /tmp
for B in 100 65536; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_secondary_$B -o sec -- python3 $R/tools/run_secondary.py 20 $B > $O/prof_secondary_$B.log 2>&1 || exit 7
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_s1_$B -- python3 $R/tools/run_secondary.py 4 $B > $O/pmc_s1_$B.log 2>&1 || exit 8
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_s2_$B -- python3 $R/tools/run_secondary.py 4 $B > $O/pmc_s2_$B.log 2>&1 || exit 8
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_s3_$B -- python3 $R/tools/run_secondary.py 4 $B > $O/pmc_s3_$B.log 2>&1 || exit 8
done
cd $R
timeout -k 10 400 python tools/bench_secondary.py > $O/secondary.log 2>&1 || exit 9
cp $R/gpurun_out/secondary.json $O/secondary.json
echo round3_measure_ok
