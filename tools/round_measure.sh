#!/bin/bash
# One GPU-box session that produces every artefact profiles/README.md lists for a round (run through gpurun):
# parity suite, headline bench in the three arithmetic modes, rocprofv3 kernel stats of the bench command, PMC passes
# (HBM traffic, MFMA busy), the secondary timings and the synthetic training iterations.  Outputs: gpurun_out/round/.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || exit 2
timeout -k 10 400 python bench.py --math fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err || exit 3
timeout -k 10 400 python bench.py --math bf16x3 --no-cpu-baseline > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err || exit 3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 300 --warmup 100 --no-cpu-baseline > $O/prof_bench.log 2>&1 || exit 4
cd $R
timeout -k 10 600 bash tools/pmc_fwd.sh > $O/pmc.log 2>&1 || exit 5
rm -rf $O/pmc && mv $R/gpurun_out/pmc $O/pmc
timeout -k 10 400 python tools/bench_secondary.py > $O/secondary.log 2>&1 || exit 6
cp $R/gpurun_out/secondary.json $O/secondary.json
for d in svhn celeba cifar10 celeba_hq256; do
  b=100; [ $d = celeba_hq256 ] && b=16
  timeout -k 10 300 python examples/train_synthetic.py --dataset $d --batch $b --iters 5 --warmup 2 2> $O/train_$d.err | tail -1 >> $O/train_configs.jsonl || exit 7
done
echo round_measure_ok
