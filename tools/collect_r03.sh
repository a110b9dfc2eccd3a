#!/bin/bash
# Round-3 evidence, run on the GPU box through gpurun from the repo root: kernel-trace stats of the bench workloads,
# PMC passes (HBM traffic of the headline product; issue / LDS counters of the M1 kernels; each in its own run, never
# combined with other trace domains), and the un-profiled bench lines.  Results: gpurun_out/r03/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03
mkdir -p $OUT
for WL in ${WORKLOADS:-default m2b m4}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -- python3 bench.py --workload $WL --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline > $OUT/under_rocprof_$WL.log 2>&1
  cp $OUT/trace_$WL/*/*kernel_stats.csv $OUT/kernel_stats_$WL.csv 2>/dev/null
  rm -rf $OUT/trace_$WL
  echo "== $WL"; head -8 $OUT/kernel_stats_$WL.csv | cut -c1-150
done
pmc() { # workload tag counters
  rocprofv3 --pmc $3 --kernel-trace --output-format csv -d $OUT/pmc_$1_$2 -- python3 bench.py --workload $1 --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline > $OUT/pmc_$1_$2.log 2>&1
}
for WL in ${PMC_WORKLOADS:-m2a m1}; do
  pmc $WL fetch "FETCH_SIZE"
  pmc $WL write "WRITE_SIZE"
  pmc $WL sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
  python3 tools/pmc_summary.py $OUT "" > $OUT/pmc_summary_$WL.txt
  rm -rf $OUT/pmc_${WL}_fetch $OUT/pmc_${WL}_write $OUT/pmc_${WL}_sq
done
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for WL in m2b m4 m3b; do python3 bench.py --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; done
MXX_BENCH_FORCE_DIST=1 python3 bench.py --steps 10 --repeats 1 --no-cpu-baseline > $OUT/bench_world1_rccl.json 2> $OUT/bench_world1_rccl.err
MXX_BENCH_INPROC_SHARE_DEVICES=1 python3 bench.py --gpus 2 --inproc --steps 10 --repeats 1 > $OUT/bench_inproc2_shared_device.json 2> $OUT/bench_inproc2.err
python3 bench.py --gpus 1 --inproc --steps 10 --repeats 1 > $OUT/bench_inproc1_rccl.json 2> $OUT/bench_inproc1.err
tail -c 300 $OUT/bench_default.json
