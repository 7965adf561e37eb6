#!/bin/bash
# Developer tool: the rocprofv3 runs behind profiles/ (run through gpurun from the repo root; outputs land in gpurun_out/).
#   kernel trace + stats of the default bench, PMC passes (FETCH_SIZE, WRITE_SIZE separately) of the headline kernel,
#   kernel trace of a graph-replayed decode and of a chunked prefill.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python $ROOT/bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py --steps 288 --warmup 72 --no-extras --no-cpu-baseline --launch eager > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
# the graph-replayed legs alone (no extras, no CPU baseline): start-to-start intervals of the headline kernel under the profiler
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/roofleg -o roofleg -- python $ROOT/bench.py --no-extras --no-cpu-baseline > $OUT/roofleg.json 2> $OUT/roofleg.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode -o decode -- python $ROOT/tools/profile_decode.py 64 > $OUT/decode.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode_sampled -o decode_sampled -- python $ROOT/tools/profile_decode.py 64 sampled > $OUT/decode_sampled.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefill -o prefill -- python $ROOT/tools/profile_prefill.py > $OUT/prefill.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w8a8 -o w8a8 -- python $ROOT/tools/w8a8_config3.py > $OUT/w8a8.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python $ROOT/tools/profile_w8a8_c3.py > $OUT/c3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gemm -o gemm -- python $ROOT/tools/vendor_probe.py all > $OUT/gemm.log 2>&1
find $OUT/w8a8 $OUT/c3 $OUT/gemm -name "*kernel_trace.csv" -delete
# condense on the box (the bench kernel trace alone exceeds the 64 MiB return budget), then drop the raw traces
PROFILE_DST=$ROOT/gpurun_out/profiles_out python $ROOT/tools/make_profile_summary.py ${PROFILE_TAG:-r06} > $OUT/summary.log 2>&1
find $OUT/bench $OUT/roofleg $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name "*kernel_trace.csv" -delete
# keep only what fits the 64 MiB return budget: stats and (for the PMC / bench runs) the counter / kernel trace tables
find $OUT -name "*_agent_info.csv" -delete
find $OUT/decode $OUT/decode_sampled $OUT/prefill -name "*kernel_trace.csv" -delete
ls -la $OUT $OUT/*/* 2>/dev/null | head -60
du -sh $OUT
