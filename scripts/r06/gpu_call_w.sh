#!/bin/bash
# round 6, call W: the closing state once more -- whole GPU suite, smoke(), the driver's bench command
set -u
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_final_pytest_gpu.log 2>&1; tail -3 gpurun_out/r06_final_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v Warn | tail -3
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_line_driver_cmd.json 2> gpurun_out/r06_w_bench.err; echo "bench.py exit $? after $SECONDS s"; cut -c1-300 gpurun_out/r06_bench_line_driver_cmd.json
cp gpurun_out/bench_full.json gpurun_out/r06_bench_full.json
