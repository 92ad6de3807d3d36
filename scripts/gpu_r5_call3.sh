#!/bin/bash
# Round 5: exact window energies (quiet windows) + GPU suite + C2 bench line
set -o pipefail
O=gpurun_out/r05_call3
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py -x -q -k "quiet or zero_energy" 2>&1 | tee $O/test_quiet.log || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_api.py -x -q -k "zero_energy or cztxcorr" 2>&1 | tee -a $O/test_quiet.log || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee $O/gpu_suite.log || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 3 2>&1 | tee $O/bench_c2.json
