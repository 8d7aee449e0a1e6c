#!/bin/bash
# the in-kernel loop with and without the draw wave (spec_main.h, SPEC_DRAW_WAVE) at several sample counts
run() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['number_samples_per_gpu'], 'samples', d['config']['grid']['n_threads'], 'threads', round(d['ms_per_step']*1e3,3), 'us', round(d['value']), 'it/s')"; }
for n in ${SAMPLES:-64 128 192 256 300 384 448}; do
run --samples $n
BSVI_SPEC_DRAW_WAVE=0 run --samples $n
done
