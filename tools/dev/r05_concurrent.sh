#!/bin/bash
# dev: two processes launching the projection-inside score kernel on the SAME GPU at once (the scheduler time-slices their queues: waves
# are context-saved and restored mid-kernel) -- does the failure rate of tools/dev/r05_ipa_repeat.py change?
mkdir -p gpurun_out
N=${N:-8000} timeout 600 python tools/dev/r05_ipa_repeat.py > gpurun_out/r05r_conc_a.txt 2>&1 &
N=${N:-8000} timeout 600 python tools/dev/r05_ipa_repeat.py > gpurun_out/r05r_conc_b.txt 2>&1 &
wait
tail -1 gpurun_out/r05r_conc_a.txt; tail -1 gpurun_out/r05r_conc_b.txt
