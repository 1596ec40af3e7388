#!/bin/bash
# gpurun_out/set_<tag>/ -> profiles/<round>/a_* and the traffic files bench.py links (run in the dev container after tools/gpu_collect.sh)
# Usage: tools/publish_profiles.sh <tag> <round dir, e.g. r06>
set -eu
TAG=$1; R=$2
mkdir -p profiles/$R
rm -f profiles/$R/a_*
for f in gpurun_out/set_$TAG/*; do b=$(basename $f); case $b in profile*.log|err.log) ;; *) cp $f profiles/$R/a_$b;; esac; done
python3 tools/make_traffic.py profiles/$R/a - c2 1024 > /dev/null
python3 tools/make_traffic.py profiles/$R/a_c2_i1024 - c2_i1024 1024 > /dev/null
python3 tools/make_traffic.py profiles/$R/a_c2_beta1 - c2_beta1 1024 > /dev/null
python3 tools/make_traffic.py profiles/$R/a_c4 - c4 256 > /dev/null
python3 tools/make_traffic.py profiles/$R/a_c4_i2048 - c4_i2048 256 > /dev/null
if [ -f profiles/$R/a_c4_b1024_pmc_fetch_size.txt ]; then python3 tools/make_traffic.py profiles/$R/a_c4_b1024 - c4_b1024 1024 > /dev/null; fi
ls profiles/$R | wc -l
