#!/bin/bash
OUT=gpurun_out/r04z
mkdir -p $OUT
timeout 300 python scripts/micro/eager_host.py 2>&1 | cut -c1-200 | tee $OUT/eager_host.txt
