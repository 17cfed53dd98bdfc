#!/bin/bash
set -u
O=gpurun_out/r3d; mkdir -p $O
GTN_HOST_SAMPLE=$O/samples.txt tools/nullhip/_bin/region_step 600 512 256 > $O/region_step.log 2>&1
cat $O/region_step.log
python tools/nullhip/report.py $O/samples.txt 60 > $O/report.txt 2>&1
rm -f $O/samples.txt
head -150 $O/report.txt
