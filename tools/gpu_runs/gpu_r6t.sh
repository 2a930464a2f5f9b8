#!/bin/bash
# Round 6, second session: second half's action upload as one DMA command (engine key pipeline_upload) A/B + timelines
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6t; mkdir -p $O
for rep in 1 2 3 4; do for pu in 0 1; do
  echo "pipeline_upload=$pu rep=$rep $(EPA_PARAMS=pipeline_upload=$pu python tools/numpy_step_ab.py HalfCheetah 65536 32768 6 bind 2>/dev/null | tail -1)"
done; done | tee $O/pipeline_upload_ab.txt
R=$PWD
for pu in 0 1; do
  cd /tmp; rm -rf /tmp/tl$pu
  EPA_PARAMS=pipeline_upload=$pu rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl$pu -o t -- python $R/tools/numpy_step_ab.py HalfCheetah 65536 32768 6 bind > /dev/null 2>&1
  cd $R; python tools/numpy_step_timeline.py /tmp/tl$pu > $O/timeline_upload$pu.txt 2>&1; head -24 $O/timeline_upload$pu.txt
done
