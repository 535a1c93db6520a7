#!/bin/bash
# round 4: a hardware queue per evaluation lane WITHOUT the process-wide GPU_MAX_HW_QUEUES: CU-masked lane streams (default)
# against plain streams on the runtime's default pool (MACHIP_LANE_QUEUES=shared) and against round 3's env setting
mkdir -p gpurun_out; out=gpurun_out/r4_queues.txt; : > $out
unset GPU_MAX_HW_QUEUES
for nm in intel sphere2500 city10000; do
  echo "== $nm: CU-masked lane streams (default), GPU_MAX_HW_QUEUES unset" >> $out
  timeout 300 python tools/sweep_probe.py $nm 16 20 2>&1 | tail -6 >> $out
  echo "== $nm: MACHIP_LANE_QUEUES=shared (plain streams, runtime default of 4 queues)" >> $out
  MACHIP_LANE_QUEUES=shared timeout 300 python tools/sweep_probe.py $nm 16 20 2>&1 | tail -6 >> $out
  echo "== $nm: MACHIP_LANE_QUEUES=shared + GPU_MAX_HW_QUEUES=16 (round 3's setting)" >> $out
  MACHIP_LANE_QUEUES=shared GPU_MAX_HW_QUEUES=16 timeout 300 python tools/sweep_probe.py $nm 16 20 2>&1 | tail -6 >> $out
done
cat $out
