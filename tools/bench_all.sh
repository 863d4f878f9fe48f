#!/bin/bash
# every bench workload once (no CPU baseline) + the host-API timings; run on the GPU box
cd $GRAFT_REPO_ROOT
out=gpurun_out/bench_all.txt
: > $out
for wl in kitti_stereo_1241x376_1000feat kitti_stereo_1241x376_2000feat euroc_stereo_752x480_1000feat mono_1241x376_1000feat mono_640x480_1000feat; do
  python3 bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s %10.1f %s  %.3f ms/step  stages %s' % (d['config']['workload'], d['value'], d['unit'], d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_call'].items() if isinstance(v,float)}))" >> $out
done
python3 bench.py --workload mono_1920x1080_4000feat --batch 32 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s %10.1f %s  %.3f ms/step (batch 32)  stages %s' % (d['config']['workload'], d['value'], d['unit'], d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_call'].items() if isinstance(v,float)}))" >> $out
python3 tools/bench_host_api.py >> $out 2>/dev/null
python3 tools/bench_matchers.py >> $out 2>/dev/null
cat $out
