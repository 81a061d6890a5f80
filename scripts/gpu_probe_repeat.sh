#!/bin/bash
# fresh-process repeats of the switch-matrix probe (first overlapped launch of a process): how often does it deviate?
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
N=${1:-40}
python tests/switch_probe.py --save /tmp/default.npz > /dev/null 2>&1
for i in $(seq 1 $N); do
  env $2 python tests/switch_probe.py --compare /tmp/default.npz 2>/dev/null | grep SWITCH_PROBE | python -c "
import sys, json
d = json.loads(sys.stdin.read()[len('SWITCH_PROBE '):])
b, s = d['medium']['big'], d['medium']['small']
flag = 'BAD' if (b['fwd'] > 2e-5 or b['grad'] > 2e-4 or s['fwd'] > 2e-5 or s['grad'] > 2e-4) else 'ok'
print($i, flag, 'big fwd %.2e grad %.2e' % (b['fwd'], b['grad']), 'small fwd %.2e grad %.2e' % (s['fwd'], s['grad']), d['overlap'])"
done 2>&1 | tee gpurun_out/probe_repeat.log | grep -c BAD
grep BAD gpurun_out/probe_repeat.log | head; tail -3 gpurun_out/probe_repeat.log
