#!/bin/bash
# developer helper (build container): launch a gpurun call in the background and return once the snapshot of /root/repo has been
# taken (relatime: the push's read of a freshly touched sentinel moves its atime past its mtime) -- the tree may be edited again.
# usage: scripts/gpu_launch.sh <log file> <gpurun timeout> <command string>
cd /root/repo
touch .snap_sentinel
(gpurun --timeout "$2" -- "$3" > "$1" 2>&1 &)
for i in $(seq 1 240); do
  sleep 5
  a=$(stat -c %X .snap_sentinel); m=$(stat -c %Y .snap_sentinel)
  if [ "$a" -gt "$m" ]; then echo "snapshot taken after $((i * 5)) s"; exit 0; fi
  if grep -q "status=" "$1" 2>/dev/null; then echo "call ended early"; tail -3 "$1"; exit 0; fi
done
echo "no snapshot seen in 20 min"
