# Run a long, quiet command on the GPU box with a progress line every 50 s (gpurun takes 7 silent minutes for a hang):
#   bash scripts/with_heartbeat.sh <logfile> <command...>
log=$1; shift
( while true; do sleep 50; echo "$(date +%T) still running: $1 $2 $3" >> "$log"; done ) &
hb=$!
"$@"
rc=$?
kill $hb 2>/dev/null
exit $rc
