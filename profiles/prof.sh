#!/bin/bash
# rocprofv3 kernel stats of a command, exported as CSV next to gpurun_out (run on the GPU box):
#   profiles/prof.sh <out.csv> <command...>
# (rocprofv3 wants a writable cwd/TMPDIR: /tmp.)
out=$(realpath -m "$1"); shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=$(mktemp -d /tmp/catre_prof.XXXX)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$d" -o run -- "$@" > "$d/stdout.log" 2> "$d/stderr.log"
db=$(find "$d" -name '*.db' | head -1)
if [ -n "$db" ]; then python "$root/profiles/export_rocprof.py" "$db" "$out" $PROF_TRACE; else
  csv=$(find "$d" -name '*kernel_stats.csv' | head -1); [ -n "$csv" ] && cp "$csv" "$out" || { echo "no rocprof output"; tail -5 "$d/stderr.log"; }; fi
rm -rf "$d"
