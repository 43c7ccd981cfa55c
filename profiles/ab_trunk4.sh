#!/bin/bash
# A/B of the one-wave-per-SIMD trunk (k_trunk4, CATRE_TRUNK4=1, default) against the 8-wave k_trunk on one box:
#   profiles/ab_trunk4.sh [extra lib ...] -> gpurun_out/r05_ab_trunk4.jsonl, gpurun_out/r05_trunk4_phases.txt
o=gpurun_out
F="--no-cpu-baseline --no-train-extra --no-small-extra --no-split-extra"
line() { python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(json.dumps({'variant': '$1', 'value': r['value'], 'ms_per_step': r['ms_per_step'], 'trunk_ms': r['roofline']['avg_launch_ms'], 'frac': r['roofline']['frac']}))"; }
: > $o/r05_ab_trunk4.jsonl
for rep in 1 2; do
  for v in 0 1; do CATRE_TRUNK4=$v python bench.py $F 2>/dev/null | grep '^{' | line trunk4=$v >> $o/r05_ab_trunk4.jsonl; done
  for lib in "$@"; do CATRE_HIP_LIB=$PWD/$lib python bench.py $F 2>/dev/null | grep '^{' | line $lib >> $o/r05_ab_trunk4.jsonl; done
done
cat $o/r05_ab_trunk4.jsonl
{ for v in 0 1; do echo "== CATRE_TRUNK4=$v"; CATRE_TRUNK4=$v CATRE_HIP_LIB=$PWD/catre_amd/csrc/libcatre_hip_trace.so python profiles/trace_trunk.py; done; } > $o/r05_trunk4_phases.txt 2>&1
cat $o/r05_trunk4_phases.txt
