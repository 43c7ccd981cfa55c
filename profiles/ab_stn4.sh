#!/bin/bash
# A/B of the one-wave-per-SIMD STN kernels (CATRE_STN4=1, default) on one box: per-kernel rocprof averages + bench lines
o=gpurun_out
F="--no-cpu-baseline --no-train-extra --no-small-extra --no-split-extra"
: > $o/r05_ab_stn4.txt
for v in 0 1; do
  echo "== CATRE_STN4=$v" >> $o/r05_ab_stn4.txt
  CATRE_STN4=$v profiles/prof.sh $o/r05_ab_stn4_$v.csv python $PWD/bench.py --steps 3 --warmup 1 $F
  grep -E "k_stn|k_trunk" $o/r05_ab_stn4_$v.csv >> $o/r05_ab_stn4.txt
  for rep in 1 2; do CATRE_STN4=$v python bench.py $F 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(json.dumps({'stn4': $v, 'value': r['value'], 'ms_per_step': r['ms_per_step']}))" >> $o/r05_ab_stn4.txt; done
done
cat $o/r05_ab_stn4.txt
