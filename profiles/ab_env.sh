#!/bin/bash
# one-box A/B of an environment switch of the library: profiles/ab_env.sh VAR  -> bench lines + per-kernel rocprof averages
# for VAR=0 and VAR=1 (gpurun_out/r05_ab_$VAR.txt)
v=$1; o=gpurun_out; out=$o/r05_ab_$v.txt
F="--no-cpu-baseline --no-train-extra --no-small-extra --no-split-extra"
: > $out
for x in 0 1; do
  echo "== $v=$x" >> $out
  env $v=$x profiles/prof.sh $o/r05_ab_${v}_$x.csv python $PWD/bench.py --steps 3 --warmup 1 $F > /dev/null
  grep -E "k_stn|k_trunk|k_rot_l1" $o/r05_ab_${v}_$x.csv | sed 's/(catre_points.*)"//; s/(float const.*)"//' | cut -c1-90 >> $out
  for rep in 1 2; do env $v=$x python bench.py $F 2>/dev/null | grep '^{' | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(json.dumps({'$v': $x, 'value': r['value'], 'ms_per_step': r['ms_per_step']}))" >> $out; done
done
cat $out
