#!/bin/bash
# A/B of the one-wave-per-SIMD rotation-head kernel (k_rot_l1w, CATRE_ROTW=1, default) against k_rot_l1<1> on one box:
#   profiles/ab_rotw.sh -> gpurun_out/r06_ab_rotw.jsonl (bench lines), gpurun_out/r06_ab_rotw_kernels.txt (rocprof averages)
o=gpurun_out
F="--no-cpu-baseline --no-train-extra --no-small-extra --no-split-extra"
line() { python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(json.dumps({'variant': '$1', 'value': r['value'], 'ms_per_step': r['ms_per_step'], 'trunk_ms': r['roofline']['avg_launch_ms'], 'frac': r['roofline']['frac']}))"; }
: > $o/r06_ab_rotw.jsonl
for rep in 1 2; do
  for v in 0 1; do CATRE_ROTW=$v python bench.py $F 2>/dev/null | grep '^{' | line rotw=$v >> $o/r06_ab_rotw.jsonl; done
done
cat $o/r06_ab_rotw.jsonl
: > $o/r06_ab_rotw_kernels.txt
for v in 0 1; do
  CATRE_ROTW=$v profiles/prof.sh $o/_ab_rotw_$v.csv python $PWD/bench.py --steps 5 --warmup 2 $F
  echo "== CATRE_ROTW=$v" >> $o/r06_ab_rotw_kernels.txt
  python - $o/_ab_rotw_$v.csv >> $o/r06_ab_rotw_kernels.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0][-40:]
    if any(k in n for k in ("k_rot_l1", "k_trunk4", "k_stn", "k_rot_out")):
        print(f"{n:42s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
  rm -f $o/_ab_rotw_$v.csv
done
cat $o/r06_ab_rotw_kernels.txt
