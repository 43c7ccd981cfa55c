#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by a few percent): the shipped one (packed fp32 ops compiled out)
# against build_tmp/libcatre_hip_pk.so (same sources, default flags).   profiles/ab_build.sh > gpurun_out/r03_ab_build.jsonl
for rep in 1 2; do
for lib in "" "$PWD/build_tmp/libcatre_hip_pk.so"; do
  tag=$([ -z "$lib" ] && echo nopk || echo pk)
  for args in "--steps 10 --warmup 3" "--dtype split --steps 10 --warmup 3" "--dtype bf16 --steps 10 --warmup 3" \
              "--mode train --steps 4 --warmup 3" "--mode train --dtype bf16 --steps 4 --warmup 3" "--mode train --dtype split --steps 4 --warmup 3"; do
    CATRE_HIP_LIB=$lib python bench.py $args --no-cpu-baseline --no-train-extra --no-small-extra 2>/dev/null | grep '^{' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'build':'$tag','args':'$args','value':d['value'],'ms_per_step':d['ms_per_step'],'roofline':d.get('roofline',{}).get('frac')}))"
  done
done
done
